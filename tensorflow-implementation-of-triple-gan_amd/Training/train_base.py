"""Train_base — loss / optimiser library, counterpart of the reference's Training/train_base.py.

`_loss_GAN` (train_base.py:113-154) is kept as the one loss the entry point calls; it is evaluated by three fused
single-launch kernels (tg_d_loss_f32 / tg_g_loss_f32 / tg_c_loss_f32) that also write d(loss)/d(logits) into the
logits' gradient buffers.  `_Adam_optimizer` (:91-97) returns the TF-form Adam configuration applied by
tg_adam_f32 over a network's flat buffers; `_train_op` applies it.
"""
from tg import lib
from tg.runtime import ctx


class AdamOptimizer(object):
    """tf.train.AdamOptimizer(learning_rate, beta1, beta2=0.999, epsilon=1e-8); lr is a DEVICE scalar."""

    def __init__(self, lr_dev, beta1, beta2=0.999, epsilon=1e-8, name='Adam_optimizer'):
        self.lr_dev, self.beta1, self.beta2, self.epsilon, self.name = lr_dev, beta1, beta2, epsilon, name

    def apply(self, store, grad_scale=1.0):
        cx = ctx()
        lib.call('tg_adam_f32', lib.ptr(store.p), lib.ptr(store.g), lib.ptr(store.m), lib.ptr(store.v), store.n_p,
                 lib.ptr(self.lr_dev), self.beta1, self.beta2, self.epsilon, lib.ptr(store.step), grad_scale, cx.stream)


class Train_base(object):
    def __init__(self):
        pass

    def _input_fn(self):
        raise NotImplementedError('metirc() is implemented in Model sub classes')

    def _build_train_graph(self):
        raise NotImplementedError('loss() is implemented in Model sub classes')

    def _Adam_optimizer(self, lr, beta1, name='Adam_optimizer'):
        return AdamOptimizer(lr, beta1, name=name)

    def _train_op(self, optimizer, store, grad_scale=1.0):
        """optimizer.minimize(loss, var_list) (train_base.py:64-68): the gradients are already in store.g."""
        ctx().prep_invalidate(store)
        optimizer.apply(store, grad_scale)

    # ---- _loss_GAN split by solver (each writes value + d/dlogits) --------------------------------
    def _d_loss(self, d_logits, n_real, n_fake, n_unl, loss_out):
        """d_loss = BCE(D_real,1) + .5 BCE(D_fake,0) + .5 BCE(D_unl,0) (train_base.py:123-126); rows [real|fake|unl]."""
        cx = ctx()
        g = cx.new_act(d_logits.n, 1, 1, 1, 32, tag='dl')
        lib.call('tg_d_loss_f32', d_logits.ptr, d_logits.ld, n_real, n_fake, n_unl, g.ptr, g.ld, lib.ptr(loss_out), cx.stream)
        d_logits.grad = g

    def _g_loss(self, d_fake_logits, loss_out):
        """g_loss = 1/2 BCE(D_fake,1) (train_base.py:128)."""
        cx = ctx()
        g = cx.new_act(d_fake_logits.n, 1, 1, 1, 32, tag='dl')
        lib.call('tg_g_loss_f32', d_fake_logits.ptr, d_fake_logits.ld, d_fake_logits.n, g.ptr, g.ld, lib.ptr(loss_out), cx.stream)
        d_fake_logits.grad = g

    def _c_loss(self, c_logits, n_real, n_unl, n_rep, n_fake, y_l_c, y_g, d_unl_logits, lambdas_dev, loss_out):
        """c_loss (train_base.py:118,130-152); rows of c_logits [real|unl|unl_rep|fake]."""
        cx = ctx()
        g = cx.new_act(c_logits.n, 1, 1, c_logits.c, c_logits.ld, tag='dl')
        lib.call('tg_c_loss_f32', c_logits.ptr, c_logits.ld, n_real, n_unl, n_rep, n_fake, y_l_c.ptr, y_g.ptr,
                 d_unl_logits.ptr, d_unl_logits.ld, lib.ptr(lambdas_dev), g.ptr, g.ld, lib.ptr(loss_out), cx.stream)
        c_logits.grad = g

    def _loss_GAN(self, D, C, Y, Lambda, loss_out):
        """All three losses of train_base.py:113-154 on the outputs of Model.forward_pass (evaluation / tests).
        D = [_, D_real_logits, _, D_fake_logits, _, D_unl_logits]; C = [C_real, C_unl, C_unl_d, C_fake(, C_unl_rep)];
        Y = [y_g, y_l_c]; Lambda = device tensor {lambda_1, lambda_2}; loss_out = device tensor of 3 floats."""
        from tg.batching import concat_acts
        _, d_real, _, d_fake, _, d_unl = D
        dcat = concat_acts([d_real, d_fake, d_unl])
        self._d_loss(dcat, d_real.n, d_fake.n, d_unl.n, loss_out[0:1])
        self._g_loss(d_fake, loss_out[1:2])
        c_real, c_unl, _, c_fake = C[:4]
        c_rep = C[4] if len(C) > 4 else None
        ccat = concat_acts([c_real, c_unl] + ([c_rep] if c_rep is not None else []) + [c_fake])
        y_g, y_l_c = Y
        self._c_loss(ccat, c_real.n, c_unl.n, c_rep.n if c_rep is not None else 0, c_fake.n, y_l_c, y_g, d_unl, Lambda, loss_out[2:3])
        return loss_out
