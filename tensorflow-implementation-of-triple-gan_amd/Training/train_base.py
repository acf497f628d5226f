"""Train_base — loss / optimiser library, counterpart of the reference's Training/train_base.py.

`_loss_GAN(D, C, Y, Lambda)` (train_base.py:113-154) is kept as the one loss the entry point calls; it is evaluated by three fused
single-launch kernels (tg_d_loss_f32 / tg_g_loss_f32 / tg_c_loss_f32) that also write d(loss)/d(logits) into the
logits' gradient buffers.  The helper methods it is written with in the reference — `_entropy`, `_balance_entropy` (:43-57),
`_softmax_cross_entropy_loss_w_logits`, `_sigmoid_cross_entopy_w_logits` (:75-84), `_accuracy_metric` (:107) — keep their names and
argument order as stand-alone single-launch heads.  `_Adam_optimizer` (:91-97) returns the TF-form Adam configuration applied by
tg_adam_f32 over a network's flat buffers; `_train_op` applies it.

Eager-mode conventions: a loss value is a 1-element DEVICE tensor (float(t) synchronises); every head also leaves d(value)/d(logits)
in `logits.grad` — written when the tensor has no gradient yet, ADDED when it has, so a loss summed from several heads accumulates
its gradient the way TensorFlow's autodiff would; `weight` scales value and gradient.
"""
import torch

from tg import lib
from tg.runtime import ctx


class StreamingAccuracy(object):
    """tf.metrics.accuracy (train_base.py:107): device counters {correct, total}; float(metric) = the accuracy so far."""

    def __init__(self, num_classes=10):
        self.k = num_classes
        self.counters = torch.zeros(2, dtype=torch.float32, device=ctx().device)

    def update(self, labels, logits):
        """labels: one-hot Act [n,k]; logits: Act [n,k] — arg-max of both is taken in the kernel."""
        lib.call('tg_accuracy_count_f32', logits.ptr, logits.ld, labels.ptr, logits.n, self.k, lib.ptr(self.counters), ctx().stream)
        return self

    def reset(self):
        lib.call('tg_fill_f32', lib.ptr(self.counters), 0.0, 2, ctx().stream)
        return self

    def result(self):
        correct, total = self.counters.cpu().numpy()
        return float(correct) / max(float(total), 1.0)

    __float__ = result


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

    # ---- helper heads (train_base.py:43-57,75-84,107) -----------------------------------------------
    @staticmethod
    def _grad_of_logits(logits):
        """(gradient Act, accumulate flag) — see the module docstring."""
        cx = ctx()
        if logits.grad is None:
            logits.grad = cx.new_act(logits.n, logits.h, logits.w, logits.c, logits.ld, tag='dl')
            return logits.grad, 0
        return logits.grad, 1

    def _entropy(self, logits, weight=1.0):
        """mean_n(logsumexp(l) - sum_k softmax_k l_k) (train_base.py:43-48)."""
        cx = ctx()
        out = cx.scratch('lossv', 4)
        g, acc = self._grad_of_logits(logits)
        lib.call('tg_entropy_terms_f32', logits.ptr, logits.ld, logits.n, logits.c, float(weight), 0.0, g.ptr, g.ld, acc, lib.ptr(out), cx.stream)
        return out[0:1]

    def _balance_entropy(self, logits, weight=1.0):
        """-sum_k (1/K) log(mean_n softmax_k + 1e-12) (train_base.py:50-57)."""
        cx = ctx()
        out = cx.scratch('lossv', 4)
        g, acc = self._grad_of_logits(logits)
        lib.call('tg_entropy_terms_f32', logits.ptr, logits.ld, logits.n, logits.c, 0.0, float(weight), g.ptr, g.ld, acc, lib.ptr(out), cx.stream)
        return out[0:1]

    def _softmax_cross_entropy_loss_w_logits(self, labels, logits, weight=1.0):
        """reduce_mean(softmax_cross_entropy_with_logits_v2(labels, logits)) (train_base.py:75-79); labels: dense Act [n,k]."""
        cx = ctx()
        assert labels.ld == labels.c == logits.c and labels.n == logits.n
        out = cx.scratch('lossv', 4)
        g, acc = self._grad_of_logits(logits)
        lib.call('tg_softmax_ce_f32', logits.ptr, logits.ld, labels.ptr, logits.n, logits.c, float(weight), g.ptr, g.ld, acc, lib.ptr(out), cx.stream)
        return out[0:1]

    def _sigmoid_cross_entopy_w_logits(self, labels, logits, weight=1.0):
        """reduce_mean(sigmoid_cross_entropy_with_logits(labels, logits)) (train_base.py:81-84).  labels: an Act of the logits' shape, or a
        number standing for tf.ones_like(logits) / tf.zeros_like(logits) (:123-128)."""
        cx = ctx()
        out = cx.scratch('lossv', 4)
        g, acc = self._grad_of_logits(logits)
        const = not hasattr(labels, 'ptr')
        lib.call('tg_bce_logits_f32', logits.ptr, logits.ld, None if const else labels.ptr, 0 if const else labels.ld, float(labels) if const else 0.0,
                 logits.rows, logits.c, float(weight), g.ptr, g.ld, acc, lib.ptr(out), cx.stream)
        return out[0:1]

    def _accuracy_metric(self, labels, predictions, metric=None):
        """tf.metrics.accuracy(labels, predictions) (train_base.py:107) -> (accuracy, update_op): one update with this batch on `metric`
        (a new StreamingAccuracy when None); float(accuracy) reads the running value, update_op(labels, predictions) adds a batch.
        labels: one-hot Act; predictions: logits Act (the arg-max of Train._metric, Train_goodGAN.py:432-433, happens in the kernel)."""
        metric = metric if metric is not None else StreamingAccuracy(predictions.c)
        metric.update(labels, predictions)
        return metric, metric.update

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

    def _loss_GAN(self, D, C, Y, Lambda):
        """train_base.py:113-154 on the outputs of Model.forward_pass, the reference's arguments:
        D = [D_real, D_real_logits, D_fake, D_fake_logits, D_unl, D_unl_logits]; C = [C_real_logits, C_unl_logits, C_unl_d_logits,
        C_fake_logits(, C_unl_logits_rep — config.DATA_NAME 'cifar10')]; Y = [y_g, y_l_c]; Lambda = [lambda_1(, lambda_2)] as numbers or
        a device tensor.  Returns (d_loss, g_loss, c_loss), 1-element device tensors."""
        from tg.batching import concat_acts
        cx = ctx()
        loss_out = cx.scratch('loss_gan', 4)
        if isinstance(Lambda, torch.Tensor):
            lam = Lambda
        else:
            vals = [float(v) for v in Lambda] + [0.0, 0.0]
            lam = cx.scratch('loss_lambda', 2)
            lam.copy_(torch.tensor(vals[:2], dtype=torch.float32))
        _, d_real, _, d_fake, _, d_unl = D
        dcat = concat_acts([d_real, d_fake, d_unl])
        self._d_loss(dcat, d_real.n, d_fake.n, d_unl.n, loss_out[0:1])
        self._g_loss(d_fake, loss_out[1:2])
        c_real, c_unl, _c_unl_d, c_fake = C[:4]
        c_rep = C[4] if len(C) > 4 else None
        ccat = concat_acts([c_real, c_unl] + ([c_rep] if c_rep is not None else []) + [c_fake])
        y_g, y_l_c = Y
        self._c_loss(ccat, c_real.n, c_unl.n, c_rep.n if c_rep is not None else 0, c_fake.n, y_l_c, y_g, d_unl, lam, loss_out[2:3])
        self.last_loss_inputs = (dcat, ccat)                   # the concatenated logits that carry d(loss)/d(logits)
        return loss_out[0:1], loss_out[1:2], loss_out[2:3]

    # ---- loss variants of train_base.py:156-574 -----------------------------------------------------------------
    # No trainer of the reference repository calls these (Train_goodGAN.py uses _loss_GAN); they are kept for the sibling trainers'
    # models: the same argument tuples (Acts instead of tensors), the same return nesting (Python floats: ONE device->host copy of the
    # term values at the end), and d(loss)/d(input) left in the inputs' `.grad`:
    #     D logits <- d_loss (concatenated copy in `self.last_d_cat`), D_fake_logits.grad <- gG_loss, C_bG_fake_feat.grad <- bG_loss,
    #     every classifier logit tensor <- c_loss.
    # Lambda: Python floats.  Every variant is a weighted sum of the same terms: tg_d_loss_terms_f32, tg_g_loss_f32,
    # tg_c_loss_terms_f32, tg_true_fake_loss_f32, tg_sqdiff_rows_loss_f32, tg_feature_match_f32, tg_pull_away_f32.

    def _variant_terms(self, D, c_real, c_unl, c_rep, c_gfake, c_bfake, c_pert, f_bfake, f_unl, y_l_c, y_g, w6, w_bad, w_pert, pt):
        """runs the term kernels; returns the host array [d, d_real, d_fake, d_unl, gG, c_head, T_real, T_unl, T_H, T_bal, T_gfake, T_mse,
        tf_w, T_bad_unl, T_bfake, sq_w, T_sq, fm, pt]."""
        import ctypes as C
        import torch
        from tg.batching import concat_acts
        from tg.runtime import Act
        cx = ctx()
        lv = torch.zeros(24, dtype=torch.float32, device=cx.device)
        P = lambda i: lib.ptr(lv[i:])
        if D is not None:
            _, d_real, _, d_fake, _, d_unl = D
            dcat = concat_acts([d_real, d_fake, d_unl])
            g = cx.new_act(dcat.n, 1, 1, 1, 32, tag='dl')
            lib.call('tg_d_loss_terms_f32', dcat.ptr, dcat.ld, d_real.n, d_fake.n, d_unl.n, g.ptr, g.ld, P(0), P(1), cx.stream)
            dcat.grad = g
            self.last_d_cat = dcat
            gg = cx.new_act(d_fake.n, 1, 1, 1, 32, tag='dl')
            lib.call('tg_g_loss_f32', d_fake.ptr, d_fake.ld, d_fake.n, gg.ptr, gg.ld, P(4), cx.stream)
            d_fake.grad = gg
        else:
            d_unl = None
        members = [c_real, c_unl] + ([c_rep] if c_rep is not None else []) + ([c_gfake] if c_gfake is not None else [])
        ccat = concat_acts(members)
        g = cx.new_act(ccat.n, 1, 1, ccat.c, ccat.ld, tag='dl')
        n_rep = c_rep.n if c_rep is not None else 0
        n_gf = c_gfake.n if c_gfake is not None else 0
        lib.call('tg_c_loss_terms_f32', ccat.ptr, ccat.ld, c_real.n, c_unl.n, n_rep, n_gf, y_l_c.ptr, y_g.ptr if c_gfake is not None else None,
                 d_unl.ptr if d_unl is not None else None, d_unl.ld if d_unl is not None else 0, (C.c_float * 6)(*w6), g.ptr, g.ld, P(5), P(6),
                 cx.stream)
        ccat.grad = g
        off = 0
        for m in members:                                              # the members' gradients are row ranges of the concatenated one
            m.grad = g.view_rows(off, off + m.n)
            off += m.n
        g_unl = c_unl.grad
        gb = cx.new_act(c_bfake.n, 1, 1, c_bfake.c, c_bfake.ld, tag='dl')
        unl_rows = ccat.view_rows(c_real.n, c_real.n + c_unl.n)
        lib.call('tg_true_fake_loss_f32', unl_rows.ptr, ccat.ld, c_unl.n, c_bfake.ptr, c_bfake.ld, c_bfake.n, w_bad, w_bad, g_unl.ptr, g_unl.ld, 1,
                 gb.ptr, gb.ld, 0, P(12), cx.stream)
        c_bfake.grad = gb
        if c_pert is not None:
            gp = cx.new_act(c_pert.n, 1, 1, c_pert.c, c_pert.ld, tag='dl')
            lib.call('tg_sqdiff_rows_loss_f32', c_pert.ptr, c_pert.ld, c_bfake.ptr, c_bfake.ld, c_pert.n, c_pert.c, w_pert, gp.ptr, gp.ld, 0,
                     gb.ptr, gb.ld, 1, P(15), cx.stream)
            c_pert.grad = gp
        # bad generator: feature matching (+ pull-away) on dense [n][c] features
        assert f_bfake.ld == f_bfake.c and f_unl.ld == f_unl.c, "features must be dense [n][c]"
        gf = cx.new_act(f_bfake.n, 1, 1, f_bfake.c, f_bfake.c, tag='dl')
        gu = cx.scratch('dfu', f_unl.n * f_unl.c)
        lib.call('tg_feature_match_f32', f_bfake.ptr, f_bfake.n, f_unl.ptr, f_unl.n, f_bfake.c, gf.ptr, lib.ptr(gu), P(17), cx.stream)
        if pt is not None:
            n, c = f_bfake.n, f_bfake.c
            gpt = cx.scratch('dpt', n * c)
            lib.call('tg_pull_away_f32', f_bfake.ptr, n, c, 1 if pt == 'masked' else 0, lib.ptr(cx.scratch('pts', n * c + n * n + n)), lib.ptr(gpt),
                     P(18), cx.stream)
            lib.call('tg_add_f32', gf.ptr, gf.ptr, lib.ptr(gpt), n * c, cx.stream)
        f_bfake.grad = gf
        return [float(v) for v in lv.cpu().numpy()]

    def _loss_BGAN(self, C, Y, Lambda=None):
        """train_base.py:156-184 -> (g_loss, c_loss); C = [C_real, C_unl, C_fake logits, feat_real, feat_unl, feat_fake]."""
        c_real, c_unl, c_fake, _f_real, f_unl, f_fake = C
        v = self._variant_terms(None, c_real, c_unl, None, None, c_fake, None, f_fake, f_unl, Y[0], None, [1.0, 0.0, 0.1, 1e-3, 0.0, 0.0], 1.0, 0.0,
                                'masked')
        return v[17] + v[18], v[5] + v[12]

    def _good_bad(self, D, C, Y, Lambda, perturb):
        if perturb:
            c_real, c_unl, _cud, c_gfake, c_bfake, c_pert, _fr, f_unl, f_bfake, _fp = C
        else:
            (c_real, c_unl, _cud, c_gfake, c_bfake, _fr, f_unl, f_bfake), c_pert = C, None
        y_g, y_l_c = Y
        lam1 = float(Lambda[0])
        v = self._variant_terms(D, c_real, c_unl, None, c_gfake, c_bfake, c_pert, f_bfake, f_unl, y_l_c, y_g,
                                [1.0, 0.01 * 0.5, 0.3, 1e-3, lam1, 0.0], 1.0, 1e-3, 'unmasked')
        return v[0], v[4], v[17] + v[18], v[5] + v[12] + (v[15] if perturb else 0.0)

    def _loss_GoodBadGAN(self, D, C, Y, Lambda):
        """train_base.py:186-240 -> (d_loss, gG_loss, bG_loss, c_loss)."""
        return self._good_bad(D, C, Y, Lambda, False)

    def _loss_GoodRegBadGAN(self, D, C, Y, Lambda):
        """train_base.py:519-574 -> (d_loss, gG_loss, bG_loss, c_loss)."""
        return self._good_bad(D, C, Y, Lambda, True)

    def _good_reg(self, D, C, Y, Lambda, variant):
        y_g, y_l_c = Y
        c_rep = None
        if variant == 'plain':
            c_real, c_unl, _cud, c_gfake, c_bfake, c_pert, _fr, f_unl, f_bfake, _fp = C
        elif variant == 'cifar10':
            c_real, c_unl, _cud, c_gfake, c_bfake, c_pert, _fr, f_unl, f_bfake, _fp, c_rep = C
        elif variant == 'BS':
            c_real, c_unl, _cud, c_gfake, c_bfake, c_pert, _cub, _fr, _fu, f_bfake, _fp, f_unl = C
        else:
            c_real, c_unl, _cud, c_gfake, c_bfake, c_pert, _cub, _fr, _fu, f_bfake, _fp, f_unl, _c_rep = C
        fast = variant == 'BS_cifar10' and getattr(getattr(self, 'config', None), 'FAST_MODE', False)
        w_h = {'plain': 0.3, 'cifar10': 0.3, 'BS': 1e-5, 'BS_cifar10': 1e-7}[variant]
        lam = [float(x) for x in Lambda]
        l1, l2, l3 = lam[:3]
        l4 = lam[3] if len(lam) > 3 else 0.0
        w6 = [1.0, l2 * 0.01 * 0.5, l2 * w_h, l2 * 1e-3, l2 * l1, l4 if variant == 'cifar10' else 0.0]
        v = self._variant_terms(D, c_real, c_unl, c_rep if variant == 'cifar10' else None, None if fast else c_gfake, c_bfake, c_pert, f_bfake,
                                f_unl, y_l_c, y_g, w6, l3, l3 * 1e-3, 'masked' if variant in ('plain', 'cifar10') else None)
        t_real, t_unl, t_h, t_bal, t_gf, t_mse = v[6:12]
        confid, unl, bal = w_h * t_h, 0.01 * 0.5 * t_unl, 1e-3 * t_bal
        c_gG = confid + unl + l1 * t_gf + bal
        pert = 1e-3 * v[16]
        c_bG = v[13] + v[14] + pert
        c_list = [v[5] + v[12] + v[15], t_real, c_gG, confid, unl, bal, t_gf, c_bG, v[13], v[14], pert]
        if variant == 'cifar10':
            c_list.append(l4 * t_mse)
        elif variant == 'BS_cifar10':
            c_list.append(l4)                                        # train_base.py:503: the constant lambda_4
            c_list[0] += l4
        return [v[0], v[1], v[2], v[3]], v[4], v[17] + (v[18] if variant in ('plain', 'cifar10') else 0.0), c_list

    def _loss_GoodRegGAN(self, D, C, Y, Lambda):
        """train_base.py:242-305."""
        return self._good_reg(D, C, Y, Lambda, 'plain')

    def _loss_GoodRegGAN_cifar10(self, D, C, Y, Lambda):
        """train_base.py:307-374."""
        return self._good_reg(D, C, Y, Lambda, 'cifar10')

    def _loss_GoodRegGAN_BS(self, D, C, Y, Lambda):
        """train_base.py:376-442."""
        return self._good_reg(D, C, Y, Lambda, 'BS')

    def _loss_GoodRegGAN_BS_cifar10(self, D, C, Y, Lambda):
        """train_base.py:444-515 (config.FAST_MODE drops the generated-sample cross-entropy)."""
        return self._good_reg(D, C, Y, Lambda, 'BS_cifar10')
