"""Good_GAN — MI355X-native counterpart of the reference's Model/Good_GAN.py (MNIST and SVHN configs; its unused
cifar10 branches are the svhn ones verbatim and are served by the same code).

Same class and method protocol (`Model(config)`, `.good_generator(z, y)`, `.discriminator(image, y)`,
`.classifier(image, train_ph)`, `.good_sampler(z, y)`, `.forward_pass(...)`); line references are to the reference
file.  Layers run in the gfx950 kernels of csrc/ through Model/model_base.py.

Eager-execution notes: tensors are tg.runtime.Act handles, `train_ph` is a Python bool; variables are
created in the constructor with the initialisers the reference passes — including its quirk that
`tf.random_normal_initializer(0.02)` / `tf.truncated_normal_initializer(0.02)` set the MEAN to 0.02 with stddev 1.0
(Model/modle_base.py:28,159,248); nonlinearities are fused into the producing kernel.
`segments=` (extension): the classifier uses full batch norm, whose statistics are per application, so a batched
call runs the applications one after the other and concatenates the logits.
"""
import numpy as np

from Model import model_base
from tg import ops
from tg.runtime import Act, ParamStore, ctx


def _trunc_normal(rng, shape):
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2
    return x


class Good_GAN(model_base.NN_Base):
    def __init__(self, config):
        super(Good_GAN, self).__init__(config.BATCH_NORM_DECAY, config.BATCH_NORM_EPSILON)
        self.config = config
        if config.DATA_NAME not in ('mnist', 'svhn', 'cifar10'):
            raise ValueError("The specified dataset is not yet implemented!")
        self.mnist = config.DATA_NAME == 'mnist'
        self._create_variables(getattr(config, 'SEED', 0))

    # ------------------------------------------------------------------ variables
    def param_specs(self):
        """{network: [(name, shape, trainable, init)]} in TF creation order."""
        k, zd = self.config.NUM_CLASSES, self.config.Z_DIM
        g, d, c = [], [], []

        def dense(L, p, cin, cout):
            L += [(p + '/kernel', (cin, cout), True, 'n02'), (p + '/bias', (cout,), True, 0.)]

        def wn(L, p, shape):
            L += [(p + '/V', shape, True, 'n05'), (p + '/g', (shape[-1] if len(shape) != 4 or shape[0] == 3 else shape[2],), True, 1.),
                  (p + '/b', (shape[-1] if len(shape) != 4 or shape[0] == 3 else shape[2],), True, 0.)]

        def bn(L, p, ch):
            L += [(p + '/beta', (ch,), True, 0.), (p + '/gamma', (ch,), True, 1.), (p + '/moving_mean', (ch,), False, 0.),
                  (p + '/moving_variance', (ch,), False, 1.)]

        G, D, C = 'good_generator/', 'discriminator/', 'classifier/'
        if self.mnist:
            dense(g, G + 'gg_h0_lin/gg_h0_lin', zd + k, 500); bn(g, G + 'gg_bn0', 500)
            dense(g, G + 'gg_h1_lin/gg_h1_lin', 500 + k, 500); bn(g, G + 'gg_bn1', 500)
            wn(g, G + 'gg_h2_lin', (500 + k, 784))
            cin = 784
            for i, n in enumerate((1000, 500, 250, 250, 250, 1)):
                wn(d, D + 'd_h%d_wndense0' % i, (cin + k, n))
                cin = n
            cin = 1
            for name, bname, cout in (('c_h0_conv0', 'c_h0_bn0', 32), ('c_h1_conv0', 'c_h1_bn0', 64), ('c_h1_conv1', 'c_h1_bn1', 64),
                                      ('c_h2_conv0', 'c_h2_bn0', 128), ('c_h2_conv1', 'c_h2_bn1', 128)):
                c += [(C + '%s/%s/kernel' % (name, name), (3, 3, cin, cout), True, 'tn02'), (C + '%s/%s/bias' % (name, name), (cout,), True, 0.)]
                bn(c, C + bname, cout)
                cin = cout
        else:
            dense(g, G + 'gg_h0_lin/gg_h0_lin', zd + k, 8192); bn(g, G + 'gg_bn0', 512)
            cin = 512
            for i, cout in enumerate((256, 128)):
                p = G + 'gg_dconv%d/gg_dconv%d' % (i, i)
                g += [(p + '/kernel', (5, 5, cout, cin + k), True, 'n02'), (p + '/bias', (cout,), True, 0.)]
                bn(g, G + 'gg_bn%d' % (i + 1), cout)
                cin = cout
            wn(g, G + 'gg_wndconv0', (5, 5, 3, cin + k))
            cin = 3
            for name, cout, extra in (('d_h0_wnconv0', 32, k), ('d_h0_wnconv1', 32, k), ('d_h1_wnconv0', 64, k), ('d_h1_wnconv1', 64, k),
                                      ('d_h2_wnconv0', 128, k), ('d_h2_wnconv1', 128, 2 * k)):
                wn(d, D + name, (3, 3, cin + extra, cout))
                cin = cout
            if getattr(self.config, 'MINIBATCH_DIS', False):                              # Good_GAN.py:159-162
                d += [(D + 'w', (cin + k, 100 * 5), True, 'xavier'), (D + 'b', (100,), True, 0.)]
                dense(d, D + 'd_h3_lin/d_h3_lin', cin + k + 100, 1)
            else:
                wn(d, D + 'd_h3_wndense', (cin + k, 1))
            cin = 3
            for name, bname, cout in (('c_h0_conv0', 'c_h0_bn0', 128), ('c_h0_conv1', 'c_h0_bn1', 128), ('c_h0_conv2', 'c_h0_bn2', 128),
                                      ('c_h1_conv0', 'c_h1_bn0', 256), ('c_h1_conv1', 'c_h1_bn1', 256), ('c_h1_conv2', 'c_h1_bn2', 256),
                                      ('c_h2_conv0', 'c_h2_bn0', 512)):
                c += [(C + '%s/%s/kernel' % (name, name), (3, 3, cin, cout), True, 'tn02'), (C + '%s/%s/bias' % (name, name), (cout,), True, 0.)]
                bn(c, C + bname, cout)
                cin = cout
            for name, bname, cout in (('c_h2_nin0', 'c_h2_bn1', 256), ('c_h2_nin1', 'c_h2_bn2', 128)):
                wn(c, C + name, (cin, cout))
                bn(c, C + bname, cout)
                cin = cout
        dense(c, C + 'c_h2_lin/c_h2_lin', cin, k)
        bn(c, C + 'c_h3_bn0', k)
        return {'good_generator': g, 'discriminator': d, 'classifier': c}

    def _create_variables(self, seed):
        cx = ctx()
        rng = np.random.default_rng(seed)
        for net, specs in self.param_specs().items():
            if net in cx.stores:
                continue
            st = ParamStore(net, [(n, s, t) for n, s, t, _ in specs], cx.device)
            for name, shape, _, init in specs:
                if init == 'n02':
                    st.set(name, 0.02 + rng.standard_normal(shape))
                elif init == 'tn02':
                    st.set(name, 0.02 + _trunc_normal(rng, shape))
                elif init == 'n05':
                    st.set(name, 0.05 * rng.standard_normal(shape))
                elif init == 'xavier':                                   # tf.contrib.layers.xavier_initializer(): uniform, fan_avg
                    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
                    st.set(name, rng.uniform(-lim, lim, shape))
                else:
                    st.set(name, np.full(shape, init, np.float32))
            cx.stores[net] = st
        cx.stores['classifier'].enable_ema()

    # ------------------------------------------------------------------ helpers
    def as_image(self, a):
        """common per-image layout for batch concatenation: MNIST images travel flattened ([N,784], as the generator emits them)."""
        if self.mnist and (a.h, a.w) != (1, 1):
            return ops.view(a, 1, 1, a.h * a.w * a.c)
        return a

    def zca(self):
        return None

    # ------------------------------------------------------------------ networks
    def good_generator(self, z, y, reuse=False):
        """:15-83."""
        cx = ctx()
        with cx.variable_scope('good_generator'):
            zy = ops.cond_concat(z, y.t, y.c)
            if self.mnist:                                                                     # :19-33
                h0 = self._linear_fc(zy, 500, 'gg_h0_lin', activation=self._softplus)
                h0 = self._batch_norm_contrib(h0, 'gg_bn0', train=True)
                h1 = self._linear_fc(ops.cond_concat(h0, y.t, y.c), 500, 'gg_h1_lin', activation=self._softplus)
                h1 = self._batch_norm_contrib(h1, 'gg_bn1', train=True)
                return self._WN_dense(ops.cond_concat(h1, y.t, y.c), 28 * 28, 'gg_h2_lin', activation=self._sigmoid, narrow=True)
            h0 = self._linear_fc(zy, 4 * 4 * 512, 'gg_h0_lin', activation=self._relu)           # relu commutes with the reshape (:40-42)
            h0 = self._batch_norm_contrib(ops.reshape(h0, z.n, 4, 4, 512), 'gg_bn0', train=True)
            h0 = self._deconv2d(self._conv_cond_concat(h0, y), 256, k_w=5, k_h=5, d_w=2, d_h=2, name='gg_dconv0', activation=self._relu)
            h0 = self._batch_norm_contrib(h0, 'gg_bn1', train=True)
            h1 = self._deconv2d(self._conv_cond_concat(h0, y), 128, k_w=5, k_h=5, d_w=2, d_h=2, name='gg_dconv1', activation=self._relu)
            h1 = self._batch_norm_contrib(h1, 'gg_bn2', train=True)
            return self._WN_deconv2d(self._conv_cond_concat(h1, y), 3, k_w=5, k_h=5, d_w=2, d_h=2, init_scale=0.1, init=False,
                                     name='gg_wndconv0', activation=self._tanh, narrow=True)

    def good_sampler(self, z, y, reuse=True):
        """:356-426 — the generator graph with reuse."""
        return self.good_generator(z, y, reuse=True)

    def _d_out(self, logits, want_prob):
        """(tf.nn.sigmoid(logits), logits) (:124,206); no loss differentiates through the sigmoid."""
        if not want_prob:
            return None, logits
        with ctx().no_record():
            return ops.activation(logits, 'sigmoid'), logits

    def discriminator(self, image, y, reuse=False, want_prob=True):
        """:89-206.  Returns (sigmoid(logits), logits [N,1]); want_prob=False (extension, the trainer's solver runs): (None, logits)."""
        cx = ctx()
        lre = self._leaky_relu
        with cx.variable_scope('discriminator'):
            if self.mnist:                                                                     # :93-124
                h = self._add_noise(self.as_image(image), stddev=0.2)
                for i in range(5):
                    h = self._WN_dense(ops.cond_concat(h, y.t, y.c), (1000, 500, 250, 250, 250)[i], 'd_h%d_wndense0' % i, init=False, activation=lre)
                    h = self._add_noise(h, stddev=0.2)
                return self._d_out(self._WN_dense(ops.cond_concat(h, y.t, y.c), 1, 'd_h5_wndense0', init=False, narrow=True), want_prob)
            image = self._drop_out(image, 0.2, True, fuse_next=True)                           # :126-165
            # (layers whose output goes straight into the next concat write that concatenation themselves: then_concat, ops.conv2d(concat=...))
            h0 = self._WN_conv2d(self._conv_cond_concat(image, y), 32, k_h=3, k_w=3, d_h=1, d_w=1, init=False, name="d_h0_wnconv0", activation=lre,
                                 then_concat=y)
            h0 = self._WN_conv2d(self._conv_cond_concat(h0, y), 32, k_h=3, k_w=3, d_h=2, d_w=2, init=False, name="d_h0_wnconv1", activation=lre)
            h0 = self._drop_out(h0, 0.2, True, fuse_next=True)
            h1 = self._WN_conv2d(self._conv_cond_concat(h0, y), 64, k_h=3, k_w=3, d_h=1, d_w=1, init=False, name="d_h1_wnconv0", activation=lre,
                                 then_concat=y)
            h1 = self._WN_conv2d(self._conv_cond_concat(h1, y), 64, k_h=3, k_w=3, d_h=2, d_w=2, init=False, name="d_h1_wnconv1", activation=lre)
            h1 = self._drop_out(h1, 0.2, True, fuse_next=True)
            y2 = _twice(y)
            h2 = self._WN_conv2d(self._conv_cond_concat(h1, y), 128, k_h=3, k_w=3, d_h=1, d_w=1, init=False, name="d_h2_wnconv0", activation=lre,
                                 then_concat=(y2, 2 * y.c))
            h2 = ops.cond_concat(h2, y2, 2 * y.c)                                               # y is concatenated twice (:151-153)
            h2 = self._WN_conv2d(h2, 128, k_h=3, k_w=3, d_h=1, d_w=1, init=False, name="d_h2_wnconv1", activation=lre)
            h3 = ops.global_avgpool_concat(h2, y.t, y.c)                                        # reduce_mean + concat y
            if self.config.MINIBATCH_DIS:                                                      # :159-162 (off in every config of the reference)
                h3 = self._minibatch_discrimination(h3, 100, concat_input=True)               # f = ...; h3 = tf.concat([h3, f], 1)
                return self._d_out(self._linear_fc(h3, 1, 'd_h3_lin', narrow=True), want_prob)
            return self._d_out(self._WN_dense(h3, 1, 'd_h3_wndense', narrow=True), want_prob)

    def classifier(self, image, train_ph, reuse=False, segments=None):
        """:212-350.  Returns (logits [N,10], feature).  `segments` (extension): image counts of the applications batched into
        `image` — the convolutions run once over the whole batch, every batch norm keeps per-application statistics and
        updates its moving statistics application by application (tg_bn_train_f32)."""
        cx = ctx()
        lre = self._leaky_relu

        def cbr(x, cname, bname, cout, k=3):
            # conv -> leaky relu -> batch norm: in training the batch-norm statistics are taken in the convolution's epilogue
            x = self._conv2d(x, cout, k_h=k, k_w=k, d_h=1, d_w=1, name=cname, activation=lre,
                             bn_segments=(segments or [x.n]) if train_ph else None)
            return self._batch_norm_contrib(x, name=bname, train=train_ph, segments=segments)

        def pool_drop(x, key):
            mask = cx.rng.keep_mask(cx, key, x.rows // 4 * x.c, 0.5) if train_ph else None
            return ops.maxpool2_dropout(x, mask, 2.0)

        with cx.variable_scope('classifier'):
            if self.mnist:                                                                         # :216-247
                img = ops.view(image, 28, 28, 1) if (image.h, image.w) == (1, 1) else image
                noise = cx.rng.normal(cx, 'noise', img.rows * img.c, 0.3)
                x = ops.im2col3x3_add(img, noise)                    # _add_noise + the 1-channel 3x3 window gathered once
                x = cbr(x, 'c_h0_conv0', 'c_h0_bn0', 32, k=1)
                x = pool_drop(x, 'drop1')
                x = cbr(x, 'c_h1_conv0', 'c_h1_bn0', 64)
                x = cbr(x, 'c_h1_conv1', 'c_h1_bn1', 64)
                x = pool_drop(x, 'drop2')
                x = cbr(x, 'c_h2_conv0', 'c_h2_bn0', 128)
                x = cbr(x, 'c_h2_conv1', 'c_h2_bn1', 128)
            else:                                                                                  # :249-299
                image = self._drop_out(image, 0.2, train_ph, name='drop0')
                x = ops.im2col3x3_add(image, None)
                x = cbr(x, 'c_h0_conv0', 'c_h0_bn0', 128, k=1)
                x = cbr(x, 'c_h0_conv1', 'c_h0_bn1', 128)
                x = cbr(x, 'c_h0_conv2', 'c_h0_bn2', 128)
                x = pool_drop(x, 'drop1')
                x = cbr(x, 'c_h1_conv0', 'c_h1_bn0', 256)
                x = cbr(x, 'c_h1_conv1', 'c_h1_bn1', 256)
                x = cbr(x, 'c_h1_conv2', 'c_h1_bn2', 256)
                x = pool_drop(x, 'drop2')
                x = cbr(x, 'c_h2_conv0', 'c_h2_bn0', 512)
                x = self._batch_norm_contrib(self._nin(x, 256, name='c_h2_nin0', activation=lre), name='c_h2_bn1', train=train_ph, segments=segments)
                x = self._batch_norm_contrib(self._nin(x, 128, name='c_h2_nin1', activation=lre), name='c_h2_bn2', train=train_ph, segments=segments)
            fm = ops.global_avgpool(x)                                                             # tf.reduce_mean(axis=[1,2])
            h = self._linear_fc(fm, self.config.NUM_CLASSES, 'c_h2_lin')
            return self._batch_norm_contrib(h, name='c_h3_bn0', train=train_ph, segments=segments), fm

    def forward_pass(self, z_g, y_g, x_l_c, y_l_c, x_l_d, y_l_d, x_u_d, x_u_c, train):
        """:428-472 (evaluation / tests; the trainer runs per-solver sub-graphs)."""
        from tg.batching import concat_acts
        cx = ctx()
        k = self.config.NUM_CLASSES
        G = self.good_generator(z_g, y_g)
        parts = [self.as_image(a) for a in (x_l_c, x_u_c, x_u_d, G)]
        segs = [p.n for p in parts]
        with cx.rng_scoped(cx.phase + '/C'):
            logits, _ = self.classifier(concat_acts(parts), train, segments=segs)
        offs = [int(v) for v in np.cumsum([0] + segs)]
        C_real, C_unl, C_unl_d, C_fake = [logits.view_rows(offs[i], offs[i + 1]) for i in range(4)]
        oh_d = Act(ops.argmax_onehot(C_unl_d, k), C_unl_d.n, 1, 1, k, k)
        oh_u = Act(ops.argmax_onehot(C_unl, k), C_unl.n, 1, 1, k, k)
        ximg = concat_acts([self.as_image(a) for a in (x_l_d, x_u_d, G, x_u_c)])
        yall = concat_acts([y_l_d, oh_d, y_g, oh_u])
        with cx.rng_scoped(cx.phase + '/D'):
            dp, dl = self.discriminator(ximg, yall)
        n_p = x_l_d.n + x_u_d.n
        cut = lambda a: [a.view_rows(0, n_p), a.view_rows(n_p, n_p + G.n), a.view_rows(n_p + G.n, a.n)]
        (p_real, p_fake, p_unl), (l_real, l_fake, l_unl) = cut(dp), cut(dl)
        return [G, [p_real, l_real, p_fake, l_fake, p_unl, l_unl], [C_real, C_unl, C_unl_d, C_fake]]


def _twice(y):
    """device tensor [N, 2k] = [y, y] for the doubled cond-concat of the SVHN discriminator."""
    from tg.batching import concat_acts
    cx = ctx()
    out = cx.new_act(y.n, 1, 1, 2 * y.c, 2 * y.c, tag='yy')
    ops.copy2d(out.t, 2 * y.c, 0, y.t, y.c, y.n, y.c)
    ops.copy2d(out.t, 2 * y.c, y.c, y.t, y.c, y.n, y.c)
    return out.t
