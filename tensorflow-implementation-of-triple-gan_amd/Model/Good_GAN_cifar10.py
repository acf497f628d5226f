"""Good_GAN_cifar10 — MI355X-native counterpart of the reference's Model/Good_GAN_cifar10.py.

Same class, method names and call protocol (`Model(config)`, `.good_generator(z, y)`,
`.discriminator(image, y)`, `.classifier(inp, is_training)`, `.good_sampler(z, y)`,
`.forward_pass(...)`, `cifar10_ZCA(config).apply(image)`); every layer runs in the hand-written
gfx950 kernels of csrc/ through Model/nn.py and Model/model_base.py.  Line references are to the
reference file.  Differences forced by eager execution:
  * tensors are tg.runtime.Act handles, `is_training` is a Python bool;
  * variables are created (with the reference's initialisers) when the Model
    is constructed — the reference's throw-away `init=True` builds (:216,227,263) only create variables;
  * `segments=` (extension) batches several applications of C or D into one call: identical maths,
    since D has no batch statistics and C's mean-only BN is computed per segment.
"""
import os

import numpy as np

from Model import model_base
from Model import nn
from tg import ops
from tg.runtime import ParamStore, ctx


def _he_trunc_normal(rng, shape):
    """variance_scaling_initializer(): factor 2, FAN_IN, truncated normal, std sqrt(1.3*2/fan_in) (:8-9)."""
    fan_in = shape[-2] * int(np.prod(shape[:-2])) if len(shape) > 2 else shape[0]
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2
    return (x * np.sqrt(1.3 * 2.0 / fan_in)).astype(np.float32)


# layer tables (class attributes so that a deeper variant only lists more rows, Model/Good_GAN_stress64.py)
C_CONVS = [  # name, filters, padding, max-pool + dropout after       (:106-149)
    ('conv1_1', 128, 'SAME', False), ('conv1_2', 128, 'SAME', False), ('conv1_3', 128, 'SAME', True),
    ('conv2_1', 256, 'SAME', False), ('conv2_2', 256, 'SAME', False), ('conv2_3', 256, 'SAME', True), ('conv3', 512, 'VALID', False)]
D_CONVS = [  # name, filters, stride, dropout after                 (:66-91)
    ('conv2d_00', 32, 1, False), ('conv2d_01', 32, 2, True), ('conv2d_10', 64, 1, False), ('conv2d_11', 64, 2, True),
    ('conv2d_20', 128, 1, False), ('conv2d_21', 128, 1, False)]
G_DECONVS = [('gg_dconv0', 256), ('gg_dconv1', 128), ('gg_dconv2', 3)]      # (:44-57); the last one is the tanh image layer


class Good_GAN_cifar10(model_base.NN_Base):
    C_CONVS, D_CONVS, G_DECONVS = C_CONVS, D_CONVS, G_DECONVS
    # Data-parallel gradient buckets: per network, the first variable (creation order) of every bucket after the first.
    # A bucket is one contiguous slice of the network's flat gradient buffer; the forward pass marks the matching boundary right before
    # the layer that owns the variable (Context.grad_bucket_boundary), so that the backward pass — run bucket by bucket — has the slice
    # from that variable on final when it comes back to the mark, and its all-reduce runs on the exchange stream beside the rest:
    #   classifier     [conv2_1 ... output_dense] (87 % of 12.5 MB) beside the backward pass of the first block, then [conv1_1 ... conv1_3]
    #   generator      [gg_dconv0 ... gg_dconv2] (17 of 20.5 MB; the 13 MB gg_dconv0 filter gradient is final here) beside bn0 / gg_h0_lin
    #   discriminator  one bucket per resolution stage: [conv2d_20 ... lin], [conv2d_10, conv2d_11], [conv2d_00, conv2d_01]
    GRAD_BUCKETS = {'classifier': ['classifier/conv2_1/V'],
                    'good_generator': ['good_generator/gg_dconv0/gg_dconv0/kernel'],
                    'discriminator': ['discriminator/conv2d_10/conv2d_10/kernel', 'discriminator/conv2d_20/conv2d_20/kernel']}

    def __init__(self, config):
        super(Good_GAN_cifar10, self).__init__(config.BATCH_NORM_DECAY, config.BATCH_NORM_EPSILON)
        self.config = config
        self._create_variables(getattr(config, 'SEED', 0))
        self._zca = None

    # ------------------------------------------------------------------ variables
    @classmethod
    def param_specs(cls, z_dim=100, ncls=10):
        """(network, name, shape, trainable, init) in TF creation order."""
        g, d, c = [], [], []
        C_CONVS, D_CONVS, G_DECONVS = cls.C_CONVS, cls.D_CONVS, cls.G_DECONVS
        p = 'good_generator/'
        g += [(p + 'gg_h0_lin/gg_h0_lin/kernel', (z_dim + ncls, 8192), True, 'he'), (p + 'gg_h0_lin/gg_h0_lin/bias', (8192,), True, 0.)]
        cin = 512
        for i, (name, cout) in enumerate([('bn0', 8192)] + G_DECONVS):
            if i > 0:
                g += [(p + '%s/%s/kernel' % (name, name), (5, 5, cout, cin + ncls), True, 'he'),
                      (p + '%s/%s/bias' % (name, name), (cout,), True, 0.)]
                cin = cout
            if i < len(G_DECONVS):
                bn = p + 'gg_bn%d/' % i
                g += [(bn + 'beta', (cout,), True, 0.), (bn + 'gamma', (cout,), True, 1.),
                      (bn + 'moving_mean', (cout,), False, 0.), (bn + 'moving_variance', (cout,), False, 1.)]
        cin = 3
        for name, cout, _, _ in D_CONVS:
            q = 'discriminator/%s/%s/' % (name, name)
            d += [(q + 'kernel', (3, 3, cin + ncls, cout), True, 'he'), (q + 'bias', (cout,), True, 0.)]
            cin = cout
        d += [('discriminator/lin/lin/kernel', (cin + ncls, 1), True, 'he'), ('discriminator/lin/lin/bias', (1,), True, 0.)]
        cin = 3
        layers = [('classifier/%s/' % n, (3, 3, cin_, co)) for (n, co, _, _), cin_ in
                  zip(C_CONVS, [3] + [co for _, co, _, _ in C_CONVS[:-1]])]
        layers += [('classifier/NiN1/NiN1/', (512, 256)), ('classifier/NiN2/NiN2/', (256, 128)), ('classifier/output_dense/', (128, ncls))]
        for q, shape in layers:
            c += [(q + 'V', shape, True, 'n05'), (q + 'b', (shape[-1],), True, 0.),
                  (q + 'meanOnlyBatchNormalization/pop_mean', (shape[-1],), False, 0.), (q + 'g', (shape[-1],), True, 1.)]
        return {'good_generator': g, 'discriminator': d, 'classifier': c}

    def _create_variables(self, seed):
        cx = ctx()
        rng = np.random.default_rng(seed)
        specs = self.param_specs(self.config.Z_DIM, self.config.NUM_CLASSES)
        for net in ('good_generator', 'discriminator', 'classifier'):
            if net in cx.stores:
                continue                                  # reuse=True
            st = ParamStore(net, [(n, s, t) for n, s, t, _ in specs[net]], cx.device)
            for name, shape, _, init in specs[net]:
                if init == 'he':
                    st.set(name, _he_trunc_normal(rng, shape))
                elif init == 'n05':
                    st.set(name, (rng.standard_normal(shape) * 0.05).astype(np.float32))   # nn.py:478
                else:
                    st.set(name, np.full(shape, init, np.float32))
            cx.stores[net] = st
        cx.stores['classifier'].enable_ema()              # Train_goodGAN.py:101-103

    def _bucket_mark(self, net, first_variable):
        """a gradient-bucket boundary in front of the layer that owns `first_variable` (GRAD_BUCKETS)."""
        if first_variable in self.GRAD_BUCKETS.get(net, ()):
            ctx().grad_bucket_boundary(net)

    # ------------------------------------------------------------------ activations
    def leakyReLu(self, x, alpha=0.2, name=None):
        """relu(x) - alpha*relu(-x) (:19-27).  Called on a tensor it is one elementwise launch; passed as `nonlinearity=` /
        `activation=` (what the networks below do) it is fused into the producing kernel's epilogue with the default slope."""
        return self._leakyReLu_impl(x, alpha)
    leakyReLu.tg_act = ('lrelu', 0.2)

    def _leakyReLu_impl(self, x, alpha):
        return ops.activation(x, 'lrelu', alpha)

    def gaussian_noise_layer(self, input_layer, std):
        """input_layer + N(0, std) (:29-31)."""
        return self._add_noise(input_layer, stddev=std)

    # ------------------------------------------------------------------ networks
    def good_generator(self, z, y, init=False, reuse=False):
        """:33-58.  z: Act [N,Z_DIM]; y: Act [N,NUM_CLASSES] one-hot.  Returns Act [N,32,32,3]."""
        cx = ctx()
        with cx.variable_scope('good_generator'):
            zy = ops.cond_concat(z, y.t, y.c)                                        # tf.concat([z, y], 1)
            h0 = self._linear_fc(zy, 4 * 4 * 512, 'gg_h0_lin', activation=self._relu)   # dense + relu (gg_rl0)
            h0 = ops.reshape(self._batch_norm_contrib(_dense_view(h0), 'gg_bn0', train=True), z.n, 4, 4, 512)
            h = h0
            for i, (name, cout) in enumerate(self.G_DECONVS):                        # [8,8], [16,16], [32,32]
                self._bucket_mark('good_generator', 'good_generator/%s/%s/kernel' % (name, name))
                h = self._conv_cond_concat(h, y)
                if i + 1 < len(self.G_DECONVS):
                    h = self._deconv2d(h, cout, k_w=5, k_h=5, d_w=2, d_h=2, name=name, activation=self._relu)
                    h = self._batch_norm_contrib(h, 'gg_bn%d' % (i + 1), train=True)
                else:
                    h = self._deconv2d(h, cout, k_w=5, k_h=5, d_w=2, d_h=2, name=name, activation=self._tanh, narrow=True)
            h2 = h
        return h2

    def good_sampler(self, z, y):
        """:176-202 — the same graph as good_generator with reuse=True (BN stays in training mode)."""
        return self.good_generator(z, y, reuse=True)

    def discriminator(self, image, y, init=False, reuse=False, getter=None, want_prob=True):
        """:60-99.  image: Act [N,32,32,3]; y: Act [N,10].  Returns (tf.nn.sigmoid(h3), h3) = (Act [N,1], logits Act [N,1]).
        Dropout 0.2 is ALWAYS active (training=True literal, :63,73,83).  want_prob=False (extension; the trainer's solver runs, which
        fetch only the losses): the sigmoid launch is skipped and None returned in its place."""
        cx = ctx()
        lre = self.leakyReLu
        with cx.variable_scope('discriminator'):
            image = self._drop_out(image, 0.2, True, fuse_next=True)
            h2 = image
            for i, (name, cout, stride, drop) in enumerate(self.D_CONVS):
                self._bucket_mark('discriminator', 'discriminator/%s/%s/kernel' % (name, name))
                # a layer whose output goes straight into the next _conv_cond_concat (no dropout in between) writes that concatenation itself
                cat = y if (not drop and i + 1 < len(self.D_CONVS)) else None
                h2 = self._conv2d(self._conv_cond_concat(h2, y), cout, k_h=3, k_w=3, d_h=stride, d_w=stride, name=name, activation=lre,
                                  then_concat=cat)
                if drop:
                    h2 = self._drop_out(_dense_view(h2), 0.2, True, fuse_next=True)
            assert h2.pending is None, "a dropout behind the last convolution has no concat to fuse into"
            h3 = ops.global_avgpool_concat(h2, y.t, y.c)                             # avg_pool 8 + squeeze + concat y
            h3 = self._linear_fc(h3, 1, 'lin', narrow=True)
        return (self._sigmoid_no_grad(h3) if want_prob else None), h3

    def _sigmoid_no_grad(self, logits):
        """tf.nn.sigmoid(logits) as an output: no loss of the reference differentiates through it (they all take the logits)."""
        with ctx().no_record():
            return ops.activation(logits, 'sigmoid')

    def classifier(self, inp, is_training, init=False, reuse=False, getter=None, segments=None):
        """:101-174.  inp: Act [N,32,32,3] (ZCA-whitened).  Returns (logits Act [N,10], feature Act [N,128]).
        The Gaussian input noise is always on (also at evaluation, :104); dropout only when training."""
        cx = ctx()
        kw = dict(init=init, use_weight_normalization=True, use_batch_normalization=False,
                  use_mean_only_batch_normalization=True, deterministic=not is_training, nonlinearity=self.leakyReLu,
                  segments=segments)
        with cx.variable_scope('classifier'):
            # x = self._add_noise(x, stddev=0.15) followed by conv1_1 (:104-110): with 3 input channels the 3x3 window is
            # gathered once (x + noise -> [N,32,32,27]) and conv1_1 runs as a 1x1 product on it with the same variable V
            # ([3,3,3,128] and [1,1,27,128] are the same bytes)
            noise = cx.rng.normal(cx, 'noise', inp.rows * inp.c, 0.15)
            x = ops.im2col3x3_add(inp, noise)
            for i, (name, cout, pad, pool) in enumerate(self.C_CONVS):
                self._bucket_mark('classifier', 'classifier/%s/V' % name)
                then_pool = None
                if pool:                                                             # max_pool_k + dropout_k (:123-124,142-143): in the layer's apply launch
                    mask = None
                    if is_training:
                        mask = cx.rng.keep_mask(cx, 'drop' + name[4], x.rows // 4 * cout, 0.5)
                    then_pool = (mask, 2.0)
                x = nn.conv2d_WN(x, num_filters=cout, name=name, pad=pad, filter_size=[1, 1] if i == 0 else [3, 3], then_pool=then_pool, **kw)
            x = nn.NiN_WN(x, num_units=256, name='NiN1', **kw)
            x = nn.NiN_WN(x, num_units=128, name='NiN2', **kw)
            x = ops.global_maxpool(x)                                                # tf.layers.max_pooling2d(pool 6) named avg_pool_0
            intermediate_layer = x
            kw['nonlinearity'] = None
            logits = nn.dense_WN(x, num_units=self.config.NUM_CLASSES, name='output_dense', **kw)
        return logits, intermediate_layer

    # ------------------------------------------------------------------ whole graph (evaluation / tests)
    CONSISTENCY = True        # forward_pass returns C_unl_logits_rep and _loss_GAN adds lambda_2 * MSE (:232-235, train_base.py:118)

    def as_image(self, a):
        return a

    def zca(self):
        if self._zca is None:
            self._zca = cifar10_ZCA(self.config)
        return self._zca

    def forward_pass(self, z_g, y_g, x_l_c, y_l_c, x_l_d, y_l_d, x_u_d, x_u_c, train):
        """:204-278.  Executes every application eagerly (the trainer runs per-solver sub-graphs instead,
        Training/Train_goodGAN.py).  Returns [G, [D_real, D_real_logits, D_fake, D_fake_logits, D_unl, D_unl_logits],
        [C_real_logits, C_unl_logits, C_unl_d_logits, C_fake_logits, C_unl_logits_rep]] as Act handles; the three discriminator
        applications run as one batched call (no batch statistics in D: identical arithmetic), the five classifier ones as one call
        with per-application mean-only-BN statistics and pop_mean updates in the reference's call-site order (real, unl, unl_rep, unl_d,
        fake: :228-240)."""
        from tg.batching import concat_acts
        cx = ctx()
        G = self.good_generator(z_g, y_g)
        w = self.zca()
        segs = [x_l_c.n, x_u_c.n, x_u_c.n, x_u_d.n, G.n]
        xc = concat_acts([w.apply(x_l_c), w.apply(x_u_c), w.apply(x_u_c), w.apply(x_u_d), w.apply(G)])
        with cx.rng_scoped(cx.phase + '/C'):
            logits, _ = self.classifier(xc, train, segments=segs)
        offs = [int(v) for v in np.cumsum([0] + segs)]
        C_real, C_unl, C_rep, C_unl_d, C_fake = [logits.view_rows(offs[i], offs[i + 1]) for i in range(5)]
        oh_d = _onehot_act(C_unl_d, self.config.NUM_CLASSES)
        oh_u = _onehot_act(C_unl, self.config.NUM_CLASSES)
        ximg = concat_acts([x_l_d, x_u_d, G, x_u_c])
        yall = concat_acts([y_l_d, oh_d, y_g, oh_u])
        with cx.rng_scoped(cx.phase + '/D'):
            dp, dl = self.discriminator(ximg, yall)
        n_p = x_l_d.n + x_u_d.n
        cut = lambda a: [a.view_rows(0, n_p), a.view_rows(n_p, n_p + G.n), a.view_rows(n_p + G.n, a.n)]
        (p_real, p_fake, p_unl), (l_real, l_fake, l_unl) = cut(dp), cut(dl)
        return [G, [p_real, l_real, p_fake, l_fake, p_unl, l_unl], [C_real, C_unl, C_unl_d, C_fake, C_rep]]


def _dense_view(a):
    """an Act whose channel stride equals its channel count (true for every multiple-of-32 layer here)."""
    assert a.ld == a.c, "layer width must be a multiple of 32 here"
    return a


def _onehot_act(logits, k):
    from tg.runtime import Act
    return Act(ops.argmax_onehot(logits, k), logits.n, 1, 1, k, k)


class cifar10_ZCA():
    """:287-299: (flatten(x) - mean) @ mat.  Constants come from DATA_DIR/cifar10_zca_{mean,mat}.npy as in the
    reference; when the files are absent (they are not part of the reference repository) `config.ZCA` may
    supply (mean, mat) arrays, e.g. the synthetic orthogonal matrix of SURVEY §8d."""

    def __init__(self, config):
        cx = ctx()
        zc = getattr(config, 'ZCA', None)
        if zc is None:
            m = np.load(os.path.join(config.DATA_DIR, "cifar10_zca_mean.npy"))
            mat = np.load(os.path.join(config.DATA_DIR, "cifar10_zca_mat.npy"))
        else:
            m, mat = zc
        m = np.asarray(m, np.float32).reshape(-1)
        mat = np.asarray(mat, np.float32)
        import torch
        self.dim = mat.shape[0]
        # (x - mean) @ mat = x @ mat + (-mean @ mat): weights as Wt[n][k] for the MFMA kernel, bias folded
        self.wt = torch.from_numpy(np.ascontiguousarray(mat.T)).to(cx.device).reshape(-1)
        self.bias = torch.from_numpy((-(m.astype(np.float64) @ mat.astype(np.float64))).astype(np.float32)).to(cx.device)

    def apply(self, image):
        """image: Act [N,32,32,3] (ld 3) -> Act of the same shape; forward only (no gradient ever flows through ZCA)."""
        from tg import geom, lib
        cx = ctx()
        assert image.ld == image.c and image.h * image.w * image.c == self.dim
        out = cx.new_act(image.n, image.h, image.w, image.c, image.c)
        splits = 4 if self.dim % 128 == 0 and image.n <= 1024 and os.environ.get('TG_ZCA_SPLITK', '1') != '0' else 1
        if splits > 1:
            # few rows, long reduction (3072): four K-ranges as sub-problems of one launch, then one add-up pass
            import ctypes as C
            part = cx.scratch('zcap', image.n * splits * self.dim)
            dds = lib.desc_array(geom.dense_fwd_splitk(image.n, self.dim, self.dim, splits))
            lib.call('tg_igemm_multi_f32', dds, len(dds), image.ptr, lib.ptr(self.wt), None, lib.ptr(part), None, 0, cx.stream)
            lib.call('tg_splitk_reduce_f32', lib.ptr(part), lib.ptr(self.bias), out.ptr, self.dim, image.n, splits, self.dim, cx.stream)
        else:
            d = geom.dense_fwd(image.n, self.dim, self.dim)
            lib.call('tg_igemm_f32', d, image.ptr, lib.ptr(self.wt), lib.ptr(self.bias), out.ptr, None, 0, cx.stream)
        return out
