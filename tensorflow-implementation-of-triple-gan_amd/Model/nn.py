"""Layer primitives (free functions) — MI355X-native counterpart of the reference's Model/nn.py.

Same names and keyword arguments as Model/nn.py:147-187 (mean_only_batch_norm_impl), :192-217 (batch_norm_impl), :220-340 (the
Salimans-style dense / conv2d / deconv2d / nin), :469-520 (conv2d_WN), :525-572 (dense_WN), :577-589 (NiN_WN); the arithmetic runs
in hand-written gfx950 kernels (csrc/) instead of TensorFlow ops.  Tensors are `tg.runtime.Act` handles (NHWC device buffers);
variables come from the active `tg.runtime.Context` under the scope names the reference's tf.variable_scope calls produce
('classifier/conv1_1/V', 'classifier/NiN1/NiN1/V') and are created on first use with the reference's initialisers
(Context.get_variable = tf.get_variable; the models create theirs up front, which is the reuse=True case).

Differences forced by eager execution: `deterministic` is a Python bool (not a tf.bool tensor): True = evaluation (use the
running statistics), False = training.  `init=True` (data-dependent initialisation) returns what the reference's init branch returns —
scale_init * (x - m_init), forward only; its g / b assigns are "created but never run" in the reference (nn.py:497-499) and do not
exist here.  `ema` of the Salimans layers (variable averages substituted at evaluation, nn.py:98-110) is accepted and must be None.
Extension: `segments` = image counts of the classifier applications batched into one call; mean-only-BN statistics are then per
application (the models' batching, Training/Train_goodGAN.py of this package).
"""
import numpy as np

from tg import ops
from tg.runtime import ctx


def int_shape(x):
    """Model/nn.py:12-13."""
    return [x.n, x.h, x.w, x.c] if (x.h, x.w) != (1, 1) else [x.n, x.c]


def get_name(layer_name, counters):
    """Model/nn.py:138-144: 'dense_0', 'dense_1', ... — a utility for keeping track of layer names."""
    if layer_name not in counters:
        counters[layer_name] = 0
    name = layer_name + '_' + str(counters[layer_name])
    counters[layer_name] += 1
    return name


def _normal(std, seed_key):
    def init(shape):
        cx = ctx()
        cx.init_rng = getattr(cx, 'init_rng', None) or np.random.default_rng(20190430)
        return (cx.init_rng.standard_normal(shape) * std).astype(np.float32)
    return init


def _tg_act(fn):
    """(name, alpha) when `fn` is one of the package's activations (fusable into a kernel epilogue), else None."""
    return getattr(fn, 'tg_act', None) if fn is not None else None


def _apply_nonlinearity(x, nonlinearity):
    if nonlinearity is None:
        return x
    a = _tg_act(nonlinearity)
    return ops.activation(x, a[0], a[1]) if a else nonlinearity(x)


def mean_only_batch_norm_impl(x, pop_mean, b, is_conv_out=True, deterministic=False, decay=0.9, name='meanOnlyBatchNormalization',
                              b_grad=None, segments=None):
    """Model/nn.py:147-187 as a stand-alone function: x - mean + b with the running mean updated (training) or x - pop_mean + b
    (deterministic).  pop_mean / b: device tensors (e.g. ctx().var(...)); is_conv_out only selects the reduction axes in the
    reference ([0,1,2] vs [0]) — here every row of the activation is one sample of the statistic either way.  The models use
    the version fused into the convolution (conv2d_WN / dense_WN)."""
    with ctx().variable_scope(name):
        return ops.mean_only_batch_norm(x, pop_mean, b, b_grad=b_grad, train=not deterministic, decay=decay, segments=segments)


def batch_norm_impl(x, is_conv_out=True, deterministic=False, decay=0.9, name='BatchNormalization'):
    """Model/nn.py:192-217: variables scale (1), beta (0), pop_mean (0), pop_var (1) under <scope>/<name>/; training: batch moments
    (biased variance), running statistics <- running*decay + batch*(1-decay), tf.nn.batch_normalization with epsilon 0.001;
    deterministic: the running statistics."""
    cx = ctx()
    with cx.variable_scope(name):
        c = (x.c,)
        scale = cx.get_variable('scale', c, 1.0)
        beta = cx.get_variable('beta', c, 0.0)
        pop_mean = cx.get_variable('pop_mean', c, 0.0, trainable=False)
        pop_var = cx.get_variable('pop_var', c, 1.0, trainable=False)
        tr = cx.trains()
        return ops.batch_norm_moments(x, scale, beta, pop_mean, pop_var, 0.001, decay, train=not deterministic,
                                      scale_grad=cx.var_grad('scale') if tr else None, beta_grad=cx.var_grad('beta') if tr else None)


def _wn_layer(x, num_out, k, pad, stride, nonlinearity, init_scale, init, use_weight_normalization, use_batch_normalization,
              use_mean_only_batch_normalization, deterministic, segments, init_eps, then_pool=None):
    """body shared by conv2d_WN (k x k) and dense_WN (k = 1 on [n,1,1,c]): every flag combination of nn.py:476-518,529-570.
    then_pool (extension): (keep-mask or None, 1/keep) of a max-pool 2x2 + dropout right behind the layer — the pooled activation is returned;
    on the weight-norm + mean-only-BN path with a fusable nonlinearity the pooling rides in the layer's apply launch (ops.conv2d(pool=...))."""
    if then_pool is not None and not (use_weight_normalization and use_mean_only_batch_normalization and not init
                                      and (_tg_act(nonlinearity) is not None or nonlinearity is None)):
        y = _wn_layer(x, num_out, k, pad, stride, nonlinearity, init_scale, init, use_weight_normalization, use_batch_normalization,
                      use_mean_only_batch_normalization, deterministic, segments, init_eps)
        return ops.maxpool2_dropout(y, then_pool[0], then_pool[1])
    cx = ctx()
    if use_weight_normalization and use_batch_normalization:
        raise ValueError("use_weight_normalization with use_batch_normalization: the reference creates no bias for that combination and "
                         "fails on it (nn.py:480-482,503-509)")
    V = cx.get_variable('V', (k, k, x.c, num_out) if k > 1 else (x.c, num_out), _normal(0.05, 'V'))
    b = cx.get_variable('b', (num_out,), 0.0) if not use_batch_normalization else None
    pop = cx.get_variable('meanOnlyBatchNormalization/pop_mean', (num_out,), 0.0, trainable=False) if use_mean_only_batch_normalization else None
    g = cx.get_variable('g', (num_out,), 1.0) if use_weight_normalization else None
    trains = cx.trains()
    gr = (lambda leaf: cx.var_grad(leaf)) if trains else (lambda leaf: None)
    a = _tg_act(nonlinearity)
    fuse = dict(act=a[0], alpha=a[1]) if a else {}
    if use_weight_normalization and init:
        # x_init = conv(x, l2_normalize(V)); scale_init*(x_init - m_init)   (nn.py:494-500,545-551) — forward only
        ones = cx.ws('const:ones', max(num_out, 1024))
        ops._call('tg_fill_f32', ops._p(ones), 1.0, ones.numel(), cx.stream)
        with cx.no_record():
            y = ops.conv2d(x, V, None, num_out, k, stride, pad, wn=(ones, None))
            y = ops.moments_normalize(y, init_eps, init_scale)
            return _apply_nonlinearity(y, nonlinearity)
    if use_weight_normalization and use_mean_only_batch_normalization:
        if a is None and nonlinearity is not None:
            y = ops.conv2d(x, V, None, num_out, k, stride, pad, wn=(g, gr('g')), mobn=(b, gr('b'), pop), segments=segments,
                           train=not deterministic, kernel_grad=gr('V'))
            return nonlinearity(y)
        return ops.conv2d(x, V, None, num_out, k, stride, pad, wn=(g, gr('g')), mobn=(b, gr('b'), pop), segments=segments,
                          train=not deterministic, kernel_grad=gr('V'), pool=then_pool, **fuse)
    if use_weight_normalization:                       # just weight normalisation: g*conv(x, V/||V||) + b
        y = ops.conv2d(x, V, b, num_out, k, stride, pad, wn=(g, gr('g')), kernel_grad=gr('V'), bias_grad=gr('b'), **fuse)
        return y if (a or nonlinearity is None) else nonlinearity(y)
    if use_batch_normalization:                        # conv(x, V) -> batch_norm_impl -> nonlinearity
        y = ops.conv2d(x, V, None, num_out, k, stride, pad, kernel_grad=gr('V'))
        y = batch_norm_impl(y, is_conv_out=k > 1, deterministic=deterministic)
        return _apply_nonlinearity(y, nonlinearity)
    y = ops.conv2d(x, V, b, num_out, k, stride, pad, kernel_grad=gr('V'), bias_grad=gr('b'), **fuse)
    return y if (a or nonlinearity is None) else nonlinearity(y)


def conv2d_WN(x, num_filters, filter_size=[3, 3], pad='SAME', stride=[1, 1], nonlinearity=None, init_scale=1., init=False,
              use_weight_normalization=False, use_batch_normalization=False, use_mean_only_batch_normalization=False,
              deterministic=False, name='', segments=None, then_pool=None):
    """Model/nn.py:469-520: W = g*l2_normalize(V,[0,1,2]); conv; mean-only BN (+b) | +b | batch_norm_impl; nonlinearity.
    segments, then_pool: extensions (see _wn_layer)."""
    assert filter_size[0] == filter_size[1] and stride[0] == stride[1]
    with ctx().variable_scope(name):
        return _wn_layer(x, num_filters, filter_size[0], pad, stride[0], nonlinearity, init_scale, init, use_weight_normalization,
                         use_batch_normalization, use_mean_only_batch_normalization, deterministic, segments, 1e-8, then_pool=then_pool)


def dense_WN(x, num_units, nonlinearity=None, init_scale=1., init=False, use_weight_normalization=False,
             use_batch_normalization=False, use_mean_only_batch_normalization=False, deterministic=False, name='',
             segments=None):
    """Model/nn.py:525-572: (x@V)*g/sqrt(sum V^2,0); mean-only BN over axis 0 (+b) | +b | batch_norm_impl; nonlinearity.
    Runs as a 1x1 convolution of the MFMA implicit-GEMM kernel."""
    with ctx().variable_scope(name):
        return _wn_layer(x, num_units, 1, 'SAME', 1, nonlinearity, init_scale, init, use_weight_normalization, use_batch_normalization,
                         use_mean_only_batch_normalization, deterministic, segments, 1e-10)


def NiN_WN(x, num_units, nonlinearity=None, init=False, use_weight_normalization=False, use_batch_normalization=False,
           use_mean_only_batch_normalization=False, deterministic=False, name='', segments=None):
    """Model/nn.py:577-589: reshape [N,H,W,C]->[NHW,C], dense_WN, reshape back (variable scope doubled:
    '<name>/<name>/V')."""
    with ctx().variable_scope(name):
        return dense_WN(x, num_units, nonlinearity=nonlinearity, init=init, use_weight_normalization=use_weight_normalization,
                        use_batch_normalization=use_batch_normalization,
                        use_mean_only_batch_normalization=use_mean_only_batch_normalization, deterministic=deterministic,
                        name=name, segments=segments)


# ---------------------------------------------------------------------------------------------------------------------------------
# Salimans-style weight-normalised layers (Model/nn.py:220-340; Salimans & Kingma 2016).  Not called by the reference's models
# (they use NN_Base._WN_dense / _WN_conv2d / _WN_deconv2d, which are the same arithmetic); layer names come from `counters`.
# ---------------------------------------------------------------------------------------------------------------------------------

def _salimans(x, num_out, k, stride, pad, nonlinearity, init_scale, init, ema, init_eps, init_w=None, train_scale=True, transposed=False):
    if ema is not None:
        raise ValueError("ema: substituting averaged variables (nn.py:98-110) is not supported; evaluate with the raw variables "
                         "as the reference's Train_goodGAN.py does (SURVEY App. C.7)")
    cx = ctx()
    if transposed:
        V = cx.get_variable('V', (k, k, num_out, x.c), init_w or _normal(0.05, 'V'))
    else:
        V = cx.get_variable('V', (k, k, x.c, num_out) if k > 1 else (x.c, num_out), init_w or _normal(0.05, 'V'))
    g = cx.get_variable('g', (num_out,), 1.0, trainable=train_scale)
    b = cx.get_variable('b', (num_out,), 0.0)
    trains = cx.trains()
    gr = (lambda leaf: cx.var_grad(leaf)) if trains else (lambda leaf: None)
    g_grad = gr('g') if train_scale else None
    a = _tg_act(nonlinearity)
    if init:
        ones = cx.ws('const:ones', max(num_out, 1024))
        ops._call('tg_fill_f32', ops._p(ones), 1.0, ones.numel(), cx.stream)
        with cx.no_record():
            if transposed:
                y = ops.deconv2d(x, V, None, num_out, wn=(ones, None))
            else:
                y = ops.conv2d(x, V, None, num_out, k, stride, pad, wn=(ones, None))
            return _apply_nonlinearity(ops.moments_normalize(y, init_eps, init_scale), nonlinearity)
    if g_grad is None and trains:
        g_grad = cx.scratch('g_frozen_grad', num_out)              # train_scale=False: the gradient is computed and dropped
    if transposed:
        y = ops.deconv2d(x, V, b, num_out, act=a[0] if a else None, kernel_grad=gr('V'), bias_grad=gr('b'), wn=(g, g_grad))
    else:
        y = ops.conv2d(x, V, b, num_out, k, stride, pad, wn=(g, g_grad), kernel_grad=gr('V'), bias_grad=gr('b'),
                       **(dict(act=a[0], alpha=a[1]) if a else {}))
    return y if (a or nonlinearity is None) else nonlinearity(y)


def dense(x, num_units, nonlinearity=None, init_scale=1., counters={}, init=False, ema=None, train_scale=True, init_w=None, **kwargs):
    """Model/nn.py:220-252, fully connected layer: x@V * g/sqrt(sum V^2,[0]) + b."""
    with ctx().variable_scope(get_name('dense', counters)):
        return _salimans(x, num_units, 1, 1, 'SAME', nonlinearity, init_scale, init, ema, 1e-10, init_w, train_scale)


def conv2d(x, num_filters, filter_size=[3, 3], stride=[1, 1], pad='SAME', nonlinearity=None, init_scale=1., counters={}, init=False,
           ema=None, **kwargs):
    """Model/nn.py:254-289, convolutional layer: conv(x, g*l2_normalize(V,[0,1,2])) + b."""
    assert filter_size[0] == filter_size[1] and stride[0] == stride[1]
    with ctx().variable_scope(get_name('conv2d', counters)):
        return _salimans(x, num_filters, filter_size[0], stride[0], pad, nonlinearity, init_scale, init, ema, 1e-8)


def deconv2d(x, num_filters, filter_size=[3, 3], stride=[1, 1], pad='SAME', nonlinearity=None, init_scale=1., counters={}, init=False,
             ema=None, **kwargs):
    """Model/nn.py:291-331, transposed convolutional layer: conv2d_transpose(x, g*l2_normalize(V,[0,1,3])) + b, V [kh,kw,Cout,Cin].
    The transposed-conv kernel of this package implements the geometry the reference's models use — 5x5, stride 2, 'SAME'
    (Model/modle_base.py:130-155, Good_GAN.py:81) — any other filter_size / stride / pad is refused."""
    if (list(filter_size), list(stride), pad) != ([5, 5], [2, 2], 'SAME'):
        raise ValueError("deconv2d: only filter_size=[5,5], stride=[2,2], pad='SAME' is implemented (got %r, %r, %r)" % (filter_size, stride, pad))
    with ctx().variable_scope(get_name('deconv2d', counters)):
        return _salimans(x, num_filters, 5, 2, 'SAME', nonlinearity, init_scale, init, ema, 1e-8, transposed=True)


def nin(x, num_units, **kwargs):
    """Model/nn.py:333-339, network in network (1x1 conv): reshape to [N*H*W, C], dense, reshape back — the NHWC rows already are
    that matrix, so the dense product runs on the activation as it lies."""
    return dense(x, num_units, **kwargs)
