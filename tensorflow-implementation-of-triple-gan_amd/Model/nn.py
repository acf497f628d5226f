"""Layer primitives (free functions) — MI355X-native counterpart of the reference's Model/nn.py.

Same names and keyword arguments as Model/nn.py:147-187 (mean_only_batch_norm_impl),
:469-520 (conv2d_WN), :525-572 (dense_WN), :577-589 (NiN_WN); the arithmetic runs in hand-written
gfx950 kernels (csrc/) instead of TensorFlow ops.  Tensors are `tg.runtime.Act` handles (NHWC
device buffers); variables come from the active `tg.runtime.Context` under the same scope names
the reference's tf.variable_scope calls produce ('classifier/conv1_1/V', 'classifier/NiN1/NiN1/V').

Graph-mode keywords with no eager meaning are accepted and ignored the way the reference
effectively ignores them: `init` (its assign ops are "created but never run", nn.py:497-499),
`init_scale`.  `deterministic` is a Python bool here (not a tf.bool tensor): True = evaluation
(use pop_mean), False = training.  Extension: `segments` = image counts of the classifier
applications batched into one call; mean-only-BN statistics are per application.
"""
from tg import ops
from tg.runtime import ctx


def int_shape(x):
    """Model/nn.py:12-13."""
    return [x.n, x.h, x.w, x.c] if (x.h, x.w) != (1, 1) else [x.n, x.c]


def mean_only_batch_norm_impl(x, pop_mean, b, is_conv_out=True, deterministic=False, decay=0.9, name='meanOnlyBatchNormalization',
                              b_grad=None, segments=None):
    """Model/nn.py:147-187 as a stand-alone function: x - mean + b with the running mean updated (training) or x - pop_mean + b
    (deterministic).  pop_mean / b: device tensors (e.g. ctx().var(...)); is_conv_out only selects the reduction axes in the
    reference ([0,1,2] vs [0]) — here every row of the activation is one sample of the statistic either way.  The models use
    the version fused into the convolution (conv2d_WN / dense_WN)."""
    with ctx().variable_scope(name):
        return ops.mean_only_batch_norm(x, pop_mean, b, b_grad=b_grad, train=not deterministic, decay=decay, segments=segments)


def _wn_layer(x, num_out, k, pad, stride, nonlinearity, use_weight_normalization, use_batch_normalization,
              use_mean_only_batch_normalization, deterministic, segments):
    if use_batch_normalization or not use_weight_normalization or not use_mean_only_batch_normalization:
        raise NotImplementedError("only the weight-norm + mean-only-BN path is executed by the reference's models "
                                  "(Model/Good_GAN_cifar10.py:106-172); batch_norm_impl (nn.py:192-218) is never enabled")
    cx = ctx()
    act = getattr(nonlinearity, 'tg_act', None) if nonlinearity is not None else None
    if nonlinearity is not None and act is None:
        raise ValueError("nonlinearity must be one of the tg activations (e.g. Good_GAN_cifar10.leakyReLu)")
    trains = cx.trains()
    return ops.conv2d(
        x, cx.var('V'), None, num_out, k, stride, pad, act=act[0] if act else None, alpha=act[1] if act else 0.2,
        wn=(cx.var('g'), cx.var_grad('g') if trains else None),
        mobn=(cx.var('b'), cx.var_grad('b') if trains else None, cx.var('meanOnlyBatchNormalization/pop_mean')),
        segments=segments, train=not deterministic, kernel_grad=cx.var_grad('V') if trains else None)


def conv2d_WN(x, num_filters, filter_size=[3, 3], pad='SAME', stride=[1, 1], nonlinearity=None, init_scale=1., init=False,
              use_weight_normalization=False, use_batch_normalization=False, use_mean_only_batch_normalization=False,
              deterministic=False, name='', segments=None):
    """Model/nn.py:469-520: W = g*l2_normalize(V,[0,1,2]); conv; mean-only BN (+b); nonlinearity."""
    assert filter_size[0] == filter_size[1] and stride[0] == stride[1]
    with ctx().variable_scope(name):
        return _wn_layer(x, num_filters, filter_size[0], pad, stride[0], nonlinearity, use_weight_normalization,
                         use_batch_normalization, use_mean_only_batch_normalization, deterministic, segments)


def dense_WN(x, num_units, nonlinearity=None, init_scale=1., init=False, use_weight_normalization=False,
             use_batch_normalization=False, use_mean_only_batch_normalization=False, deterministic=False, name='',
             segments=None):
    """Model/nn.py:525-572: (x@V)*g/sqrt(sum V^2,0); mean-only BN over axis 0 (+b); nonlinearity.
    Runs as a 1x1 convolution of the MFMA implicit-GEMM kernel."""
    with ctx().variable_scope(name):
        return _wn_layer(x, num_units, 1, 'SAME', 1, nonlinearity, use_weight_normalization, use_batch_normalization,
                         use_mean_only_batch_normalization, deterministic, segments)


def NiN_WN(x, num_units, nonlinearity=None, init=False, use_weight_normalization=False, use_batch_normalization=False,
           use_mean_only_batch_normalization=False, deterministic=False, name='', segments=None):
    """Model/nn.py:577-589: reshape [N,H,W,C]->[NHW,C], dense_WN, reshape back (variable scope doubled:
    '<name>/<name>/V')."""
    with ctx().variable_scope(name):
        return dense_WN(x, num_units, nonlinearity=nonlinearity, init=init, use_weight_normalization=use_weight_normalization,
                        use_batch_normalization=use_batch_normalization,
                        use_mean_only_batch_normalization=use_mean_only_batch_normalization, deterministic=deterministic,
                        name=name, segments=segments)
