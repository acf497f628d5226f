"""NN_Base — MI355X-native counterpart of the reference's Model/modle_base.py (sic; the reference's
models import it as `model_base`, Model/Good_GAN.py:8 — both spellings are provided here).

Method names and arguments follow Model/modle_base.py:27-48 (_linear_fc), :157-168 (_conv2d),
:190-202 (_drop_out, _add_noise), :229-237 (_batch_norm_contrib), :239-244 (_conv_cond_concat),
:246-259 (_deconv2d).  Tensors are tg.runtime.Act handles; variables are looked up in the active
Context under the names TF would create ('<scope>/<name>/<name>/kernel' for tf.layers.* called
inside `with tf.variable_scope(name)` with `name=name`).

Extensions (keyword-only, default = reference behaviour): `activation` fuses the nonlinearity the
model applies next into the MFMA kernel's epilogue; `narrow` stores only the logical channels.
Initialisers are applied when the Model creates its variables (see Good_GAN_cifar10.param_specs),
so `kernel_initializer` is accepted and ignored here.
"""
from tg import ops
from tg.runtime import ctx


def _act_of(fn):
    if fn is None:
        return None, 0.2
    a = getattr(fn, 'tg_act', None)
    if a is None:
        raise ValueError("activation must be a tg activation (leakyReLu, tf-like relu/tanh helpers)")
    return a


def _cat_of(then_concat):
    """(label tensor, number of label channels) of a `then_concat=` argument: a label Act, such a tuple, or None."""
    if then_concat is None or isinstance(then_concat, tuple):
        return then_concat
    return (then_concat.t, then_concat.c)


class NN_Base(object):
    def __init__(self, batch_norm_decay=0.9, batch_norm_epsilon=1e-5):
        self._batch_norm_decay = batch_norm_decay
        self._batch_norm_epsilon = batch_norm_epsilon

    def forward_pass(self, x):
        raise NotImplementedError('forward_pass() is implemented in Model sub classes')

    # ---- activations: callable on a tensor (one elementwise launch, modle_base.py:170-188) AND usable as `activation=` /
    # `nonlinearity=` of a layer, which fuses them into the producing kernel's epilogue (what the models of this package do)
    def _relu(self, x):
        return ops.activation(x, 'relu')
    _relu.tg_act = ('relu', 0.0)

    def _tanh(self, x):
        return ops.activation(x, 'tanh')
    _tanh.tg_act = ('tanh', 0.0)

    def _leaky_relu(self, x, alpha=0.2):
        """tf.nn.leaky_relu (modle_base.py:181-182); as `activation=` the slope is the default 0.2."""
        return ops.activation(x, 'lrelu', alpha)
    _leaky_relu.tg_act = ('lrelu', 0.2)

    def _softplus(self, x):
        return ops.activation(x, 'softplus')
    _softplus.tg_act = ('softplus', 0.0)

    def _sigmoid(self, x):
        return ops.activation(x, 'sigmoid')
    _sigmoid.tg_act = ('sigmoid', 0.0)

    # ---- layers -----------------------------------------------------------------------------------
    def _linear_fc(self, input_, output_size, scope=None, bias_start=0.0, use_bias=True, kernel_initializer=None,
                   activation=None, narrow=False):
        """tf.layers.dense + bias (Model/modle_base.py:27-48); runs as a 1-tap MFMA implicit GEMM."""
        cx = ctx()
        act, alpha = _act_of(activation)
        with cx.variable_scope(scope), cx.variable_scope(scope):
            tr = cx.trains()
            return ops.conv2d(input_, cx.var('kernel'), cx.var('bias') if use_bias else None, output_size, 1, 1, 'SAME',
                              act=act, alpha=alpha, kernel_grad=cx.var_grad('kernel') if tr else None,
                              bias_grad=cx.var_grad('bias') if (tr and use_bias) else None,
                              n_store_ld=(output_size, output_size) if narrow else None)

    def _conv2d(self, input_, output_dim, k_h=5, k_w=5, d_h=2, d_w=2, kernel_initializer=None, name="conv2d",
                activation=None, bn_segments=None, then_concat=None):
        """tf.layers.conv2d 'same' + bias (Model/modle_base.py:157-168).  bn_segments (extension): a training-mode _batch_norm_contrib over
        these application segments follows directly — its statistics pass is taken in this layer's launch (ops.conv2d(bn_stats=True)).
        then_concat (extension): the label Act a _conv_cond_concat right behind this layer will append — the launch writes the concatenated
        tensor itself (ops.conv2d(concat=...)) and that _conv_cond_concat becomes a view."""
        assert k_h == k_w and d_h == d_w
        cx = ctx()
        act, alpha = _act_of(activation)
        with cx.variable_scope(name), cx.variable_scope(name):
            tr = cx.trains()
            return ops.conv2d(input_, cx.var('kernel'), cx.var('bias'), output_dim, k_h, d_h, 'SAME', act=act, alpha=alpha,
                              kernel_grad=cx.var_grad('kernel') if tr else None, bias_grad=cx.var_grad('bias') if tr else None,
                              segments=bn_segments if bn_segments else None, bn_stats=bn_segments is not None,
                              concat=_cat_of(then_concat))

    def _deconv2d(self, input_, output_shape, k_h=5, k_w=5, d_h=2, d_w=2, name="deconv2d", use_bias=True,
                  kernel_initializer=None, activation=None, narrow=False):
        """tf.layers.conv2d_transpose 'same' + bias (Model/modle_base.py:246-259); output_shape = channels."""
        assert (k_h, k_w, d_h, d_w) == (5, 5, 2, 2) and use_bias
        cx = ctx()
        act, _ = _act_of(activation)
        with cx.variable_scope(name), cx.variable_scope(name):
            tr = cx.trains()
            return ops.deconv2d(input_, cx.var('kernel'), cx.var('bias'), output_shape, act=act,
                                kernel_grad=cx.var_grad('kernel') if tr else None, bias_grad=cx.var_grad('bias') if tr else None,
                                narrow_out=narrow)

    def _batch_norm_contrib(self, x, name, train=False, segments=None):
        """tf.contrib.layers.batch_norm(decay, eps, scale=True, updates_collections=None) (modle_base.py:229-237).
        train=True: batch statistics, moving statistics updated in place; train=False: moving statistics.
        segments (extension): image counts of the applications batched into x — statistics per application."""
        cx = ctx()
        with cx.variable_scope(name):
            if not train:
                return ops.batch_norm_eval(x, cx.var('gamma'), cx.var('beta'), cx.var('moving_mean'), cx.var('moving_variance'),
                                           self._batch_norm_epsilon)
            tr = cx.trains()
            return ops.batch_norm_train(x, cx.var('gamma'), cx.var('beta'), cx.var('moving_mean'), cx.var('moving_variance'),
                                        self._batch_norm_epsilon, self._batch_norm_decay,
                                        gamma_grad=cx.var_grad('gamma') if tr else None, beta_grad=cx.var_grad('beta') if tr else None,
                                        segments=segments)

    def _WN_dense(self, input_, output_size, scope, init_scale=1.0, init=False, activation=None, narrow=False):
        """g * (x @ l2_normalize(V,[0])) + b (modle_base.py:50-73; the data-dependent init branch is never taken)."""
        cx = ctx()
        act, alpha = _act_of(activation)
        with cx.variable_scope(scope):
            tr = cx.trains()
            return ops.conv2d(input_, cx.var('V'), cx.var('b'), output_size, 1, 1, 'SAME', act=act, alpha=alpha,
                              wn=(cx.var('g'), cx.var_grad('g') if tr else None), kernel_grad=cx.var_grad('V') if tr else None,
                              bias_grad=cx.var_grad('b') if tr else None, n_store_ld=(output_size, output_size) if narrow else None)

    def _WN_conv2d(self, input_, output_dim, k_h=5, k_w=5, d_h=2, d_w=2, padding='SAME', init_scale=1.0, init=False, name="conv2d",
                   activation=None, then_concat=None):
        """g * conv(x, l2_normalize(V,[0,1,2])) + b (modle_base.py:75-108).  then_concat (extension): as in _conv2d — (label tensor, count) or a
        label Act the _conv_cond_concat / cond_concat right behind this layer appends."""
        assert k_h == k_w and d_h == d_w
        cx = ctx()
        act, alpha = _act_of(activation)
        with cx.variable_scope(name):
            tr = cx.trains()
            return ops.conv2d(input_, cx.var('V'), cx.var('b'), int(output_dim), k_h, d_h, padding, act=act, alpha=alpha,
                              wn=(cx.var('g'), cx.var_grad('g') if tr else None), kernel_grad=cx.var_grad('V') if tr else None,
                              bias_grad=cx.var_grad('b') if tr else None, concat=_cat_of(then_concat))

    def _WN_deconv2d(self, input_, output_dim, k_h=3, k_w=3, d_h=2, d_w=2, padding='SAME', init_scale=1.0, init=False, name="deconv2d",
                     activation=None, narrow=False):
        """g * conv2d_transpose(x, l2_normalize(V,[0,1,3])) + b, V [kh,kw,Cout,Cin] (modle_base.py:130-155); 5x5 s2 'SAME' only
        (the only configuration the reference's models use)."""
        assert (k_h, k_w, d_h, d_w, padding) == (5, 5, 2, 2, 'SAME')
        cx = ctx()
        act, _ = _act_of(activation)
        with cx.variable_scope(name):
            tr = cx.trains()
            return ops.deconv2d(input_, cx.var('V'), cx.var('b'), int(output_dim), act=act, kernel_grad=cx.var_grad('V') if tr else None,
                                bias_grad=cx.var_grad('b') if tr else None, narrow_out=narrow,
                                wn=(cx.var('g'), cx.var_grad('g') if tr else None))

    def _minibatch_discrimination(self, input, num_kernels, dim_per_kernel=5, name="minibatch_discrim", concat_input=False):
        """modle_base.py:110-128: variables `w` [features, num_kernels*dim_per_kernel] (Xavier) and `b` [num_kernels] of the enclosing
        variable scope (tf.name_scope does not prefix tf.get_variable).  concat_input (extension): return concat([input, f], 1), the
        only use the reference makes of the result (Good_GAN.py:160-161), from the same launch."""
        cx = ctx()
        tr = cx.trains()
        return ops.minibatch_discrimination(input, cx.var('w'), cx.var('b'), num_kernels, dim_per_kernel,
                                            w_grad=cx.var_grad('w') if tr else None, b_grad=cx.var_grad('b') if tr else None,
                                            concat_input=concat_input)

    def _nin(self, input, num_units, name, activation=None):
        """network-in-network (1x1 conv): reshape + _WN_dense + reshape (modle_base.py:204-209)."""
        return self._WN_dense(input, num_units, name, activation=activation)

    def _conv_cond_concat(self, x, y):
        """Concatenate conditioning vector on feature map axis (modle_base.py:239-244); y: Act [N,ncls]."""
        return ops.cond_concat(x, y.t, y.c)

    def _drop_out(self, x, rate=0.5, train=False, name=None, fuse_next=False):
        """tf.layers.dropout (modle_base.py:190-191): x*mask/keep with a floor(keep+U) keep-mask.  fuse_next (extension): the
        result goes straight into _conv_cond_concat, which applies the mask in its own launch (ops.scale_mask(defer=True))."""
        if not train:
            return x
        cx = ctx()
        mask = cx.rng.keep_mask(cx, name or cx.next_rng_name('drop'), x.rows * x.c, 1.0 - rate)
        assert x.ld == x.c
        return ops.scale_mask(x, mask, 1.0 / (1.0 - rate), defer=fuse_next)

    def _add_noise(self, inputs, mean=0.0, stddev=0.001, name=None):
        """inputs + N(mean, stddev) (modle_base.py:193-202) on a dense activation; the gradient passes through."""
        assert mean == 0.0
        cx = ctx()
        noise = cx.rng.normal(cx, name or cx.next_rng_name('noise'), inputs.rows * inputs.c, stddev)
        return ops.add_noise(inputs, noise)
