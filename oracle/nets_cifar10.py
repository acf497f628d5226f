"""ORACLE — test infrastructure only (see oracle/tf_ops.py header; PARITY UNPINNED).

CPU restatement of the three CIFAR-10 networks of the reference
(Model/Good_GAN_cifar10.py:33-58 generator, :60-99 discriminator, :101-174
classifier, :287-299 ZCA), forward and hand-derived backward.

Parameters live in a flat dict keyed by the TF variable names the reference's
variable scopes produce (e.g. 'classifier/conv1_1/V').  All randomness (dropout
keep-masks, Gaussian noise) is injected through `rnd` dicts so that the HIP path
can be fed the identical draws.
"""
import numpy as np
from . import tf_ops as T

NUM_CLASSES = 10

# ---------------------------------------------------------------- classifier
C_CONVS = [  # name, cout, padding       (Good_GAN_cifar10.py:106-149)
    ('conv1_1', 128, 'SAME'), ('conv1_2', 128, 'SAME'), ('conv1_3', 128, 'SAME'),
    ('conv2_1', 256, 'SAME'), ('conv2_2', 256, 'SAME'), ('conv2_3', 256, 'SAME'),
    ('conv3', 512, 'VALID'),
]
C_POOL_AFTER = {'conv1_3': 'drop1', 'conv2_3': 'drop2'}


def classifier_param_shapes():
    """name -> shape, in TF creation order (nn.py:476-492: V, b, pop_mean, g)."""
    out = []
    cin = 3
    for name, cout, _ in C_CONVS:
        p = 'classifier/%s/' % name
        out += [(p + 'V', (3, 3, cin, cout)), (p + 'b', (cout,)),
                (p + 'meanOnlyBatchNormalization/pop_mean', (cout,)), (p + 'g', (cout,))]
        cin = cout
    for name, cout in (('NiN1', 256), ('NiN2', 128)):
        p = 'classifier/%s/%s/' % (name, name)            # doubled scope, nn.py:581-588
        out += [(p + 'V', (cin, cout)), (p + 'b', (cout,)),
                (p + 'meanOnlyBatchNormalization/pop_mean', (cout,)), (p + 'g', (cout,))]
        cin = cout
    p = 'classifier/output_dense/'
    out += [(p + 'V', (cin, NUM_CLASSES)), (p + 'b', (NUM_CLASSES,)),
            (p + 'meanOnlyBatchNormalization/pop_mean', (NUM_CLASSES,)), (p + 'g', (NUM_CLASSES,))]
    return out


def _wn_mobn(x2, P, p, train, out_axis_dense, pop_updates):
    """shared tail of conv2d_WN / dense_WN: mean-only BN (nn.py:505-506,559-561)."""
    pm = P[p + 'meanOnlyBatchNormalization/pop_mean']
    if train:
        y, new_pop = T.mobn_train(x2, pop_updates.get(p, pm), P[p + 'b'])
        pop_updates[p] = new_pop
    else:
        y = T.mobn_eval(x2, pm, P[p + 'b'])
    return y


def classifier_fwd(P, inp, train, rnd, pop_updates=None):
    """Good_GAN_cifar10.py:101-174.  rnd: 'noise' [N,32,32,3] (already scaled by 0.15),
    'drop1' [N,16,16,128], 'drop2' [N,8,8,256] keep masks (used only if train).
    pop_updates: dict collecting the sequential pop_mean updates (call-site order).
    Returns logits, feature, cache."""
    if pop_updates is None:
        pop_updates = {}
    cache = {'train': train}
    x = inp.reshape(-1, 32, 32, 3) + rnd['noise']          # :103-104 noise is ALWAYS on
    for name, cout, pad in C_CONVS:
        p = 'classifier/%s/' % name
        W = T.wn_weight(P[p + 'V'], P[p + 'g'])           # nn.py:502
        cache[name + '/x'] = x
        x = T.conv2d(x, W, (1, 1), pad)
        x = T.lrelu(_wn_mobn(x, P, p, train, None, pop_updates))
        cache[name + '/y'] = x
        if name in C_POOL_AFTER:
            x, idx = T.maxpool2(x)                          # :123,142
            cache[name + '/pool_idx'] = idx
            if train:                                       # :124,143 dropout 0.5
                x = T.dropout(x, rnd[C_POOL_AFTER[name]], 0.5)
    n = x.shape[0]
    for name, cout in (('NiN1', 256), ('NiN2', 128)):       # nn.py:577-589
        p = 'classifier/%s/%s/' % (name, name)
        s = x.shape
        x2 = x.reshape(-1, s[-1])
        cache[name + '/x'] = x2
        V, g = P[p + 'V'], P[p + 'g']
        scaler = g / np.sqrt(np.sum(np.square(V), axis=0))  # nn.py:554 (no eps)
        x2 = T.matmul(x2, V * scaler) if T.MFMA_BF16 else (x2 @ V) * scaler      # bf16 path: the MFMA sees the effective weight
        x2 = T.lrelu(_wn_mobn(x2, P, p, train, None, pop_updates))
        cache[name + '/y'] = x2
        x = x2.reshape(s[:-1] + (cout,))
    cache['gpool/shape'] = x.shape
    feat, idx = T.global_maxpool(x)                         # :163 MAX pool named avg_pool_0
    cache['gpool/idx'] = idx
    p = 'classifier/output_dense/'
    cache['output_dense/x'] = feat
    V, g = P[p + 'V'], P[p + 'g']
    scaler = g / np.sqrt(np.sum(np.square(V), axis=0))
    logits = _wn_mobn(T.matmul(feat, V * scaler) if T.MFMA_BF16 else (feat @ V) * scaler, P, p, train, None, pop_updates)
    return logits, feat, cache


def classifier_bwd(P, cache, dlogits, rnd, dfeat=None):
    """Gradient of classifier_fwd (train mode) wrt its trainable variables.
    Returns dict name -> grad.  No input gradient (nothing upstream of C is trained
    through it: SURVEY §3.2)."""
    assert cache['train']
    G = {}

    def dense_wn_bwd(p, x2, dy):
        dy, db = T.mobn_train_bwd(dy)
        G[p + 'b'] = db
        V, g = P[p + 'V'], P[p + 'g']
        dW = T.matmul(x2.T, dy)
        dv, dg = T.wn_weight_bwd(V, g, dW)
        G[p + 'V'], G[p + 'g'] = dv, dg
        return T.matmul(dy, T.wn_weight(V, g).T)

    dx = dense_wn_bwd('classifier/output_dense/', cache['output_dense/x'], dlogits)
    if dfeat is not None:
        dx = dx + dfeat
    dx = T.global_maxpool_bwd(dx, cache['gpool/idx'], cache['gpool/shape'])
    for name in ('NiN2', 'NiN1'):
        p = 'classifier/%s/%s/' % (name, name)
        s = dx.shape
        d2 = T.lrelu_bwd_from_out(cache[name + '/y'], dx.reshape(-1, s[-1]))
        d2 = dense_wn_bwd(p, cache[name + '/x'], d2)
        dx = d2.reshape(s[:-1] + (d2.shape[-1],))
    for name, cout, pad in reversed(C_CONVS):
        p = 'classifier/%s/' % name
        if name in C_POOL_AFTER:
            dx = T.dropout_bwd(dx, rnd[C_POOL_AFTER[name]], 0.5)
            dx = T.maxpool2_bwd(dx, cache[name + '/pool_idx'], cache[name + '/y'].shape)
        dx = T.lrelu_bwd_from_out(cache[name + '/y'], dx)
        dx, db = T.mobn_train_bwd(dx)
        G[p + 'b'] = db
        V, g = P[p + 'V'], P[p + 'g']
        x = cache[name + '/x']
        dW = T.conv2d_bwd_filter(x, dx, V.shape, (1, 1), pad)
        G[p + 'V'], G[p + 'g'] = T.wn_weight_bwd(V, g, dW)
        if name != 'conv1_1':
            dx = T.conv2d_bwd_input(x.shape, T.wn_weight(V, g), dx, (1, 1), pad)
    return G


# ----------------------------------------------------------------- generator
G_DECONVS = [('gg_dconv0', 256), ('gg_dconv1', 128), ('gg_dconv2', 3)]


def generator_param_shapes(z_dim=100):
    out = [('good_generator/gg_h0_lin/gg_h0_lin/kernel', (z_dim + NUM_CLASSES, 8192)),
           ('good_generator/gg_h0_lin/gg_h0_lin/bias', (8192,)),
           ('good_generator/gg_bn0/beta', (8192,)), ('good_generator/gg_bn0/gamma', (8192,))]
    cin = 512
    for i, (name, cout) in enumerate(G_DECONVS):
        p = 'good_generator/%s/%s/' % (name, name)
        out += [(p + 'kernel', (5, 5, cout, cin + NUM_CLASSES)), (p + 'bias', (cout,))]
        if i < 2:
            out += [('good_generator/gg_bn%d/beta' % (i + 1), (cout,)),
                    ('good_generator/gg_bn%d/gamma' % (i + 1), (cout,))]
        cin = cout
    return out


BN_DECAY = 0.9      # config.BATCH_NORM_DECAY (reference config.py)


def generator_moving_shapes():
    """the non-trainable variables tf.contrib.layers.batch_norm creates next to beta / gamma (modle_base.py:229-237)."""
    out = []
    for i, cout in enumerate([8192] + [co for _, co in G_DECONVS[:2]]):
        out += [('good_generator/gg_bn%d/moving_mean' % i, (cout,)), ('good_generator/gg_bn%d/moving_variance' % i, (cout,))]
    return out


def _bn_moving(P, i, cache, count, moving):
    """updates_collections=None: every execution of a training-mode batch_norm updates its moving statistics in place — once per
    sess.run that evaluates the generator, i.e. THREE times per iteration (Train_goodGAN.py:267,270,275).  Dead state for the losses
    (the generator's BN is always in training mode) but checkpoint content."""
    if moving is None:
        return
    km, kv = 'good_generator/gg_bn%d/moving_mean' % i, 'good_generator/gg_bn%d/moving_variance' % i
    if km in moving:
        moving[km], moving[kv] = T.batch_norm_moving_update(moving[km], moving[kv], cache[2], cache[3], count, BN_DECAY, fused=True)


def generator_fwd(P, z, y, eps=1e-5, moving=None):
    """Good_GAN_cifar10.py:33-58 (= good_sampler :176-202). BN always in train mode.  moving: the variable dict whose
    gg_bn*/moving_mean / moving_variance entries this execution updates (None: a pure function, as the tests of the nets use it)."""
    c = {}
    zy = np.concatenate([z, y], axis=1)
    c['zy'] = zy
    h = T.matmul(zy, P['good_generator/gg_h0_lin/gg_h0_lin/kernel']) + P['good_generator/gg_h0_lin/gg_h0_lin/bias']
    h = T.relu(h)
    c['r0'] = h
    h, c['bn0'] = T.batch_norm_train(h, P['good_generator/gg_bn0/gamma'], P['good_generator/gg_bn0/beta'], eps)
    _bn_moving(P, 0, c['bn0'], h.shape[0], moving)
    h = T.conv_cond_concat(h.reshape(-1, 4, 4, 512), y)
    for i, (name, cout) in enumerate(G_DECONVS):
        p = 'good_generator/%s/%s/' % (name, name)
        c['x%d' % i] = h
        h = T.conv2d_transpose(h, P[p + 'kernel'], (2, 2)) + P[p + 'bias']
        if i < 2:
            h = T.relu(h)
            c['r%d' % (i + 1)] = h
            h, c['bn%d' % (i + 1)] = T.batch_norm_train(
                h, P['good_generator/gg_bn%d/gamma' % (i + 1)], P['good_generator/gg_bn%d/beta' % (i + 1)], eps)
            _bn_moving(P, i + 1, c['bn%d' % (i + 1)], h.size // h.shape[-1], moving)
            h = T.conv_cond_concat(h, y)
        else:
            h = np.tanh(h)
            c['out'] = h
    return h, c


def generator_bwd(P, c, dout):
    G = {}
    d = dout * (1 - np.square(c['out']))
    for i in (2, 1, 0):
        name, cout = G_DECONVS[i]
        p = 'good_generator/%s/%s/' % (name, name)
        if i < 2:
            d = d[..., :cout]                                # drop the label channels of the concat
            d, dg, db = T.batch_norm_train_bwd(d, P['good_generator/gg_bn%d/gamma' % (i + 1)], c['bn%d' % (i + 1)])
            G['good_generator/gg_bn%d/gamma' % (i + 1)] = dg
            G['good_generator/gg_bn%d/beta' % (i + 1)] = db
            d = T.relu_bwd_from_out(c['r%d' % (i + 1)], d)
        G[p + 'bias'] = d.sum(axis=(0, 1, 2))
        G[p + 'kernel'] = T.conv2d_transpose_bwd_filter(c['x%d' % i], d, P[p + 'kernel'].shape, (2, 2))
        d = T.conv2d_transpose_bwd_input(P[p + 'kernel'], d, (2, 2))
    d = d[..., :512].reshape(-1, 8192)
    d, dg, db = T.batch_norm_train_bwd(d, P['good_generator/gg_bn0/gamma'], c['bn0'])
    G['good_generator/gg_bn0/gamma'], G['good_generator/gg_bn0/beta'] = dg, db
    d = T.relu_bwd_from_out(c['r0'], d)
    G['good_generator/gg_h0_lin/gg_h0_lin/bias'] = d.sum(axis=0)
    G['good_generator/gg_h0_lin/gg_h0_lin/kernel'] = T.matmul(c['zy'].T, d)
    return G


# ------------------------------------------------------------- discriminator
D_CONVS = [  # name, cout, stride, dropout-after key       (Good_GAN_cifar10.py:66-91)
    ('conv2d_00', 32, 1, None), ('conv2d_01', 32, 2, 'drop1'),
    ('conv2d_10', 64, 1, None), ('conv2d_11', 64, 2, 'drop2'),
    ('conv2d_20', 128, 1, None), ('conv2d_21', 128, 1, None),
]


def discriminator_param_shapes():
    out = []
    cin = 3
    for name, cout, _, _ in D_CONVS:
        p = 'discriminator/%s/%s/' % (name, name)
        out += [(p + 'kernel', (3, 3, cin + NUM_CLASSES, cout)), (p + 'bias', (cout,))]
        cin = cout
    out += [('discriminator/lin/lin/kernel', (cin + NUM_CLASSES, 1)), ('discriminator/lin/lin/bias', (1,))]
    return out


def discriminator_fwd(P, image, y, rnd):
    """Good_GAN_cifar10.py:60-99. rnd: keep masks 'drop0' [N,32,32,3], 'drop1' [N,16,16,32],
    'drop2' [N,8,8,64] (dropout 0.2, ALWAYS on).  Returns logits [N,1], cache."""
    c = {}
    h = T.dropout(image, rnd['drop0'], 0.2)
    for name, cout, s, drop in D_CONVS:
        p = 'discriminator/%s/%s/' % (name, name)
        h = T.conv_cond_concat(h, y)
        c[name + '/x'] = h
        h = T.lrelu(T.conv2d(h, P[p + 'kernel'], (s, s), 'SAME') + P[p + 'bias'])
        c[name + '/y'] = h
        if drop:
            h = T.dropout(h, rnd[drop], 0.2)
    c['pool/shape'] = h.shape
    h = np.concatenate([T.global_avgpool(h), y], axis=1)   # :94-97 avg-pool 8x8 -> [N,128]; concat y
    c['lin/x'] = h
    logits = T.matmul(h, P['discriminator/lin/lin/kernel']) + P['discriminator/lin/lin/bias']
    return logits, c


def discriminator_bwd(P, c, dlogits, rnd, want_weight_grads=True, want_input_grad=False):
    G = {}
    if want_weight_grads:
        G['discriminator/lin/lin/bias'] = dlogits.sum(axis=0)
        G['discriminator/lin/lin/kernel'] = T.matmul(c['lin/x'].T, dlogits)
    d = T.matmul(dlogits, P['discriminator/lin/lin/kernel'].T)[:, :128]
    d = T.global_avgpool_bwd(d, c['pool/shape'])
    for li in range(len(D_CONVS) - 1, -1, -1):
        name, cout, s, drop = D_CONVS[li]
        p = 'discriminator/%s/%s/' % (name, name)
        if drop:
            d = T.dropout_bwd(d, rnd[drop], 0.2)
        d = T.lrelu_bwd_from_out(c[name + '/y'], d)
        x = c[name + '/x']
        if want_weight_grads:
            G[p + 'bias'] = d.sum(axis=(0, 1, 2))
            G[p + 'kernel'] = T.conv2d_bwd_filter(x, d, P[p + 'kernel'].shape, (s, s), 'SAME')
        if li > 0 or want_input_grad:
            d = T.conv2d_bwd_input(x.shape, P[p + 'kernel'], d, (s, s), 'SAME')
            d = d[..., :x.shape[-1] - NUM_CLASSES]
    dimage = T.dropout_bwd(d, rnd['drop0'], 0.2) if want_input_grad else None
    return G, dimage


# ----------------------------------------------------------------------- ZCA

def zca_apply(x, mean, mat):
    """Good_GAN_cifar10.py:294-299: (flatten(x) - mean) @ mat, reshaped back."""
    s = x.shape
    return ((x.reshape(s[0], -1) - mean) @ mat).reshape(s)
