"""ORACLE — test infrastructure only (see oracle/tf_ops.py header; PARITY UNPINNED).

CPU restatement of the MNIST and SVHN networks of the reference's Model/Good_GAN.py
(generator :15-83, discriminator :89-206, classifier :212-350; the unused cifar10 branches
are the svhn ones verbatim), forward and hand-derived backward, written as layer lists run by
one small sequential interpreter.  Variables are keyed by the TF names the reference's scopes
produce; randomness (noise, dropout keep-masks) is injected through `rnd`.

TF semantics encoded here beyond oracle/tf_ops.py (all [UNVERIFIED-TF]):
  * tf.layers.* default initialisers as the reference passes them: `random_normal_initializer(0.02)`
    / `truncated_normal_initializer(0.02)` = MEAN 0.02, stddev 1.0 (Model/modle_base.py:28,159,248);
  * tf.nn.leaky_relu default alpha 0.2;
  * contrib batch_norm(is_training=False) normalises with moving_mean / moving_variance; in training the
    moving variance is fed the Bessel-corrected batch variance (fused kernel), decay 0.9.
"""
import numpy as np
from . import tf_ops as T

NCLS = 10
BN_EPS, BN_DECAY = 1e-5, 0.9


# ---------------------------------------------------------------------------- layer lists

def _bn(name):
    return ('bn', name)


def generator_layers(data):
    g = 'good_generator/'
    if data == 'mnist':                                             # Good_GAN.py:19-33
        return [('concat_y',), ('dense', g + 'gg_h0_lin/gg_h0_lin', 500), ('act', 'softplus'), _bn(g + 'gg_bn0'),
                ('concat_y',), ('dense', g + 'gg_h1_lin/gg_h1_lin', 500), ('act', 'softplus'), _bn(g + 'gg_bn1'),
                ('concat_y',), ('wn_dense', g + 'gg_h2_lin', 784), ('act', 'sigmoid')]
    return [('concat_y',), ('dense', g + 'gg_h0_lin/gg_h0_lin', 8192), ('reshape', (4, 4, 512)), ('act', 'relu'), _bn(g + 'gg_bn0'),   # :35-59
            ('cond_concat',), ('deconv', g + 'gg_dconv0/gg_dconv0', 256), ('act', 'relu'), _bn(g + 'gg_bn1'),
            ('cond_concat',), ('deconv', g + 'gg_dconv1/gg_dconv1', 128), ('act', 'relu'), _bn(g + 'gg_bn2'),
            ('cond_concat',), ('wn_deconv', g + 'gg_wndconv0', 3), ('act', 'tanh')]


def discriminator_layers(data):
    d = 'discriminator/'
    if data == 'mnist':                                             # :93-124
        L = [('reshape', (784,)), ('noise', 'noise0', 0.2), ('concat_y',)]
        for i, n in enumerate((1000, 500, 250, 250, 250)):
            L += [('wn_dense', d + 'd_h%d_wndense0' % i, n), ('act', 'lrelu'), ('noise', 'noise%d' % (i + 1), 0.2), ('concat_y',)]
        return L + [('wn_dense', d + 'd_h5_wndense0', 1)]
    return [('dropout', 'drop0', 0.2, True), ('cond_concat',),     # :126-165
            ('wn_conv', d + 'd_h0_wnconv0', 32, 1), ('act', 'lrelu'), ('cond_concat',),
            ('wn_conv', d + 'd_h0_wnconv1', 32, 2), ('act', 'lrelu'), ('dropout', 'drop1', 0.2, True), ('cond_concat',),
            ('wn_conv', d + 'd_h1_wnconv0', 64, 1), ('act', 'lrelu'), ('cond_concat',),
            ('wn_conv', d + 'd_h1_wnconv1', 64, 2), ('act', 'lrelu'), ('dropout', 'drop2', 0.2, True), ('cond_concat',),
            ('wn_conv', d + 'd_h2_wnconv0', 128, 1), ('act', 'lrelu'), ('cond_concat',), ('cond_concat',),   # y twice (:151-153)
            ('wn_conv', d + 'd_h2_wnconv1', 128, 1), ('act', 'lrelu'),
            ('gmean',), ('concat_y',), ('wn_dense', d + 'd_h3_wndense', 1)]


def classifier_layers(data):
    c = 'classifier/'

    def cbr(name, bn, cout):
        return [('conv', c + '%s/%s' % (name, name), cout, 1), ('act', 'lrelu'), _bn(c + bn)]
    if data == 'mnist':                                             # :216-247
        L = [('reshape', (28, 28, 1)), ('noise', 'noise', 0.3)] + cbr('c_h0_conv0', 'c_h0_bn0', 32) + \
            [('maxpool',), ('dropout', 'drop1', 0.5, False)] + cbr('c_h1_conv0', 'c_h1_bn0', 64) + cbr('c_h1_conv1', 'c_h1_bn1', 64) + \
            [('maxpool',), ('dropout', 'drop2', 0.5, False)] + cbr('c_h2_conv0', 'c_h2_bn0', 128) + cbr('c_h2_conv1', 'c_h2_bn1', 128)
    else:                                                           # :249-299
        L = [('dropout', 'drop0', 0.2, False)] + cbr('c_h0_conv0', 'c_h0_bn0', 128) + cbr('c_h0_conv1', 'c_h0_bn1', 128) + \
            cbr('c_h0_conv2', 'c_h0_bn2', 128) + [('maxpool',), ('dropout', 'drop1', 0.5, False)] + \
            cbr('c_h1_conv0', 'c_h1_bn0', 256) + cbr('c_h1_conv1', 'c_h1_bn1', 256) + cbr('c_h1_conv2', 'c_h1_bn2', 256) + \
            [('maxpool',), ('dropout', 'drop2', 0.5, False)] + cbr('c_h2_conv0', 'c_h2_bn0', 512) + \
            [('nin', c + 'c_h2_nin0', 256), ('act', 'lrelu'), _bn(c + 'c_h2_bn1'), ('nin', c + 'c_h2_nin1', 128), ('act', 'lrelu'), _bn(c + 'c_h2_bn2')]
    return L + [('gmean',), ('feature',), ('dense', c + 'c_h2_lin/c_h2_lin', NCLS), _bn(c + 'c_h3_bn0')]


def image_shape(data):
    return (28, 28, 1) if data == 'mnist' else (32, 32, 3)


def param_shapes(data, z_dim=100):
    """[(name, shape, kind)] in creation order; kind in he? no: 'n02' (mean .02 std 1), 'tn02' (truncated), 'n05' (N(0,.05^2)),
    'one', 'zero'; non-trainable names contain 'moving_'."""
    out = []

    def walk(layers, x_shape):
        s = x_shape
        for l in layers:
            k = l[0]
            if k in ('concat_y',):
                s = s[:-1] + (s[-1] + NCLS,)
            elif k == 'cond_concat':
                s = s[:-1] + (s[-1] + NCLS,)
            elif k == 'reshape':
                s = tuple(l[1])
            elif k == 'dense':
                out.extend([(l[1] + '/kernel', (s[-1], l[2]), 'n02'), (l[1] + '/bias', (l[2],), 'zero')])
                s = (l[2],)
            elif k in ('wn_dense', 'nin'):
                out.extend([(l[1] + '/V', (s[-1], l[2]), 'n05'), (l[1] + '/g', (l[2],), 'one'), (l[1] + '/b', (l[2],), 'zero')])
                s = s[:-1] + (l[2],)
            elif k == 'conv':
                out.extend([(l[1] + '/kernel', (3, 3, s[-1], l[2]), 'tn02'), (l[1] + '/bias', (l[2],), 'zero')])
                s = s[:-1] + (l[2],)
            elif k == 'wn_conv':
                out.extend([(l[1] + '/V', (3, 3, s[-1], l[2]), 'n05'), (l[1] + '/g', (l[2],), 'one'), (l[1] + '/b', (l[2],), 'zero')])
                h = -(-s[0] // l[3])
                s = (h, h, l[2])
            elif k == 'deconv':
                out.extend([(l[1] + '/kernel', (5, 5, l[2], s[-1]), 'n02'), (l[1] + '/bias', (l[2],), 'zero')])
                s = (s[0] * 2, s[1] * 2, l[2])
            elif k == 'wn_deconv':
                out.extend([(l[1] + '/V', (5, 5, l[2], s[-1]), 'n05'), (l[1] + '/g', (l[2],), 'one'), (l[1] + '/b', (l[2],), 'zero')])
                s = (s[0] * 2, s[1] * 2, l[2])
            elif k == 'bn':
                c = s[-1]
                out.extend([(l[1] + '/beta', (c,), 'zero'), (l[1] + '/gamma', (c,), 'one'),
                            (l[1] + '/moving_mean', (c,), 'zero'), (l[1] + '/moving_variance', (c,), 'one')])
            elif k == 'maxpool':
                s = (s[0] // 2, s[1] // 2, s[2])
            elif k == 'gmean':
                s = (s[-1],)
    walk(generator_layers(data), (z_dim,))
    walk(discriminator_layers(data), image_shape(data))
    walk(classifier_layers(data), image_shape(data))
    return out


def init_params(data, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    P = {}
    for name, shape, kind in param_shapes(data):
        if kind == 'n02':
            P[name] = (0.02 + rng.standard_normal(shape)).astype(dtype)
        elif kind == 'tn02':
            x = rng.standard_normal(shape)
            bad = np.abs(x) > 2
            while bad.any():
                x[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(x) > 2
            P[name] = (0.02 + x).astype(dtype)
        elif kind == 'n05':
            P[name] = (0.05 * rng.standard_normal(shape)).astype(dtype)
        else:
            P[name] = np.full(shape, 1.0 if kind == 'one' else 0.0, dtype)
    return P


# ---------------------------------------------------------------------------- interpreter

_ACT = {
    'relu': (T.relu, lambda y, d: T.relu_bwd_from_out(y, d)),
    'lrelu': (lambda x: T.lrelu(x, 0.2), lambda y, d: T.lrelu_bwd_from_out(y, d, 0.2)),
    'softplus': (T.softplus, lambda y, d: d * (1 - np.exp(-y))),
    'sigmoid': (T.sigmoid, lambda y, d: d * y * (1 - y)),
    'tanh': (np.tanh, lambda y, d: d * (1 - y * y)),
}


def _wn_mat(V, g):
    """g * V/||V|| with the norm over all axes but the last (dense / conv)."""
    return T.wn_weight(V, g, -1)


def seq_fwd(P, layers, x, y, rnd, train, bn_updates=None):
    """Runs the layer list.  bn_updates: dict collecting moving-statistics updates of training-mode BN (sequential).
    Returns (out, caches, feature)."""
    caches, feat = [], None
    for l in layers:
        k = l[0]
        c = None
        if k == 'concat_y':
            c = x.shape[-1]
            x = np.concatenate([x, y.astype(x.dtype)], axis=1)
        elif k == 'cond_concat':
            c = x.shape[-1]
            x = T.conv_cond_concat(x, y)
        elif k == 'reshape':
            c = x.shape
            x = x.reshape((x.shape[0],) + tuple(l[1]))
        elif k == 'dense':
            c = x
            x = T.matmul(x, P[l[1] + '/kernel']) + P[l[1] + '/bias']
        elif k in ('wn_dense', 'nin'):
            c = (x, x.shape)
            x2 = x.reshape(-1, x.shape[-1])
            x = (T.matmul(x2, _wn_mat(P[l[1] + '/V'], P[l[1] + '/g'])) + P[l[1] + '/b']).reshape(x.shape[:-1] + (l[2],))
        elif k == 'conv':
            c = x
            x = T.conv2d(x, P[l[1] + '/kernel'], (l[3], l[3]), 'SAME') + P[l[1] + '/bias']
        elif k == 'wn_conv':
            c = x
            x = T.conv2d(x, _wn_mat(P[l[1] + '/V'], P[l[1] + '/g']), (l[3], l[3]), 'SAME') + P[l[1] + '/b']
        elif k == 'deconv':
            c = x
            x = T.conv2d_transpose(x, P[l[1] + '/kernel']) + P[l[1] + '/bias']
        elif k == 'wn_deconv':
            c = x
            x = T.conv2d_transpose(x, T.wn_weight(P[l[1] + '/V'], P[l[1] + '/g'], 2)) + P[l[1] + '/b']   # norm over axes 0,1,3
        elif k == 'act':
            x = _ACT[l[1]][0](x)
            c = x
        elif k == 'bn':
            g_, b_ = P[l[1] + '/gamma'], P[l[1] + '/beta']
            if train:
                x_in = x
                x, bc = T.batch_norm_train(x, g_, b_, BN_EPS)
                c = bc
                if bn_updates is not None:
                    mm, mv = bn_updates.get(l[1], (P[l[1] + '/moving_mean'], P[l[1] + '/moving_variance']))
                    cnt = x_in.size // x_in.shape[-1]
                    bn_updates[l[1]] = T.batch_norm_moving_update(mm, mv, bc[2], bc[3], cnt, BN_DECAY, fused=True)
            else:
                x = g_ * (x - P[l[1] + '/moving_mean']) / np.sqrt(P[l[1] + '/moving_variance'] + BN_EPS) + b_
        elif k == 'noise':
            x = x + rnd[l[1]]
        elif k == 'dropout':
            if l[3] or train:
                x = T.dropout(x, rnd[l[1]], l[2])
                c = True
        elif k == 'maxpool':
            shp = x.shape
            x, idx = T.maxpool2(x)
            c = (idx, shp)
        elif k == 'gmean':
            c = x.shape
            x = x.mean(axis=(1, 2))
        elif k == 'feature':
            feat = x
        caches.append(c)
    return x, caches, feat


def seq_bwd(P, layers, caches, d, y, rnd, want_params=True, dfeat=None):
    """Gradient of seq_fwd (training mode).  Returns (grads, d_input)."""
    G = {}
    for l, c in zip(reversed(layers), reversed(caches)):
        k = l[0]
        if k in ('concat_y', 'cond_concat'):
            d = d[..., :c]
        elif k == 'reshape':
            d = d.reshape(c)
        elif k == 'dense':
            if want_params:
                G[l[1] + '/kernel'], G[l[1] + '/bias'] = T.matmul(c.T, d), d.sum(0)
            d = T.matmul(d, P[l[1] + '/kernel'].T)
        elif k in ('wn_dense', 'nin'):
            x, shp = c
            x2, d2 = x.reshape(-1, shp[-1]), d.reshape(-1, d.shape[-1])
            V, g = P[l[1] + '/V'], P[l[1] + '/g']
            if want_params:
                G[l[1] + '/V'], G[l[1] + '/g'] = T.wn_weight_bwd(V, g, T.matmul(x2.T, d2))
                G[l[1] + '/b'] = d2.sum(0)
            d = T.matmul(d2, _wn_mat(V, g).T).reshape(shp)
        elif k in ('conv', 'wn_conv'):
            s = (l[3], l[3])
            if k == 'conv':
                W = P[l[1] + '/kernel']
                if want_params:
                    G[l[1] + '/kernel'] = T.conv2d_bwd_filter(c, d, W.shape, s, 'SAME')
                    G[l[1] + '/bias'] = d.sum(axis=(0, 1, 2))
            else:
                V, g = P[l[1] + '/V'], P[l[1] + '/g']
                W = _wn_mat(V, g)
                if want_params:
                    G[l[1] + '/V'], G[l[1] + '/g'] = T.wn_weight_bwd(V, g, T.conv2d_bwd_filter(c, d, V.shape, s, 'SAME'))
                    G[l[1] + '/b'] = d.sum(axis=(0, 1, 2))
            d = T.conv2d_bwd_input(c.shape, W, d, s, 'SAME')
        elif k in ('deconv', 'wn_deconv'):
            if k == 'deconv':
                W = P[l[1] + '/kernel']
                if want_params:
                    G[l[1] + '/kernel'] = T.conv2d_transpose_bwd_filter(c, d, W.shape)
                    G[l[1] + '/bias'] = d.sum(axis=(0, 1, 2))
            else:
                V, g = P[l[1] + '/V'], P[l[1] + '/g']
                W = T.wn_weight(V, g, 2)
                if want_params:
                    G[l[1] + '/V'], G[l[1] + '/g'] = T.wn_weight_bwd(V, g, T.conv2d_transpose_bwd_filter(c, d, V.shape), 2)
                    G[l[1] + '/b'] = d.sum(axis=(0, 1, 2))
            d = T.conv2d_transpose_bwd_input(W, d)
        elif k == 'act':
            d = _ACT[l[1]][1](c, d)
        elif k == 'bn':
            d, dg, db = T.batch_norm_train_bwd(d, P[l[1] + '/gamma'], c)
            if want_params:
                G[l[1] + '/gamma'], G[l[1] + '/beta'] = dg, db
        elif k == 'dropout':
            if c:
                d = T.dropout_bwd(d, rnd[l[1]], l[2])
        elif k == 'maxpool':
            d = T.maxpool2_bwd(d, c[0], c[1])
        elif k == 'gmean':
            d = T.global_avgpool_bwd(d, c)
        elif k == 'feature' and dfeat is not None:
            d = d + dfeat
    return G, d


def commit_bn(P, bn_updates):
    for name, (mm, mv) in bn_updates.items():
        P[name + '/moving_mean'], P[name + '/moving_variance'] = mm, mv
