"""ORACLE — test infrastructure only (see oracle/tf_ops.py header).  PARITY UNPINNED (TensorFlow absent, no reference fixtures).

NumPy (float64-capable) restatement of the loss variants of Training/train_base.py:156-574 — SURVEY §8f N4: `_loss_BGAN`,
`_loss_GoodBadGAN`, `_loss_GoodRegGAN`, `_loss_GoodRegGAN_cifar10`, `_loss_GoodRegGAN_BS`, `_loss_GoodRegGAN_BS_cifar10`,
`_loss_GoodRegBadGAN` — and of NN_Base._minibatch_discrimination (Model/modle_base.py:110-128).  No trainer of the reference
repository calls these losses (Train_goodGAN.py uses `_loss_GAN` only), so they are restated term by term with the file:line of every
expression.  Each loss function returns (out, grads):
   out   — the reference's return value (same nesting and order);
   grads — what each solver differentiates: d_loss wrt the D logits, gG_loss wrt D_fake_logits, bG_loss wrt the bad generator's
           features, c_loss wrt every classifier logit tensor.
"""
import numpy as np

from . import tf_ops as T


def true_fake_unl(logits):
    """-0.5 mean(lse) + 0.5 mean(softplus(lse))   (train_base.py:162-163, 226-227, 296-297)."""
    lse = T._logsumexp(logits)
    val = -0.5 * lse.mean() + 0.5 * T.softplus(lse).mean()
    grad = ((-0.5 + 0.5 * T.sigmoid(lse)) / logits.shape[0])[:, None] * T._softmax(logits)
    return val, grad


def true_fake_fake(logits):
    """0.5 mean(softplus(lse))   (train_base.py:166, 228, 298)."""
    lse = T._logsumexp(logits)
    val = 0.5 * T.softplus(lse).mean()
    return val, (0.5 * T.sigmoid(lse) / logits.shape[0])[:, None] * T._softmax(logits)


def sqdiff_rows(a, b):
    """mean_n sum_k (a - b)^2   (train_base.py:299); grads wrt a and b."""
    d = a - b
    return np.mean(np.sum(np.square(d), axis=1)), 2 * d / a.shape[0], -2 * d / a.shape[0]


def minibatch_discrimination(x, w, b, dim_per_kernel=5):
    """Model/modle_base.py:110-128: f[i,k] = sum_j exp(-sum_d |A[i,k,d] - A[j,k,d]|) + b[k], A = reshape(x @ W, [N, K, D])."""
    n = x.shape[0]
    k = b.shape[0]
    a = (x @ w).reshape(n, k, dim_per_kernel)
    diff = a[:, None, :, :] - a[None, :, :, :]                  # [i, j, k, d]
    e = np.exp(-np.abs(diff).sum(axis=3))                       # [i, j, k]
    return e.sum(axis=1) + b, (a, diff, e)


def minibatch_discrimination_bwd(x, w, cache, df):
    """gradients of the layer above wrt x, W, b given d loss / d f."""
    a, diff, e = cache
    n, k, d = a.shape
    gg = df[:, None, :] + df[None, :, :]                        # [i, j, k]: f_i and f_j both contain e_ij
    da = -(gg * e)[:, :, :, None] * np.sign(diff)               # d/d a_i
    da = da.sum(axis=1).reshape(n, k * d)
    return da @ w.T, x.T @ da, df.sum(axis=0)


def _d_losses(D):
    """train_base.py:193-196 and alike: (d_loss, [real, fake, unl] unweighted, grads wrt the three logit tensors)."""
    _, d_real, _, d_fake, _, d_unl = D
    lr, gr = T.bce_mean(d_real, np.ones_like(d_real))
    lf, gf = T.bce_mean(d_fake, np.zeros_like(d_fake))
    lu, gu = T.bce_mean(d_unl, np.zeros_like(d_unl))
    return lr + 0.5 * lf + 0.5 * lu, (lr, lf, lu), (gr, 0.5 * gf, 0.5 * gu)


def _gG(D):
    lg, gg = T.bce_mean(D[3], np.ones_like(D[3]))               # 1/2 BCE(D_fake, 1)   (train_base.py:199)
    return 0.5 * lg, 0.5 * gg


def _c_good_terms(c_real, c_unl, c_gfake, d_unl, y_l_c, y_g):
    out = {}
    out['real'] = T.softmax_ce_mean(c_real, y_l_c)
    out['gfake'] = T.softmax_ce_mean(c_gfake, y_g) if c_gfake is not None else (0.0, None)
    out['unl'] = T.c_unl_loss(c_unl, d_unl)
    out['H'] = T.entropy(c_unl)
    out['bal'] = T.balance_entropy(c_unl)
    return out


def loss_BGAN(C, Y, Lambda=None):
    """train_base.py:156-184."""
    y_l_c = Y[0]
    c_real, c_unl, c_fake, feat_real, feat_unl, feat_fake = C
    real = T.softmax_ce_mean(c_real, y_l_c)
    bad_unl = true_fake_unl(c_unl)
    h, bal = T.entropy(c_unl), T.balance_entropy(c_unl)
    fake = true_fake_fake(c_fake)
    c_loss = real[0] + bad_unl[0] + fake[0] + 0.1 * h[0] + 1e-3 * bal[0]
    fm = T.feature_match(feat_fake, feat_unl)
    pt = T.pull_away_masked(feat_fake)
    g_loss = fm[0] + pt[0]
    grads = {'c_real': real[1], 'c_unl': bad_unl[1] + 0.1 * h[1] + 1e-3 * bal[1], 'c_fake': fake[1], 'feat_fake': fm[1] + pt[1]}
    return (g_loss, c_loss), grads


def _good_bad(D, C, Y, Lambda, perturb):
    """train_base.py:186-240 (_loss_GoodBadGAN) and :519-574 (_loss_GoodRegBadGAN: + the perturbation term)."""
    if perturb:
        c_real, c_unl, _c_unl_d, c_gfake, c_bfake, c_pert, _f_real, f_unl, f_bfake, _f_pert = C
    else:
        c_real, c_unl, _c_unl_d, c_gfake, c_bfake, _f_real, f_unl, f_bfake = C
    y_g, y_l_c = Y
    d_loss, _, dg = _d_losses(D)
    gG, ggG = _gG(D)
    fm = T.feature_match(f_bfake, f_unl)
    pt = T.pull_away_unmasked(f_bfake)                           # 0.8 * mean cos (:204-207, :536-539)
    bG = fm[0] + pt[0]
    t = _c_good_terms(c_real, c_unl, c_gfake, D[5], y_l_c, y_g)
    lam1 = Lambda[0]
    bad_unl, bfake = true_fake_unl(c_unl), true_fake_fake(c_bfake)
    c_loss = t['real'][0] + 0.3 * t['H'][0] + 0.01 * 0.5 * t['unl'][0] + lam1 * t['gfake'][0] + 1e-3 * t['bal'][0] + bad_unl[0] + bfake[0]
    g_unl = 0.3 * t['H'][1] + 0.005 * t['unl'][1] + 1e-3 * t['bal'][1] + bad_unl[1]
    grads = {'d_real': dg[0], 'd_fake': dg[1], 'd_unl': dg[2], 'gG_d_fake': ggG, 'feat_bfake': fm[1] + pt[1], 'c_real': t['real'][1],
             'c_unl': g_unl, 'c_gfake': lam1 * t['gfake'][1], 'c_bfake': bfake[1]}
    if perturb:
        pv, pa, pb = sqdiff_rows(c_pert, c_bfake)
        c_loss += 1e-3 * pv
        grads['c_pert'] = 1e-3 * pa
        grads['c_bfake'] = grads['c_bfake'] + 1e-3 * pb
    return (d_loss, gG, bG, c_loss), grads


def loss_GoodBadGAN(D, C, Y, Lambda):
    return _good_bad(D, C, Y, Lambda, False)


def loss_GoodRegBadGAN(D, C, Y, Lambda):
    return _good_bad(D, C, Y, Lambda, True)


def _good_reg(D, C, Y, Lambda, variant, fast_mode=False):
    """_loss_GoodRegGAN (:242-305), _cifar10 (:307-374), _BS (:376-442), _BS_cifar10 (:444-515)."""
    y_g, y_l_c = Y
    c_rep = None
    if variant == 'plain':
        c_real, c_unl, _cud, c_gfake, c_bfake, c_pert, _fr, f_unl, f_bfake, _fp = C
        f_unl_bg = f_unl
    elif variant == 'cifar10':
        c_real, c_unl, _cud, c_gfake, c_bfake, c_pert, _fr, f_unl, f_bfake, _fp, c_rep = C
        f_unl_bg = f_unl
    elif variant == 'BS':
        c_real, c_unl, _cud, c_gfake, c_bfake, c_pert, _c_unl_bg, _fr, _f_unl, f_bfake, _fp, f_unl_bg = C
    else:
        c_real, c_unl, _cud, c_gfake, c_bfake, c_pert, _c_unl_bg, _fr, _f_unl, f_bfake, _fp, f_unl_bg, c_rep = C
    d_loss, (d_r, d_f, d_u), dg = _d_losses(D)
    gG, ggG = _gG(D)
    fm = T.feature_match(f_bfake, f_unl_bg)
    if variant in ('plain', 'cifar10'):
        pt = T.pull_away_masked(f_bfake)                         # :278-284, :326-334
        bG, g_feat = fm[0] + pt[0], fm[1] + pt[1]
    else:
        bG, g_feat = fm[0], fm[1]                                # :404-405, :469 (the pull-away term is commented out)
    t = _c_good_terms(c_real, c_unl, None if (fast_mode and variant == 'BS_cifar10') else c_gfake, D[5], y_l_c, y_g)
    w_h = {'plain': 0.3, 'cifar10': 0.3, 'BS': 1e-5, 'BS_cifar10': 1e-7}[variant]       # :291, :346, :421, :489
    lam1, lam2, lam3 = Lambda[:3]
    confid, unl, bal = w_h * t['H'][0], 0.01 * 0.5 * t['unl'][0], 1e-3 * t['bal'][0]
    c_gG = confid + unl + lam1 * t['gfake'][0] + bal
    bad_unl, bfake = true_fake_unl(c_unl), true_fake_fake(c_bfake)
    pv, pa, pb = sqdiff_rows(c_pert, c_bfake)
    pert = 1e-3 * pv
    c_bG = bad_unl[0] + bfake[0] + pert
    c_loss = t['real'][0] + lam2 * c_gG + lam3 * c_bG
    g_unl = lam2 * (w_h * t['H'][1] + 0.005 * t['unl'][1] + 1e-3 * t['bal'][1]) + lam3 * bad_unl[1]
    grads = {'d_real': dg[0], 'd_fake': dg[1], 'd_unl': dg[2], 'gG_d_fake': ggG, 'feat_bfake': g_feat, 'c_real': t['real'][1],
             'c_gfake': (lam2 * lam1 * t['gfake'][1]) if t['gfake'][1] is not None else np.zeros_like(c_gfake),
             'c_bfake': lam3 * (bfake[1] + 1e-3 * pb), 'c_pert': lam3 * 1e-3 * pa}
    c_list = [None, t['real'][0], c_gG, confid, unl, bal, t['gfake'][0], c_bG, bad_unl[0], bfake[0], pert]
    if variant == 'cifar10':
        mv, ga, gb = T.mse_mean(c_unl, c_rep)                    # :363  lambda_4 * MSE(C_unl, C_unl_rep)
        unsup = Lambda[3] * mv
        c_loss += unsup
        g_unl = g_unl + Lambda[3] * ga
        grads['c_rep'] = Lambda[3] * gb
        c_list.append(unsup)
    elif variant == 'BS_cifar10':
        unsup = Lambda[3]                                        # :503  c_loss_unsup = lambda_4 (a constant: no gradient)
        c_loss += unsup
        grads['c_rep'] = np.zeros_like(c_rep)
        c_list.append(unsup)
    grads['c_unl'] = g_unl
    c_list[0] = c_loss
    return ([d_loss, d_r, 0.5 * d_f, 0.5 * d_u], gG, bG, c_list), grads


def loss_GoodRegGAN(D, C, Y, Lambda):
    return _good_reg(D, C, Y, Lambda, 'plain')


def loss_GoodRegGAN_cifar10(D, C, Y, Lambda):
    return _good_reg(D, C, Y, Lambda, 'cifar10')


def loss_GoodRegGAN_BS(D, C, Y, Lambda):
    return _good_reg(D, C, Y, Lambda, 'BS')


def loss_GoodRegGAN_BS_cifar10(D, C, Y, Lambda, fast_mode=False):
    return _good_reg(D, C, Y, Lambda, 'BS_cifar10', fast_mode)
