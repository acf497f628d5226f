"""Test reference for the descriptor builders of csrc/geom.cpp (tg_conv2d_desc_* ... in include/tg_kernels.h): the Python builders
the package used in round 1, kept here as an independent restatement — tests/test_geom.py compares the C ABI's descriptors with
these field by field over a sweep of shapes.

Padding arithmetic is TensorFlow's (SAME: total = max((ceil(in/s)-1)*s + k - in, 0),
before = total//2, the extra pixel goes after) — SURVEY App. C.1/C.2.
"""
from tg.lib import IgemmDesc, ACT


def pad32(c):
    return (c + 31) // 32 * 32


def same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def out_size(n, k, s, padding):
    if padding == 'SAME':
        return same_pad(n, k, s)[0], same_pad(n, k, s)[1]
    return (n - k) // s + 1, 0


def _desc(n_img, h_in, w_in, ld_in, h_v, w_v, s, h_out, w_out, ld_out, os_, oo, c_out, n_store,
          taps, w_sn, w_st, act=None, alpha=0.2):
    d = IgemmDesc()
    d.n_img, d.h_in, d.w_in, d.ld_in = n_img, h_in, w_in, ld_in
    d.h_v, d.w_v, d.s_y, d.s_x = h_v, w_v, s[0], s[1]
    d.h_out, d.w_out, d.ld_out = h_out, w_out, ld_out
    d.os_y, d.os_x, d.oo_y, d.oo_x = os_[0], os_[1], oo[0], oo[1]
    d.c_out, d.n_store, d.n_taps = c_out, n_store, len(taps)
    assert 1 <= len(taps) <= 25
    for i, (dy, dx, tw) in enumerate(taps):
        d.dy[i], d.dx[i], d.tapw[i] = dy, dx, tw
    d.w_sn, d.w_st = w_sn, w_st
    d.act, d.alpha = ACT[act], alpha
    return d


def conv_fwd(n, h, w, ld_in, c_out, k, stride, padding, ld_out=None, n_store=None, act=None, alpha=0.2):
    """y = conv(x, W) with W re-laid out as [c_out][k*k][ld_in] (OTI)."""
    ho, pt = out_size(h, k, stride, padding)
    wo, pl = out_size(w, k, stride, padding)
    taps = [(ky - pt, kx - pl, ky * k + kx) for ky in range(k) for kx in range(k)]
    ld_out = c_out if ld_out is None else ld_out
    n_store = c_out if n_store is None else n_store
    return _desc(n, h, w, ld_in, ho, wo, (stride, stride), ho, wo, ld_out, (1, 1), (0, 0), c_out, n_store,
                 taps, k * k * ld_in, ld_in, act, alpha)


def conv_dgrad(n, h, w, c_in_pad, ld_dy, k, stride, padding, ld_out=None, n_store=None):
    """dx[n,h,w,:] from dy of conv_fwd; W in padded HWIO [k*k][c_in_pad][ld_dy].
    Returns one descriptor per input parity class (1 for stride 1, 4 for stride 2)."""
    ho, pt = out_size(h, k, stride, padding)
    wo, pl = out_size(w, k, stride, padding)
    ld_out = c_in_pad if ld_out is None else ld_out
    n_store = c_in_pad if n_store is None else n_store
    descs = []
    for py in range(stride):
        for px in range(stride):
            taps = []
            for ky in range(k):
                if (py + pt - ky) % stride:
                    continue
                for kx in range(k):
                    if (px + pl - kx) % stride:
                        continue
                    taps.append(((py + pt - ky) // stride, (px + pl - kx) // stride, ky * k + kx))
            hv = (h - py + stride - 1) // stride
            wv = (w - px + stride - 1) // stride
            if not taps or hv <= 0 or wv <= 0:
                continue  # caller must zero-fill such outputs (does not occur for the nets here)
            descs.append(_desc(n, ho, wo, ld_dy, hv, wv, (1, 1), h, w, ld_out, (stride, stride), (py, px),
                               c_in_pad, n_store, taps, ld_dy, c_in_pad * ld_dy))
    return descs


def deconv_fwd(n, h, w, ld_in, c_out_pad, k=5, stride=2, ld_out=None, n_store=None, act=None):
    """tf conv2d_transpose 'same': out[s*i + k - pt] += in[i] W[k]; filter padded to
    [k*k][c_out_pad][ld_in].  One descriptor per output parity."""
    _, pt, _ = same_pad(h * stride, k, stride)
    _, pl, _ = same_pad(w * stride, k, stride)
    ld_out = c_out_pad if ld_out is None else ld_out
    n_store = c_out_pad if n_store is None else n_store
    descs = []
    for py in range(stride):
        for px in range(stride):
            taps = [((py + pt - ky) // stride, (px + pl - kx) // stride, ky * k + kx)
                    for ky in range(k) if (py + pt - ky) % stride == 0
                    for kx in range(k) if (px + pl - kx) % stride == 0]
            descs.append(_desc(n, h, w, ld_in, h, w, (1, 1), h * stride, w * stride, ld_out, (stride, stride),
                               (py, px), c_out_pad, n_store, taps, ld_in, c_out_pad * ld_in, act))
    return descs


def deconv_fwd_merged(n, h, w, ld_in, c_out, ld_out, n_store=None, act=None):
    """The same transposed conv (5x5, stride 2) as ONE 3x3 problem over the input grid whose GEMM columns are (output parity,
    channel): the parities have 9/6/6/4 of the 25 taps, so four separate sub-problems leave the matrix pipes unevenly loaded;
    here every workgroup does 9 taps and the missing ones are zero weights (36/25 of the arithmetic, but balanced — and for a
    3-channel image layer the four parities share one 32-column tile instead of padding 3 to 32 four times).
    Returns (descriptor, n_group, tapmap[36]) — tapmap[g*9 + t9] = index of the 5x5 tap, -1 where the parity has none."""
    k, stride = 5, 2
    _, pt, _ = same_pad(h * stride, k, stride)
    _, pl, _ = same_pad(w * stride, k, stride)
    n_group = c_out if c_out % 4 == 0 else c_out        # channels per parity group (vector stores need a multiple of 4)
    n_pad = pad32(4 * n_group)
    tapmap = [-1] * 36
    for py in range(stride):
        for px in range(stride):
            g = py * stride + px
            for ky in range(k):
                if (py + pt - ky) % stride:
                    continue
                for kx in range(k):
                    if (px + pl - kx) % stride:
                        continue
                    dy, dx = (py + pt - ky) // stride, (px + pl - kx) // stride
                    assert -1 <= dy <= 1 and -1 <= dx <= 1
                    tapmap[g * 9 + (dy + 1) * 3 + (dx + 1)] = ky * k + kx
    taps = [(dy, dx, (dy + 1) * 3 + (dx + 1)) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    n_store = c_out if n_store is None else n_store
    d = _desc(n, h, w, ld_in, h, w, (1, 1), h * stride, w * stride, ld_out, (stride, stride), (0, 0), n_pad, n_store, taps,
              9 * ld_in, ld_in, act)
    d.n_group = n_group
    return d, n_group, tapmap


def deconv_dgrad(n, h, w, c_in_pad, ld_dy, k=5, stride=2, ld_out=None, n_store=None):
    """d(in) of deconv_fwd = strided conv of dy [n,2h,2w,ld_dy]; W as [k*k][c_in_pad][ld_dy]
    (per-tap transpose of the filter)."""
    _, pt, _ = same_pad(h * stride, k, stride)
    _, pl, _ = same_pad(w * stride, k, stride)
    taps = [(ky - pt, kx - pl, ky * k + kx) for ky in range(k) for kx in range(k)]
    ld_out = c_in_pad if ld_out is None else ld_out
    n_store = c_in_pad if n_store is None else n_store
    return _desc(n, h * stride, w * stride, ld_dy, h, w, (stride, stride), h, w, ld_out, (1, 1), (0, 0),
                 c_in_pad, n_store, taps, ld_dy, c_in_pad * ld_dy)


def conv_wgrad(n, h, w, ld_in, c_out_pad, k, stride, padding, ld_dy=None):
    """slab[t][c][n] geometry of conv_fwd's filter gradient (in = x, dout = dy)."""
    ld_dy = c_out_pad if ld_dy is None else ld_dy
    return conv_fwd(n, h, w, ld_in, c_out_pad, k, stride, padding, ld_out=ld_dy, n_store=c_out_pad)


def deconv_wgrad(n, h, w, ld_dy, c_in_pad, k=5, stride=2, ld_x=None):
    """filter gradient of deconv_fwd, laid out [t][c_out(ld_dy)][c_in_pad]: the gathered tensor
    is dy [n,2h,2w,ld_dy], 'dout' is the deconv input x [n,h,w,ld_x]."""
    ld_x = c_in_pad if ld_x is None else ld_x
    return conv_fwd(n, h * stride, w * stride, ld_dy, c_in_pad, k, stride, 'SAME', ld_out=ld_x, n_store=c_in_pad)


def dense_fwd(m, ld_in, c_out, ld_out=None, n_store=None, act=None, w_sn=None):
    """y[m, c_out] = x[m, ld_in] @ Wt[c_out][ld_in]^T (1 tap)."""
    ld_out = c_out if ld_out is None else ld_out
    n_store = c_out if n_store is None else n_store
    return _desc(m, 1, 1, ld_in, 1, 1, (1, 1), 1, 1, ld_out, (1, 1), (0, 0), c_out, n_store,
                 [(0, 0, 0)], ld_in if w_sn is None else w_sn, 0, act)


def dense_fwd_splitk(m, k_dim, n_out, splits):
    """y = x @ Wt.T with the reduction dimension cut into `splits` ranges: sub-problem s reads x[:, s*k/splits:(s+1)*k/splits]
    (the input viewed as `splits` pixels of k/splits channels) against the matching slice of Wt[n][k] and writes partial
    products to part[m][s][n]; tg_splitk_reduce_f32 adds them up.  For skinny products (few rows, long K: the ZCA matmul of
    130 / 250 images has only ~150 tiles of 96 K-steps) this multiplies the number of workgroups by `splits`."""
    assert k_dim % (32 * splits) == 0
    kc = k_dim // splits
    return [_desc(m, 1, splits, kc, 1, 1, (1, 1), 1, splits, n_out, (1, 1), (0, s), n_out, n_out, [(0, s, s)], k_dim, kc)
            for s in range(splits)]
