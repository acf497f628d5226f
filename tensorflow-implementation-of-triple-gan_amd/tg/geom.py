"""tg_igemm_desc builders: thin callers of the C ABI (csrc/geom.cpp, include/tg_kernels.h "descriptor builders") — the padding
arithmetic, tap tables, pixel splits and tile rules live in the library, in one copy, for every host language."""
import ctypes as C

from . import lib
from .lib import ACT, IgemmDesc


def pad32(c):
    return (c + 31) // 32 * 32


def _same(padding):
    if padding not in ('SAME', 'VALID'):
        raise ValueError("padding must be 'SAME' or 'VALID', got %r" % (padding,))
    return 1 if padding == 'SAME' else 0


def _opt(v, neg=False):
    """None -> the ABI's "use the default" value."""
    return (-1 if neg else 0) if v is None else int(v)


def _many(fn, *args):
    arr = (IgemmDesc * 4)()
    n = C.c_int32()
    lib.call(fn, *args, C.cast(arr, C.c_void_p), C.byref(n))
    return [arr[i] for i in range(n.value)]


def conv_fwd(n, h, w, ld_in, c_out, k, stride, padding, ld_out=None, n_store=None, act=None, alpha=0.2):
    """y = conv(x, W) with W re-laid out as [c_out][k*k][ld_in] (OTI)."""
    d = IgemmDesc()
    lib.call('tg_conv2d_desc_fwd', n, h, w, ld_in, c_out, k, stride, _same(padding), _opt(ld_out), _opt(n_store, True), ACT[act], alpha, C.byref(d))
    return d


def conv_dgrad(n, h, w, c_in_pad, ld_dy, k, stride, padding, ld_out=None, n_store=None):
    """dx[n,h,w,:] from dy of conv_fwd; W in padded HWIO [k*k][c_in_pad][ld_dy].  One descriptor per input parity class (1 for
    stride 1, 4 for stride 2)."""
    return _many('tg_conv2d_desc_dgrad', n, h, w, c_in_pad, ld_dy, k, stride, _same(padding), _opt(ld_out), _opt(n_store, True))


def conv_wgrad(n, h, w, ld_in, c_out_pad, k, stride, padding, ld_dy=None):
    """slab[t][c][n] geometry of conv_fwd's filter gradient (in = x, dout = dy)."""
    d = IgemmDesc()
    lib.call('tg_conv2d_desc_wgrad', n, h, w, ld_in, c_out_pad, k, stride, _same(padding), _opt(ld_dy), C.byref(d))
    return d


def deconv_fwd(n, h, w, ld_in, c_out_pad, ld_out=None, n_store=None, act=None):
    """tf conv2d_transpose 5x5 s2 'same'; filter padded to [25][c_out_pad][ld_in].  One descriptor per output parity."""
    return _many('tg_deconv5x5s2_desc_fwd', n, h, w, ld_in, c_out_pad, _opt(ld_out), _opt(n_store, True), ACT[act])


def deconv_fwd_merged(n, h, w, ld_in, c_out, ld_out, n_store=None, act=None):
    """the same transposed conv as ONE 3x3 problem over (output parity, channel) columns.  Returns (descriptor, n_group, tapmap[36])."""
    d, ng, tapmap = IgemmDesc(), C.c_int32(), (C.c_int32 * 36)()
    lib.call('tg_deconv5x5s2_desc_fwd_merged', n, h, w, ld_in, c_out, ld_out, _opt(n_store, True), ACT[act], C.byref(d), C.byref(ng), tapmap)
    return d, ng.value, list(tapmap)


def deconv_dgrad(n, h, w, c_in_pad, ld_dy, ld_out=None, n_store=None):
    """d(in) of deconv_fwd = strided conv of dy [n,2h,2w,ld_dy]; W as [25][c_in_pad][ld_dy] (per-tap transpose of the filter)."""
    d = IgemmDesc()
    lib.call('tg_deconv5x5s2_desc_dgrad', n, h, w, c_in_pad, ld_dy, _opt(ld_out), _opt(n_store, True), C.byref(d))
    return d


def deconv_wgrad(n, h, w, ld_dy, c_in_pad, ld_x=None):
    """filter gradient of deconv_fwd, laid out [t][c_out(ld_dy)][c_in_pad]: gathered tensor = dy, 'dout' = the deconv input x."""
    d = IgemmDesc()
    lib.call('tg_deconv5x5s2_desc_wgrad', n, h, w, ld_dy, c_in_pad, _opt(ld_x), C.byref(d))
    return d


def dense_fwd(m, ld_in, c_out, ld_out=None, n_store=None, act=None, w_sn=None):
    """y[m, c_out] = x[m, ld_in] @ Wt[c_out][ld_in]^T (1 tap)."""
    d = IgemmDesc()
    lib.call('tg_dense_desc', m, ld_in, c_out, _opt(ld_out), _opt(n_store, True), ACT[act], _opt(w_sn), C.byref(d))
    return d


def dense_fwd_splitk(m, k_dim, n_out, splits):
    """y = x @ Wt.T with the reduction cut into `splits` sub-problems of one launch (partials part[m][s][n]; tg_splitk_reduce_f32)."""
    arr = (IgemmDesc * 4)()
    lib.call('tg_dense_splitk_desc', m, k_dim, n_out, splits, C.cast(arr, C.c_void_p))
    return [arr[i] for i in range(splits)]


def wgrad_splits(desc, bf16=False):
    """pixel splits tg_wgrad_f32 (bf16: tg_wgrad_bf16) should be given for `desc`."""
    return lib.call('tg_wgrad_splits_bf16' if bf16 else 'tg_wgrad_splits', C.byref(desc))


def wgrad_slab_floats(desc, n_split):
    return lib.call('tg_wgrad_workspace_bytes', C.byref(desc), n_split) // 4


def colsum_supported(desc, seg_rows):
    """may tg_igemm_colsum_* / tg_igemm_actsum_* take (desc, seg_rows)?"""
    return bool(lib.call('tg_igemm_colsum_supported', C.byref(desc), (C.c_int32 * len(seg_rows))(*seg_rows), len(seg_rows)))
