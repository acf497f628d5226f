"""GPU parity of the MFMA implicit-GEMM family (tg_igemm_f32 / tg_wgrad_f32) against the
oracle's conv / transposed-conv / dense, through the C ABI.

Tolerance (round 4): element-wise |got - ref| <= TOL * S, ref = the oracle evaluated in FLOAT64 on the same operands, S = the same
operator applied to the operands' absolute values = the ACTUAL sum |a||b| behind that output (+ |bias|).  fp32 accumulation of K
random-sign products in any order errs by ~0.6 u * S rms (u = 2^-24), ~3 u * S at the five-sigma tail of a million outputs; TOL = 1e-6
~ 17 u leaves a factor five.  A kernel that rounds its operands to bf16 misses this bound by two orders of magnitude
(test_tolerance_tells_fp32_from_bf16_operands and the negative controls inside the layer tests) — the previous max-based bound
(3e-5 * max|x| * max|w| * K) accepted it."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T

pytestmark = pytest.mark.gpu


def _tg():
    from tg import lib, geom
    lib.load()
    return lib, geom


@pytest.fixture(autouse=True)
def _halo_kernel_wherever_it_applies():
    """kernel tests run the halo-tiled 3x3 kernel on every shape it accepts (policy 1); the default policy routes launches too small
    to fill whole rounds of one workgroup per CU to the generic kernel (include/tg_kernels.h: tg_conv3x3_policy)."""
    lib, _ = _tg()
    was = lib.call('tg_conv3x3_policy', 1)
    yield
    lib.call('tg_conv3x3_policy', was)


def _takes_halo_kernel(prec, h, w, ci_p, co_p, k, s, pad):
    return (k == 3 and s == 1 and pad == 'SAME' and w in (16, 32, 64) and h % (256 // w) == 0 and ci_p % (64 if prec == 'bf16' else 32) == 0
            and co_p % 128 == 0)


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def padc(x, ld):
    out = np.zeros(x.shape[:-1] + (ld,), np.float32)
    out[..., :x.shape[-1]] = x
    return out


TOL = 1e-6          # of the per-output sum |a||b| (module docstring)
f8 = lambda a: np.asarray(a, np.float64)


def worst(got, ref64, sabs):
    """max over the outputs of |got - ref| / (TOL * sum|a||b|): <= 1 passes."""
    err = np.abs(f8(got) - ref64)
    return float((err / (TOL * f8(sabs) + 1e-30)).max())


def close(got, ref64, sabs):
    r = worst(got, ref64, sabs)
    assert r <= 1.0, "error is %.2f x the fp32 accumulation bound (TOL %.0e of the per-output sum |a||b|)" % (r, TOL)


def rejected(got, ref64, sabs, factor=10.0):
    """the bound tells `got` from `ref64` with room to spare (negative controls)."""
    return worst(got, ref64, sabs) > factor


# 'bf16': the tg_*_bf16 variants (operands rounded to bfloat16 inside the kernel, fp32 accumulation) against the oracle
# run on operands rounded the same way (T.bf16_round) — the products are then exact in fp32 and the tolerance is the
# fp32 accumulation-order bound of the f32 case.
PRECS = ['f32', 'bf16']


def _q(prec, x):
    return T.bf16_round(x) if prec == 'bf16' else x


CONV_CASES = [  # n,h,w,cin,cout,k,stride,pad
    (3, 8, 8, 32, 32, 3, 1, 'SAME'),
    (5, 16, 16, 64, 128, 3, 1, 'SAME'),
    (2, 32, 32, 13, 32, 3, 1, 'SAME'),      # cond-concat channel count, padded to 32
    (3, 32, 32, 42, 64, 3, 2, 'SAME'),      # stride 2, asymmetric SAME
    (4, 8, 8, 96, 64, 3, 1, 'VALID'),
    (2, 6, 6, 64, 96, 1, 1, 'SAME'),        # NiN
    (7, 9, 7, 40, 32, 3, 2, 'SAME'),        # ragged sizes, M not a tile multiple
    (3, 8, 8, 138, 138, 3, 1, 'SAME'),      # 160 padded channels both ways: 64-wide tiles whose last one overhangs (columns and reduction rows)
    (2, 16, 16, 266, 96, 3, 2, 'SAME'),     # 288 = 4.5 x 64 reduction channels, 96 = 1.5 x 64 columns
    # classifier-shaped 3x3 layers: with bf16 operands the forward pass and the input gradient take the halo-tiled kernel
    # (csrc/conv3x3_bf16.hip: widths 16 / 32 / 64, 64 | channels in, 128 | channels out)
    (3, 32, 32, 128, 128, 3, 1, 'SAME'),
    (2, 16, 16, 128, 256, 3, 1, 'SAME'),
    (1, 64, 64, 64, 128, 3, 1, 'SAME'),
    (2, 16, 16, 64, 120, 3, 1, 'SAME'),      # 120 logical output channels in 128 padded ones on the halo kernels (zero filter rows / bias beyond 120; pads come out exactly 0)
]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("n,h,w,cin,cout,k,s,pad", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(n, h, w, cin, cout, k, s, pad, prec):
    lib, geom = _tg()
    q = lambda a: _q(prec, a)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((k, k, cin, cout)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    ci_p, co_p = geom.pad32(cin), geom.pad32(cout)
    y_ref = T.lrelu(T.conv2d(f8(q(x)), f8(q(wt)), (s, s), pad) + f8(bias))
    y_abs = T.conv2d(np.abs(f8(q(x))), np.abs(f8(q(wt))), (s, s), pad) + np.abs(f8(bias))
    other = T.bf16_round if prec == 'f32' else (lambda a: a)          # negative control: the oracle on the OTHER operand type
    y_other = T.lrelu(T.conv2d(f8(other(x)), f8(other(wt)), (s, s), pad) + f8(bias))
    ho, wo = y_ref.shape[1:3]

    # forward: OTI weights [co_p][k*k][ci_p]
    w_oti = np.zeros((co_p, k * k, ci_p), np.float32)
    w_oti[:cout, :, :cin] = wt.reshape(k * k, cin, cout).transpose(2, 0, 1)
    xd, wd, bd = dev(padc(x, ci_p)), dev(w_oti), dev(padc(bias, co_p))
    yd = torch.full((n, ho, wo, co_p), 7.0, device='cuda')
    d = geom.conv_fwd(n, h, w, ci_p, co_p, k, s, pad, act='lrelu')
    halo0 = lib.call('tg_conv3x3_launches')
    lib.call_igemm("tg_igemm_" + prec, d, lib.ptr(xd), lib.ptr(wd), lib.ptr(bd), lib.ptr(yd), lib.cur_stream())
    assert lib.call('tg_conv3x3_launches') - halo0 == int(_takes_halo_kernel(prec, h, w, ci_p, co_p, k, s, pad))
    y = yd.cpu().numpy()
    close(y[..., :cout], y_ref, y_abs)
    assert rejected(y[..., :cout], y_other, y_abs)          # the bound separates fp32 operands from bf16-rounded ones
    assert (y[..., cout:] == 0).all()

    # input gradient: padded HWIO weights [k*k][ci_p][co_p]
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = T.conv2d_bwd_input(x.shape, f8(q(wt)), f8(q(dy)), (s, s), pad)
    dx_abs = T.conv2d_bwd_input(x.shape, np.abs(f8(q(wt))), np.abs(f8(q(dy))), (s, s), pad)
    w_hwio = np.zeros((k * k, ci_p, co_p), np.float32)
    w_hwio[:, :cin, :cout] = wt.reshape(k * k, cin, cout)
    dyd, whd = dev(padc(dy, co_p)), dev(w_hwio)
    dxd = torch.full((n, h, w, ci_p), 7.0, device='cuda')
    descs = geom.conv_dgrad(n, h, w, ci_p, co_p, k, s, pad)
    assert len(descs) == s * s
    halo0 = lib.call('tg_conv3x3_launches')
    for dd in descs:
        lib.call_igemm("tg_igemm_" + prec, dd, lib.ptr(dyd), lib.ptr(whd), None, lib.ptr(dxd), lib.cur_stream())
    assert lib.call('tg_conv3x3_launches') - halo0 == int(_takes_halo_kernel(prec, h, w, co_p, ci_p, k, s, pad))
    dx = dxd.cpu().numpy()
    close(dx[..., :cin], dx_ref, dx_abs)
    assert rejected(dx[..., :cin], T.conv2d_bwd_input(x.shape, f8(other(wt)), f8(other(dy)), (s, s), pad), dx_abs)
    assert (dx[..., cin:] == 0).all()

    # filter gradient, 3 pixel splits summed on the host
    dw_ref = T.conv2d_bwd_filter(f8(q(x)), f8(q(dy)), wt.shape, (s, s), pad)
    dw_abs = T.conv2d_bwd_filter(np.abs(f8(q(x))), np.abs(f8(q(dy))), wt.shape, (s, s), pad)
    nsplit = 3
    slab = torch.full((nsplit, k * k, ci_p, co_p), 7.0, device='cuda')
    dw_desc = geom.conv_wgrad(n, h, w, ci_p, co_p, k, s, pad)
    halo0 = lib.call('tg_conv3x3_launches')
    lib.call("tg_wgrad_" + prec, dw_desc, lib.ptr(xd), lib.ptr(dyd), lib.ptr(slab), nsplit, lib.cur_stream())
    # csrc/wgrad3x3.hip (activation tile read once for the nine taps): 3x3 / stride 1 / SAME, width 16 / 32 / 64, 128 | output channels
    assert lib.call('tg_conv3x3_launches') - halo0 == int(k == 3 and s == 1 and pad == 'SAME' and w in (16, 32, 64) and co_p % 128 == 0)
    dw = slab.cpu().numpy().sum(0)
    close(dw[:, :cin, :cout].reshape(wt.shape), dw_ref, dw_abs)
    assert rejected(dw[:, :cin, :cout].reshape(wt.shape), T.conv2d_bwd_filter(f8(other(x)), f8(other(dy)), wt.shape, (s, s), pad), dw_abs)
    assert (dw[:, cin:, :] == 0).all() and (dw[:, :, cout:] == 0).all()


DECONV_CASES = [(3, 4, 4, 42, 64), (2, 8, 8, 74, 3), (5, 4, 4, 522, 256)]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("n,h,w,cin,cout", DECONV_CASES)
def test_deconv5x5s2_fwd_dgrad_wgrad(n, h, w, cin, cout, prec):
    lib, geom = _tg()
    q = lambda a: _q(prec, a)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((5, 5, cout, cin)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    ci_p, co_p = geom.pad32(cin), geom.pad32(cout)
    y_ref = np.tanh(T.conv2d_transpose(f8(q(x)), f8(q(wt))) + f8(bias))
    y_abs = T.conv2d_transpose(np.abs(f8(q(x))), np.abs(f8(q(wt)))) + np.abs(f8(bias)) + 0.5          # + tanhf's own error (<= 2 ulp of a value <= 1)
    w_pad = np.zeros((25, co_p, ci_p), np.float32)
    w_pad[:, :cout, :cin] = wt.reshape(25, cout, cin)
    xd, wd, bd = dev(padc(x, ci_p)), dev(w_pad), dev(padc(bias, co_p))
    # store only the logical channels (ld_out = cout) — the generator's last layer writes [N,32,32,3]
    yd = torch.full((n, 2 * h, 2 * w, cout), 7.0, device='cuda')
    for d in geom.deconv_fwd(n, h, w, ci_p, co_p, ld_out=cout, n_store=cout, act='tanh'):
        lib.call_igemm("tg_igemm_" + prec, d, lib.ptr(xd), lib.ptr(wd), lib.ptr(bd), lib.ptr(yd), lib.cur_stream())
    close(yd.cpu().numpy(), y_ref, y_abs)

    # the same forward as ONE 3x3 problem with (output parity, channel) columns (tg_igemm_desc.n_group) and the merged filter
    import ctypes as C
    dm, ng, tapmap = geom.deconv_fwd_merged(n, h, w, ci_p, cout, cout, n_store=cout, act='tanh')
    assert dm.n_group == ng == cout and sorted(t for t in tapmap if t >= 0) == list(range(25))
    wraw = dev(wt.reshape(25, cout, cin))
    wm = torch.full((dm.c_out * 9 * ci_p,), 7.0, device='cuda')
    lib.call("tg_deconv_merge_prep_f32", lib.ptr(wraw), None, cout, cin, ng, dm.c_out, ci_p, (C.c_int32 * 36)(*tapmap), lib.ptr(wm), lib.cur_stream())
    ym = torch.full((n, 2 * h, 2 * w, cout), 7.0, device='cuda')
    lib.call_igemm("tg_igemm_" + prec, dm, lib.ptr(xd), lib.ptr(wm), lib.ptr(bd), lib.ptr(ym), lib.cur_stream())
    close(ym.cpu().numpy(), y_ref, y_abs)

    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dyd = dev(padc(dy, co_p))
    dx_ref = T.conv2d_transpose_bwd_input(f8(q(wt)), f8(q(dy)))
    dx_abs = T.conv2d_transpose_bwd_input(np.abs(f8(q(wt))), np.abs(f8(q(dy))))
    w_t = np.zeros((25, ci_p, co_p), np.float32)
    w_t[:, :cin, :cout] = wt.reshape(25, cout, cin).transpose(0, 2, 1)
    dxd = torch.full((n, h, w, ci_p), 7.0, device='cuda')
    wtd = dev(w_t)
    lib.call_igemm("tg_igemm_" + prec, geom.deconv_dgrad(n, h, w, ci_p, co_p), lib.ptr(dyd), lib.ptr(wtd), None,
             lib.ptr(dxd), lib.cur_stream())
    close(dxd.cpu().numpy()[..., :cin], dx_ref, dx_abs)

    dw_ref = T.conv2d_transpose_bwd_filter(f8(q(x)), f8(q(dy)), wt.shape)
    dw_abs = T.conv2d_transpose_bwd_filter(np.abs(f8(q(x))), np.abs(f8(q(dy))), wt.shape)
    slab = torch.full((2, 25, co_p, ci_p), 7.0, device='cuda')
    lib.call("tg_wgrad_" + prec, geom.deconv_wgrad(n, h, w, co_p, ci_p), lib.ptr(dyd), lib.ptr(xd), lib.ptr(slab), 2,
             lib.cur_stream())
    dw = slab.cpu().numpy().sum(0)
    close(dw[:, :cout, :cin].reshape(wt.shape), dw_ref, dw_abs)


@pytest.mark.parametrize("prec", PRECS)
def test_dense_and_identity_layout(prec):
    """A = I with an asymmetric B catches a transposed C-write (cdna_hip_programming.md §3)."""
    lib, geom = _tg()
    m, kdim, nout = 200, 128, 96
    rng = np.random.default_rng(2)
    x = np.zeros((m, kdim), np.float32)
    x[np.arange(128), np.arange(128)] = 1
    x[128:] = rng.standard_normal((m - 128, kdim))
    wt = rng.standard_normal((nout, kdim)).astype(np.float32)      # Wt[n][k]
    yd = torch.zeros((m, nout), device='cuda')
    xd, wd = dev(x), dev(wt)      # keep the device buffers alive across the asynchronous launch
    lib.call_igemm("tg_igemm_" + prec, geom.dense_fwd(m, kdim, nout), lib.ptr(xd), lib.ptr(wd), None, lib.ptr(yd),
             lib.cur_stream())
    y = yd.cpu().numpy()
    np.testing.assert_array_equal(y[:128], _q(prec, wt).T)                    # exact: one product per output
    close(y[128:], f8(_q(prec, x[128:])) @ f8(_q(prec, wt)).T, np.abs(f8(_q(prec, x[128:]))) @ np.abs(f8(_q(prec, wt))).T)


def test_bad_descriptor_is_rejected():
    lib, geom = _tg()
    d = geom.dense_fwd(4, 32, 32)
    d.ld_in = 30
    x = torch.zeros(4, 32, device='cuda')
    with pytest.raises(lib.TgError, match="ld_in"):
        lib.call("tg_igemm_f32", d, lib.ptr(x), lib.ptr(x), None, lib.ptr(x), None, 0, lib.cur_stream())


# ---- work-unit schedule (csrc/geom.cpp tg::igemm_schedule): tiles cut along K, partial sums through caller-owned scratch, fix-up launch ----
# Each case names what the schedule is expected to cut; the launch WITH scratch must equal the one-workgroup-per-tile launch (scratch =
# NULL) up to the fp32 order of the K segments' sum, twice in a row bit for bit (the fix-up adds the segments in a fixed order), and both
# must equal the oracle (the cases are also shapes of the training step: profiles/r02_launches.csv).
def _rand(rng, *shape):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()


def _run_both(lib, name, args, out, colsum=None):
    """-> (out with scratch, out without, [colsum with, without]); also checks run-to-run bit identity of the cut schedule."""
    res = []
    for scratch in (True, True, False):
        out.fill_(7.0)
        if colsum is not None:
            colsum.zero_()
        lib.call_igemm(name, *args, scratch=scratch)
        torch.cuda.synchronize()
        res.append((out.clone(), None if colsum is None else colsum.clone()))
    assert torch.equal(res[0][0], res[1][0]), "the cut schedule is not deterministic"
    if colsum is not None:
        assert torch.equal(res[0][1], res[1][1])
    return res[0], res[2]


@pytest.mark.parametrize("prec", PRECS)
def test_cut_tiles_of_under_filled_launches(prec):
    import ctypes as C
    lib, geom = _tg()
    rng = np.random.default_rng(5)
    st = lib.cur_stream()
    bf = 1 if prec == 'bf16' else 0
    cases = [
        # the generator's first input gradient: 225 tiles of 64 x 64 on 512 slots -> every tile cut in two
        ('under-filled', geom.deconv_dgrad(100, 4, 4, 544, 256), (100, 8, 8, 256), (25 * 544 * 256,), (100, 4, 4, 544)),
        # a discriminator layer on 50 images of 8x8 (160 -> 128): 100 tiles of 45 K-tiles -> cut in four
        ('few tiles', geom.conv_fwd(50, 8, 8, 160, 128, 3, 1, 'SAME', act='lrelu'), (50, 8, 8, 160), (128 * 9 * 160,), (50, 8, 8, 128)),
        # the 2-image tail of a split 130-image classifier launch: 8 x 4 tiles, 72 K-tiles each
        ('tail of a split launch', geom.conv_fwd(2, 16, 16, 256, 256, 3, 1, 'SAME'), (2, 16, 16, 256), (256 * 9 * 256,), (2, 16, 16, 256)),
    ]
    was = lib.call('tg_conv3x3_policy', 2)        # the generic kernel for every shape here
    try:
        # launches that fill the chip are left alone (cutting the tiles of a last partial round was measured and rejected: profiles/r03_split_ab.txt)
        for full in (geom.conv_fwd(153, 16, 16, 256, 64, 3, 1, 'SAME'), geom.conv_fwd(130, 8, 8, 256, 512, 3, 1, 'VALID')):
            assert lib.call('tg_igemm_workspace_bytes', C.byref(full), 1, None, 0, bf) == 0
        for tag, d, xs, ws_, ys in cases:
            need = lib.call('tg_igemm_workspace_bytes', C.byref(d), 1, None, 0, bf)
            assert need > 0, tag
            x, w = _rand(rng, *xs), _rand(rng, *ws_) * 0.05
            bias = _rand(rng, d.c_out)
            y = torch.empty(ys, device='cuda')
            (a, _), (b, _) = _run_both(lib, 'tg_igemm_' + prec, (d, lib.ptr(x), lib.ptr(w), lib.ptr(bias), lib.ptr(y), st), y)
            scale = float(x.abs().max() * w.abs().max()) * d.n_taps * d.ld_in
            assert float((a - b).abs().max()) <= 3e-5 * scale, (tag, float((a - b).abs().max()), scale)
            assert float(a.abs().max()) > 0 and not bool((a == 7.0).any()), tag
    finally:
        lib.call('tg_conv3x3_policy', was)


@pytest.mark.parametrize("prec", PRECS)
def test_cut_long_parities_of_a_transposed_conv_launch(prec):
    """four sub-problems of 9 / 6 / 6 / 4 taps in ONE launch (the generator's 5x5 stride-2 transposed convs): the long parities are cut;
    against the oracle and against the uncut launch."""
    import ctypes as C
    lib, geom = _tg()
    q = lambda a: _q(prec, a)
    rng = np.random.default_rng(6)
    n, h, w, cin, cout = 100, 4, 4, 522, 256
    ci_p, co_p = geom.pad32(cin), geom.pad32(cout)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((5, 5, cout, cin)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    w_pad = np.zeros((25, co_p, ci_p), np.float32)
    w_pad[:, :cout, :cin] = wt.reshape(25, cout, cin)
    xd, wd, bd = dev(padc(x, ci_p)), dev(w_pad), dev(padc(bias, co_p))
    dds = lib.desc_array(geom.deconv_fwd(n, h, w, ci_p, co_p, act='relu'))
    need = lib.call('tg_igemm_workspace_bytes', C.cast(dds, C.c_void_p), len(dds), None, 0, 1 if prec == 'bf16' else 0)
    assert need > 0                                             # the schedule cuts something
    y = torch.empty((n, 2 * h, 2 * w, co_p), device='cuda')
    (a, _), (b, _) = _run_both(lib, 'tg_igemm_multi_' + prec, (C.cast(dds, C.c_void_p), len(dds), lib.ptr(xd), lib.ptr(wd), lib.ptr(bd), lib.ptr(y),
                                                              lib.cur_stream()), y)
    y_ref = T.relu(T.conv2d_transpose(f8(q(x)), f8(q(wt))) + f8(bias))
    y_abs = T.conv2d_transpose(np.abs(f8(q(x))), np.abs(f8(q(wt)))) + np.abs(f8(bias))
    close(a.cpu().numpy()[..., :cout], y_ref, y_abs)
    close(b.cpu().numpy()[..., :cout], y_ref, y_abs)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("n,hw,ci,co,segs", [(6, 8, 256, 256, [128, 256]), (5, 16, 256, 256, [768, 512]), (5, 6, 512, 256, [108, 72])])
def test_cut_tiles_with_column_sums_and_activation_gradient(prec, n, hw, ci, co, segs):
    """the mean-only-BN launches (tg_igemm_colsum_* / tg_igemm_actsum_*): the fix-up launch runs their LDS-staged epilogue, column sums per
    application segment included."""
    import ctypes as C
    lib, geom = _tg()
    rng = np.random.default_rng(7)
    # (6, 8, 256, 256): 384 rows = 6 x 4 tiles of 64 x 64, 72 K-tiles -> cut; the others: the classifier's layers on five images in two
    # applications (tests/test_gpu_nets.py), the last one with application boundaries inside tiles
    sa = (C.c_int32 * 2)(*segs)
    st = lib.cur_stream()
    bf = 1 if prec == 'bf16' else 0
    x, w = _rand(rng, n, hw, hw, ci), _rand(rng, co * 9 * ci) * 0.05
    yact = _rand(rng, n, hw, hw, co)
    y = torch.empty((n, hw, hw, co), device='cuda')
    sums = torch.zeros(2 * len(segs) * co, device='cuda')        # nseg x co doubles
    d = geom.conv_fwd(n, hw, hw, ci, co, 3, 1, 'SAME')
    was = lib.call('tg_conv3x3_policy', 2)
    try:
        assert lib.call('tg_igemm_workspace_bytes', C.byref(d), 1, sa, len(segs), bf) > 0
        for name, args in (('tg_igemm_colsum_' + prec, (d, lib.ptr(x), lib.ptr(w), lib.ptr(y), sa, len(segs), lib.ptr(sums), 0, st)),
                           ('tg_igemm_actsum_' + prec, (d, lib.ptr(x), lib.ptr(w), lib.ptr(yact), lib.ACT['lrelu'], 0.2, lib.ptr(y), sa, len(segs),
                                                        lib.ptr(sums), 0, st))):
            (a, sa_), (b, sb_) = _run_both(lib, name, args, y, colsum=sums)
            scale = float(x.abs().max() * w.abs().max()) * 9 * ci
            assert float((a - b).abs().max()) <= 3e-5 * scale, name
            cs_a, cs_b = sa_.view(torch.float64).cpu().numpy(), sb_.view(torch.float64).cpu().numpy()
            assert np.abs(cs_a - cs_b).max() <= 3e-5 * scale * max(segs), name
            ref = np.stack([a.cpu().numpy().reshape(-1, co)[:segs[0]].astype(np.float64).sum(0), a.cpu().numpy().reshape(-1, co)[segs[0]:].astype(np.float64).sum(0)])
            assert np.abs(cs_a.reshape(2, co) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), name      # the sums are of the stored values
    finally:
        lib.call('tg_conv3x3_policy', was)


@pytest.mark.parametrize("n,h,w,cin,cout", [(5, 16, 16, 138, 3), (2, 32, 32, 74, 3), (3, 4, 16, 40, 1), (2, 8, 16, 120, 4)])
def test_narrow_transposed_conv_backward_k_packed(n, h, w, cin, cout):
    """csrc/narrow.hip: input and filter gradient of a 5x5 / stride-2 transposed conv with <= 4 output channels (the generator's image
    layer) against the oracle — K-packed fp32 MFMA products (exact fp32 products), so the fp32 bound of the MFMA kernels applies."""
    lib, geom = _tg()
    rng = np.random.default_rng(11)
    ci_p, co_p = geom.pad32(cin), geom.pad32(cout)
    assert lib.call('tg_deconv5x5s2_narrow_supported', n, h, w, cout, ci_p) == 1
    assert lib.call('tg_deconv5x5s2_narrow_supported', n, h, w, 5, ci_p) == 0 and lib.call('tg_deconv5x5s2_narrow_supported', n, h, 24, cout, ci_p) == 0
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((5, 5, cout, cin)) * 0.1).astype(np.float32)
    dy = rng.standard_normal((n, 2 * h, 2 * w, cout)).astype(np.float32)
    st = lib.cur_stream()
    # input gradient, from the [5,5,Cout,Cin] variable itself (and with a per-output-channel weight-norm scale)
    dyd, wtd, xd = dev(padc(dy, co_p)), dev(wt), dev(padc(x, ci_p))
    scale = (1 + 0.3 * rng.standard_normal(cout)).astype(np.float32)
    for sc in (None, scale):
        dxd = torch.full((n, h, w, ci_p), 7.0, device='cuda')
        lib.call('tg_deconv5x5s2_narrow_dgrad_f32', lib.ptr(dyd), co_p, lib.ptr(wtd), lib.ptr(dev(sc)) if sc is not None else None, n, h, w, cout, cin, ci_p,
                 lib.ptr(dxd), ci_p, st)
        dx = dxd.cpu().numpy()
        w_eff = wt if sc is None else wt * sc[None, None, :, None]
        close(dx[..., :cin], T.conv2d_transpose_bwd_input(f8(w_eff), f8(dy)), T.conv2d_transpose_bwd_input(np.abs(f8(w_eff)), np.abs(f8(dy))))
        assert (dx[..., cin:] == 0).all()
    # filter gradient straight into the [5,5,Cout,Cin] variable's layout
    need = lib.call('tg_deconv5x5s2_narrow_wgrad_workspace_bytes', n, h, w, cout, ci_p)
    ws = torch.empty(need // 4, device='cuda')
    dwd = torch.full((25, cout, cin), 7.0, device='cuda')
    lib.call('tg_deconv5x5s2_narrow_wgrad_f32', lib.ptr(dyd), co_p, lib.ptr(xd), ci_p, n, h, w, cout, cin, ci_p, lib.ptr(ws), lib.ptr(dwd), st)
    close(dwd.cpu().numpy().reshape(wt.shape), T.conv2d_transpose_bwd_filter(f8(x), f8(dy), wt.shape),
          T.conv2d_transpose_bwd_filter(np.abs(f8(x)), np.abs(f8(dy)), wt.shape))


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("n,h,cin,cout,s", [(5, 16, 42, 32, 1), (3, 16, 42, 64, 2), (7, 8, 74, 128, 1), (130, 8, 138, 128, 1)])
def test_conv_writes_the_cond_concat_behind_it(n, h, cin, cout, s, prec):
    """tg_igemm_labels_*: conv -> bias -> leaky relu whose output buffer IS the tensor _conv_cond_concat would build (Model/modle_base.py:239-244,
    the discriminators' conv -> concat pairs): channels [cout, cout + 10) = the image's label vector, zeros up to the 32-padded stride, written by
    the workgroups of the last column tile; the convolution channels as tg_igemm_* writes them.  Against oracle conv + conv_cond_concat; the
    130-image case has several column tiles per row tile and tiles cut along K (the fix-up launch runs the same epilogue)."""
    lib, geom = _tg()
    q = lambda a: _q(prec, a)
    rng = np.random.default_rng(31)
    x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    lab = rng.random((n, 10)).astype(np.float32)                     # soft labels (the C-update feeds the classifier's one-hots, any vector works)
    ci_p, co_p, ld = geom.pad32(cin), geom.pad32(cout), geom.pad32(cout + 10)
    y_ref = T.lrelu(T.conv2d(f8(q(x)), f8(q(wt)), (s, s), 'SAME') + f8(bias))
    y_abs = T.conv2d(np.abs(f8(q(x))), np.abs(f8(q(wt))), (s, s), 'SAME') + np.abs(f8(bias))
    cat_ref = T.conv_cond_concat(y_ref, f8(lab))
    ho = y_ref.shape[1]
    w_oti = np.zeros((co_p, 9, ci_p), np.float32)
    w_oti[:cout, :, :cin] = wt.reshape(9, cin, cout).transpose(2, 0, 1)
    xd, wd, bd, ld_ = dev(padc(x, ci_p)), dev(w_oti), dev(padc(bias, co_p)), dev(lab)
    yd = torch.full((n, ho, ho, ld), 7.0, device='cuda')
    d = geom.conv_fwd(n, h, h, ci_p, co_p, 3, s, 'SAME', ld_out=ld, n_store=cout, act='lrelu')
    lib.call_igemm("tg_igemm_labels_" + prec, d, lib.ptr(xd), lib.ptr(wd), lib.ptr(bd), lib.ptr(ld_), 10, lib.ptr(yd), lib.cur_stream())
    y = yd.cpu().numpy()
    close(y[..., :cout], y_ref, y_abs)
    np.testing.assert_array_equal(y[..., cout:cout + 10], cat_ref[..., cout:].astype(np.float32))      # the labels, bit for bit
    assert (y[..., cout + 10:] == 0).all()
    # too narrow an output stride for the labels is refused
    bad = geom.conv_fwd(n, h, h, ci_p, co_p, 3, s, 'SAME', ld_out=co_p, n_store=cout, act='lrelu')
    with pytest.raises(lib.TgError, match='igemm_labels'):
        lib.call_igemm("tg_igemm_labels_" + prec, bad, lib.ptr(xd), lib.ptr(wd), lib.ptr(bd), lib.ptr(ld_), 10, lib.ptr(yd), lib.cur_stream())


@pytest.mark.parametrize("n,h,w,cin,cout,nlab", [(5, 32, 32, 13, 32, 10), (3, 16, 16, 3, 64, 0), (2, 8, 32, 16, 32, 10), (2, 12, 16, 7, 32, 4)])
def test_few_channel_3x3_conv_k_packed(n, h, w, cin, cout, nlab):
    """csrc/packed_conv.hip: forward (+ bias, leaky relu, the cond-concat's label channels and the channel padding behind them) and filter
    gradient of a 3x3 / stride-1 / SAME convolution with <= 16 input channels (the discriminators' first layer, 13 -> 32) as K-packed fp32
    MFMA products straight from / into the [3,3,Cin,Cout] variable, against the oracle — exact fp32 products: the fp32 bound of the MFMA
    kernels, and the negative control (bf16-rounded operands) misses it."""
    lib, geom = _tg()
    rng = np.random.default_rng(41)
    assert lib.call('tg_conv3x3_packed_supported', n, h, w, cin, cout) == 1
    assert lib.call('tg_conv3x3_packed_supported', n, h, w, 17, cout) == 0 and lib.call('tg_conv3x3_packed_supported', n, h, w, cin, 128) == 0
    assert lib.call('tg_conv3x3_packed_supported', n, h, 24, cin, cout) == 0
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    lab = rng.random((n, max(nlab, 1))).astype(np.float32)
    dy = rng.standard_normal((n, h, w, cout)).astype(np.float32)
    ci_p, ld = geom.pad32(cin), geom.pad32(cout + nlab)
    st = lib.cur_stream()
    xd, wd, bd, labd = dev(padc(x, ci_p)), dev(wt), dev(bias), dev(lab)
    other = lambda a: T.bf16_round(a)
    for act, fn in (('lrelu', T.lrelu), (None, lambda v: v)):
        y_ref = fn(T.conv2d(f8(x), f8(wt), (1, 1), 'SAME') + f8(bias))
        y_abs = T.conv2d(np.abs(f8(x)), np.abs(f8(wt)), (1, 1), 'SAME') + np.abs(f8(bias))
        yd = torch.full((n, h, w, ld), 7.0, device='cuda')
        lib.call('tg_conv3x3_packed_fwd_f32', lib.ptr(xd), ci_p, cin, lib.ptr(wd), lib.ptr(bd), lib.ACT[act], 0.2, lib.ptr(labd) if nlab else None, nlab,
                 lib.ptr(yd), ld, n, h, w, cout, st)
        y = yd.cpu().numpy()
        close(y[..., :cout], y_ref, y_abs)
        assert rejected(y[..., :cout], fn(T.conv2d(f8(other(x)), f8(other(wt)), (1, 1), 'SAME') + f8(bias)), y_abs)
        if nlab:
            np.testing.assert_array_equal(y[..., cout:cout + nlab], np.broadcast_to(lab[:, None, None, :], (n, h, w, nlab)))      # the labels, bit for bit
        assert (y[..., cout + nlab:] == 0).all()
    # filter gradient straight into the [3,3,Cin,Cout] variable's layout
    need = lib.call('tg_conv3x3_packed_wgrad_workspace_bytes', n, h, w, cin, cout)
    ws = torch.empty(need // 4, device='cuda')
    dwd = torch.full((3, 3, cin, cout), 7.0, device='cuda')
    dyd = dev(padc(dy, geom.pad32(cout)))
    lib.call('tg_conv3x3_packed_wgrad_f32', lib.ptr(xd), ci_p, cin, lib.ptr(dyd), geom.pad32(cout), n, h, w, cout, lib.ptr(ws), lib.ptr(dwd), st)
    dw_abs = T.conv2d_bwd_filter(np.abs(f8(x)), np.abs(f8(dy)), wt.shape, (1, 1), 'SAME')
    close(dwd.cpu().numpy(), T.conv2d_bwd_filter(f8(x), f8(dy), wt.shape, (1, 1), 'SAME'), dw_abs)
    assert rejected(dwd.cpu().numpy(), T.conv2d_bwd_filter(f8(other(x)), f8(other(dy)), wt.shape, (1, 1), 'SAME'), dw_abs)
    with pytest.raises(lib.TgError, match='conv3x3_packed_fwd'):
        lib.call('tg_conv3x3_packed_fwd_f32', lib.ptr(xd), ci_p, cin, lib.ptr(wd), lib.ptr(bd), lib.ACT['lrelu'], 0.2, lib.ptr(labd), 40, lib.ptr(yd), ld, n, h, w, cout, st)
