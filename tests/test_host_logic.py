"""CPU tests of the host side: implicit-GEMM geometry (emulated in NumPy from the descriptor and checked against
the oracle's conv / transposed conv and their gradients), split heuristics, parameter stores, dataset protocol,
sample-grid utilities, schedules and the FLOP accounting of bench.py."""
import numpy as np
import pytest

from oracle import tf_ops as T
from tg import geom


def emulate_igemm(d, x, w_flat, out):
    """NumPy statement of tg_igemm_f32's contract (include/tg_kernels.h) for one descriptor."""
    n, hv, wv = d.n_img, d.h_v, d.w_v
    for t in range(d.n_taps):
        dy, dx, tw = d.dy[t], d.dx[t], d.tapw[t]
        wt = np.stack([w_flat[nn * d.w_sn + tw * d.w_st: nn * d.w_sn + tw * d.w_st + d.ld_in] for nn in range(d.c_out)])  # [c_out, ld_in]
        for vy in range(hv):
            iy = vy * d.s_y + dy
            if not 0 <= iy < d.h_in:
                continue
            for vx in range(wv):
                ix = vx * d.s_x + dx
                if not 0 <= ix < d.w_in:
                    continue
                out[:, vy * d.os_y + d.oo_y, vx * d.os_x + d.oo_x, :d.n_store] += (x[:, iy, ix, :] @ wt.T)[:, :d.n_store]
    return out


def padc(a, ld):
    o = np.zeros(a.shape[:-1] + (ld,), a.dtype)
    o[..., :a.shape[-1]] = a
    return o


@pytest.mark.parametrize("h,w,cin,cout,k,s,pad", [(8, 8, 5, 7, 3, 1, 'SAME'), (8, 8, 5, 7, 3, 2, 'SAME'), (9, 7, 4, 6, 3, 2, 'SAME'),
                                                   (8, 8, 4, 6, 3, 1, 'VALID'), (6, 6, 3, 4, 1, 1, 'SAME')])
def test_conv_geometry(h, w, cin, cout, k, s, pad):
    rng = np.random.default_rng(0)
    n = 2
    x = rng.standard_normal((n, h, w, cin))
    wt = rng.standard_normal((k, k, cin, cout))
    ci_p, co_p = geom.pad32(cin), geom.pad32(cout)
    y_ref = T.conv2d(x, wt, (s, s), pad)
    d = geom.conv_fwd(n, h, w, ci_p, co_p, k, s, pad)
    w_oti = np.zeros((co_p, k * k, ci_p))
    w_oti[:cout, :, :cin] = wt.reshape(k * k, cin, cout).transpose(2, 0, 1)
    y = emulate_igemm(d, padc(x, ci_p), w_oti.reshape(-1), np.zeros((n, d.h_out, d.w_out, co_p)))
    np.testing.assert_allclose(y[..., :cout], y_ref, atol=1e-10)
    dy = rng.standard_normal(y_ref.shape)
    w_hwio = np.zeros((k * k, ci_p, co_p))
    w_hwio[:, :cin, :cout] = wt.reshape(k * k, cin, cout)
    dx = np.zeros((n, h, w, ci_p))
    descs = geom.conv_dgrad(n, h, w, ci_p, co_p, k, s, pad)
    assert len(descs) == s * s
    for dd in descs:
        emulate_igemm(dd, padc(dy, co_p), w_hwio.reshape(-1), dx)
    np.testing.assert_allclose(dx[..., :cin], T.conv2d_bwd_input(x.shape, wt, dy, (s, s), pad), atol=1e-10)


def test_deconv_geometry():
    rng = np.random.default_rng(1)
    n, h, cin, cout = 2, 4, 6, 3
    x = rng.standard_normal((n, h, h, cin))
    wt = rng.standard_normal((5, 5, cout, cin))
    ci_p, co_p = geom.pad32(cin), geom.pad32(cout)
    w_pad = np.zeros((25, co_p, ci_p))
    w_pad[:, :cout, :cin] = wt.reshape(25, cout, cin)
    y = np.zeros((n, 2 * h, 2 * h, cout))
    descs = geom.deconv_fwd(n, h, h, ci_p, co_p, ld_out=cout, n_store=cout)
    assert sorted(d.n_taps for d in descs) == [4, 6, 6, 9]          # 25 taps over the four output parities
    for d in descs:
        emulate_igemm(d, padc(x, ci_p), w_pad.reshape(-1), y)
    np.testing.assert_allclose(y, T.conv2d_transpose(x, wt), atol=1e-10)
    dy = rng.standard_normal(y.shape)
    w_tr = np.zeros((25, ci_p, co_p))
    w_tr[:, :cin, :cout] = wt.reshape(25, cout, cin).transpose(0, 2, 1)
    dx = emulate_igemm(geom.deconv_dgrad(n, h, h, ci_p, co_p), padc(dy, co_p), w_tr.reshape(-1), np.zeros((n, h, h, ci_p)))
    np.testing.assert_allclose(dx[..., :cin], T.conv2d_transpose_bwd_input(wt, dy), atol=1e-10)


def test_same_padding_matches_tf_rule():
    """TF's SAME rule as the library's descriptors encode it: (output size, padding before) = first tap's offset negated."""
    def out_before(n, k, s, pad):
        d = geom.conv_fwd(1, n, n, 32, 32, k, s, pad)
        return d.h_out, -d.dy[0]
    assert out_before(32, 3, 2, 'SAME') == (16, 0)      # total 1: the extra pixel goes after
    assert out_before(32, 3, 1, 'SAME') == (32, 1)
    assert out_before(8, 5, 2, 'SAME') == (4, 1)        # total 3: 1 before, 2 after
    assert out_before(8, 3, 1, 'VALID') == (6, 0)
    from tg.lib import TgError
    with pytest.raises(TgError, match="conv2d_desc_fwd"):
        geom.conv_fwd(1, 2, 2, 32, 32, 3, 1, 'VALID')     # window larger than the image


def test_wgrad_splits_fill_one_round():
    d = geom.conv_wgrad(250, 32, 32, 128, 128, 3, 1, 'SAME')
    assert geom.wgrad_splits(d) == 64                                # csrc/wgrad3x3.hip: 4 channel chunks x 64 splits = one workgroup per CU (256 without a device)
    assert geom.wgrad_slab_floats(d, 64) == 64 * 9 * 128 * 128
    d = geom.conv_wgrad(250, 16, 16, 256, 256, 3, 1, 'SAME')
    assert geom.wgrad_splits(d) == 16                                # 8 chunks x 2 column tiles x 16
    d = geom.conv_wgrad(250, 8, 8, 256, 512, 3, 1, 'VALID')         # 8-wide images: the generic kernel, 9 taps x 2 x 4 tiles
    assert geom.wgrad_splits(d) == 7                                 # 72 * 7 = 504 <= 512 resident workgroups
    d = geom.conv_wgrad(100, 32, 32, 32, 64, 3, 2, 'SAME')
    assert geom.wgrad_splits(d) == 56                                # stride 2: generic, 9 tiles * 56 = 504
    d = geom.dense_fwd(100, 128, 8192)
    assert geom.wgrad_splits(d) == 1                                 # never an empty split


def test_param_store_layout_and_roundtrip():
    import torch
    from tg.runtime import ParamStore
    st = ParamStore('net', [('net/a', (3, 5), True), ('net/pop', (7,), False), ('net/b', (33,), True)], torch.device('cpu'))
    assert st.index['net/a'][1] == 0 and st.index['net/b'][1] == 32 and st.n_p == 96      # 32-float aligned
    a = np.arange(15, dtype=np.float32).reshape(3, 5)
    st.set('net/a', a)
    st.set('net/pop', np.ones(7))
    np.testing.assert_array_equal(st.get('net/a'), a)
    assert st.names(True) == ['net/a', 'net/b'] and st.names(False) == ['net/pop']
    assert float(st.p[15:32].abs().sum()) == 0                       # padding stays zero
    d = st.to_dict()
    assert set(d) == {'net/a', 'net/pop', 'net/b'}


def test_param_store_growth_keeps_pointers_inside_the_reserve_and_refuses_to_move_captured_buffers():
    """advisor (round 2): variables appended within the reserve keep every device pointer; beyond it the buffers move, which is an
    error once a hipGraph has captured them (ParamStore.frozen, set by Train._capture)."""
    import torch
    from tg import lib
    from tg.runtime import ParamStore
    st = ParamStore('net', [('net/a', (8,), True)], torch.device('cpu'), capacity=64)
    st.enable_ema()
    ptr = st.p.data_ptr()
    st.extend([('net/b', (40,), True)])                               # 32 + 64 floats needed, 96 reserved
    assert st.p.data_ptr() == ptr and st.n_p == 96 and st.ema.numel() == 96
    st.frozen = True
    with pytest.raises(lib.TgError, match='after a hipGraph captured'):
        st.extend([('net/c', (64,), True)])


def test_model_param_specs_match_reference_counts():
    from Model.Good_GAN_cifar10 import Good_GAN_cifar10
    specs = Good_GAN_cifar10.param_specs()
    cnt = lambda net: sum(int(np.prod(s)) for _, s, tr, _ in specs[net] if tr)
    assert (cnt('good_generator'), cnt('discriminator'), cnt('classifier')) == (5129201, 327467, 3121812)
    names = [n for n, *_ in specs['classifier']]
    assert 'classifier/NiN1/NiN1/V' in names and 'classifier/conv1_1/meanOnlyBatchNormalization/pop_mean' in names


def test_synthetic_dataset_protocol():
    from config import Config
    from Input_Pipeline.syntheticDataset import syntheticDataset as Dataset

    class Cfg(Config):
        DATA_NAME, NUM_CLASSES, BATCH_SIZE = 'cifar10', 10, 100
        IMAGE_HEIGHT, IMAGE_WIDTH, CHANNEL = 32, 32, 3
        BATCH_SIZE_L_C, BATCH_SIZE_U_C, BATCH_SIZE_L_D, BATCH_SIZE_U_D = 50, 50, 20, 80
    c = Cfg()
    tr, va = Dataset(None, c, 4000, 'train', True), Dataset(None, c, 4000, 'test', False)
    init_train, init_val, nnio = tr.inputpipline_train_val(va)
    init_train(), init_val()
    b = nnio.next()
    assert b['x_l_c'].shape == (50, 32, 32, 3) and b['x_l_d'].shape == (20, 32, 32, 3)
    assert b['x_u_d'].shape == (80, 32, 32, 3) and b['x_u_c'].shape == (50, 32, 32, 3)
    assert b['x_l_c'].min() >= -1 and b['x_l_c'].max() <= 1 and (b['y_l_c'].sum(1) == 1).all()
    vb = list(nnio.val_batches())
    assert len(vb) == 10 and vb[0][0].shape == (100, 32, 32, 3)


def test_sample_grid_utils(tmp_path):
    import utils
    imgs = np.random.default_rng(0).uniform(-1, 1, (64, 32, 32, 3))
    assert utils.image_manifold_size(64) == (8, 8)
    grid = utils.merge(utils.inverse_transform(imgs), (8, 8))
    assert grid.shape == (256, 256, 3)
    np.testing.assert_allclose(grid[32:64, 0:32], (imgs[8] + 1) / 2)          # row-major tiling
    p = tmp_path / "train_01.png"
    utils.save_images(imgs, (8, 8), str(p))
    assert p.read_bytes()[:8] == b'\x89PNG\r\n\x1a\n'


def test_bench_flop_accounting_matches_survey():
    import bench
    f = bench.algorithmic_flops()
    assert abs(f['total'] / 1e9 - 1464.0) < 0.1                     # SURVEY §8d / BASELINE.md §4
    assert abs(f['igemm'] + f['wgrad'] - f['total']) < 1


def test_rampup_rampdown():
    pytest.importorskip("torch")
    from Training import Train_goodGAN as TG
    assert TG.rampup(300) == 1.0 and TG.rampdown(0) == 1.0
    assert abs(TG.rampup(0) - np.exp(-5.0)) < 1e-12


def test_tape_split_at_gradient_bucket_boundary():
    """Context.backward(stop_at_boundary=True): the closures recorded after the last boundary run, the head of the tape is
    returned in order and runs later through run_tape (markers skipped) — the DP bucket protocol of SURVEY §8e."""
    from tg import runtime
    cx = runtime.Context.__new__(runtime.Context)          # tape logic only: no device needed
    log = []
    cx.tape = []
    for i in range(3):
        cx.record(lambda i=i: log.append('a%d' % i))
    cx.grad_bucket_boundary()
    for i in range(2):
        cx.record(lambda i=i: log.append('b%d' % i))
    cx.grad_bucket_boundary()
    cx.record(lambda: log.append('c0'))
    rest = cx.backward(stop_at_boundary=True)
    assert log == ['c0'] and cx.tape == []
    assert len(rest) == 6                                   # 3 + marker + 2
    cx.run_tape(rest)
    assert log == ['c0', 'b1', 'b0', 'a2', 'a1', 'a0'] and rest == []
    # without the flag the markers are transparent
    log[:] = []
    cx.tape = []
    cx.record(lambda: log.append('x'))
    cx.grad_bucket_boundary()
    cx.record(lambda: log.append('y'))
    assert cx.backward() is None and log == ['y', 'x']
    cx.tape = None
    cx.grad_bucket_boundary()                               # no tape: no-op


def test_bench_reads_the_committed_traffic_record_without_a_gpu(tmp_path, monkeypatch):
    """bench.measured_traffic(): the committed PMC record gives (bytes, detail) when it was collected for the kernel sources in the tree,
    (None, reason) when it is stale or incomplete — never an exception (a crash here would cost the round's bench line)."""
    import json
    import bench
    traffic, detail = bench.measured_traffic()
    assert isinstance(detail, dict)
    assert traffic is None or traffic > 1e8
    # an incomplete record (a PMC pass whose kernel names did not match) degrades to null
    prof = tmp_path / 'profiles'
    prof.mkdir()
    (prof / 'r03_traffic.json').write_text(json.dumps({'kernel_sources_sha256': 'x'}))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    traffic, detail = bench.measured_traffic()
    assert traffic is None and 'incomplete' in detail
