"""hipGraph replay of the three solver runs (tg_graph_* through the C ABI) must be bit-identical to launching every
kernel eagerly: same kernels, same order, Philox state advanced on the device inside the graph."""
import numpy as np
import pytest

from oracle import step_cifar10 as S
import gpu_common as G

pytestmark = pytest.mark.gpu


SMALL = dict(B_G=16, L_C=8, U_C=8, L_D=4, U_D=12)


def run(use_graph, steps=4, sizes=SMALL, _holder=None, **over):
    tr = G.fresh_trainer(G.make_config(sizes, USE_HIP_GRAPH=use_graph, SEED=3, **over))
    if _holder is not None:
        _holder['tr'] = tr
    tr.set_hyper(lambda_1=0.3, lambda_2=0.5)
    full = dict(S.SIZES, **sizes)
    losses = []
    for it in range(steps):
        tr.feed(S.synth_batch(50 + it, full))
        tr.sample_latent()
        tr.train_iteration()
        losses.append(tr.losses())
    params = {net: st.p.detach().cpu().numpy().copy() for net, st in tr.cx.stores.items()}
    graphs = tr._graphs
    return losses, params, graphs


def test_graph_replay_equals_eager():
    l_e, p_e, g_e = run(False)
    l_g, p_g, g_g = run(True)
    assert all(h is None for h in g_e['full']) and all(h is not None for h in g_g['full'])   # the graphs were really used
    assert l_e == l_g
    for net in p_e:
        np.testing.assert_array_equal(p_e[net], p_g[net])
    # dropout / noise differ between iterations (the RNG step advances inside the graph): losses are not constant
    assert len({l[0] for l in l_g}) == len(l_g)


def test_graph_replay_equals_eager_at_the_benchmark_sizes():
    """BASELINE configs[1] (100/50/50/20/80, Training/Train_goodGAN.py:566-572): the launch configurations the bench line runs
    (128x128 tiles with fused column sums, split filter gradients, batched preps / draws).  Replay == eager bit for bit, and a second
    replayed run reproduces the first (no atomics-order or uninitialised-scratch dependence in any result)."""
    sizes = dict(S.SIZES)
    l_e, p_e, _ = run(False, 3, sizes)
    l_g, p_g, g_g = run(True, 3, sizes)
    l_h, p_h, _ = run(True, 3, sizes)
    assert all(h is not None for h in g_g['full'])
    assert l_e == l_g == l_h and all(np.isfinite(v) for l in l_g for v in l)
    for net in p_e:
        np.testing.assert_array_equal(p_e[net], p_g[net])
        np.testing.assert_array_equal(p_g[net], p_h[net])


def test_graph_replay_equals_eager_with_bf16_operands_at_the_benchmark_sizes():
    """configs[3]'s operand type on the bench shapes: the launches that exist only there — the pipelined 3x3 kernel with its per-launch
    filter pack into the scratch ring and LDS-DMA fetches, the split 130-image launches, wgrad3x3 with tg_wgrad_splits_bf16 — captured
    and replayed: bit-identical to the eager run (the ring slots baked into the graph are the ones the eager order would use)."""
    from tg import lib
    sizes = dict(S.SIZES)
    halo0 = lib.call('tg_conv3x3_launches')
    l_e, p_e, _ = run(False, 3, sizes, MFMA_DTYPE='bf16')
    assert lib.call('tg_conv3x3_launches') - halo0 >= 3 * 20          # the halo kernels really ran (27+ launches per iteration)
    l_g, p_g, g_g = run(True, 3, sizes, MFMA_DTYPE='bf16')
    assert all(h is not None for h in g_g['full'])
    assert l_e == l_g and all(np.isfinite(v) for l in l_g for v in l)
    for net in p_e:
        np.testing.assert_array_equal(p_e[net], p_g[net])


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_plan_replay_equals_eager_at_the_benchmark_sizes(dtype):
    """config.EXEC_MODE = 'plan' at BASELINE configs[1] / configs[3] sizes: iteration 0 launches eagerly on two streams, iteration 1 records every
    segment into a native launch plan (include/tg_plan.h) while it runs, iterations 2.. are tg_plan_replay alone — bit-identical weights
    and losses to the one-stream eager run, Philox state advancing on the device."""
    sizes = dict(S.SIZES)
    l_e, p_e, _ = run(False, 5, sizes, MFMA_DTYPE=dtype)
    tr_holder = {}
    l_p, p_p, _ = run(None, 5, sizes, MFMA_DTYPE=dtype, EXEC_MODE='plan', _holder=tr_holder)
    plans = tr_holder['tr']._plans['full']
    assert all(p is not None for p in plans) and sum(p.launches for p in plans) > 200, [p and p.launches for p in plans]
    assert l_e == l_p and all(np.isfinite(v) for l in l_p for v in l)
    for net in p_e:
        np.testing.assert_array_equal(p_e[net], p_p[net])


def test_execution_modes_compute_the_same_numbers_and_auto_decides():
    """config.EXEC_MODE: 'eager' (one stream), 'overlap' (filter gradients and the D-update's generator forward on a second stream),
    'plan' (the two-stream launch sequence recorded once and re-issued by tg_plan_replay, include/tg_plan.h), 'graph' (hipGraph replay)
    and 'auto' (times the last two over its first AUTO_ITERS iterations, then keeps the faster one) launch
    the same kernels on the same operands — bit-identical weights after the decision, and the decision is recorded."""
    from Training.Train_goodGAN import Train
    steps = Train.AUTO_ITERS + 2
    ref = None
    for mode in ('eager', 'overlap', 'plan', 'graph', 'auto'):
        tr = G.fresh_trainer(G.make_config(SMALL, USE_HIP_GRAPH=None, EXEC_MODE=mode, SEED=3))
        tr.set_hyper(lambda_1=0.3, lambda_2=0.5)
        full = dict(S.SIZES, **SMALL)
        losses = []
        for it in range(steps):
            tr.feed(S.synth_batch(50 + it, full))
            tr.sample_latent()
            tr.train_iteration()
            losses.append(tr.losses())
        params = {net: st.p.detach().cpu().numpy().copy() for net, st in tr.cx.stores.items()}
        used_graphs = tr._graphs is not None and all(h is not None for h in tr._graphs.get('full', [None]))
        pick, timings = tr.exec_mode_chosen()
        if mode == 'auto':
            assert pick in ('plan', 'graph') and set(timings) == {'plan', 'graph'} and all(t > 0 for t in timings.values()), (pick, timings)
            assert used_graphs          # the graph candidate was captured and timed
            assert all(p is not None for p in tr._plans['full'])          # and so was the plan candidate
        else:
            assert pick is None
            assert used_graphs == (mode == 'graph')
            if mode == 'plan':          # every segment recorded in the second iteration and replayed natively from the third on
                plans = tr._plans['full']
                assert all(p is not None for p in plans) and sum(p.launches for p in plans) > 100, [p and p.launches for p in plans]
                assert sum(len(p) - p.launches for p in plans) >= 4          # the cross-stream events of the overlap are part of the plans
            else:
                assert not getattr(tr, '_plans', {})
        if ref is None:
            ref = (losses, params)
        else:
            assert losses == ref[0], mode
            for net in params:
                np.testing.assert_array_equal(params[net], ref[1][net], err_msg=mode)


def test_no_buffer_is_allocated_once_the_execution_mode_is_decided():
    """Every buffer of the step is keyed by its call site and allocated in the first eager iterations; the two-stream and the graph mode
    share those keys (tg/runtime.py Context._event).  At the bench sizes, through the alternating blocks of EXEC_MODE = 'auto' and 30
    iterations beyond: the number of call-site buffers and the bytes torch holds stay what they were after the first two iterations of
    each mode."""
    import torch
    from Training.Train_goodGAN import Train
    sizes = dict(S.SIZES)
    tr = G.fresh_trainer(G.make_config(sizes, USE_HIP_GRAPH=None, EXEC_MODE='auto', SEED=3))
    tr.set_hyper(lambda_1=0.3, lambda_2=0.5)
    b = S.synth_batch(50, sizes)
    settled = None
    first_graph_block = Train.AUTO_SETTLE + Train.AUTO_TIMED
    for it in range(Train.AUTO_ITERS + 30):
        tr.feed(b)
        tr.sample_latent()
        tr.train_iteration()
        if it == first_graph_block + 2:                     # two-stream block done, graphs captured and replayed once
            torch.cuda.synchronize()
            settled = (len(tr.cx.buffers), torch.cuda.memory_allocated())
    torch.cuda.synchronize()
    assert tr.exec_mode_chosen()[0] in ('plan', 'graph')
    assert (len(tr.cx.buffers), torch.cuda.memory_allocated()) == settled
    assert all(np.isfinite(v) for v in tr.losses())


def test_launch_plan_api_on_two_streams_and_its_error_report():
    """include/tg_plan.h driven directly: two fills on two streams ordered by an event, replayed twice (also on other streams than the ones it
    was recorded with); a recorded launch whose entry point refuses its arguments stops the replay and names the operation."""
    import ctypes as C
    import torch
    from tg import lib, plan
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.zeros(1 << 16, device='cuda')
    b = torch.zeros(1 << 16, device='cuda')
    ev = torch.cuda.Event()
    ev.record(s0)                                        # creates the event's handle
    p = plan.Plan([s0.cuda_stream, s1.cuda_stream])
    p.add_launch('tg_fill_f32', (lib.ptr(a), 3.0, a.numel(), C.c_void_p(s0.cuda_stream)))
    p.add_record(ev, s0.cuda_stream)
    p.add_wait(s1.cuda_stream, ev)
    p.add_launch('tg_copy2d_f32', (lib.ptr(a), a.numel(), lib.ptr(b), b.numel(), 1, a.numel(), C.c_void_p(s1.cuda_stream)))   # b = a, after the fill
    assert len(p) == 4 and p.launches == 2
    p.replay()
    torch.cuda.synchronize()
    assert float(b.min()) == 3.0 == float(b.max())
    a.zero_(); b.zero_()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Stream(), torch.cuda.Stream()
    p.replay([t0.cuda_stream, t1.cuda_stream])           # the slots are positions, not the recorded handles
    torch.cuda.synchronize()
    assert float(b.min()) == 3.0 == float(b.max())
    bad = plan.Plan([s0.cuda_stream])
    bad.add_launch('tg_fill_f32', (lib.ptr(a), 1.0, 16, C.c_void_p(s0.cuda_stream)))
    bad.add_launch('tg_copy2d_f32', (None, 4, lib.ptr(b), 4, 1, 4, C.c_void_p(s0.cuda_stream)))            # a null source: refused by the entry point
    with pytest.raises(lib.TgError, match=r'operation 1 \(tg_copy2d_f32\) failed: copy2d'):
        bad.replay()
    torch.cuda.synchronize()
    assert float(a[:16].max()) == 1.0                    # the operation in front of the failing one was issued


def test_capture_survives_the_garbage_of_earlier_owners():
    """Train._capture keeps Python's garbage collector out of the capture window: a collector pass can free pinned host tensors of earlier
    owners, torch's host allocator then records and queries an event on the streams they were copied on — if that is the capturing stream
    (torch's stream pool wraps around after 32) the capture is invalidated (tools/micro/capture_pinned_free.py shows the mechanism).  Here
    cyclic garbage holding streams, events, device and pinned tensors waits for collection and the collector is set to run every few
    allocations while an iteration is captured."""
    import gc
    import torch
    l_ref, p_ref, _ = run(True, steps=3)
    for _ in range(16):
        s, e = torch.cuda.Stream(), torch.cuda.Event()
        e.record(s)
        cyc = [s, e, torch.empty(1024, device='cuda'), torch.empty(1 << 16).pin_memory()]
        cyc.append(cyc)
    del s, e, cyc
    old = gc.get_threshold()
    gc.set_threshold(5, 1, 1)
    try:
        l_g, p_g, g_g = run(True, steps=3)
    finally:
        gc.set_threshold(*old)
    assert all(h is not None for h in g_g['full'])
    assert l_g == l_ref
    for net in p_ref:
        np.testing.assert_array_equal(p_ref[net], p_g[net])
