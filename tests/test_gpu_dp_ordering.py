"""Stream-ordering of the gradient exchange, checked on ONE GPU.

A one-rank all-reduce is the identity, so tests/test_gpu_rccl_single.py cannot see an exchange that reads the gradients too early
or an optimiser step that does not wait for it; gloo (tests/test_gpu_dp.py) blocks the host, which hides the same mistakes.  Here
tg.dist is replaced by an emulation of the RCCL process group's stream semantics for TWO replicas that hold identical data:

  * the "all-reduce" runs on its own stream, which first waits for what the launch stream has enqueued so far (as
    ProcessGroupNCCL does), then idles for a long while (a slow transfer) and finally doubles the buffer (g + g);
  * Work.wait() makes the launch stream wait for it — nothing blocks the host.

Adam applies grad / world = (2 g) / 2 = g exactly, so every result must equal the plain single-process run BIT FOR BIT — unless
a consumer ran before the exchange finished (it would see g instead of 2 g) or the exchange started before the gradients were final.
Runs with hipGraph segments (every bucket's exchange beside the graph launch of the next backward segment) and eagerly."""
import numpy as np
import pytest

from oracle import step_cifar10 as S
import gpu_common as G

pytestmark = pytest.mark.gpu
SIZES = dict(B_G=8, L_C=4, U_C=4, L_D=2, U_D=6)


def run(graph, steps=4):
    tr = G.fresh_trainer(G.make_config(SIZES, USE_HIP_GRAPH=graph, SEED=5))
    tr.set_hyper(lambda_1=0.3, lambda_2=0.5)
    full = dict(S.SIZES, **SIZES)
    losses = []
    for it in range(steps):
        tr.feed(S.synth_batch(it, full))
        tr.sample_latent()
        tr.train_iteration()
        losses.append(tr.losses())
    return tr, losses, {k: st.p.detach().cpu().numpy().copy() for k, st in tr.cx.stores.items()}


class _Exchange(object):
    """what tg.dist offers the trainer, for two identical replicas on one device."""

    def __init__(self, torch, honour_wait=True):
        self.torch = torch
        self.honour_wait = honour_wait
        self.stream = torch.cuda.Stream()
        self.calls = dict(sync=0, asynchronous=0, waits=0)

    def _start(self, flat):
        torch = self.torch
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self.stream.wait_event(ready)                        # the collective sees everything enqueued before the call
        with torch.cuda.stream(self.stream):
            torch.cuda._sleep(3_000_000)                     # ~1.5 ms: a transfer much slower than the kernels that follow
            flat.mul_(2.0)
            done = torch.cuda.Event()
            done.record(self.stream)
        ex = self

        class Work(object):
            def wait(self):
                ex.calls['waits'] += 1
                if ex.honour_wait:
                    torch.cuda.current_stream().wait_event(done)
        return Work()

    def allreduce_sum_(self, flat):
        self.calls['sync'] += 1
        self._start(flat).wait()
        return flat

    def allreduce_sum_async_(self, flat):
        self.calls['asynchronous'] += 1
        return self._start(flat)


def _patch(monkeypatch, ex):
    from tg import dist as tgdist
    monkeypatch.setattr(tgdist, 'init', lambda backend=None: (2, 0, 0))
    monkeypatch.setattr(tgdist, 'active', lambda: True)
    monkeypatch.setattr(tgdist, 'world_size', lambda: 2)
    monkeypatch.setattr(tgdist, 'graphs_allowed', lambda: True)
    monkeypatch.setattr(tgdist, 'allreduce_sum_', ex.allreduce_sum_)
    monkeypatch.setattr(tgdist, 'allreduce_sum_async_', ex.allreduce_sum_async_)
    monkeypatch.setattr(tgdist, 'wait_', lambda w: w.wait() if w is not None else None)
    monkeypatch.setattr(tgdist, 'broadcast_', lambda flat, src=0: flat)
    monkeypatch.setattr(tgdist, 'barrier', lambda: None)


@pytest.mark.parametrize("graph", [True, False])
def test_exchange_is_ordered_against_producers_and_consumers(graph, monkeypatch):
    import torch
    _, l_ref, p_ref = run(graph)
    ex = _Exchange(torch)
    _patch(monkeypatch, ex)
    tr, l_dp, p_dp = run(graph)
    torch.cuda.synchronize()
    assert tr.world == 2
    # per iteration every gradient bucket goes out asynchronously behind the segment that completes it — discriminator 3 (one per
    # resolution stage), generator 2 (gg_dconv0 ... first), classifier 2 — and is waited for by the segment that applies the optimiser step
    assert ex.calls['asynchronous'] == 4 * 7 and ex.calls['sync'] == 0 and ex.calls['waits'] == 4 * 7, ex.calls
    assert len(tr._segments()) == 8 and [w for _, _, w in tr._segments()] == [False, False, False, True, False, True, False, True]
    assert l_dp == l_ref
    for k in p_ref:
        np.testing.assert_array_equal(p_dp[k], p_ref[k], err_msg=k)


def test_the_check_sees_a_missing_wait(monkeypatch):
    """sensitivity of the test above: the same emulation with Work.wait() doing nothing (the optimiser steps read the gradients before
    the slow exchange has doubled them, the doubling then lands in the next phase) must NOT reproduce the plain run."""
    import torch
    _, l_ref, p_ref = run(True)
    _patch(monkeypatch, _Exchange(torch, honour_wait=False))
    _, l_dp, p_dp = run(True)
    torch.cuda.synchronize()
    assert l_dp != l_ref or any(not np.array_equal(p_dp[k], p_ref[k]) for k in p_ref)
