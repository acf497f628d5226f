"""Data-parallel logic on CPU with gloo, world_size 2: every replica computes the gradients of ITS
image shard, tg.dist sums them, Adam applies grad/world — which must equal the gradient of the mean loss over
the union batch (the discriminator has no batch statistics, so the identity is exact up to rounding).
The per-replica compute is the oracle here (the HIP kernels need a GPU); what is under test is the host-side
protocol the trainer uses: env-driven init, flat-buffer sum all-reduce, weight broadcast, max-over-ranks timing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from tg import dist as tgdist
    from oracle import nets_cifar10 as N
    from oracle import step_cifar10 as S
    from oracle import tf_ops as T
    w, r, _ = tgdist.init(backend='gloo')
    assert (w, r) == (world, rank) and tgdist.world_size() == world and tgdist.rank() == rank
    P = {k: v for k, v in S.init_params(0).items() if k.startswith('discriminator/')}
    names = sorted(P)
    # rank 1 starts from different weights: the broadcast from rank 0 must make the replicas identical
    flat = torch.from_numpy(np.concatenate([(P[k] + (0.5 if rank else 0.0)).reshape(-1) for k in names]).astype(np.float32))
    tgdist.broadcast_(flat, src=0)
    off = 0
    for k in names:
        P[k] = flat[off:off + P[k].size].numpy().reshape(P[k].shape).copy()
        off += P[k].size
    n = 3
    rng = np.random.default_rng(100 + rank)                     # per-rank data seed
    img = np.tanh(rng.standard_normal((n, 32, 32, 3))).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
    rnd = S.synth_rnd(7 + rank, dict(S.SIZES, B_G=n))['G']['D_fake']
    logits, c = N.discriminator_fwd(P, img, y, rnd)
    _, dl = T.bce_mean(logits, np.ones_like(logits))
    g, _ = N.discriminator_bwd(P, c, dl.astype(np.float32), rnd)
    gflat = torch.from_numpy(np.concatenate([g[k].reshape(-1) for k in names if k in g]).astype(np.float32))
    tgdist.allreduce_sum_(gflat)
    t = tgdist.max_over_ranks(1.0 + rank, torch.device('cpu'))
    tgdist.barrier()
    torch.save(dict(img=img, y=y, rnd=rnd, gsum=gflat.numpy(), t=t, names=[k for k in names if k in g], P=P), out % rank)
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_replicas_average_gradients(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "rank%d.pt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r = [torch.load(out % i, weights_only=False) for i in range(world)]
    np.testing.assert_array_equal(r[0]['gsum'], r[1]['gsum'])                 # identical result on every replica
    for k in r[0]['P']:
        np.testing.assert_array_equal(r[0]['P'][k], r[1]['P'][k])             # broadcast made the weights identical
    assert r[0]['t'] == r[1]['t'] == 2.0                                        # max over ranks
    # mean of per-shard gradients == gradient of the mean loss over the union batch
    from oracle import nets_cifar10 as N
    from oracle import tf_ops as T
    P = r[0]['P']
    img = np.concatenate([r[0]['img'], r[1]['img']])
    y = np.concatenate([r[0]['y'], r[1]['y']])
    rnd = {k: np.concatenate([r[0]['rnd'][k], r[1]['rnd'][k]]) for k in r[0]['rnd']}
    logits, c = N.discriminator_fwd(P, img, y, rnd)
    _, dl = T.bce_mean(logits, np.ones_like(logits))
    g, _ = N.discriminator_bwd(P, c, dl.astype(np.float32), rnd)
    ref = np.concatenate([g[k].reshape(-1) for k in r[0]['names']])
    got = r[0]['gsum'] / world
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_single_process_is_a_noop():
    for p in (ROOT, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from tg import dist as tgdist
    t = torch.arange(4, dtype=torch.float32)
    assert tgdist.world_size() == 1 and tgdist.rank() == 0
    assert torch.equal(tgdist.allreduce_sum_(t.clone()), t)
    assert tgdist.max_over_ranks(3.5, torch.device('cpu')) == 3.5
