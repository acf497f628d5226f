"""Data-parallel trainer path on the GPU box's single MI355X: two ranks share cuda:0 and exchange gradients with
gloo (RCCL cannot put two ranks on one device).  Everything except the transport is the production path: per-rank
batches and RNG streams, hipGraph segments with the collectives between them, grad/world in Adam.
Checks: replicas hold identical weights after every iteration, and they equal a single process that averages the
two shards' gradients itself."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "tensorflow-implementation-of-triple-gan_amd"))
import torch
import gpu_common as G
from oracle import step_cifar10 as S
sizes = dict(B_G=8, L_C=4, U_C=4, L_D=2, U_D=6)
tr = G.fresh_trainer(G.make_config(sizes, USE_HIP_GRAPH={graph}, SEED=5))
rank = tr.rank
tr.set_hyper(lambda_1=0.3, lambda_2=0.5)
full = dict(S.SIZES, **sizes)
sums = []
import bench
from tg import dist as tgdist
tested = tgdist.self_test(tr.cx.device)                    # the start-up check of an N > 1 bench line
tr.measure_exposed(True)
for it in range({iters}):
    tr.feed(S.synth_batch(1000 * rank + it, full))
    tr.sample_latent()
    tr.train_iteration()
    sums.append([float(st.p.double().sum().item()) for st in tr.cx.stores.values()])
exposed = tr.exposed_ms()
tr.measure_exposed(False)
identical = bench.replicas_identical(tr.cx.stores, tr.cx.device)
keep = tr.cx.stores['discriminator'].p[5].clone()
if rank == 1:
    tr.cx.stores['discriminator'].p[5] = torch.nextafter(keep, keep + 1)      # one replica drifts by ONE ulp in one weight: the check must see it
broken = bench.replicas_identical(tr.cx.stores, tr.cx.device)
tr.cx.stores['discriminator'].p[5] = keep
# data-parallel resume: rank 0 writes, every rank restores — weights are rank 0's, the random streams stay per rank
from Training.Saver import Saver
if rank == 0:
    sv = Saver({ckpt!r})
    sv.set_save_path(comments='dp')
    sv.save(tr, 'model_0001.ckpt')
tgdist.barrier()
Saver({ckpt!r}).restore(tr)
tr.sample_latent()
torch.cuda.synchronize()
out = dict(world=tr.world, rank=rank, sums=sums, pick=tr.exec_mode_chosen()[0], losses=tr.losses(), tested=tested, exposed=exposed, identical=identical, broken=broken,
           z=tr.z_g_ph.t.cpu().numpy(), rng_state=tr.cx.rng.state.cpu().numpy(),
           p={{k: st.p.cpu().numpy() for k, st in tr.cx.stores.items()}})
torch.save(out, {out!r} % rank)
torch.distributed.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("graph", [False, True, None])
def test_two_ranks_on_one_gpu_keep_identical_weights(tmp_path, graph):
    """graph = None: config.EXEC_MODE = 'auto' — both candidates are timed with the collectives in place and the replicas decide together
    (the slowest rank's time per candidate), so the run goes past the decision point."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
    from Training.Train_goodGAN import Train
    port = _free_port()
    out = str(tmp_path / "r%d.pt")
    script = tmp_path / "worker.py"
    iters = 3 if graph is not None else Train.AUTO_ITERS + 2
    script.write_text(WORKER.format(root=ROOT, graph=graph, out=out, ckpt=str(tmp_path / 'ckpt'), iters=iters))
    os.makedirs(str(tmp_path / 'ckpt'))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TG_DIST_BACKEND="gloo", TG_DEVICE_INDEX="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    r = [torch.load(out % i, weights_only=False) for i in range(2)]
    assert r[0]['world'] == r[1]['world'] == 2
    if graph is None:
        assert r[0]['pick'] == r[1]['pick'] and r[0]['pick'] in ('plan', 'graph'), (r[0]['pick'], r[1]['pick'])
    # bench.py's N > 1 self-validation on the real trainer: exchange self-test, checksum agreement, exposed exchange time (gloo blocks the
    # host, so every wait is fully exposed: > 0)
    for q in r:
        assert q['tested'] == 2 and q['identical'] is True and q['broken'] is False and q['exposed'] > 0.0, {k: q[k] for k in ('tested', 'identical', 'broken', 'exposed')}
    assert r[0]['sums'] == r[1]['sums']                      # bit-identical weights after every iteration
    for k in r[0]['p']:
        np.testing.assert_array_equal(r[0]['p'][k], r[1]['p'][k])
    assert r[0]['losses'] != r[1]['losses']                  # ... although each rank trained on its own batch
    # after the restore: same step counter, each rank's own seed -> different latents (round-1 advisor finding: rank 0's seed on all)
    assert r[0]['rng_state'][1] == r[1]['rng_state'][1] and r[0]['rng_state'][0] != r[1]['rng_state'][0]
    assert not np.array_equal(r[0]['z'], r[1]['z'])


def test_rccl_is_usable_on_this_box(tmp_path):
    """backend 'nccl' (= RCCL) initialises, all-reduces and tears down on the launch stream the trainer uses (world 1)."""
    code = r"""
import os, sys, torch, torch.distributed as dist
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='%d')
torch.cuda.set_device(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
t = torch.arange(1 << 20, dtype=torch.float32, device='cuda')
dist.all_reduce(t); dist.broadcast(t, src=0); dist.barrier()
m = torch.tensor([2.5], dtype=torch.float64, device='cuda'); dist.all_reduce(m, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert float(t[12345]) == 12345.0 and float(m) == 2.5
dist.destroy_process_group()
print('rccl ok')
""" % _free_port()
    out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0 and b'rccl ok' in out.stdout, out.stdout.decode()[-2000:]
