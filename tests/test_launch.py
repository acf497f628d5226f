"""`bench.py --gpus N` starts its own ranks (tg/launch.py) — exercised on the CPU with gloo as the transport, world_size 2:
the job's single JSON line comes from rank 0 and reports N replicas; a failing rank fails the job; without N devices the bench
refuses instead of printing a one-GPU figure; the communicator-id rendezvous of the rccl-direct backend (tg/comm.py:exchange_id)
hands every rank the same bytes, both when rank 0 serves the store and under a launcher's agent store."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd")

PARENT = r'''
import sys
sys.path.insert(0, {pkg!r})
from tg import launch
sys.exit(launch.spawn_ranks({n}, [{script!r}], need_devices=False))
'''

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, {pkg!r})
sys.path.insert(0, {root!r})
import torch
from tg import dist as tgdist
import bench
mode = {mode!r}
rank = int(os.environ['RANK'])
if mode == 'fail' and rank == int(os.environ['WORLD_SIZE']) - 1:
    sys.exit(3)
world, rank, local = tgdist.init(backend='gloo')
t = torch.full((4,), float(rank + 1))
tgdist.allreduce_sum_(t)
slow = tgdist.max_over_ranks(1.0 + rank, torch.device('cpu'))
# the self-validation fields of an N > 1 bench line (bench.py): exchange self-test, replica checksums
tested = tgdist.self_test('cpu')
lo, hi = tgdist.minmax_over_ranks([float(rank), 5.0], 'cpu')
class Store(object):
    def __init__(self, t):
        self.p = t
same = bench.replicas_identical(dict(a=Store(torch.arange(1000, dtype=torch.float32) * 0.37), b=Store(torch.ones(33))), 'cpu')
differ = bench.replicas_identical(dict(a=Store(torch.arange(1000, dtype=torch.float32) * 0.37 + (1e-7 if rank else 0.0))), 'cpu')
# the execution-mode decision of EXEC_MODE = 'auto' (Train._auto_mode -> tg.dist.decide_together): locally every rank but the last finds
# 'plan' faster; the last rank's plan time is the slowest of all -> every replica must take 'graph', with the same timings
local = dict(plan=14.5e-3 + (2e-3 if rank == world - 1 else 0.0), graph=15.0e-3 + 1e-5 * rank)
pick, worst = tgdist.decide_together(local, 'cpu')
picks = tgdist.minmax_over_ranks([float(pick == 'graph'), worst['plan'], worst['graph']], 'cpu')
tgdist.barrier()
if mode == 'fail':
    time.sleep(120)                       # never reached by a healthy job: the launcher stops this rank when rank 1 fails
if rank == 0:
    print(json.dumps(dict(n_gpus=world, ranks=tgdist.rccl_ranks(), backend=tgdist.backend_name(), sum=t.tolist(), slow=slow,
                          spawned=os.environ.get('TG_SPAWNED'), tested=tested, lo=lo, hi=hi, same=same, differ=differ, pick=pick,
                          picks_agree=picks[0] == picks[1], worst=worst)), flush=True)
else:
    print("rank 1 must stay silent on stdout", flush=True)
tgdist.shutdown()
'''


def _job(tmp_path, mode, n=2):
    w = tmp_path / 'worker.py'
    w.write_text(WORKER.format(pkg=PKG, root=ROOT, mode=mode))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'TG_DIST_BACKEND')}
    return subprocess.run([sys.executable, '-c', PARENT.format(pkg=PKG, n=n, script=str(w))], env=env, capture_output=True, text=True,
                          timeout=300)


def test_spawned_ranks_report_one_line_from_rank_zero(tmp_path):
    r = _job(tmp_path, 'ok')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip() and not l.startswith('[Gloo]')]       # gloo's own connection notice
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == dict(n_gpus=2, ranks=2, backend='gloo', sum=[3.0] * 4, slow=2.0, spawned='1',
                       tested=2, lo=[0.0, 5.0], hi=[1.0, 5.0], same=True, differ=False, pick='graph', picks_agree=True,
                       worst=dict(plan=16.5e-3, graph=15.0e-3 + 1e-5))


def test_eight_spawned_ranks(tmp_path):
    """the driver's N = 8 launch shape, rehearsed on the CPU over gloo: rendezvous, the exchange self-test, replica checksums, the
    collective execution-mode decision (one straggler rank decides for all) — one JSON line from rank 0."""
    r = _job(tmp_path, 'ok', n=8)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip() and not l.startswith('[Gloo]')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == dict(n_gpus=8, ranks=8, backend='gloo', sum=[36.0] * 4, slow=8.0, spawned='1',
                       tested=8, lo=[0.0, 5.0], hi=[7.0, 5.0], same=True, differ=False, pick='graph', picks_agree=True,
                       worst=dict(plan=16.5e-3, graph=15.0e-3 + 7e-5))


def test_a_failing_rank_of_eight_fails_the_job(tmp_path):
    t0 = time.time()
    r = _job(tmp_path, 'fail', n=8)
    assert r.returncode == 3 and '{' not in r.stdout, (r.returncode, r.stdout, r.stderr[-1000:])
    assert time.time() - t0 < 120         # the seven healthy ranks were terminated, not waited for


def test_communicator_id_rendezvous_of_eight(tmp_path):
    w = tmp_path / 'idw.py'
    w.write_text(ID_WORKER.format(pkg=PKG))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'TORCHELASTIC_USE_AGENT_STORE')}
    r = subprocess.run([sys.executable, '-c', PARENT.format(pkg=PKG, n=8, script=str(w))], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == 'id ok', (r.stdout, r.stderr[-2000:])


def test_a_failing_rank_fails_the_job_and_stops_the_others(tmp_path):
    t0 = time.time()
    r = _job(tmp_path, 'fail')
    assert r.returncode == 3 and '{' not in r.stdout, (r.returncode, r.stdout, r.stderr[-1000:])
    assert time.time() - t0 < 90          # rank 0 was terminated, not waited for


def test_bench_refuses_more_replicas_than_devices():
    """no GPU in the build container: --gpus 2 must exit non-zero and print no JSON line (round 1 printed n_gpus = 1)."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box could run two replicas")
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip() and 'HIP device' in r.stderr
    # and a world that disagrees with --gpus is an error as well, also for WORLD_SIZE=1
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=dict(env, WORLD_SIZE='1', RANK='0'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip() and 'WORLD_SIZE=1' in r.stderr


def test_device_count_reads_sysfs_and_the_launcher_never_opens_the_gpu(tmp_path):
    """advisor (round 2): the launcher must not initialise HIP before it forks — devices are counted from the KFD topology files."""
    sys.path.insert(0, PKG)
    from tg import launch
    nodes = tmp_path / 'nodes'
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):            # two CPU nodes, three GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / 'properties').write_text('cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n' % (64 if simd == 0 else 0, simd))
    none = str(tmp_path / 'no-render-nodes-*')
    assert launch.visible_devices({}, str(nodes), none) == 3
    assert launch.visible_devices({'HIP_VISIBLE_DEVICES': '0,2'}, str(nodes), none) == 2
    assert launch.visible_devices({'HIP_VISIBLE_DEVICES': ''}, str(nodes), none) == 0
    assert launch.visible_devices({'ROCR_VISIBLE_DEVICES': '1', 'HIP_VISIBLE_DEVICES': '0,1'}, str(nodes), none) == 1
    assert launch.visible_devices({'HIP_VISIBLE_DEVICES': '0,7,1'}, str(nodes), none) == 1      # an out-of-range index ends the list
    assert launch.visible_devices({'HIP_VISIBLE_DEVICES': '-1'}, str(nodes), none) == 0         # the usual way to hide every device
    assert launch.visible_devices({'HIP_VISIBLE_DEVICES': '1,-1,2'}, str(nodes), none) == 1     # a negative index ends the list
    assert launch.visible_devices({'HIP_VISIBLE_DEVICES': '2,2,0,2'}, str(nodes), none) == 2    # a repeated index is one device
    (tmp_path / 'renderD128').write_text('')                                                    # container: one device file passed through
    assert launch.visible_devices({}, str(nodes), str(tmp_path / 'renderD*')) == 1
    assert launch.visible_devices({}, str(tmp_path / 'absent'), str(tmp_path / 'renderD*')) == 1
    # the parent of a job holds no /dev/kfd descriptor when it spawns (checked inside spawn_ranks as well)
    code = ("import sys; sys.path.insert(0, %r); from tg import launch; n = launch.visible_devices(); "
            "assert not launch.holds_gpu(); assert 'torch' not in sys.modules; print(n)" % PKG)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().isdigit(), (r.stdout, r.stderr[-1000:])


ID_WORKER = r'''
import os, sys
sys.path.insert(0, {pkg!r})
from tg import comm
world, rank = int(os.environ['WORLD_SIZE']), int(os.environ['RANK'])
uid, store = comm.exchange_id(world, rank, lambda: bytes(range(128)) if rank == 0 else b'never')
assert uid == bytes(range(128)), uid
if rank == 0:
    print('id ok', flush=True)
'''


def test_communicator_id_rendezvous(tmp_path):
    w = tmp_path / 'idw.py'
    w.write_text(ID_WORKER.format(pkg=PKG))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'TORCHELASTIC_USE_AGENT_STORE')}
    r = subprocess.run([sys.executable, '-c', PARENT.format(pkg=PKG, n=2, script=str(w))], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == 'id ok', (r.stdout, r.stderr[-2000:])


def test_rendezvous_names_the_rank_that_never_arrived(tmp_path):
    """world_size 2 with only rank 0 (and only rank 1) started: a bounded wait and an error that says who is missing."""
    sys.path.insert(0, PKG)
    from tg import launch
    w = tmp_path / 'idw.py'
    w.write_text(ID_WORKER.format(pkg=PKG))
    base = {k: v for k, v in os.environ.items() if k not in ('TORCHELASTIC_USE_AGENT_STORE', 'TORCHELASTIC_RUN_ID')}
    env = dict(base, RANK='0', WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(launch.free_port()),
               TG_RENDEZVOUS_TIMEOUT='3')
    t0 = time.time()
    r = subprocess.run([sys.executable, str(w)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'rank(s) [1] of 2 did not arrive' in r.stderr, r.stderr[-1500:]
    assert time.time() - t0 < 60
    env = dict(env, RANK='1', LOCAL_RANK='1', MASTER_PORT=str(launch.free_port()))
    r = subprocess.run([sys.executable, str(w)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and ('no rendezvous store' in r.stderr or 'did not publish' in r.stderr or 'never confirmed' in r.stderr), r.stderr[-1500:]


def test_communicator_id_rendezvous_under_an_agent_store(tmp_path):
    """torchrun's elastic agent already listens on MASTER_PORT (TORCHELASTIC_USE_AGENT_STORE=True): every rank connects as a client."""
    import datetime
    import torch.distributed as dist
    sys.path.insert(0, PKG)
    from tg import launch
    port = launch.free_port()
    agent = dist.TCPStore('127.0.0.1', port, None, is_master=True, timeout=datetime.timedelta(seconds=60), wait_for_workers=False)
    w = tmp_path / 'idw.py'
    w.write_text(ID_WORKER.format(pkg=PKG))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   TORCHELASTIC_USE_AGENT_STORE='True', TORCHELASTIC_RUN_ID='job7')
        procs.append(subprocess.Popen([sys.executable, str(w)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert outs[0][0].strip() == 'id ok'
    # an elastic restart: the agent's store still holds the dead incarnation's id under restart count 0; the new group (count 1)
    # must read only what ITS rank 0 writes.  Rank 1 starts first and has to wait for rank 0 instead of taking the stale id.
    stale_w = tmp_path / 'idw2.py'
    stale_w.write_text(ID_WORKER.format(pkg=PKG).replace('bytes(range(128))', 'bytes(range(1, 129))'))
    procs = []
    for rank in (1, 0):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   TORCHELASTIC_USE_AGENT_STORE='True', TORCHELASTIC_RUN_ID='job7', TORCHELASTIC_RESTART_COUNT='1')
        procs.append(subprocess.Popen([sys.executable, str(stale_w)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        if rank == 1:
            time.sleep(3.0)
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    del agent
