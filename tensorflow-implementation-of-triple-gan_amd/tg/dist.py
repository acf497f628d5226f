"""Data-parallel plumbing: one process per GPU.  The Triple-GAN step shards by image batch: every replica runs the
three solver phases on its own batch and the three networks' flat gradient buffers are sum-all-reduced in buckets; Adam then
applies grad/world_size identically on every replica, so weights stay bit-identical.

Backends (TG_DIST_BACKEND, default 'rccl-direct' on a GPU, 'gloo' on the CPU):

  rccl-direct  RCCL through the C ABI of include/tg_comm.h (libtg_comm.so).  Every collective of the communicator is issued on ONE
               HIP stream owned by this module (the exchange stream) and ordered against the launch stream with events that only this
               thread touches: there is no watchdog thread, so nothing polls an event while the launch stream captures a hipGraph.
               torch.distributed is used for the rendezvous only (a TCPStore carries the communicator id).
  nccl         torch.distributed's RCCL process group.  Its watchdog thread calls hipEventQuery on every collective still on its
               work list; on ROCm that query fails with hipErrorCapturedEvent while ANY stream of the process captures (also in
               thread-local capture mode) and takes the process down (round 1: gpurun_out/rccl_single_stderr.txt).  This torch build
               has no call that waits for the watchdog's list to drain, so there is no deterministic way to make a capture safe:
               with this backend the trainer does not capture graphs (graphs_allowed() is False) and launches eagerly.
  gloo         CPU tests and the N-ranks-on-one-GPU rehearsal (host-blocking collectives, no watchdog).
"""
import os

import torch
import torch.distributed as dist


# TG_DIST_SINGLE=1: create the communicator / process group even for ONE replica and run every collective on it — the RCCL code
# path (communicator bound to the device, bucketed exchange on its own stream beside a graph launch, broadcast, barrier) exercised
# on a one-GPU box; results are unchanged (a one-rank sum is the identity).
SINGLE = os.environ.get('TG_DIST_SINGLE') == '1'

_direct = None            # tg.comm.Communicator
_direct_store = None      # keeps the rendezvous TCPStore alive
_xstream = None           # the exchange stream of the direct backend


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


def default_backend():
    return os.environ.get('TG_DIST_BACKEND') or ('rccl-direct' if torch.cuda.is_available() else 'gloo')


def init(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  Returns (world, rank, local_rank)."""
    global _direct, _direct_store, _xstream
    world, rank, local = env_world()
    if (world > 1 or SINGLE) and not dist.is_initialized() and _direct is None:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = default_backend()
        if backend == 'rccl-direct':
            from . import comm
            if torch.cuda.device_count() <= local:
                raise RuntimeError("rank %d wants HIP device %d but only %d are visible" % (rank, local, torch.cuda.device_count()))
            torch.cuda.set_device(local)
            _direct, _direct_store = comm.rendezvous(world, rank, local)
            _xstream = torch.cuda.Stream(device=local)
            return world, rank, local
        if 'TG_DEVICE_INDEX' in os.environ:        # rehearsal of N ranks on one GPU (gloo): every rank uses this device
            local = int(os.environ['TG_DEVICE_INDEX'])
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local)      # bind the communicator to this rank's GPU up front
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    if 'TG_DEVICE_INDEX' in os.environ:
        local = int(os.environ['TG_DEVICE_INDEX'])
    return world, rank, local


def world_size():
    if _direct is not None:
        return _direct.world
    return dist.get_world_size() if dist.is_initialized() else 1


def backend_name():
    if _direct is not None:
        return 'rccl-direct'
    return dist.get_backend() if dist.is_initialized() else None


def rccl_ranks():
    """number of ranks the RCCL communicator itself reports (tg_comm_count / the process group's size); 1 without one."""
    if _direct is not None:
        return _direct.count()[0]
    return dist.get_world_size() if dist.is_initialized() else 1


def active():
    """collectives are issued: more than one replica (or the one-replica RCCL rehearsal)."""
    if _direct is not None:
        return _direct.world > 1 or SINGLE
    return dist.is_initialized() and (dist.get_world_size() > 1 or SINGLE)


def graphs_allowed():
    """May the trainer capture hipGraphs while this backend is active?  Not with the 'nccl' process group (module docstring)."""
    return not (dist.is_initialized() and dist.get_backend() == 'nccl')


def rank():
    if _direct is not None:
        return _direct.rank
    return dist.get_rank() if dist.is_initialized() else 0


class _StreamWork(object):
    """handle of a collective issued on the exchange stream: `done` is recorded behind it."""
    __slots__ = ('done',)

    def __init__(self, done):
        self.done = done

    def wait(self):
        torch.cuda.current_stream().wait_event(self.done)


def _on_xstream(fn):
    """run fn(stream handle) on the exchange stream after everything the current stream has enqueued so far."""
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream())
    _xstream.wait_event(ready)
    fn(_xstream.cuda_stream)
    done = torch.cuda.Event()
    done.record(_xstream)
    return _StreamWork(done)


def allreduce_sum_async_(flat):
    """Start the in-place sum over the replicas: it waits for what the current stream has enqueued so far and runs on the exchange
    stream beside whatever is enqueued next (the remaining backward pass).  Returns a handle for wait_(); None on one replica."""
    if not active():
        return None
    if _direct is not None:
        return _on_xstream(lambda s: _direct.allreduce_sum_(flat, stream=s))
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)


def wait_(work):
    """make the CURRENT stream wait for an asynchronous collective (no host block on the RCCL backends)."""
    if work is not None:
        work.wait()


def allreduce_sum_(flat):
    """in-place sum over replicas of a flat gradient buffer, ordered on the current stream."""
    wait_(allreduce_sum_async_(flat))
    return flat


def allreduce_mean_(flat):
    if active():
        allreduce_sum_(flat)
        flat.div_(world_size())
    return flat


def broadcast_(flat, src=0):
    if active():
        if _direct is not None:
            _on_xstream(lambda s: _direct.broadcast_(flat, src, stream=s)).wait()
        else:
            dist.broadcast(flat, src=src)
    return flat


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if active():
        if _direct is not None:
            _on_xstream(lambda s: _direct.allreduce_max_f64_(t, stream=s)).wait()
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def minmax_over_ranks(values, device):
    """(min, max) over the replicas of each entry of `values` (floats exactly representable in float64) — one max-all-reduce of
    [v, -v].  What bench.py's `replicas_identical` is computed from."""
    v = torch.tensor([float(x) for x in values], dtype=torch.float64, device=device)
    t = torch.cat([v, -v])
    if active():
        if _direct is not None:
            _on_xstream(lambda s: _direct.allreduce_max_f64_(t, stream=s)).wait()
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t = t.cpu()
    n = len(values)
    return [-float(x) for x in t[n:]], [float(x) for x in t[:n]]


def decide_together(times, device):
    """{candidate: seconds on THIS rank} -> (the candidate every replica takes, {candidate: the slowest rank's seconds}): replicas
    must launch the same way (their collectives interleave with the segments of an iteration), so the choice is made from the
    max over the ranks of every candidate's time — identical on all ranks by construction — and ties go to the first candidate in
    `times`' order.  Every rank must call this in the same iteration (one max-all-reduce)."""
    names = list(times)
    _, worst = minmax_over_ranks([times[k] for k in names], device)
    pick = min(range(len(names)), key=lambda i: (worst[i], i))
    return names[pick], dict(zip(names, worst))


def self_test(device, n=4096):
    """Start-up check of the exchange every replica is about to rely on: a sum-all-reduce of a vector holding rank + 1 must give
    world (world + 1) / 2 in every element on every rank, and a broadcast from rank 0 must arrive.  Raises RuntimeError naming what
    failed; returns the number of ranks that took part (1 without replicas)."""
    if not active():
        return 1
    w, r = world_size(), rank()
    t = torch.full((n,), float(r + 1), dtype=torch.float32, device=device)
    allreduce_sum_(t)
    want = w * (w + 1) / 2.0
    bad = int((t != want).sum().item())
    if bad:
        raise RuntimeError("exchange self-test: sum-all-reduce over %d ranks gave %r in %d of %d elements on rank %d, expected %r "
                           "(backend %s)" % (w, float(t[0].item()), bad, n, r, want, backend_name()))
    b = torch.full((n,), float(1000 + r), dtype=torch.float32, device=device)
    broadcast_(b, 0)
    if torch.device(device).type == 'cuda':
        torch.cuda.synchronize()
    if float(b.min().item()) != 1000.0 or float(b.max().item()) != 1000.0:
        raise RuntimeError("exchange self-test: broadcast from rank 0 did not arrive on rank %d" % r)
    return w


def barrier():
    if active():
        if _direct is not None:                      # a one-element sum every rank must reach, then drain the device
            allreduce_sum_(torch.zeros(1, dtype=torch.float32, device='cuda'))
            torch.cuda.synchronize()
        else:
            dist.barrier()


def shutdown():
    global _direct, _direct_store, _xstream
    if _direct is not None:
        torch.cuda.synchronize()
        _direct.destroy()
        _direct = _direct_store = _xstream = None
    if dist.is_initialized():
        dist.destroy_process_group()
