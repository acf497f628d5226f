"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend 'nccl' on ROCm) or gloo (CPU
tests).  The Triple-GAN step shards by image batch: every replica runs the three solver phases on
its own batch and the three networks' flat gradient buffers are sum-all-reduced once per phase; Adam then applies
grad/world_size identically on every replica, so weights stay bit-identical."""
import os

import torch
import torch.distributed as dist


# TG_DIST_SINGLE=1: create the process group even for ONE replica and run every collective on it — the RCCL code path
# (communicator bound to the device, asynchronous bucket all-reduce beside a graph launch, broadcast, barrier) exercised on a
# one-GPU box; results are unchanged (a one-rank sum is the identity).
SINGLE = os.environ.get('TG_DIST_SINGLE') == '1'

# TG_DIST_BACKEND=rccl-direct: no torch process group — the collectives are tg_comm.h calls (libtg_comm.so) on the launch stream.
_direct = None            # tg.comm.Communicator
_direct_store = None      # keeps the rendezvous TCPStore alive


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


def init(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  Returns (world, rank, local_rank)."""
    global _direct, _direct_store
    world, rank, local = env_world()
    if (world > 1 or SINGLE) and not dist.is_initialized() and _direct is None:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('TG_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'rccl-direct':
            from . import comm
            torch.cuda.set_device(local)
            _direct, _direct_store = comm.rendezvous(world, rank, local)
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


def active():
    """collectives are issued: more than one replica (or the one-replica RCCL rehearsal)."""
    if _direct is not None:
        return _direct.world > 1 or SINGLE
    return dist.is_initialized() and (dist.get_world_size() > 1 or SINGLE)


def quiet_capture_needed():
    """a process group's watchdog thread polls events of pending collectives — it must be idle while a stream captures
    (Training/Train_goodGAN.py:_capture).  The direct backend has no such thread."""
    return active() and _direct is None


def rank():
    if _direct is not None:
        return _direct.rank
    return dist.get_rank() if dist.is_initialized() else 0


def allreduce_sum_(flat):
    """in-place sum over replicas of a flat gradient buffer (one collective per network per iteration)."""
    if active():
        if _direct is not None:
            return _direct.allreduce_sum_(flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_sum_async_(flat):
    """Start the in-place sum on RCCL's own stream — it waits for what the current stream has enqueued so far and runs beside
    whatever is enqueued next (the remaining backward pass).  Returns a handle for wait_(); None on one replica."""
    if active():
        if _direct is not None:                     # stream-ordered on the launch stream: nothing to wait for
            _direct.allreduce_sum_(flat)
            return None
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    return None


def wait_(work):
    """make the CURRENT stream wait for an asynchronous collective (no host block with the nccl backend)."""
    if work is not None:
        work.wait()


def allreduce_mean_(flat):
    if active():
        allreduce_sum_(flat)
        flat.div_(world_size())
    return flat


def broadcast_(flat, src=0):
    if active():
        if _direct is not None:
            return _direct.broadcast_(flat, src)
        dist.broadcast(flat, src=src)
    return flat


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if active():
        if _direct is not None:
            _direct.allreduce_max_f64_(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if active():
        if _direct is not None:                      # a one-element sum every rank must reach, then drain the stream
            _direct.allreduce_sum_(torch.zeros(1, dtype=torch.float32, device='cuda'))
            torch.cuda.synchronize()
        else:
            dist.barrier()


def shutdown():
    global _direct, _direct_store
    if _direct is not None:
        torch.cuda.synchronize()
        _direct.destroy()
        _direct = _direct_store = None
    if dist.is_initialized():
        dist.destroy_process_group()
