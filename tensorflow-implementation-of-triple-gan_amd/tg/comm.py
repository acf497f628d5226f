"""ctypes binding of include/tg_comm.h (libtg_comm.so): RCCL collectives called directly on the launch stream — no process group,
no watchdog thread, legal inside hipStream capture.  Selected by TG_DIST_BACKEND=rccl-direct (tg/dist.py); the communicator id
travels from rank 0 to the other ranks through a torch.distributed.TCPStore at MASTER_ADDR:MASTER_PORT (rendezvous only).

libtg_comm.so is built WITHOUT a link to librccl so that the process holds exactly one RCCL: the copy PyTorch ships (already
mapped once `import torch` has run on a ROCm build) is promoted to the global symbol scope here, /opt/rocm's otherwise."""
import ctypes as C
import os

from . import lib

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libtg_comm.so")
HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "tg_comm.h")
ID_BYTES = 128

_lib = None


def _rccl_candidates():
    try:
        import torch
        yield os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    except ImportError:
        pass
    yield "/opt/rocm/lib/librccl.so.1"
    yield "librccl.so.1"


def load():
    """dlopen RCCL (RTLD_GLOBAL) then libtg_comm.so; raises TgError when either is missing — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise lib.TgError("%s not built: run `make -C %s` (or __graft_entry__.build())" % (LIB_PATH, os.path.dirname(LIB_PATH)))
    err = None
    for cand in _rccl_candidates():
        if os.path.isabs(cand) and not os.path.exists(cand):
            continue
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            err = None
            break
        except OSError as e:
            err = e
    if err is not None:
        raise lib.TgError("no RCCL library could be loaded: %s" % err)
    l = C.CDLL(LIB_PATH)
    vp, i64, ci = C.c_void_p, C.c_int64, C.c_int
    l.tg_comm_last_error_string.restype = C.c_char_p
    l.tg_comm_last_error_string.argtypes = []
    sigs = {
        'tg_comm_unique_id': [vp],
        'tg_comm_init_rank': [C.POINTER(vp), ci, vp, ci, ci],
        'tg_comm_count': [vp, C.POINTER(ci), C.POINTER(ci)],
        'tg_allreduce_sum_f32': [vp, i64, vp, vp],
        'tg_allreduce_max_f64': [vp, i64, vp, vp],
        'tg_broadcast_f32': [vp, i64, ci, vp, vp],
        'tg_comm_destroy': [vp],
    }
    for name, argt in sigs.items():
        f = getattr(l, name)
        f.argtypes = argt
        f.restype = ci
    _lib = l
    return l


def call(name, *args):
    l = load()
    rc = getattr(l, name)(*args)
    if rc != 0:
        raise lib.TgError("%s failed (%d): %s" % (name, rc, l.tg_comm_last_error_string().decode()))


class Communicator(object):
    """one RCCL communicator of `world` ranks bound to HIP device `device`."""

    def __init__(self, world, rank, device, unique_id):
        assert len(unique_id) == ID_BYTES
        self.world, self.rank, self.device = world, rank, device
        self.handle = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), ID_BYTES)
        call('tg_comm_init_rank', C.byref(self.handle), world, C.cast(buf, C.c_void_p), rank, device)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(ID_BYTES)
        call('tg_comm_unique_id', C.cast(buf, C.c_void_p))
        return buf.raw

    def allreduce_sum_(self, t, stream=None):
        assert t.dtype.is_floating_point and t.element_size() == 4 and t.is_contiguous()
        call('tg_allreduce_sum_f32', lib.ptr(t), t.numel(), self.handle, lib.cur_stream() if stream is None else stream)
        return t

    def allreduce_max_f64_(self, t, stream=None):
        assert t.element_size() == 8 and t.is_contiguous()
        call('tg_allreduce_max_f64', lib.ptr(t), t.numel(), self.handle, lib.cur_stream() if stream is None else stream)
        return t

    def broadcast_(self, t, root=0, stream=None):
        assert t.element_size() == 4 and t.is_contiguous()
        call('tg_broadcast_f32', lib.ptr(t), t.numel(), root, self.handle, lib.cur_stream() if stream is None else stream)
        return t

    def count(self):
        """(nranks, rank) as the communicator itself reports them (tg_comm_count)."""
        n, r = C.c_int(), C.c_int()
        call('tg_comm_count', self.handle, C.byref(n), C.byref(r))
        return n.value, r.value

    def destroy(self):
        if self.handle:
            call('tg_comm_destroy', self.handle)
            self.handle = C.c_void_p()


def exchange_id(world, rank, make_id, key='tg_comm_id', timeout=None):
    """Rendezvous only: rank 0 calls make_id() and publishes the bytes in a TCPStore at MASTER_ADDR:MASTER_PORT, every rank
    reads them.  Under torchrun the elastic agent already serves a store on that port (TORCHELASTIC_USE_AGENT_STORE=True): every
    rank then connects as a client, exactly as torch's own env:// rendezvous does.  Returns (id bytes, store).

    Bounded (TG_RENDEZVOUS_TIMEOUT seconds, default 300): every rank announces itself under '<prefix>/here/<rank>'; when the time is
    up, rank 0 fails with the list of ranks that never arrived and the other ranks with "rank 0 did not publish" / "rank 0 never confirmed
    the group" (they wait for a 'go' key rank 0 sets once every '<prefix>/here/<r>' is present) — instead of a communicator initialisation
    that hangs on a rank that is not there."""
    import datetime
    import torch.distributed as dist
    timeout = float(os.environ.get('TG_RENDEZVOUS_TIMEOUT', '300')) if timeout is None else float(timeout)
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '29500'))
    agent = os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True'
    try:
        store = dist.TCPStore(addr, port, world, is_master=(rank == 0 and not agent), timeout=datetime.timedelta(seconds=timeout),
                              wait_for_workers=False)
    except Exception as e:
        raise lib.TgError("rank %d of %d: no rendezvous store at %s:%d within %.0f s (%s)" % (rank, world, addr, port, timeout, e))
    # the agent's store outlives a restarted worker group: key the id by the incarnation, or ranks of the new group could read the
    # communicator id of the dead one before rank 0 has overwritten it
    prefix = '%s/%s' % (os.environ.get('TORCHELASTIC_RUN_ID', 'run'), os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'))
    key = '%s/%s' % (prefix, key)
    here = lambda r: '%s/here/%d' % (prefix, r)
    store.set(here(rank), b'1')
    go = '%s/go' % prefix
    if rank == 0:
        store.set(key, make_id())
        try:
            store.wait([here(r) for r in range(world)], datetime.timedelta(seconds=timeout))
        except Exception:
            missing = [r for r in range(world) if not store.check([here(r)])]
            raise lib.TgError("rendezvous at %s:%d: rank(s) %s of %d did not arrive within %.0f s" % (addr, port, missing, world, timeout))
        store.set(go, b'1')                # everyone is here: only now may any rank walk into the communicator initialisation
    try:
        store.wait([key], datetime.timedelta(seconds=timeout))
    except Exception:
        raise lib.TgError("rendezvous at %s:%d: rank 0 did not publish the communicator id within %.0f s (rank %d of %d waiting)"
                          % (addr, port, timeout, rank, world))
    if rank != 0:
        # the other ranks wait (bounded) for rank 0's word that ALL ranks arrived — a rank that read the id at once would otherwise sit in
        # ncclCommInitRank, which has no timeout, waiting for one that never comes
        try:
            store.wait([go], datetime.timedelta(seconds=timeout + 5))
        except Exception:
            missing = [r for r in range(world) if not store.check([here(r)])]
            raise lib.TgError("rendezvous at %s:%d: rank 0 never confirmed the group within %.0f s (rank %d of %d waiting; not arrived: %s)"
                              % (addr, port, timeout, rank, world, missing))
    return store.get(key), store


def rendezvous(world, rank, device):
    """every rank joins ONE communicator whose id rank 0 draws (exchange_id)."""
    if world == 1:
        return Communicator(1, 0, device, Communicator.unique_id()), None
    uid, store = exchange_id(world, rank, Communicator.unique_id)
    return Communicator(world, rank, device, uid), store
