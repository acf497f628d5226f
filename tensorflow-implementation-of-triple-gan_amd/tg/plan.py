"""Launch plans: ctypes binding of include/tg_plan.h + the recorder tg.lib.call feeds.

A solver run's launches have fixed arguments once its buffers exist (call-site workspaces, device-resident hyper-parameters, a
counter-based RNG): `Plan.recording()` runs one segment eagerly while every launch entry point that goes through tg.lib.call, and
every event operation of Context's second-stream overlap, is appended to a native plan; `Plan.replay()` then re-issues the segment
from one C loop (tg_plan_replay) — eager launches on two streams without ~15 us of interpreter per launch.  The reference's
counterpart is the TF session executing a cached sub-graph natively (Training/Train_goodGAN.py:266-276)."""
import contextlib
import ctypes as C

from . import lib


class PlanWord(C.Union):
    _fields_ = [("p", C.c_void_p), ("i", C.c_int64), ("f", C.c_float)]


_bound = False


def _bind():
    global _bound
    h = lib.load()
    if _bound:
        return h
    vp = C.c_void_p
    sigs = {
        'tg_plan_create': (C.c_int, [C.POINTER(vp)]),
        'tg_plan_destroy': (C.c_int, [vp]),
        'tg_plan_hold': (C.c_int, [vp, vp, C.c_int64, C.POINTER(vp)]),
        'tg_plan_add_launch': (C.c_int, [vp, C.c_char_p, C.POINTER(PlanWord), C.c_int, C.c_int]),
        'tg_plan_add_event_record': (C.c_int, [vp, vp, C.c_int]),
        'tg_plan_add_stream_wait': (C.c_int, [vp, C.c_int, vp]),
        'tg_plan_length': (C.c_int64, [vp]),
        'tg_plan_launches': (C.c_int64, [vp]),
        'tg_plan_signature': (C.c_char_p, [C.c_char_p]),
        'tg_plan_replay': (C.c_int, [vp, C.POINTER(vp), C.c_int]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    _bound = True
    return h


def _check(h, rc, what):
    if rc != 0:
        raise lib.TgError("%s failed (%d): %s" % (what, rc, h.tg_last_error_string().decode()))


_SIG_CACHE = {}


def signature(name):
    """'pif...' per parameter of launch entry point `name` (without the stream) or None when it is not one."""
    if name not in _SIG_CACHE:
        s = _bind().tg_plan_signature(name.encode())
        _SIG_CACHE[name] = s.decode() if s is not None else None
    return _SIG_CACHE[name]


class Plan(object):
    """one recorded segment.  streams: the hipStream_t handles (ints) of slot 0, 1, ... while recording; replay may pass others."""

    def __init__(self, streams):
        self.h = _bind()
        p = C.c_void_p()
        _check(self.h, self.h.tg_plan_create(C.byref(p)), 'tg_plan_create')
        self.p = p
        self.slots = {int(s): k for k, s in enumerate(streams)}
        self._streams = (C.c_void_p * len(streams))(*[C.c_void_p(int(s)) for s in streams])
        self._keep = []                # event objects the plan names (their handles must outlive it)

    def __del__(self):
        try:
            if getattr(self, 'p', None):
                self.h.tg_plan_destroy(self.p)
                self.p = None
        except Exception:
            pass

    # ---- recording --------------------------------------------------------------------------------
    def _slot(self, stream):
        s = stream.value if isinstance(stream, C.c_void_p) else stream
        s = int(s or 0)
        if s not in self.slots:
            raise lib.TgError("launch plan: a launch went to stream %#x, which is not one of the plan's streams %s"
                              % (s, [hex(k) for k in self.slots]))
        return self.slots[s]

    def _hold(self, obj):
        out = C.c_void_p()
        _check(self.h, self.h.tg_plan_hold(self.p, C.addressof(obj), C.sizeof(obj), C.byref(out)), 'tg_plan_hold')
        return out.value

    def add_launch(self, name, args):
        """args: the Python-side arguments of tg.lib.call(name, *args), stream last."""
        kinds = signature(name)
        if kinds is None:
            raise lib.TgError("launch plan: %s is not a launch entry point" % name)
        if len(args) != len(kinds) + 1:
            raise lib.TgError("launch plan: %s called with %d arguments, the header declares %d" % (name, len(args), len(kinds) + 1))
        words = (PlanWord * max(len(kinds), 1))()
        for k, (kind, v) in enumerate(zip(kinds, args)):
            if kind == 'p':
                if v is None:
                    words[k].p = None
                elif isinstance(v, (C.Array, C.Structure)):
                    words[k].p = self._hold(v)             # host data read at issue time (descriptor, segment table, job array): the plan's copy
                elif isinstance(v, C.c_void_p):
                    words[k].p = v.value
                elif isinstance(v, int):
                    words[k].p = v
                else:
                    raise lib.TgError("launch plan: %s argument %d: cannot record a %s (pass host arrays as ctypes arrays, not cast pointers)"
                                      % (name, k, type(v).__name__))
            elif kind == 'f':
                words[k].i = 0
                words[k].f = float(v.value if isinstance(v, C.c_float) else v)
            else:
                words[k].i = int(getattr(v, 'value', v))
        _check(self.h, self.h.tg_plan_add_launch(self.p, name.encode(), words, len(kinds), self._slot(args[-1])), 'tg_plan_add_launch(%s)' % name)

    def on_call(self, name, args):
        """tg.lib.call hook: launches are recorded, host-side queries (no stream parameter) are not — replay never runs them."""
        if name.startswith('tg_graph_'):
            raise lib.TgError("launch plan: %s inside a recording" % name)
        if signature(name) is not None:
            self.add_launch(name, args)

    def add_record(self, event, stream):
        """event: torch.cuda.Event already recorded once (its handle exists); stream: hipStream_t handle."""
        self._keep.append(event)
        _check(self.h, self.h.tg_plan_add_event_record(self.p, C.c_void_p(event.cuda_event), self._slot(stream)), 'tg_plan_add_event_record')

    def add_wait(self, stream, event):
        self._keep.append(event)
        _check(self.h, self.h.tg_plan_add_stream_wait(self.p, self._slot(stream), C.c_void_p(event.cuda_event)), 'tg_plan_add_stream_wait')

    @contextlib.contextmanager
    def recording(self):
        """every tg.lib.call launch and every Context event operation executed inside is ALSO appended to this plan."""
        if lib._recorder is not None:
            raise lib.TgError("launch plan: a recording is already open")
        lib._recorder = self
        try:
            yield self
        finally:
            lib._recorder = None

    # ---- replay -----------------------------------------------------------------------------------
    def __len__(self):
        return int(self.h.tg_plan_length(self.p))

    @property
    def launches(self):
        return int(self.h.tg_plan_launches(self.p))

    def replay(self, streams=None):
        arr = self._streams if streams is None else (C.c_void_p * len(streams))(*[C.c_void_p(int(s)) for s in streams])
        _check(self.h, self.h.tg_plan_replay(self.p, arr, len(arr)), 'tg_plan_replay')
