"""Eager execution context under the reference's op surface.

TensorFlow graph-mode concepts have no equivalent here: instead of placeholders,
variable scopes and a Session there is ONE `Context` holding
  * the persistent device workspace (every activation / gradient buffer is allocated once, keyed by
    the call-site sequence of a phase, so that a captured hipGraph can replay the step),
  * the flat per-network parameter / gradient / Adam-slot buffers (`ParamStore`),
  * the backward tape of the phase being executed (explicit closures; no autograd engine),
  * the RNG provider (Philox kernels, or tensors injected by the parity tests).
torch is used only as the device-buffer substrate (allocation, H2D copies, streams).
"""
import contextlib
import os
import ctypes as C

import numpy as np
import torch

from . import lib


def pad32(c):
    return (c + 31) // 32 * 32


class Act(object):
    """Handle of an NHWC activation buffer: logical shape [n,h,w,c], channel stride ld >= c
    (channels c..ld are zero when the buffer feeds an MFMA kernel)."""
    __slots__ = ('t', 'n', 'h', 'w', 'c', 'ld', 'grad', 'requires_grad', 'strided_grad_ok', 'grad_sink', 'grad_fused', 'pending', 'bias_sink',
                 'grad_is_dpre', 'bn_sums', 'bn_bwd_sink', 'bn_bwd_sums', 'contribs', 'labels')

    def __init__(self, t, n, h, w, c, ld, requires_grad=False):
        self.t, self.n, self.h, self.w, self.c, self.ld = t, n, h, w, c, ld
        self.grad = None
        self.requires_grad = requires_grad
        self.strided_grad_ok = False      # set by producers whose backward reads .grad through (pointer, channel stride) only
        self.grad_sink = None             # (act, alpha, seg_rows): the producer is a mean-only-BN layer — a consumer whose input-gradient
        self.grad_fused = None            # launch can apply act'(y) and sum the columns stores t = dx*act'(y) in .grad and the sums here
        self.bias_sink = None             # (act, alpha, bias gradient): the producer is conv / deconv + bias + relu-like activation — a batch norm
        self.grad_is_dpre = False         # consuming it folds act'(y) and the bias-gradient sums into its backward pass and sets this flag
        self.bn_sums = None               # (fp64 sums buffer, seg_rows): the producing convolution took the batch-norm statistics of this tensor
        self.bn_bwd_sink = None           # (batch norm's input Act, seg_rows): this tensor is a training-mode batch norm's output — the launch that
        self.bn_bwd_sums = None           # produces its gradient can take the backward statistics (sum dy, sum dy*x) and leaves the buffer here
        self.labels = None                # (label tensor address, n labels): the producing convolution wrote the cond_concat behind it into this buffer
        self.contribs = 0                 # on a GRADIENT handle: how many backward closures have written into it (Context.grad_of) — the fused
                                          # backward paths above are only valid for exactly one
        self.pending = None               # (source Act, keep-mask, scale): a dropout the NEXT op applies in its own launch (ops.scale_mask(defer=True));
                                          # such a handle has no storage of its own (t is None) until that op has run

    @property
    def rows(self):
        return self.n * self.h * self.w

    @property
    def ptr(self):
        return C.c_void_p(self.t.data_ptr())

    def numpy(self):
        """logical [n,h,w,c] (or [n,c] when h=w=1) host copy — tests / evaluation only."""
        a = self.t.detach().cpu().numpy().reshape(self.n, self.h, self.w, self.ld)[..., :self.c]
        return a.reshape(self.n, self.c) if self.h == 1 and self.w == 1 else a

    def view_rows(self, r0, r1):
        """sub-batch [r0:r1) of images sharing the same storage."""
        per = self.h * self.w * self.ld
        return Act(self.t[r0 * per:r1 * per], r1 - r0, self.h, self.w, self.c, self.ld, self.requires_grad)


class ParamStore(object):
    """One network's variables in flat fp32 device buffers: `p` (trainable values), `g` (gradients),
    `m`, `v` (Adam slots), `s` (non-trainable state: pop_mean, BN moving statistics).  Offsets are
    32-float aligned; the padding stays zero (zero gradient => Adam leaves it at zero)."""

    def __init__(self, name, specs, device, capacity=0):
        """capacity: floats reserved per buffer beyond what `specs` need (stores that grow through Context.get_variable: variables
        appended within the reserve keep every device pointer valid)."""
        self.name = name
        self.specs = list(specs)                      # (name, shape, trainable)
        self.index = {}
        off_p = off_s = 0
        for nm, shape, trainable in self.specs:
            n = int(np.prod(shape))
            if trainable:
                self.index[nm] = ('p', off_p, n, tuple(shape))
                off_p += pad32(n)
            else:
                self.index[nm] = ('s', off_s, n, tuple(shape))
                off_s += pad32(n)
        self.n_p, self.n_s = off_p, max(off_s, 32)
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=device)
        cap_p, cap_s = self.n_p + capacity, self.n_s + capacity // 16
        # p / g / m / v / s are the LIVE prefixes of the reserved buffers (one Adam / all-reduce launch covers exactly n_p floats)
        self._full = {k: z(cap_s if k == 's' else cap_p) for k in ('p', 'g', 'm', 'v', 's')}
        self._narrow()
        self.step = torch.zeros(1, dtype=torch.int32, device=device)
        self.ema = None
        self.frozen = False          # set by Train._capture: a hipGraph holds this store's device pointers, re-allocation is an error

    def _narrow(self):
        f = self._full
        self.p, self.g, self.m, self.v, self.s = f['p'][:self.n_p], f['g'][:self.n_p], f['m'][:self.n_p], f['v'][:self.n_p], f['s'][:self.n_s]
        if 'ema' in f:
            self.ema = f['ema'][:self.n_p]

    def names(self, trainable=None):
        return [nm for nm, _, tr in self.specs if trainable is None or tr == trainable]

    def _slice(self, buf, nm):
        kind, off, n, shape = self.index[nm]
        return buf[off:off + n]

    def value(self, nm):
        kind = self.index[nm][0]
        return self._slice(self.p if kind == 'p' else self.s, nm)

    def grad(self, nm):
        assert self.index[nm][0] == 'p', nm
        return self._slice(self.g, nm)

    def shape(self, nm):
        return self.index[nm][3]

    def set(self, nm, array):
        v = self.value(nm)
        a = np.ascontiguousarray(array, np.float32).reshape(-1)
        assert a.size == v.numel(), (nm, a.size, v.numel())
        v.copy_(torch.from_numpy(a))

    def get(self, nm, which='value'):
        t = {'value': self.value, 'grad': self.grad}.get(which)
        if t is None:
            t = lambda n: self._slice(getattr(self, which), n)
        return t(nm).detach().cpu().numpy().reshape(self.shape(nm))

    def load_dict(self, d):
        for nm in self.names():
            if nm in d:
                self.set(nm, d[nm])

    def to_dict(self, which='value'):
        return {nm: self.get(nm, which) for nm in self.names(None if which == 'value' else True)}

    def enable_ema(self):
        self._full['ema'] = self._full['p'].clone()
        self.ema = self._full['ema'][:self.n_p]

    def extend(self, specs):
        """Append variables (tf.get_variable creating on first use, Context.get_variable).  Offsets of existing variables never
        move; device pointers stay valid as long as the new variables fit the reserve (`capacity`) — beyond it the buffers are
        re-allocated (doubling) and tensors handed out earlier go stale, so variables are to be created at graph-build time, before
        the first Train.train_iteration captures anything (TensorFlow likewise finalises its graph before running it)."""
        off_p, off_s = self.n_p, (self.n_s if any(not t for _, _, t in self.specs) else 0)
        for nm, shape, trainable in specs:
            assert nm not in self.index, nm
            n = int(np.prod(shape))
            if trainable:
                self.index[nm] = ('p', off_p, n, tuple(shape))
                off_p += pad32(n)
            else:
                self.index[nm] = ('s', off_s, n, tuple(shape))
                off_s += pad32(n)
            self.specs.append((nm, tuple(shape), trainable))

        self.n_p, self.n_s = off_p, max(off_s, 32)
        for k, buf in list(self._full.items()):
            need = self.n_s if k == 's' else self.n_p
            if buf.numel() < need:                     # beyond the reserve: re-allocate with room to double (pointers change)
                if self.frozen:
                    raise lib.TgError("ParamStore %r: variable(s) %s created after a hipGraph captured this store's device pointers and "
                                      "beyond its reserve (%d > %d floats): create every variable before the first Train.train_iteration"
                                      % (self.name, [nm for nm, _, _ in specs], need, buf.numel()))
                new = torch.zeros(max(need, 2 * buf.numel()), dtype=buf.dtype, device=buf.device)
                new[:buf.numel()].copy_(buf)
                self._full[k] = new
        self._narrow()

    def offset(self, nm):
        """element offset of trainable variable `nm` inside p / g / m / v (the gradient-bucket boundary of the DP exchange)."""
        assert self.index[nm][0] == 'p', nm
        return self.index[nm][1]


class InjectedRNG(object):
    """Parity-test RNG: every draw is looked up in a dict of host arrays keyed '<rng_scope>/<name>'."""

    def __init__(self, arrays, device):
        self.arrays, self.device, self.cache = arrays, device, {}

    def _get(self, ctx, name, n):
        key = ctx.rng_scope + '/' + name
        if key not in self.cache:
            a = np.ascontiguousarray(self.arrays[key], np.float32).reshape(-1)
            assert a.size == n, (key, a.size, n)
            self.cache[key] = torch.from_numpy(a).to(self.device)
        return self.cache[key]

    def keep_mask(self, ctx, name, n, keep):
        return self._get(ctx, name, n)

    def normal(self, ctx, name, n, std):
        return self._get(ctx, name, n)       # injected noise is already scaled by std

    def advance(self, ctx):
        pass


class PhiloxRNG(object):
    """Production RNG: tg_rng_* kernels; (seed, step) in device memory so hipGraph replays draw fresh numbers.
    Inside Train.train_iteration the draws of a solver run are recorded the first time (buffer, size, distribution, stream id) and
    from then on generated by ONE tg_rng_multi_f32 launch at the first request of the run — the values are the same (counter-based
    generator: every draw is a function of (seed, step, stream id, element index) only)."""

    def __init__(self, seed, device):
        self.state = torch.tensor([seed, 0], dtype=torch.int64, device=device)
        self.stream_ids = {}
        self.plans = {}                  # (plan_tag, phase) -> [(out tensor, n, mode, a, b, sid)]
        self._plan = self._rec = None
        self._cursor = 0

    def _sid(self, ctx, name):
        key = ctx.rng_scope + '/' + name
        if key not in self.stream_ids:
            self.stream_ids[key] = len(self.stream_ids) + 1
        return self.stream_ids[key]

    # ---- per solver run -------------------------------------------------------------------------
    def begin_phase(self, ctx):
        key = (ctx.plan_tag, ctx.phase)
        on = ctx.prep_cache is not None and os.environ.get('TG_RNG_MULTI', '1') != '0'   # only inside Train.train_iteration
        self._plan = self.plans.get(key) if on else None
        self._rec = [] if (on and self._plan is None) else None
        self._cursor = 0

    def end_phase(self, ctx):
        if self._rec is not None and len(self._rec) >= 2:
            self.plans[(ctx.plan_tag, ctx.phase)] = self._rec
        self._plan = self._rec = None

    def _multi(self, ctx, jobs):
        arr = (lib.RngJob * len(jobs))(*[lib.RngJob(j[0].data_ptr(), j[1], j[2], j[3], j[4], j[5]) for j in jobs])
        lib.call('tg_rng_multi_f32', arr, len(jobs), lib.ptr(self.state), ctx.stream)

    def _draw(self, ctx, name, n, mode, a, b, out=None):
        if out is None:
            out = ctx.ws('rng:' + ctx.rng_scope + '/' + name, n * (int(a) if mode == 3 else 1))
        sid = self._sid(ctx, name)
        job = (out, int(n), mode, float(a), float(b), sid)
        if self._plan is not None and self._cursor < len(self._plan):
            want = self._plan[self._cursor]
            if want[0].data_ptr() == out.data_ptr() and want[1:] == job[1:]:
                if self._cursor == 0:
                    for k in range(0, len(self._plan), 16):
                        self._multi(ctx, self._plan[k:k + 16])
                self._cursor += 1
                return out
            self._plan = None                              # a different sequence than recorded: single launches (same values)
        self._multi(ctx, [job])
        if self._rec is not None:
            self._rec.append(job)
        return out

    def keep_mask(self, ctx, name, n, keep):
        return self._draw(ctx, name, n, 1, keep, 0.0)

    def normal(self, ctx, name, n, std):
        return self._draw(ctx, name, n, 2, std, 0.0)

    def uniform(self, ctx, name, n, lo, hi, out=None):
        return self._draw(ctx, name, n, 0, lo, hi, out)

    def onehot(self, ctx, name, rows, k, out=None):
        return self._draw(ctx, name, rows, 3, k, 0.0, out)

    def latents(self, ctx, z_out, y_out, rows, k, lo=-1.0, hi=1.0):
        """z ~ U(lo,hi) and y ~ onehot(U{0..k-1}) (Training/Train_goodGAN.py:234-239) in one launch."""
        self._multi(ctx, [(z_out, z_out.numel(), 0, float(lo), float(hi), self._sid(ctx, 'z')), (y_out, int(rows), 3, float(k), 0.0, self._sid(ctx, 'y'))])

    def advance(self, ctx):
        lib.call('tg_rng_advance', lib.ptr(self.state), ctx.stream)


class Context(object):
    side_fwd_only = os.environ.get('TG_SIDE_FWD_ONLY') == '1'
    capturing = False                                 # a hipGraph capture is open on the launch stream (Train._capture)
    _wgrad_side_pending = False                       # launches on the second stream not yet joined (wgrad_on_side)

    def __init__(self, device='cuda:0', seed=0):
        lib.load()                                   # fails loudly when the HIP extension is missing
        if not torch.cuda.is_available():
            raise lib.TgError("no MI355X visible: the Triple-GAN step has no CPU fallback")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        # a dedicated non-default stream: hipStream capture is illegal on the legacy null stream
        # TG_STREAM_PRIO=1 (A/B): the launch stream above the second stream in the hardware queues' priority
        prio = os.environ.get('TG_STREAM_PRIO') == '1'
        self.torch_stream = torch.cuda.Stream(device=self.device, priority=-1) if prio else torch.cuda.Stream(device=self.device)
        torch.cuda.set_stream(self.torch_stream)
        # Optional side stream for the small dependent chains that are off the critical path (weight-norm scale + filter re-layouts
        # ahead of a convolution; slab reduce + weight-norm gradient behind a filter-gradient launch), forked from / joined to the
        # main stream by events inside every phase_scope so that a hipGraph capture records them as parallel branches.
        # MEASURED NEGATIVE (round 1, MI355X, ROCm 7.2): 5 571 images/s without it, 5 386 with the backward chains only, 5 403 with
        # both — a captured graph with cross-stream edges replays slower than the single chain even though ~1 ms of small launches
        # become concurrent.  OFF by default; TG_SIDE_STREAM=1 (forward + backward) or 2 (backward only) re-enables it.
        self.side_stream = torch.cuda.Stream(device=self.device)
        mode = os.environ.get('TG_SIDE_STREAM', '0')
        self.use_side_stream = mode != '0'
        self.side_forward = mode == '1'
        # Round 3: second-stream overlap of EAGER launches (config.EXEC_MODE = 'overlap', switched on by Train.train_iteration): every
        # filter-gradient launch of a backward pass runs on the side stream beside the input-gradient chain it does not feed (joined once
        # where the slabs are reduced, flush_tails; ops.filter_grad), and the D-update's generator forward beside the classifier's.  The
        # bandwidth-bound passes of one chain (mean-only-BN centring, pooling, activation gradients) then overlap the matrix kernels of the
        # other.  MEASURED (one MI355X, bench.py 100 steps, same box): eager one stream 14.96 ms, eager + overlap 14.59 ms; replayed
        # hipGraphs 14.98 ms, hipGraphs captured WITH the cross-stream edges 15.37 ms — graphs stay a single chain.  TG_WGRAD_SIDE=0/1/2
        # (off / small launches only / all) overrides for A/B runs.
        self.wgrad_side_env = 'TG_WGRAD_SIDE' in os.environ                    # set: the environment decides (A/B runs), Train.train_iteration does not
        self.wgrad_side = os.environ.get('TG_WGRAD_SIDE', '0') in ('1', '2')
        self.wgrad_side_all = os.environ.get('TG_WGRAD_SIDE', '0') == '2'      # also the large (wgrad3x3) launches
        self._wgrad_side_pending = False
        # fp64 statistics accumulators (fused mean-only BN / batch norm) of one solver run live in ONE arena per phase, zeroed by one
        # launch at the start of the phase instead of one memset per layer and direction (36 -> 3 launches per iteration)
        # forward filter preparation of a whole solver run in two launches: the first pass of a (mode, phase) records the layers that
        # needed preparing; from then on the layer that opens that sequence launches tg_filter_prep_multi_f32 for all of them
        self.plan_tag = None           # training mode ('pre' / 'full'), set by Train.train_iteration
        self.prep_plans = {}           # (plan_tag, phase, first layer key) -> [job dicts]
        self._planned = set()
        self._prep_rec = None
        self.tail_jobs = []            # deferred filter-gradient tails of the running backward pass (ops.filter_grad / flush_tails)
        self.prep_cache = None         # {layout key: prepared filter buffers} while Train.train_iteration runs (see ops.conv2d)
        self.state_replay = None       # list collecting state-update closures of a kept forward pass (sub_tape(replay=...))
        self._zarena = {}              # phase -> dict(sizes=[...], buf=tensor or None, cursor=int, recording=bool)
        self._events = {}
        self._side_depth = 0
        self._phase_depth = 0          # the side stream is only used between a phase's fork and join
        self.buffers = {}
        self.stores = {}
        self.tape = None
        self.phase = 'init'
        self.counter = 0
        self.scopes = []
        self.rng_scope = ''
        self.rng_counters = {}
        self.rng = PhiloxRNG(seed, self.device)
        self.train_nets = set()
        self._stream = None
        self.mfma_dtype = 'f32'          # 'bf16': MFMA operands rounded to bf16 inside the kernels (tensors stay fp32)

    # ---- streams ---------------------------------------------------------------------------------
    @property
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _event(self, tag):
        """one persistent event per call site (created in the first, eager iteration; reused inside captures).  Events are numbered BESIDE the
        buffers' call-site counter, never through it: the two-stream and the single-chain execution modes record different events, and the
        buffers of a call site must be the same in both (a graph captured after two-stream iterations would otherwise meet every key with
        another size and re-allocate inside the capture)."""
        at = (self.phase, self.counter)
        if getattr(self, '_ev_at', None) != at:
            self._ev_at, self._ev_k = at, 0
        self._ev_k += 1
        key = '%s/%s%d.%d' % (self.phase, tag, self.counter, self._ev_k)
        ev = self._events.get(key)
        if ev is None:
            ev = self._events[key] = torch.cuda.Event()
        return ev

    def _ev_record(self, ev, stream):
        """hipEventRecord on a torch stream; while a launch plan is being recorded (tg.plan.Plan.recording) the operation is appended to it."""
        ev.record(stream)
        if lib._recorder is not None:
            lib._recorder.add_record(ev, stream.cuda_stream)

    def _ev_wait(self, stream, ev):
        stream.wait_event(ev)
        if lib._recorder is not None:
            lib._recorder.add_wait(stream.cuda_stream, ev)

    def _fork_side(self):
        if self.use_side_stream:
            ev = self._event('fork')
            self._ev_record(ev, self.torch_stream)
            self._ev_wait(self.side_stream, ev)

    def _join_side(self):
        if self.use_side_stream:
            ev = self._event('join')
            self._ev_record(ev, self.side_stream)
            self._ev_wait(self.torch_stream, ev)

    @contextlib.contextmanager
    def on_side(self, after_main=False, forward=False):
        """run the enclosed launches on the side stream (after_main: only after what the main stream has enqueued so far)."""
        if not self.use_side_stream or self._side_depth or not self._phase_depth or (forward and not self.side_forward):
            yield
            return
        if after_main:
            ev = self._event('m2s')
            self._ev_record(ev, self.torch_stream)
            self._ev_wait(self.side_stream, ev)
        self._side_depth += 1
        torch.cuda.set_stream(self.side_stream)
        try:
            yield
        finally:
            torch.cuda.set_stream(self.torch_stream)
            self._side_depth -= 1

    def main_waits_side(self):
        """the main stream continues only after what the side stream has enqueued so far."""
        if self.use_side_stream and self.side_forward and not self._side_depth and self._phase_depth:
            ev = self._event('s2m')
            self._ev_record(ev, self.side_stream)
            self._ev_wait(self.torch_stream, ev)

    # ---- workspace -------------------------------------------------------------------------------
    def ws(self, key, numel, zero=False):
        t = self.buffers.get(key)
        if t is None or t.numel() < numel:
            if self.capturing:
                # a graph holds addresses: a buffer born (or re-born) inside a capture means this pass is not the one the eager pass before
                # it allocated for — replaying it would write through stale pointers
                raise lib.TgError("buffer %r (%d floats) would be allocated inside a hipGraph capture / launch-plan recording: the recorded pass "
                                  "differs from the eager pass that preceded it" % (key, int(numel)))
            t = torch.zeros(int(numel), dtype=torch.float32, device=self.device)
            self.buffers[key] = t
        elif zero:
            lib.call('tg_fill_f32', lib.ptr(t), 0.0, int(numel), self.stream)
        return t[:numel]

    def new_act(self, n, h, w, c, ld=None, requires_grad=False, tag='a'):
        ld = c if ld is None else ld
        self.counter += 1
        key = '%s/%s%d' % (self.phase, tag, self.counter)
        return Act(self.ws(key, n * h * w * ld), n, h, w, c, ld, requires_grad)

    def scratch(self, tag, numel):
        self.counter += 1
        return self.ws('%s/%s%d' % (self.phase, tag, self.counter), numel)

    def prep_invalidate(self, store):
        """drop the prepared filter layouts of `store`'s variables (its optimiser step is about to change them)."""
        if self.prep_cache:
            lo = store.p.data_ptr()
            hi = lo + store.p.numel() * 4
            for k in [k for k in self.prep_cache if lo <= k[1] < hi]:
                del self.prep_cache[k]

    def zscratch(self, tag, numel):
        """(tensor, zeroed) for a statistics accumulator of `numel` floats (= numel/2 doubles).  The first time a phase runs the
        requests are recorded and served as ordinary scratch (zeroed = 0: the kernel clears it); from then on they are slices of the
        phase's arena, cleared by _zarena_begin (zeroed = 1)."""
        numel = (int(numel) + 63) // 64 * 64
        za = self._zarena.setdefault(self.phase, dict(sizes=[], buf=None, cursor=0, off=0, recording=True))
        if za['recording'] or za['buf'] is None:
            if za['recording']:
                za['sizes'].append(numel)
            return self.scratch(tag, numel), 0
        k = za['cursor']
        if k >= len(za['sizes']) or za['sizes'][k] != numel:          # a different op sequence than recorded: stay correct
            za['cursor'] = len(za['sizes']) + 1
            return self.scratch(tag, numel), 0
        t = za['buf'][za['off']:za['off'] + numel]
        za['cursor'], za['off'] = k + 1, za['off'] + numel
        self.counter += 1                                              # keep the call-site numbering of the recording pass
        return t, 1

    def _zarena_begin(self, resume):
        za = self._zarena.get(self.phase)
        if za is None or za['recording'] or resume or za['buf'] is None:
            return
        za['cursor'], za['off'] = 0, 0
        lib.call('tg_fill_f32', lib.ptr(za['buf']), 0.0, za['buf'].numel(), self.stream)

    def _zarena_end(self):
        za = self._zarena.get(self.phase)
        if za is not None and za['recording']:
            za['recording'] = False
            if za['sizes']:                 # allocated here, in the eager recording pass — never inside a stream capture
                za['buf'] = torch.zeros(sum(za['sizes']), dtype=torch.float32, device=self.device)

    def grad_of(self, a):
        """gradient buffer of an activation (same layout), allocated on first use per call site."""
        if a.grad is None:
            self.counter += 1
            a.grad = Act(self.ws('%s/g%d' % (self.phase, self.counter), a.rows * a.ld), a.n, a.h, a.w, a.c, a.ld)
        a.grad.contribs += 1               # every backward closure that writes a gradient asks for the buffer here
        return a.grad

    def from_numpy(self, x, ld=None, key=None):
        """host array [n,h,w,c] or [n,c] -> device Act (tests, data feeding)."""
        x = np.asarray(x, np.float32)
        if x.ndim == 2:
            x = x.reshape(x.shape[0], 1, 1, x.shape[1])
        n, h, w, c = x.shape
        ld = c if ld is None else ld
        if ld != c:
            xp = np.zeros((n, h, w, ld), np.float32)
            xp[..., :c] = x
            x = xp
        if key is None:
            t = torch.from_numpy(np.ascontiguousarray(x)).to(self.device).reshape(-1)
        else:
            t = self.ws(key, x.size)
            t.copy_(torch.from_numpy(np.ascontiguousarray(x).reshape(-1)))
        return Act(t, n, h, w, c, ld)

    # ---- phases / scopes -------------------------------------------------------------------------
    @contextlib.contextmanager
    def phase_scope(self, name, train_nets=(), record=True, counter=0):
        """one solver run (sess.run of Training/Train_goodGAN.py:266-276): fresh call-site counter and tape.
        counter: resume value when a solver run is executed in two pieces (see Context.backward)."""
        prev = (self.phase, self.counter, self.tape, self.train_nets)
        self.phase, self.counter = name, counter
        self.tape = [] if record else None
        self.train_nets = set(train_nets)
        self._fork_side()
        self._phase_depth += 1
        self._zarena_begin(resume=counter != 0)
        if counter == 0 and hasattr(self.rng, 'begin_phase'):
            self.rng.begin_phase(self)
        rec_prev = self._prep_rec
        tagp = (self.plan_tag, name)
        self._prep_rec = [] if (self.prep_cache is not None and counter == 0 and tagp not in self._planned) else None
        try:
            yield self
        finally:
            if self._prep_rec is not None:
                self._planned.add(tagp)
                if len(self._prep_rec) >= 2:
                    self.prep_plans[(self.plan_tag, name, self._prep_rec[0]['key'])] = self._prep_rec
            self._prep_rec = rec_prev
            if hasattr(self.rng, 'end_phase') and counter == 0:
                self.rng.end_phase(self)
            self._zarena_end()
            self._phase_depth -= 1
            self._join_side()
            self.phase, self.counter, self.tape, self.train_nets = prev

    @contextlib.contextmanager
    def sub_tape(self, train_nets, replay=None):
        """record the ops executed inside on a SEPARATE tape (returned) with their own trainable set: lets one solver run
        keep a forward pass (and its backward closures) for the next run to finish.  replay: a list that collects, from the ops
        executed inside, closures re-applying the STATE updates a re-execution of this forward pass would make (batch-norm moving
        statistics) — the run that re-uses the pass calls them where TensorFlow would have run the pass again."""
        prev = (self.tape, self.train_nets, self.state_replay)
        tape = []
        self.tape, self.train_nets, self.state_replay = tape, set(train_nets), replay
        try:
            yield tape
        finally:
            self.tape, self.train_nets, self.state_replay = prev

    @contextlib.contextmanager
    def no_record(self):
        """execute the enclosed ops forward-only (nothing goes on the tape)."""
        prev, self.tape = self.tape, None
        try:
            yield
        finally:
            self.tape = prev

    def run_tape(self, tape, stop_at_boundary=False):
        """run a recorded tape in reverse (all of it, or down to its last bucket boundary: see backward); returns the unexecuted head
        or None, and empties the tape when everything ran."""
        rest = self._run_reverse(tape, stop_at_boundary)
        if rest is None:
            del tape[:]
        return rest

    @contextlib.contextmanager
    def variable_scope(self, name):
        self.scopes.append(name)
        try:
            yield
        finally:
            self.scopes.pop()

    @contextlib.contextmanager
    def rng_scoped(self, name):
        prev, self.rng_scope = (self.rng_scope, self.rng_counters), name
        self.rng_counters = {}
        try:
            yield
        finally:
            self.rng_scope, self.rng_counters = prev

    def next_rng_name(self, kind):
        """'drop0', 'drop1', ... in call order inside the current rng scope (one network application batch)."""
        i = self.rng_counters.get(kind, 0)
        self.rng_counters[kind] = i + 1
        return '%s%d' % (kind, i)

    def scope_name(self, leaf=''):
        return '/'.join(self.scopes + ([leaf] if leaf else []))

    def store_of(self, full_name):
        return self.stores[full_name.split('/')[0]]

    def var(self, leaf):
        """device tensor of variable '<scope>/<leaf>' (tf.get_variable with reuse=True)."""
        full = self.scope_name(leaf)
        return self.store_of(full).value(full)

    def get_variable(self, leaf, shape, initializer, trainable=True):
        """tf.get_variable(leaf, shape, initializer=..., trainable=...) under the current variable scope: the existing variable
        (reuse=True) or a new one appended to the store of the scope's root (created on first use the way the reference's layer
        functions create theirs, Model/nn.py:194-201,227-237).  initializer: float (constant) or callable(shape) -> array."""
        full = self.scope_name(leaf)
        root = full.split('/')[0]
        st = self.stores.get(root)
        if st is None:
            st = self.stores[root] = ParamStore(root, [], self.device, capacity=1 << 22)     # 16 MiB reserve per buffer
        if full not in st.index:
            st.extend([(full, tuple(int(d) for d in shape), trainable)])
            val = initializer(tuple(shape)) if callable(initializer) else np.full(shape, initializer, np.float32)
            st.set(full, val)
            if trainable and st.ema is not None:        # tf.train.ExponentialMovingAverage starts a shadow at the variable's initial value
                kind, off, n, _ = st.index[full]
                st.ema[off:off + n].copy_(st.p[off:off + n])
        # the same bytes may be declared under two views (the classifier's first filter: [3,3,3,128] = [27,128])
        assert int(np.prod(st.shape(full))) == int(np.prod(shape)), (full, st.shape(full), shape)
        return st.value(full)

    def var_grad(self, leaf):
        full = self.scope_name(leaf)
        return self.store_of(full).grad(full)

    def trains(self, leaf=''):
        return self.tape is not None and self.scope_name(leaf).split('/')[0] in self.train_nets

    def record(self, fn):
        if self.tape is not None:
            self.tape.append(fn)

    def grad_bucket_boundary(self, net=None):
        """Called by a model between two layers of its forward pass: once the backward pass has come back to this point, the
        variable gradients of every layer of network `net` recorded AFTER it are final (data-parallel bucket boundary)."""
        if self.tape is not None:
            self.tape.append(_Boundary(net))

    def _run_reverse(self, tape, stop_at_boundary):
        i = len(tape)
        while i > 0:
            i -= 1
            fn = tape[i]
            if isinstance(fn, _Boundary):
                # stop_at_boundary: True = at any mark, a network name = at that network's marks only (a solver run's tape also
                # carries the marks of the networks it merely applies)
                if stop_at_boundary is True or (stop_at_boundary and fn.net == stop_at_boundary):
                    self.flush_tails()             # the finished bucket's gradients must be complete
                    return tape[:i]
                continue
            fn()
        self.flush_tails()
        return None

    def backward(self, stop_at_boundary=False):
        """Run the recorded closures in reverse.  stop_at_boundary: stop at the last grad_bucket_boundary() mark and return the
        not-yet-executed head of the tape (continue with run_tape(head, stop_at_boundary=...) inside phase_scope(..., counter=
        self.counter) — once per remaining boundary), so that every finished bucket can be all-reduced while the rest of the backward
        pass runs; None when everything ran."""
        tape, self.tape = self.tape, []
        return self._run_reverse(tape, stop_at_boundary)

    @contextlib.contextmanager
    def wgrad_on_side(self, kind='wgrad'):
        """run the enclosed filter-gradient launch on the side stream, after what the main stream has enqueued so far (its operands are
        final); the main stream goes on with the input-gradient chain and waits for the side stream in join_wgrad_side().
        kind = 'fwd': a forward chain (the D-update's generator forward).  Context.side_fwd_only (TG_SIDE_FWD_ONLY=1, A/B runs inside
        captured graphs): only those go to the side stream — two cross-stream edges per iteration instead of one pair per filter gradient."""
        if not self.wgrad_side or self._side_depth or not self._phase_depth or (self.side_fwd_only and kind != 'fwd'):
            yield
            return
        ev = self._event('m2w')
        self._ev_record(ev, self.torch_stream)
        self._ev_wait(self.side_stream, ev)
        self._side_depth += 1
        torch.cuda.set_stream(self.side_stream)
        try:
            yield
        finally:
            torch.cuda.set_stream(self.torch_stream)
            self._side_depth -= 1
            self._wgrad_side_pending = True

    def join_wgrad_side(self):
        if self._wgrad_side_pending:
            ev = self._event('w2m')
            self._ev_record(ev, self.side_stream)
            self._ev_wait(self.torch_stream, ev)
            self._wgrad_side_pending = False

    def flush_tails(self):
        """launch the deferred filter-gradient tails (slab reduction, weight-norm gradient) of the layers whose wgrad has been
        issued since the last flush: three launches for up to 16 layers (tg_filter_grad_tail_multi_f32)."""
        self.join_wgrad_side()
        jobs, self.tail_jobs = getattr(self, 'tail_jobs', []), []
        if not jobs:
            return
        for k in range(0, len(jobs), 16):
            part = jobs[k:k + 16]
            arr = (lib.WnJob * len(part))(*part)
            lib.call('tg_filter_grad_tail_multi_f32', arr, len(part), self.stream)


class _Boundary(object):
    """tape entry of Context.grad_bucket_boundary."""
    __slots__ = ('net',)

    def __init__(self, net=None):
        self.net = net


BUCKET_BOUNDARY = _Boundary


_CTX = None


def set_context(ctx):
    global _CTX
    _CTX = ctx
    return ctx


def ctx():
    if _CTX is None:
        raise lib.TgError("no tg Context: create tg.runtime.Context() first (needs an MI355X)")
    return _CTX


def seg_array(segs):
    return (C.c_int32 * len(segs))(*segs)
