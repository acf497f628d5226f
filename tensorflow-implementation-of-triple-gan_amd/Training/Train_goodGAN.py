"""Train — the Triple-GAN training driver, counterpart of the reference's Training/Train_goodGAN.py.

Same entry point: `Train(config, log_dir, save_dir, **kwargs).train(Dataset, Model, sample_y)`
(Train_goodGAN.py:27,43) and the same iteration protocol (:230-278):

    D-update  (sess.run([d_solver, d_loss]))   G fwd, C(x_u_c), C(x_u_d) fwd, D x3 fwd, D bwd, Adam(D)
    G-update  (sess.run([g_solver, g_loss]))   G fwd, D(G) fwd, D data-bwd, G bwd, Adam(G)
    C-update  (sess.run([c_solver, c_loss]))   G fwd, C x4 fwd, D(x_u_c) fwd, C bwd, Adam(C), EMA(C)

Each solver run executes only the sub-graph its fetches need and re-samples dropout / noise, exactly as three
TF session calls on one feed do.  Instead of a TF graph + Session the three runs are captured
once into hipGraphs and replayed (config.USE_HIP_GRAPH); hyper-parameters that change (lr, lambdas, Adam step,
RNG step) live in device memory.  Applications of one network inside a solver run are batched into one call
(D: [D_real|D_fake|D_unl] = 250 images; C: [C_real|C_unl|C_unl_rep|C_fake] = 250 images, mean-only-BN per
application).  Data-parallel replicas (one process per GPU) sum-all-reduce the trained network's flat gradient
buffer over RCCL after each backward (tg/dist.py); the reference has no multi-device path.
"""
import math
import os
import time

import numpy as np
import torch

from Training.train_base import Train_base
from tg import dist as tgdist
from tg import lib, ops
from tg.batching import concat_acts
from tg.runtime import Act, Context, PhiloxRNG, ctx, set_context


class Train(Train_base):
    def __init__(self, config, log_dir, save_dir, **kwargs):
        super(Train, self).__init__()
        self.config = config
        self.save_dir = save_dir
        self.log_dir = log_dir
        self.comments = kwargs.get('comments', '')
        self.world, self.rank, self.local_rank = tgdist.init()
        try:
            self.cx = ctx()
        except lib.TgError:
            self.cx = set_context(Context('cuda:%d' % self.local_rank, seed=getattr(config, 'SEED', 0) + 7919 * self.rank))
        cx = self.cx
        # 'bf16' = BASELINE.json configs[3] "bf16 MFMA conv path": conv / deconv / dense operands rounded to bf16 inside the
        # MFMA kernels, fp32 accumulation, fp32 tensors, statistics, master weights and optimiser state
        cx.mfma_dtype = getattr(config, 'MFMA_DTYPE', 'f32')
        if cx.mfma_dtype not in ('f32', 'bf16'):
            raise ValueError("MFMA_DTYPE must be 'f32' or 'bf16', got %r" % (cx.mfma_dtype,))
        # device-resident hyper-parameters (the reference's lr_ph / cla_lr_ph / lambda placeholders, :30-31,416-420)
        self.hyper = torch.zeros(4, dtype=torch.float32, device=cx.device)       # lr, cla_lr, lambda_1, lambda_2
        self.loss_dev = torch.zeros(3, dtype=torch.float32, device=cx.device)    # d_loss, g_loss, c_loss
        self.model = None
        self._graphs = None
        self._rest = {}                  # solver run -> (unexecuted head of its backward tape, call-site counter): bucketed backward passes
        self._warm = False
        self._warm_keys = set()
        self.iteration = 0
        self._exposed = None             # [(mark before, mark after)] of the waits for gradient buckets while measure_exposed(True)
        self._label_override = {}        # see label_override()
        self.summary_train = self.summary_val = None
        if getattr(config, 'SUMMARY', False) and log_dir and self.rank == 0:          # :37-41
            from Training.Summary import Summary
            self.summary_train = Summary(log_dir, config, log_type='train', log_comments=kwargs.get('comments', ''))
            self.summary_val = Summary(log_dir, config, log_type='val', log_comments=kwargs.get('comments', ''))

    # ------------------------------------------------------------------ graph build
    def _build_train_graph(self, Model):
        """:400-426: the placeholders become persistent device buffers of the static batch sizes."""
        c, cx = self.config, self.cx
        dims = c.IMAGE_DIM

        def ph(key, n, shape):
            t = cx.ws('ph:' + key, n * int(np.prod(shape)))
            return Act(t, n, *(shape if len(shape) == 3 else (1, 1, shape[0])), ld=shape[-1])

        self.z_g_ph = ph('z_g', c.BATCH_SIZE_G, [c.Z_DIM])
        self.y_g_ph = ph('y_g', c.BATCH_SIZE_G, [c.NUM_CLASSES])
        self.x_l_c_ph = ph('x_l_c', c.BATCH_SIZE_L_C, dims)
        self.y_l_c_ph = ph('y_l_c', c.BATCH_SIZE_L_C, [c.NUM_CLASSES])
        self.x_l_d_ph = ph('x_l_d', c.BATCH_SIZE_L_D, dims)
        self.y_l_d_ph = ph('y_l_d', c.BATCH_SIZE_L_D, [c.NUM_CLASSES])
        self.x_u_d_ph = ph('x_u_d', c.BATCH_SIZE_U_D, dims)
        self.x_u_c_ph = ph('x_u_c', c.BATCH_SIZE_U_C, dims)
        self.model = Model(c)
        st = cx.stores
        # three Adam optimisers (:85-87): G and D share lr_ph / config.BETA1, C uses cla_lr_ph / 0.5
        self.d_optimizer = self._Adam_optimizer(self.hyper[0:1], c.BETA1)
        self.g_optimizer = self._Adam_optimizer(self.hyper[0:1], c.BETA1)
        self.c_optimizer = self._Adam_optimizer(self.hyper[1:2], 0.5)
        self.set_hyper(c.LEARNING_RATE, getattr(c, 'CLA_LEARNINIG_RATE', c.LEARNING_RATE), 0.0, 0.0)
        if tgdist.active():                      # identical initial weights on every replica
            for s in st.values():
                tgdist.broadcast_(s.p)
                tgdist.broadcast_(s.s)
            st['classifier'].ema.copy_(st['classifier'].p)
        PH = [self.z_g_ph, self.y_g_ph, self.x_l_c_ph, self.y_l_c_ph, self.x_l_d_ph, self.y_l_d_ph, self.x_u_d_ph,
              self.x_u_c_ph, True, self.hyper[2:4]]
        return PH, self.model

    def set_hyper(self, lr=None, cla_lr=None, lambda_1=None, lambda_2=None):
        vals = self.hyper.detach().cpu().numpy()
        for i, v in enumerate((lr, cla_lr, lambda_1, lambda_2)):
            if v is not None:
                vals[i] = v
        self.hyper.copy_(torch.from_numpy(vals))

    # ------------------------------------------------------------------ the three solver runs
    def _d_forward_backward(self, split=False):
        """split: stop the backward pass at the discriminator's last gradient-bucket boundary (the rest runs in _backward_rest)."""
        c, cx, m = self.config, self.cx, self.model
        with cx.phase_scope('D', train_nets=('discriminator',)):
            # The G-update that follows runs the generator on the same feed with the same (not yet updated) weights, and the
            # generator is deterministic (no dropout / noise): TF recomputes it in the second sess.run, here the forward
            # pass and its backward closures are kept for _g_forward_backward (bit-identical result, one G forward saved).
            g_replay = []
            # (eager launches with the side stream, Context.wgrad_side: the generator's forward pass — small launches — runs beside the
            # classifier's, which it does not depend on; the two meet where the discriminator's batch is assembled)
            with cx.wgrad_on_side('fwd'):
                with cx.sub_tape(('good_generator',), replay=g_replay) as g_tape:
                    G = m.good_generator(self.z_g_ph, self.y_g_ph)
            self._g_saved = (G, g_tape, g_replay)
            xz = concat_acts([m.as_image(self.x_u_c_ph), m.as_image(self.x_u_d_ph)])
            if m.zca() is not None:
                xz = m.zca().apply(xz)
            with cx.rng_scoped('D/C'):
                c_logits, _ = m.classifier(xz, True, segments=[c.BATCH_SIZE_U_C, c.BATCH_SIZE_U_D])
            cx.join_wgrad_side()
            oh = ops.argmax_onehot(c_logits, c.NUM_CLASSES)                       # [C_unl_hard | C_unl_d_hard]
            self._d_logits = c_logits                    # (parity tests read the logits the labels were taken from)
            if self._label_override.get('D') is not None:      # parity tests against a precomputed fixture: see label_override()
                oh.copy_(self._label_override['D'])
            k = c.NUM_CLASSES
            oh_unl = Act(oh[:c.BATCH_SIZE_U_C * k], c.BATCH_SIZE_U_C, 1, 1, k, k)
            oh_unl_d = Act(oh[c.BATCH_SIZE_U_C * k:], c.BATCH_SIZE_U_D, 1, 1, k, k)
            self._d_labels = (oh_unl, oh_unl_d)          # the labels this run's discriminator sees (parity tests read them back)
            ximg = concat_acts([m.as_image(a) for a in (self.x_l_d_ph, self.x_u_d_ph, G, self.x_u_c_ph)])   # X_P | G | x_u_c (:258-271)
            yall = concat_acts([self.y_l_d_ph, oh_unl_d, self.y_g_ph, oh_unl])
            with cx.rng_scoped('D/D'):
                _, d_logits = m.discriminator(ximg, yall, want_prob=False)
            self._d_loss(d_logits, c.BATCH_SIZE_L_D + c.BATCH_SIZE_U_D, c.BATCH_SIZE_G, c.BATCH_SIZE_U_C, self.loss_dev[0:1])
            self._rest['D'] = (cx.backward(stop_at_boundary='discriminator' if split else False), cx.counter)

    def _g_forward_backward(self, split=False):
        cx, m = self.cx, self.model
        with cx.phase_scope('G', train_nets=('good_generator',)):
            saved = getattr(self, '_g_saved', None)
            if saved is not None:                       # generator forward of the D-update on the same feed and weights
                G, g_tape, g_replay = saved
                self._g_saved = None
                for fn in g_replay:                     # the state a re-executed forward pass would have advanced: batch-norm moving
                    fn()                                # statistics get their second update of the iteration (the C-update makes the third)
            else:
                G, g_tape = m.good_generator(self.z_g_ph, self.y_g_ph), None
            with cx.rng_scoped('G/D'):
                _, d_fake = m.discriminator(G, self.y_g_ph, want_prob=False)
            self._g_loss(d_fake, self.loss_dev[1:2])
            rest = cx.backward(stop_at_boundary='good_generator' if (split and g_tape is None) else False)
            if g_tape is not None:                      # the discriminator's input gradient is complete: now the kept generator tape
                rest = cx.run_tape(g_tape, stop_at_boundary='good_generator' if split else False)
            self._rest['G'] = (rest, cx.counter)

    def _c_forward_backward(self, split=False):
        """split: stop the backward pass at the classifier's last gradient-bucket boundary (the rest runs in _backward_rest)."""
        c, cx, m = self.config, self.cx, self.model
        rep = bool(getattr(m, 'CONSISTENCY', False))       # Good_GAN_cifar10 only: second stochastic pass on x_u_c
        with cx.phase_scope('C', train_nets=('classifier',)):
            G = m.good_generator(self.z_g_ph, self.y_g_ph)
            parts = [self.x_l_c_ph, self.x_u_c_ph] + ([self.x_u_c_ph] if rep else []) + [G]
            segs = [p.n for p in parts]
            xc = concat_acts([m.as_image(p) for p in parts])
            if m.zca() is not None:
                xc = m.zca().apply(xc)
            with cx.rng_scoped('C/C'):
                c_logits, _ = m.classifier(xc, True, segments=segs)
            c_unl = c_logits.view_rows(segs[0], segs[0] + segs[1])
            k = c.NUM_CLASSES
            oh_c = ops.argmax_onehot(c_unl, k)
            self._c_logits = c_unl
            if self._label_override.get('C') is not None:
                oh_c.copy_(self._label_override['C'])
            oh_unl = Act(oh_c, c_unl.n, 1, 1, k, k)
            with cx.rng_scoped('C/D'):
                _, d_unl = m.discriminator(self.x_u_c_ph, oh_unl, want_prob=False)
            self._c_loss(c_logits, segs[0], segs[1], segs[1] if rep else 0, G.n, self.y_l_c_ph, self.y_g_ph, d_unl,
                         self.hyper[2:4], self.loss_dev[2:3])
            self._rest['C'] = (cx.backward(stop_at_boundary='classifier' if split else False), cx.counter)

    def _backward_rest(self, phase, net, split):
        """continue the backward pass of solver run `phase` — down to the trained network's next bucket boundary (split) or to its end."""
        rest, counter = self._rest[phase]
        if rest:
            with self.cx.phase_scope(phase, train_nets=(net,), counter=counter):
                rest = self.cx.run_tape(rest, stop_at_boundary=net if split else False)
                self._rest[phase] = (rest, self.cx.counter)

    def _c_apply(self):
        st = self.cx.stores['classifier']
        self._train_op(self.c_optimizer, st, 1.0 / self.world)
        # ema.apply(c_vars) under control-dependency on the C step (:101-103)
        lib.call('tg_ema_f32', lib.ptr(st.ema), lib.ptr(st.p), st.n_p, 0.9999, self.cx.stream)
        self.cx.rng.advance(self.cx)

    def _bucket_slices(self, net):
        """the flat gradient buffer of `net` cut at the model's GRAD_BUCKETS, in the order the backward pass completes them (last
        variables first); one slice — the whole buffer — without replicas."""
        st = self.cx.stores[net]
        names = getattr(self.model, 'GRAD_BUCKETS', {}).get(net, ())
        if not tgdist.active() or getattr(self.config, 'NO_GRAD_BUCKETS', False) or not names:
            return [st.g]
        offs = [0] + sorted(st.offset(n) for n in names) + [st.n_p]
        return [st.g[offs[i]:offs[i + 1]] for i in range(len(offs) - 2, -1, -1)]

    def _phase_segments(self, phase, net, first_fn, before=None):
        """[(callable, gradient slice to exchange once it has run)] of one solver run: the forward pass + the backward pass down to the
        last bucket boundary, then one segment per remaining bucket.  `before` (the previous network's optimiser step) opens the first."""
        slices = self._bucket_slices(net)
        split = len(slices) > 1
        head = (lambda: first_fn(split)) if before is None else (lambda: (before(), first_fn(split)))
        segs = [(head, slices[0])]
        for k, sl in enumerate(slices[1:]):
            last = k == len(slices) - 2
            segs.append((lambda last=last: self._backward_rest(phase, net, not last), sl))
        return segs

    def _segments(self, pre_train=False):
        """[(callable, flat gradient slice to exchange afterwards or None, wait for the pending exchanges first?)] — each callable is
        one hipGraph.  With replicas every backward pass is cut at the model's bucket boundaries: a finished bucket is all-reduced on
        the exchange stream while the next segment (the rest of the backward pass) runs; the optimiser step of a network opens the
        next solver run's first segment and waits for that network's buckets."""
        st = self.cx.stores
        w = 1.0 / self.world
        if pre_train:                                          # :182-226: pre-training runs c_solver only
            phases = [self._phase_segments('C', 'classifier', self._c_forward_backward)]
        else:
            phases = [self._phase_segments('D', 'discriminator', self._d_forward_backward),
                      self._phase_segments('G', 'good_generator', self._g_forward_backward,
                                           lambda: self._train_op(self.d_optimizer, st['discriminator'], w)),
                      self._phase_segments('C', 'classifier', self._c_forward_backward,
                                           lambda: self._train_op(self.g_optimizer, st['good_generator'], w))]
        out = []
        for k, segs in enumerate(phases):
            for j, (fn, grads) in enumerate(segs):
                out.append((fn, grads, j == 0 and k > 0))       # a solver run that opens with the previous network's optimiser step
        return out + [(self._c_apply, None, True)]

    # ------------------------------------------------------------------ one iteration
    def feed(self, batch):
        """host feed_dict (:249-263) -> placeholders.  batch keys: z_g,y_g,x_l_c,y_l_c,x_l_d,y_l_d,x_u_d,x_u_c
        (numpy arrays or device Acts)."""
        dev = []
        for key, ph in (('z_g', self.z_g_ph), ('y_g', self.y_g_ph), ('x_l_c', self.x_l_c_ph), ('y_l_c', self.y_l_c_ph),
                        ('x_l_d', self.x_l_d_ph), ('y_l_d', self.y_l_d_ph), ('x_u_d', self.x_u_d_ph), ('x_u_c', self.x_u_c_ph)):
            if key not in batch:
                continue
            v = batch[key]
            if isinstance(v, Act):
                dev.append((ph.t, 0, v.t, ph.t.numel()))
            else:
                a = np.ascontiguousarray(v, np.float32).reshape(-1)
                assert a.size == ph.t.numel(), (key, a.size, ph.t.numel())
                ph.t.copy_(torch.from_numpy(a), non_blocking=False)
        if dev:
            ops.copy_many(dev)                   # device-resident batch: all placeholders in one launch

    def label_override(self, d_labels=None, c_labels=None):
        """TEST HOOK (eager launches only).  The discriminator's labels for unlabelled images are the arg-max of the classifier's logits
        (Model/Good_GAN.py:447-455, Good_GAN_cifar10.py:243-262): a near-tie can come out differently under another summation order or operand
        rounding, and the discriminator's gradient then differs for a reason that is not the discriminator's.  A test that compares against a
        PRECOMPUTED oracle run hands in the one-hot labels that run used — d_labels: [U_C + U_D, k] host array for the D-update ([C_unl | C_unl_d]),
        c_labels: [U_C, k] for the C-update — after checking on the logits (kept in _d_logits / _c_logits) that every disagreement is a
        near-tie.  None clears."""
        cx = self.cx
        to_dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(-1)).to(cx.device)
        self._label_override = {'D': to_dev(d_labels), 'C': to_dev(c_labels)}

    def sample_latent(self):
        """z ~ U(-1,1), y ~ onehot(U{0..9}) (:234-239) drawn on the device."""
        cx = self.cx
        with cx.rng_scoped('latent'):
            if hasattr(cx.rng, 'latents'):
                cx.rng.latents(cx, self.z_g_ph.t, self.y_g_ph.t, self.y_g_ph.n, self.config.NUM_CLASSES)
            else:
                cx.rng.uniform(cx, 'z', self.z_g_ph.t.numel(), -1.0, 1.0, out=self.z_g_ph.t)
                cx.rng.onehot(cx, 'y', self.y_g_ph.n, self.config.NUM_CLASSES, out=self.y_g_ph.t)

    def train_iteration(self, pre_train=False, use_graph=None):
        """D-update, G-update, C-update on the current placeholder contents (:266-276).  No host sync."""
        cx = self.cx
        mode = getattr(self.config, 'EXEC_MODE', 'auto')
        key = 'pre' if pre_train else 'full'
        if use_graph is None:
            use_graph = getattr(self.config, 'USE_HIP_GRAPH', None)
        replayable = isinstance(cx.rng, PhiloxRNG)        # injected draws (parity tests) are host-fed per iteration: nothing to replay
        if mode == 'auto':
            if use_graph is None and replayable and tgdist.graphs_allowed():
                mode = self._auto_mode(key)               # 'plan' or 'graph': the measured faster one for this workload on this host
            elif use_graph is None and replayable:
                mode = 'plan'                             # torch's RCCL process group forbids captures (tg/dist.py): plans are plain launches
            else:
                mode = 'overlap' if not use_graph else 'graph'
        if use_graph is None:
            use_graph = mode == 'graph'
        use_graph = use_graph and replayable
        use_plan = mode == 'plan' and replayable and not use_graph
        if use_graph and not tgdist.graphs_allowed():          # torch's RCCL process group: its watchdog cannot coexist with a capture
            if not getattr(self, '_warned_eager', False) and self.rank == 0:
                print("tg: backend %r cannot run beside hipGraph capture (tg/dist.py) - launching eagerly; "
                      "use TG_DIST_BACKEND=rccl-direct for graphs" % tgdist.backend_name(), flush=True)
            self._warned_eager = True
            use_graph = False
        if (use_graph or use_plan) and any(v is not None for v in self._label_override.values()):
            raise lib.TgError("label_override() is a test hook of eager launches: a replayed graph / launch plan would not see it")
        segs = self._segments(pre_train)
        if self._graphs is None:
            self._graphs = {}
        graphs = self._graphs.setdefault(key, [None] * len(segs))
        if use_graph and key in self._warm_keys and any(g is None for g in graphs):
            self._capture(segs, graphs, key)
        plans = self.__dict__.setdefault('_plans', {}).setdefault(key, [None] * len(segs)) if use_plan else None
        # a segment's plan is recorded while it runs eagerly in the SECOND two-stream iteration of its kind (the first one allocates the
        # buffers and records the multi-launch plans of the RNG / filter preparation / statistics arena) and replayed from then on
        plan_ready = use_plan and ('plan', key) in self._warm_keys
        pending = []
        cx.prep_cache = {}                              # filter layouts stay valid between a network's optimiser steps
        cx.plan_tag = key
        # second-stream overlap (Context.wgrad_on_side): only beside eager launches — a captured graph with cross-stream edges replays slower
        # than the single chain on ROCm 7.2 (measured rounds 1 and 3), so graph replay stays one chain
        side_was = (cx.wgrad_side, cx.wgrad_side_all)
        on = cx.wgrad_side
        if not cx.wgrad_side_env:
            on = ((not use_graph) and mode in ('overlap', 'auto', 'plan')) or cx.side_fwd_only
            cx.wgrad_side, cx.wgrad_side_all = on, on
        try:
            for i, (fn, grads, wait) in enumerate(segs):
                if wait:                                    # this segment opens with an optimiser step: its network's buckets must be in
                    mark = self._mark() if (self._exposed is not None and pending) else None
                    for wk in pending:
                        tgdist.wait_(wk)
                    if mark is not None:
                        self._exposed.append((mark, self._mark()))
                    pending = []
                if use_graph and graphs[i] is not None:
                    lib.call('tg_graph_launch', graphs[i], cx.stream)
                elif use_plan and plans[i] is not None:
                    plans[i].replay()
                elif plan_ready:
                    plans[i] = self._record_plan(fn)
                else:
                    fn()
                if grads is not None and tgdist.active():
                    pending.append(tgdist.allreduce_sum_async_(grads))
        finally:
            cx.prep_cache = None
            cx.join_wgrad_side()
            cx.wgrad_side, cx.wgrad_side_all = side_was
        self._warm = True
        self._warm_keys.add(key)       # graphs of a mode are captured from its SECOND iteration on: the first one allocates its buffers eagerly
        if on and not use_graph:
            self._warm_keys.add(('plan', key))       # one two-stream iteration has run: its events and side-stream workspaces exist
        self.iteration += 1

    def _record_plan(self, fn):
        """run segment `fn` eagerly on the two streams while every launch and event operation is appended to a native launch plan
        (tg/plan.py, include/tg_plan.h); returns the plan.  Buffers may not be born during the recording (the plan holds addresses), and
        the stores are frozen like under a captured graph."""
        from tg.plan import Plan
        cx = self.cx
        for st in cx.stores.values():
            st.frozen = True
        plan = Plan([cx.torch_stream.cuda_stream, cx.side_stream.cuda_stream])
        was = cx.capturing
        cx.capturing = True
        try:
            with plan.recording():
                fn()
        finally:
            cx.capturing = was
        return plan

    # ---- EXEC_MODE = 'auto': which way of launching is faster for THIS workload on THIS host is measured, not assumed
    AUTO_TIMED = 5                     # timed iterations per block
    AUTO_SETTLE = 3                    # untimed iterations in front of each timed block (allocations, lazily loaded code objects, recorded launch plans)
    AUTO_BLOCKS = 3                    # blocks per candidate, alternating: a candidate's time is its FASTEST block
    AUTO_ITERS = 2 * AUTO_BLOCKS * (AUTO_SETTLE + AUTO_TIMED) + 1

    def _auto_mode(self, key):
        """Both candidates compute the same numbers; which is faster depends on the workload: the CIFAR-10 / SVHN steps (15 ms of large
        kernels) gain 2.5 - 4 % from the second-stream overlap that only eager launches can have, the MNIST step (2.7 ms in ~300 launches of
        a few microseconds) is bound by the host's launch rate when launched eagerly (4.1 ms) and needs graph replay.  Schedule per graph
        key: AUTO_BLOCKS x [overlap block, graph block], a block = AUTO_SETTLE untimed + AUTO_TIMED timed iterations (the first graph block
        captures); a candidate's time is its fastest block — the first block of a fresh process on a fresh machine measures page-ins of
        library code, not the candidate (seen: 28 ms for a 14.6 ms step) — then the faster candidate for good.  Costs two device
        synchronisations per block in the first AUTO_ITERS iterations, none afterwards."""
        st = self.__dict__.setdefault('_auto', {}).setdefault(key, dict(n=0, t0=None, t={'plan': [], 'graph': []}, pick=None))
        if st['pick'] is not None:
            return st['pick']
        S, N, B = self.AUTO_SETTLE, self.AUTO_TIMED, self.AUTO_BLOCKS
        n = st['n']
        st['n'] = n + 1
        now = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
        b, k = divmod(n, S + N)
        mode = 'plan' if b % 2 == 0 else 'graph'
        if k == 0 and b > 0:                                # the previous block ends here
            st['t']['graph' if mode == 'plan' else 'plan'].append((now() - st['t0']) / N)
        if b == 2 * B:
            # replicas decide together (every rank reaches this point in the same iteration): the slowest rank's time per candidate
            st['pick'], st['best'] = tgdist.decide_together(dict(plan=min(st['t']['plan']), graph=min(st['t']['graph'])), self.cx.device)
            return st['pick']
        if k == S:
            st['t0'] = now()
        return mode

    def exec_mode_chosen(self, key='full'):
        """(mode, {candidate: seconds per iteration}) of EXEC_MODE = 'auto' once decided, else (None, partial timings)."""
        st = getattr(self, '_auto', {}).get(key)
        if st is None:
            return None, {}
        return st['pick'], (dict(st['best']) if 'best' in st else {k: min(v) for k, v in st['t'].items() if v})

    # ---- how much of the gradient exchange is NOT hidden behind the backward pass (bench.py `exchange_exposed_ms`)
    def _mark(self):
        """a point of the launch stream's timeline: a timing event on a GPU (the waits are stream-side, the host does not block),
        the host clock otherwise (gloo's wait blocks the host)."""
        if self.cx.device.type == 'cuda':
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            return e
        return time.perf_counter()

    def measure_exposed(self, on=True):
        """start (or stop) bracketing every wait for a network's gradient buckets in train_iteration."""
        self._exposed = [] if on else None

    def exposed_ms(self):
        """total time the launch stream spent stalled in those waits since measure_exposed(True) — the exchange time the backward
        pass did not hide.  Synchronises the device."""
        if not self._exposed:
            return 0.0
        if isinstance(self._exposed[0][0], float):
            return 1e3 * sum(b - a for a, b in self._exposed)
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in self._exposed))

    def _capture(self, segs, graphs, key):
        """Record every segment of one iteration as a hipGraph — all of them back to back, nothing launched and no collective issued
        in between.  Only reached when tg.dist.graphs_allowed(): the exchange backends used with graphs (rccl-direct, gloo) have no
        thread that touches HIP events behind the trainer's back, so a capture cannot be disturbed (tg/dist.py docstring)."""
        import ctypes as C
        import gc
        cx = self.cx
        cx.prep_cache = {}
        cx.plan_tag = key
        for st in cx.stores.values():
            st.frozen = True                     # the graphs hold these buffers' addresses: ParamStore.extend must not re-allocate them
        # No garbage collection inside a capture window.  A collector pass can free a PINNED host tensor of an earlier owner (an input
        # pipeline's staging slots): torch's caching host allocator then records an event on every stream the tensor was copied on — torch
        # hands out streams from a pool of 32, so in a long-lived process that can be THIS trainer's capturing stream — and its next query of
        # that captured event fails with "operation not permitted when stream is capturing", which invalidates the capture (every later
        # launch: "operation failed due to a previous error during capture").  Seen once in a full test session (round 4); mechanism
        # reproduced in tools/micro/capture_pinned_free.py.
        gc_was = gc.isenabled()
        if not os.environ.get('TG_DEBUG_CAPTURE_GC'):      # (test hook: leave the collector running, to show what the guard is for)
            gc.collect()
            gc.disable()
        try:
            for i, (fn, _grads, _wait) in enumerate(segs):
                if graphs[i] is not None:
                    continue
                lib.call('tg_graph_begin_capture', cx.stream)
                cx.capturing = True
                try:
                    if os.environ.get('TG_DEBUG_CAPTURE_SLEEP'):          # test hook: widen the capture window
                        time.sleep(float(os.environ['TG_DEBUG_CAPTURE_SLEEP']))
                    fn()
                finally:
                    cx.capturing = False
                    h = C.c_void_p()
                    lib.call('tg_graph_end_capture', cx.stream, C.byref(h))
                graphs[i] = h
        finally:
            cx.prep_cache = None
            if gc_was:
                gc.enable()

    def losses(self):
        """(d_loss, g_loss, c_loss) of the last iteration — a device->host sync; call sparingly."""
        return tuple(float(v) for v in self.loss_dev.detach().cpu().numpy())

    def training_statistics(self):
        """The reference's end-of-epoch "Get the training statistics" run (:280-285): ONE forward-only sess.run of
        [merged_summary_train, d_loss, g_loss, c_loss] on the epoch's LAST feed with train_ph = True.  One session call evaluates the
        whole graph of Model.forward_pass once: a single set of fresh dropout / noise draws shared by the three losses (the training
        iterations re-draw per solver run), every classifier application of forward_pass — C_real, C_unl, (C_unl_rep,) C_unl_d, C_fake —
        updating pop_mean / the batch-norm moving statistics once more in call-site order, the generator's batch norm too; no
        gradient, no optimiser step.  THESE losses are what the reference logs and writes to the train summary (:287-293), not the
        last iteration's.  Returns (d_loss, g_loss, c_loss) as floats (a device->host sync)."""
        cx = self.cx
        with cx.phase_scope('stats', record=False):
            PH = [self.z_g_ph, self.y_g_ph, self.x_l_c_ph, self.y_l_c_ph, self.x_l_d_ph, self.y_l_d_ph, self.x_u_d_ph, self.x_u_c_ph]
            G, D, C = self.model.forward_pass(*PH, True)                                                       # :422
            d, g, c = self._goodGAN_loss(G, D, C, None, [self.y_g_ph, self.y_l_c_ph], self.hyper[2:4], self.model.discriminator)
            out = (float(d), float(g), float(c))
        cx.rng.advance(cx)                       # the next iteration draws fresh numbers again
        return out

    # ------------------------------------------------------------------ evaluation
    def _metric(self, real_lab_logits, real_lab, metric=None):
        """:428-447: streaming accuracy of argmax(real_lab_logits) against argmax(real_lab).  Returns the reference's tuple
        (accuracy, update_op, reset_op, prediction, probs): accuracy — float(accuracy) reads the running value (this batch already
        counted); update_op(labels, logits) adds a batch; reset_op() clears the counters; prediction — one-hot arg-max of the logits
        (device tensor [n,k]); probs — None (the reference's `probs` output is never fetched)."""
        accuracy, update_op = self._accuracy_metric(real_lab, real_lab_logits, metric)
        return accuracy, update_op, accuracy.reset, ops.argmax_onehot(real_lab_logits, real_lab_logits.c), None

    def _goodGAN_loss(self, G, D, C, X, Y, Lambda, discriminator=None):
        """:449-454."""
        return self._loss_GAN(D, C, Y, Lambda)

    def evaluate(self, batches):
        """streaming accuracy of argmax C_real_logits vs argmax y over test batches with train=False (:295-351, :428-447).
        batches: iterable of (x [n,h,w,c], y onehot [n,k]) host arrays.  Returns accuracy."""
        cx, m = self.cx, self.model
        metric = None
        for x, y in batches:
            with cx.phase_scope('val', record=False):
                xa = cx.from_numpy(x, key='val:x')
                ya = cx.from_numpy(y, key='val:y')
                if m.zca() is not None:
                    xa = m.zca().apply(m.as_image(xa))
                with cx.rng_scoped('val/C'):
                    logits, _ = m.classifier(xa, False)
                metric = self._accuracy_metric(ya, logits, metric)[0]        # what _metric (:428-447) counts; its one-hot prediction output is not needed here
        return float(metric) if metric is not None else 0.0

    def sync_running_state(self):
        """Replicas keep their own running statistics (pop_mean, batch-norm moving mean / variance) and EMA shadows while
        training; they are averaged over the replicas before evaluation and before a checkpoint is written."""
        if tgdist.active():
            for st in self.cx.stores.values():
                tgdist.allreduce_mean_(st.s)
                if st.ema is not None:
                    tgdist.allreduce_mean_(st.ema)

    def sample(self, sample_z, sample_y):
        """model.good_sampler on fixed latents (:68,353-364) -> host array [N,H,W,C] in [-1,1]."""
        cx = self.cx
        with cx.phase_scope('sample', record=False):
            out = self.model.good_sampler(cx.from_numpy(sample_z, key='smp:z'), cx.from_numpy(sample_y, key='smp:y'))
            return out.numpy().reshape([-1] + list(self.config.IMAGE_DIM))

    # ------------------------------------------------------------------ the reference's entry point
    def train(self, Dataset, Model, sample_y):
        """:43-381.  Dataset(data_dir, config, num_label, subset, use_augmentation).inputpipline_train_val(val)
        -> (init_op_train, init_op_val, NNIO); NNIO.next() yields the feed of one iteration, NNIO.val_batches()
        the test split."""
        c = self.config
        dataset_train = Dataset(c.DATA_DIR, c, c.NUM_LABEL, 'train', True)
        dataset_val = Dataset(c.DATA_DIR, c, c.NUM_LABEL, 'test', False)
        init_op_train, init_op_val, NNIO = dataset_train.inputpipline_train_val(dataset_val)
        self._build_train_graph(Model)
        sample_z = np.random.uniform(low=-1.0, high=1.0, size=(c.SAMPLE_SIZE, c.Z_DIM)).astype(np.float32)   # :130
        lr, cla_lr = c.LEARNING_RATE, getattr(c, 'CLA_LEARNINIG_RATE', c.LEARNING_RATE)
        start_epoch = 0
        saver = None
        if self.save_dir:
            from Training.Saver import Saver
            saver = Saver(self.save_dir)
            if c.RESTORE:                                                              # :140-147
                start_epoch = saver.restore(self, dir_names=c.RUN, epoch=c.RESTORE_EPOCH)
                if start_epoch >= 300:
                    lr = lr * 0.995 ** (start_epoch - 300)
                    cla_lr = cla_lr * 0.99 ** (start_epoch - 300)
            elif self.rank == 0:
                saver.set_save_path(comments=self.comments)                            # :150
        if self.summary_train is not None and getattr(c, 'SUMMARY_SCALAR', True):      # :105-118
            self.summary_train.add_summary({'scalar': dict.fromkeys(('g_loss', 'd_loss', 'c_loss', 'train_accuracy'))})
            self.summary_val.add_summary({'scalar': dict.fromkeys(('val_accuracy',))})
        history = []
        iters = int(c.TRAIN_SIZE / c.BATCH_SIZE)
        for epoch in range(1, c.EPOCHS + 1):
            lambda_1 = c.FAKE_G_LAMBDA if (start_epoch + epoch) > 200 else 0.          # :165
            lambda_2 = (0.5 if epoch > 67 else 0.) if getattr(self.model, 'CONSISTENCY', False) else 0.  # :171
            if start_epoch + epoch >= 300:                                             # :175-177
                lr, cla_lr = lr * 0.995, cla_lr * 0.99
            self.set_hyper(lr, cla_lr, lambda_1, lambda_2)
            init_op_train()
            pre = bool(c.PRE_TRAIN and (start_epoch + epoch <= 30))                    # :182
            t0 = time.time()
            for i in range(iters):
                self.feed(NNIO.next())
                self.sample_latent()
                self.train_iteration(pre_train=pre)
            torch.cuda.synchronize()
            dt = time.time() - t0
            d_loss, g_loss, c_loss = self.training_statistics() if iters > 0 else self.losses()        # :280-285
            init_op_val()
            self.sync_running_state()
            acc = self.evaluate(NNIO.val_batches())
            rec = dict(epoch=epoch + start_epoch, d_loss=d_loss, g_loss=g_loss, c_loss=c_loss, val_accuracy=acc,
                       images_per_sec=iters * c.BATCH_SIZE * self.world / dt)
            history.append(rec)
            if self.summary_train is not None:                                         # :293,346
                self.summary_train.write(dict(g_loss=g_loss, d_loss=d_loss, c_loss=c_loss), epoch + start_epoch)
                self.summary_val.write(dict(val_accuracy=acc), epoch + start_epoch)
            if saver is not None and self.rank == 0 and epoch % c.SAVE_PER_EPOCH == 0:  # :366-369
                saver.save(self, 'model_' + str(epoch + start_epoch).zfill(4) + '.ckpt')
            if self.rank == 0:
                print("epoch {epoch}: g_loss {g_loss:.3f} d_loss {d_loss:.3f} c_loss {c_loss:.3f} val_acc {val_accuracy:.4f} "
                      "{images_per_sec:.0f} img/s".format(**rec), flush=True)
                if c.SAMPLE_DIR and sample_y is not None:
                    from utils import save_images, image_manifold_size
                    os.makedirs(c.SAMPLE_DIR, exist_ok=True)
                    samples = self.sample(sample_z, sample_y)
                    save_images(samples, image_manifold_size(samples.shape[0]),
                                os.path.join(c.SAMPLE_DIR, 'train_{:02d}.png'.format(epoch + start_epoch)))   # :359-363
        if saver is not None and self.rank == 0 and c.EPOCHS > 0:                      # :378-379 (after all epochs)
            saver.save(self, 'model_' + str(c.EPOCHS + start_epoch).zfill(4) + '.ckpt')
        return history


def rampup(epoch):
    """:456-462 (unused by the reference's loop; kept for API parity)."""
    if epoch < 300:
        p = 1.0 - max(0.0, float(epoch)) / float(300)
        return math.exp(-p * p * 5.0)
    return 1.0


def rampdown(epoch):
    """:464-469."""
    if epoch >= (300 - 50):
        ep = (epoch - (300 - 50)) * 0.5
        return math.exp(-(ep * ep) / 50)
    return 1.0


# ---------------------------------------------------------------------------------------------------------------------
# Experiment entry points (Training/Train_goodGAN.py:472-725).  Same TempConfig attribute values as the reference; the
# datasets / sample_y files of the reference are not part of its repository, so the synthetic Dataset (same protocol) and a
# cyclic sample_y are used unless a Dataset class is passed in.  SVHN / CIFAR-10 batch composition: the reference's own
# pipelines disagree with its placeholders; the consistent protocol is L_C / L_D / U_D+U_C per iteration.
# ---------------------------------------------------------------------------------------------------------------------

def _root_dir():
    return os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _customize_config(tmp_config, FLAGS):
    """:707-720: override config attributes from an argparse-like object."""
    for k in dir(FLAGS):
        v = getattr(FLAGS, k)
        if not k.startswith('_') and not callable(v) and v is not None and hasattr(tmp_config, k.upper()):
            setattr(tmp_config, k.upper(), v)


def _run(TempConfig, Model, Dataset, FLAGS, comments, epochs=None):
    from Input_Pipeline.syntheticDataset import syntheticDataset
    tmp_config = TempConfig()
    if FLAGS:
        _customize_config(tmp_config, FLAGS)
    if epochs is not None:
        tmp_config.EPOCHS = epochs
    tmp_config.SAMPLE_DIR = os.path.join(_root_dir(), "Training", tmp_config.SAMPLE_DIR)
    if tmp_config.NUM_LABEL < 1000:
        tmp_config.PRE_TRAIN = True                                    # :537-538,614-615
    tmp_config.display()
    training = Train(tmp_config, tmp_config.LOG_DIR, tmp_config.WEIGHT_DIR, comments=comments + tmp_config.config_str())
    sample_y = np.eye(tmp_config.NUM_CLASSES, dtype=np.float32)[np.arange(tmp_config.SAMPLE_SIZE) % tmp_config.NUM_CLASSES]
    return training.train(Dataset or syntheticDataset, Model, sample_y)


def _main_training_svhn(FLAGS=None, Dataset=None, epochs=None):
    """:472-549."""
    from config import Config
    from Model.Good_GAN import Good_GAN as Model

    class TempConfig(Config):
        NAME = "Good_GAN"
        DATA_NAME = "svhn"
        DATA_DIR = os.path.join(_root_dir(), "DataSet/svhn")
        NUM_LABEL = 500
        BATCH_SIZE = 100
        BATCH_SIZE_G = BATCH_SIZE
        BATCH_SIZE_bG = 20
        BATCH_SIZE_L_C = 50
        BATCH_SIZE_U_C = 50
        BATCH_SIZE_L_D = 20
        BATCH_SIZE_U_D = 80
        IMAGE_HEIGHT, IMAGE_WIDTH, CHANNEL = 32, 32, 3
        REPEAT = -1
        FAKE_G_LAMBDA = 0.03
        CLA_LEARNINIG_RATE = 3e-4
        Z_DIM = 100
        NUM_CLASSES = 10
        MINIBATCH_DIS = False
        RESTORE = False
        LEARNING_RATE = 3e-4
        EPOCHS = 1000
        TRAIN_SIZE = 73257 - NUM_LABEL
        SAVE_PER_EPOCH = 1
        VAL_STEP = None
        SAMPLE_DIR = "good_GAN_svhn_500"
        WEIGHT_DIR = os.path.join(_root_dir(), "Training/Weight_svhn")
        LOG_DIR = os.path.join(_root_dir(), "Training/Log_svhn")

    return _run(TempConfig, Model, Dataset, FLAGS, "This training is for svhn dataset.", epochs)


def _main_training_cifar10(FLAGS=None, Dataset=None, epochs=None):
    """:551-626.  config.ZCA must carry (mean, mat) when DATA_DIR holds no cifar10_zca_*.npy."""
    from config import Config
    from Model.Good_GAN_cifar10 import Good_GAN_cifar10 as Model

    class TempConfig(Config):
        NAME = "Good_GAN"
        DATA_NAME = "cifar10"
        DATA_DIR = os.path.join(_root_dir(), "DataSet/cifar_10")
        NUM_LABEL = 4000
        BATCH_SIZE_G = 100
        BATCH_SIZE_bG = 10
        BATCH_SIZE_L_C = 50
        BATCH_SIZE_U_C = 50
        BATCH_SIZE_L_D = 20
        BATCH_SIZE_U_D = 80
        BATCH_SIZE = BATCH_SIZE_G
        IMAGE_HEIGHT, IMAGE_WIDTH, CHANNEL = 32, 32, 3
        REPEAT = -1
        FAKE_G_LAMBDA = 0.3
        Z_DIM = 100
        NUM_CLASSES = 10
        MINIBATCH_DIS = False
        RESTORE = False                      # the reference resumes a run directory that is not in its repository (:588-592)
        LEARNING_RATE = 3e-4
        CLA_LEARNINIG_RATE = 3e-3
        EPOCHS = 1000
        TRAIN_SIZE = 60000 - NUM_LABEL
        SAVE_PER_EPOCH = 1
        VAL_STEP = None
        SAMPLE_DIR = "cifar10_good_GAN_4000"
        WEIGHT_DIR = os.path.join(_root_dir(), "Training/Weight_cifar10")
        LOG_DIR = os.path.join(_root_dir(), "Training/Log_cifar10")

    if not os.path.exists(os.path.join(TempConfig.DATA_DIR, "cifar10_zca_mat.npy")):
        q, _ = np.linalg.qr(np.random.default_rng(4321).standard_normal((3072, 3072)))      # SURVEY §8d synthetic whitening
        TempConfig.ZCA = (np.zeros(3072, np.float32), q.astype(np.float32))
    return _run(TempConfig, Model, Dataset, FLAGS, "This training is for cifar10 dataset.", epochs)


def _main_training_mnist(FLAGS=None, Dataset=None, epochs=None):
    """:628-705."""
    from config import Config
    from Model.Good_GAN import Good_GAN as Model

    class TempConfig(Config):
        NAME = "Good_GAN"
        DATA_NAME = "mnist"
        DATA_DIR = os.path.join(_root_dir(), "DataSet/mnist")
        NUM_LABEL = 100
        BATCH_SIZE_G = 100
        BATCH_SIZE_bG = 100
        BATCH_SIZE_L_C = 100
        BATCH_SIZE_U_C = 100
        BATCH_SIZE_L_D = 20
        BATCH_SIZE_U_D = 80
        BATCH_SIZE = BATCH_SIZE_G
        IMAGE_HEIGHT, IMAGE_WIDTH, CHANNEL = 28, 28, 1
        REPEAT = -1
        FAKE_G_LAMBDA = 0.1
        Z_DIM = 100
        NUM_CLASSES = 10
        MINIBATCH_DIS = False
        RESTORE = False
        LEARNING_RATE = 1e-3
        CLA_LEARNINIG_RATE = 3e-4
        EPOCHS = 1000
        TRAIN_SIZE = 60000 - NUM_LABEL
        SAVE_PER_EPOCH = 1
        VAL_STEP = None
        SAMPLE_DIR = "mnist_good_GAN_100"
        WEIGHT_DIR = os.path.join(_root_dir(), "Training/Weight_mnist")
        LOG_DIR = os.path.join(_root_dir(), "Training/Log_mnist")

    return _run(TempConfig, Model, Dataset, FLAGS, "This training is for mnist dataset.", epochs)


def _main_training_stress64(FLAGS=None, Dataset=None, epochs=None):
    """Build-defined 64x64x3 / batch-256 stress configuration (not in the reference)."""
    from config import Config
    from Model.Good_GAN_stress64 import Good_GAN_stress64 as Model

    class TempConfig(Config):
        NAME = "Good_GAN"
        DATA_NAME = "stress64"
        DATA_DIR = os.path.join(_root_dir(), "DataSet/stress64")
        NUM_LABEL = 4000
        BATCH_SIZE_G = 256
        BATCH_SIZE_bG = 10
        BATCH_SIZE_L_C = 128
        BATCH_SIZE_U_C = 128
        BATCH_SIZE_L_D = 51
        BATCH_SIZE_U_D = 205
        BATCH_SIZE = BATCH_SIZE_G
        IMAGE_HEIGHT, IMAGE_WIDTH, CHANNEL = 64, 64, 3
        REPEAT = -1
        FAKE_G_LAMBDA = 0.3
        Z_DIM = 100
        NUM_CLASSES = 10
        MINIBATCH_DIS = False
        RESTORE = False
        LEARNING_RATE = 3e-4
        CLA_LEARNINIG_RATE = 3e-3
        EPOCHS = 1000
        TRAIN_SIZE = 60000 - NUM_LABEL
        SAVE_PER_EPOCH = 1
        VAL_STEP = None
        SAMPLE_DIR = "stress64_good_GAN"
        WEIGHT_DIR = os.path.join(_root_dir(), "Training/Weight_stress64")
        LOG_DIR = os.path.join(_root_dir(), "Training/Log_stress64")

    return _run(TempConfig, Model, Dataset, FLAGS, "64x64 stress configuration.", epochs)


if __name__ == "__main__":
    # :722-725 — the reference launches the MNIST experiment
    _main_training_mnist()
