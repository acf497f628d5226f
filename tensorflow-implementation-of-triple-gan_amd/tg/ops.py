"""Kernel-level building blocks (forward + recorded backward) that Model/nn.py and
Model/model_base.py compose.  Every function launches tg_* kernels through the C ABI on the
context's stream and, when a tape is active, records a closure producing the input / variable
gradients.  Gradient of an activation `a` lives in `a.grad` (an Act of identical layout)."""
import ctypes as C

from . import geom, lib
from .lib import ACT
from .runtime import Act, ctx, pad32, seg_array

import os as _os
_BNSTAT = _os.environ.get('TG_BN_STAT_FUSE', '1') != '0'   # A/B switch: batch-norm statistics taken in the producing convolution's epilogue
_BNBWDSTAT = _os.environ.get('TG_BN_BWD_STAT_FUSE', '1') != '0'   # A/B switch: batch-norm BACKWARD statistics taken in the epilogue of the launch that produces dy
_BNACT = _os.environ.get('TG_BN_ACT_FUSE', '1') != '0'      # A/B switch: activation derivative + bias gradient folded into the batch-norm backward pass
_NARROW = _os.environ.get('TG_NARROW_DECONV', '1') != '0'     # A/B switch of csrc/narrow.hip (the generator's image layer, backward)
_ACTSUM = _os.environ.get('TG_ACTSUM', '1') != '0'      # A/B switch of the input-gradient + activation-derivative + column-sum fusion
_MFMA_F32 = ('tg_igemm_f32', 'tg_igemm_multi_f32', 'tg_igemm_colsum_f32', 'tg_igemm_actsum_f32', 'tg_igemm_bnstat_f32', 'tg_igemm_bnbwdstat_f32', 'tg_wgrad_f32',
             'tg_igemm_labels_f32')
_PACKED = _os.environ.get('TG_PACKED_CONV', '1') != '0'       # A/B switch of csrc/packed_conv.hip (3x3 convolutions of <= 16 input channels: the discriminators' first layer)
_WIDE_SIDE = _os.environ.get('TG_WIDE_SIDE', '1') != '0'         # A/B switch: the many-split filter gradients of small filters (and their reduction) on the second stream
_POOL_FUSE = _os.environ.get('TG_POOL_FUSE', '1') != '0'         # A/B switch: mean-only-BN apply + max-pool 2x2 + dropout in one launch (tg_mobn_apply_pool_f32)
_CONCAT_FUSE = _os.environ.get('TG_CONCAT_FUSE', '1') != '0'     # A/B switch: conv -> cond_concat pairs written by the convolution's own epilogue (tg_igemm_labels_*)


def _call(name, *args):
    """lib.call; the MFMA launches switch to their bf16-operand variants when the context asks for the "bf16 MFMA conv path"
    (Context.mfma_dtype = 'bf16', BASELINE.json configs[3]).  The igemm entry points take caller-owned scratch (include/tg_kernels.h):
    sized by tg_igemm_workspace_bytes — the partial sums of tiles the schedule cuts along K, the packed bf16 filter of the 3x3 kernel —
    one buffer per call site like every other workspace; callers here pass the arguments WITHOUT it."""
    cx = ctx()
    if name in _MFMA_F32:
        bf16 = cx.mfma_dtype == 'bf16'
        if bf16:
            name = name[:-3] + 'bf16'
        if not name.startswith('tg_wgrad'):
            args = igemm_scratch(cx, name, args, bf16)
    return lib.call(name, *args)


def igemm_scratch(cx, name, args, bf16):
    """insert (scratch, scratch_bytes) in front of the stream argument of a tg_igemm_* call."""
    need = lib.igemm_workspace_bytes(name, args)
    buf = cx.scratch('igws', (need + 3) // 4) if need > 0 else None
    return args[:-1] + (_p(buf), need, args[-1])


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _single_consumer(y, gy):
    """The fused backward paths hand a layer its gradient ALREADY multiplied by act'(y) (Act.grad_is_dpre: a batch norm behind it did; Act.grad_fused:
    the consumer's input-gradient launch did).  That is only the layer's pre-activation gradient if that consumer was the ONLY writer of the
    buffer: a second consumer's raw contribution in the same buffer cannot be told apart afterwards — refuse instead of training on it."""
    if (y.grad_is_dpre or y.grad_fused is not None) and gy.contribs > 1:
        raise lib.TgError("fused backward: the gradient of a %dx%dx%dx%d activation was written by %d consumers, but one of them already folded "
                          "the activation derivative into it (TG_BN_ACT_FUSE=0 / TG_ACTSUM=0 run such a topology unfused)" % (y.n, y.h, y.w, y.c, gy.contribs))


def _segs(x, segments):
    """per-application image counts -> row counts of the batched activation."""
    if segments is None:
        segments = [x.n]
    assert sum(segments) == x.n, (segments, x.n)
    per = x.h * x.w
    return [s * per for s in segments]


# ------------------------------------------------------------------ statistics helpers

def colstats(mode, a_t, ld_a, b_t, ld_b, rows, c, seg_rows, act=None, alpha=0.2, s1=None, s2=None):
    cx = ctx()
    nseg = len(seg_rows)
    wsn = _call('tg_colstats_workspace_floats', rows, nseg, c)
    work = cx.scratch('cs', wsn)
    if s1 is None:
        s1 = cx.scratch('s1_', nseg * c + 4)          # +4: mode 4 reads its per-channel input in 16-B groups
    if s2 is None and mode in (1, 3):
        s2 = cx.scratch('s2_', nseg * c + 4)
    _call('tg_colstats_f32', mode, _p(a_t), ld_a, _p(b_t), ld_b, rows, c, seg_array(seg_rows), nseg, ACT[act], alpha,
          _p(work), _p(s1), _p(s2), cx.stream)
    return s1, s2


def _run_prep_plan(cx, plan):
    """tg_filter_prep_multi_f32 for the recorded layers that are not prepared yet; every result goes into the cache tagged with the
    scratch count its layer would have consumed, so that call-site numbering stays what it was in the recording pass."""
    todo = [j for j in plan if j['key'] not in cx.prep_cache]
    for k in range(0, len(todo), 24):
        part = todo[k:k + 24]
        arr = (lib.PrepJob * len(part))(*[
            lib.PrepJob(j['kernel'].data_ptr(), j['g'].data_ptr() if j['g'] is not None else None, j['scale'].data_ptr() if j['g'] is not None else None,
                        j['w_hwio'].data_ptr(), j['w_oti'].data_ptr(), j['t'] * j['a_pad'], j['a_pad'], j['t'], j['a'], j['b'], j['a_pad'], j['b_pad'])
            for j in part])
        lib.call('tg_filter_prep_multi_f32', arr, len(part), cx.stream)
    for j in todo:
        cx.prep_cache[j['key']] = (j['scale'], j['w_oti'], j['w_hwio'], j['bump'], cx.phase)


def filter_grad(desc, in_act_t, dout_t, t, c_dim, n_dim, dst, wn=None, defer=True):
    """dst[t][c_dim][n_dim] = filter gradient via tg_wgrad_f32 slabs + deterministic reduce.
    wn=(v, g, dv, dg): weight-normalised layer — dst is the gradient of the effective filter (scratch), dv / dg the variables'.
    The tail (slab reduction [+ weight-norm gradient]) is DEFERRED to Context.flush_tails (end of the backward pass / bucket
    boundary), where the tails of all layers go out as three launches; the 512-split first convolution keeps its own reduce."""
    cx = ctx()
    ns = geom.wgrad_splits(desc, cx.mfma_dtype == 'bf16')      # pixel split and slab size: the library's rule (tg_wgrad_splits[_bf16])
    slab = cx.scratch('slab', geom.wgrad_slab_floats(desc, ns))
    deferred = defer and (cx.tape is not None or cx._phase_depth > 0)
    small = desc.n_img * desc.h_v * desc.w_v * desc.ld_in * desc.c_out * desc.n_taps < (1 << 34)      # < 34 GFLOP: the generic kernel's launches
    wide = ns >= 32 and t * c_dim * n_dim <= 65536
    if deferred and (small or cx.wgrad_side_all) and not wide:
        with cx.wgrad_on_side():                                 # beside the input-gradient chain (Context.wgrad_on_side; joined in flush_tails)
            _call('tg_wgrad_f32', desc, _p(in_act_t), _p(dout_t), _p(slab), ns, cx.stream)
    elif deferred and small and wide and _WIDE_SIDE:
        # many splits of a small filter (the discriminator's first layers, the classifier's first): the launch AND its own reduction go to
        # the second stream, so the input-gradient chain does not wait for them (round 4: they were 0.18 ms of the D-update's launch stream)
        with cx.wgrad_on_side():
            _call('tg_wgrad_f32', desc, _p(in_act_t), _p(dout_t), _p(slab), ns, cx.stream)
            _call('tg_slab_reduce_f32', _p(slab), ns, t, desc.ld_in, desc.c_out, c_dim, n_dim, _p(dst), cx.stream)
            if wn is not None:
                coef = cx.scratch('coef', 2 * n_dim)
                _call('tg_wn_bwd_f32', _p(dst), _p(wn[0]), _p(wn[1]), t * c_dim, n_dim, _p(wn[2]), _p(wn[3]), _p(coef), cx.stream)
        return
    else:
        _call('tg_wgrad_f32', desc, _p(in_act_t), _p(dout_t), _p(slab), ns, cx.stream)
    if deferred and not wide:
        coef = cx.scratch('coef', 2 * n_dim) if wn is not None else None
        j = lib.WnJob(slab.data_ptr(), dst.data_ptr(), wn[0].data_ptr() if wn else None, wn[1].data_ptr() if wn else None,
                      wn[2].data_ptr() if wn else None, wn[3].data_ptr() if wn else None, coef.data_ptr() if wn else None,
                      ns, t, desc.ld_in, desc.c_out, c_dim, n_dim)
        cx.tail_jobs.append(j)
        return
    with cx.on_side(after_main=True):              # nothing on the main stream consumes dst before the phase's join
        _call('tg_slab_reduce_f32', _p(slab), ns, t, desc.ld_in, desc.c_out, c_dim, n_dim, _p(dst), cx.stream)
        if wn is not None:
            coef = cx.scratch('coef', 2 * n_dim)
            _call('tg_wn_bwd_f32', _p(dst), _p(wn[0]), _p(wn[1]), t * c_dim, n_dim, _p(wn[2]), _p(wn[3]), _p(coef), cx.stream)


# ------------------------------------------------------------------ conv / dense (plain and weight-normalised)

def conv2d(x, kernel, bias, c_out, k, stride, padding, act=None, alpha=0.2, wn=None, mobn=None, segments=None,
           train=True, kernel_grad=None, bias_grad=None, n_store_ld=None, bn_stats=False, concat=None, pool=None):
    """y = act(conv(x, W) + bias)   or, with wn=(g, g_grad) and mobn=(b, b_grad, pop_mean):
       W = g V/||V||; y = act(conv(x, W) - mean_seg + b)            (Model/nn.py:469-520,525-589).
    kernel: HWIO tensor [k,k,c_in,c_out] (flat).  Dense layers are k = 1 on [n,1,1,c].
    n_store_ld: (n_store, ld_out) override for narrow outputs (D's logit).
    bn_stats: a training-mode batch norm over `segments` follows directly — its sum / sum-of-squares pass is taken in this launch's
    epilogue (tg_igemm_bnstat_*) and left in y.bn_sums for batch_norm_train.
    concat: (label tensor [n, ncls], ncls) — a cond_concat with these labels follows directly (the discriminators' conv -> leaky relu -> concat
    pairs): the launch writes into the concatenated tensor's buffer and appends the label channels itself (tg_igemm_labels_*); the handle
    returned still describes the c_out convolution channels, y.labels tells cond_concat that its work is done.
    pool: (keep-mask tensor or None, 1/keep) — tf.nn.max_pool 2x2 + dropout follow directly (the classifier's conv1_3 / conv2_3): returns the POOLED
    activation; on the fused mean-only-BN path the apply pass and the pooling are one launch (tg_mobn_apply_pool_f32)."""
    cx = ctx()
    assert x.ld % 32 == 0, "conv input must be channel-padded to 32"
    c_in, ci_p, co_p = x.c, x.ld, pad32(c_out)
    t = k * k
    needs_w = cx.trains() and kernel_grad is not None
    needs_x = cx.tape is not None and x.requires_grad
    # MFMA-side filter layouts (+ weight-norm scale).  Inside Train.train_iteration the result is kept per variable until that
    # network's optimiser step: the classifier is prepared once for the D-update's forward passes and the C-update, the
    # discriminator once for the G- and C-updates (Context.prep_cache; None outside train_iteration: no caching).
    key = ('conv', kernel.data_ptr(), wn[0].data_ptr() if wn is not None else 0, t, c_in, c_out, ci_p, co_p)
    ent = cx.prep_cache.get(key) if cx.prep_cache is not None else None
    if ent is None and cx.prep_cache is not None:
        plan = cx.prep_plans.get((cx.plan_tag, cx.phase, key))
        if plan is not None:                               # this layer opens a recorded sequence: prepare all of its layers now
            _run_prep_plan(cx, plan)
            ent = cx.prep_cache.get(key)
    if ent is not None and (ent[2] is not None or not needs_x):
        scale, w_oti, w_hwio = ent[:3]
        if len(ent) > 3:                                   # produced by a plan in this phase: keep the call-site numbering of the
            if ent[4] == cx.phase:                         # recording pass, in which this layer allocated its own buffers here
                cx.counter += ent[3]
            cx.prep_cache[key] = ent[:3]
    else:
        scale = None
        c_before = cx.counter
        with cx.on_side(forward=True):                 # weights are final since the phase's fork: runs ahead, beside the previous layer's launch
            # (buffers written on the side stream are also ALLOCATED under it: a first-use zero fill is a launch on the current stream)
            w_oti = cx.scratch('woti', co_p * t * ci_p)
            w_hwio = cx.scratch('whwio', t * ci_p * co_p) if (needs_x or cx.prep_cache is not None) else None
            if wn is not None:
                scale = cx.scratch('wns', c_out)
                _call('tg_wn_scale_f32', _p(kernel), _p(wn[0]), t * c_in, c_out, _p(scale), cx.stream)
            _call('tg_filter_prep_f32', _p(kernel), _p(scale), None, t, c_in, c_out, ci_p, co_p, _p(w_hwio), _p(w_oti), t * ci_p, ci_p, cx.stream)
        cx.main_waits_side()
        if cx.prep_cache is not None:
            cx.prep_cache[key] = (scale, w_oti, w_hwio)
            if cx._prep_rec is not None and w_hwio is not None and not cx.use_side_stream:
                cx._prep_rec.append(dict(key=key, kernel=kernel, g=wn[0] if wn is not None else None, scale=scale, w_oti=w_oti, w_hwio=w_hwio,
                                         t=t, a=c_in, b=c_out, a_pad=ci_p, b_pad=co_p, bump=cx.counter - c_before))
    fuse_cat = (_CONCAT_FUSE and concat is not None and mobn is None and not bn_stats and n_store_ld is None and c_out == co_p)
    if fuse_cat:
        n_store, ld_out = c_out, pad32(c_out + concat[1])
    elif n_store_ld is None:
        n_store, ld_out = c_out, co_p
    else:
        n_store, ld_out = n_store_ld
    fused_act = act if mobn is None else None
    d = geom.conv_fwd(x.n, x.h, x.w, ci_p, co_p, k, stride, padding, ld_out=ld_out, n_store=n_store, act=fused_act, alpha=alpha)
    y = cx.new_act(x.n, d.h_out, d.w_out, c_out, ld_out, requires_grad=needs_w or needs_x)
    y.strided_grad_ok = True
    pooled = None
    seg_rows = _segs(y, segments)
    fused = (mobn is not None and train and c_out == co_p and c_out <= 512 and stride == 1 and geom.colsum_supported(d, seg_rows))
    packed = (_PACKED and cx.mfma_dtype != 'bf16' and k == 3 and stride == 1 and padding == 'SAME' and wn is None and mobn is None and not bn_stats
              and n_store_ld is None and c_out == co_p and act in (None, 'relu', 'lrelu') and c_in <= 16
              and bool(_call('tg_conv3x3_packed_supported', x.n, x.h, x.w, c_in, c_out)))
    if fused:
        # convolution + per-(application, channel) sums in one launch, then one fused apply pass (mean, +b, activation, pop_mean)
        b, b_grad, pop = mobn
        sums, zd = cx.zscratch('cs64', 2 * len(seg_rows) * c_out)     # fp64 accumulators
        _call('tg_igemm_colsum_f32', d, x.ptr, _p(w_oti), y.ptr, seg_array(seg_rows), len(seg_rows), _p(sums), zd, cx.stream)
        if _POOL_FUSE and pool is not None and y.h % 2 == 0 and y.w % 2 == 0 and all(r % (y.h * y.w) == 0 for r in seg_rows):
            pooled = cx.new_act(y.n, y.h // 2, y.w // 2, c_out, co_p, requires_grad=needs_w or needs_x)
            _call('tg_mobn_apply_pool_f32', y.ptr, y.ld, y.n, y.h, y.w, c_out, seg_array(seg_rows), len(seg_rows), _p(sums), _p(b), _p(pop), 0.9, ACT[act],
                  alpha, pooled.ptr, pooled.ld, _p(pool[0]), c_out, pool[1], cx.stream)
        else:
            _call('tg_mobn_apply_f32', y.ptr, y.ld, y.rows, c_out, seg_array(seg_rows), len(seg_rows), _p(sums), _p(b), _p(pop), 0.9, ACT[act],
                  alpha, cx.stream)
    elif (_BNSTAT and bn_stats and mobn is None and act in (None, 'relu', 'lrelu') and c_out == co_p == ld_out and len(seg_rows) <= 8
          and geom.colsum_supported(d, seg_rows)):
        bsum, zd = cx.zscratch('bn64', 32 * len(seg_rows) * c_out)     # the batch norm's buffer: 8 replicas x nseg x 2 x c doubles
        _call('tg_igemm_bnstat_f32', d, x.ptr, _p(w_oti), _p(bias), y.ptr, seg_array(seg_rows), len(seg_rows), _p(bsum), zd, cx.stream)
        y.bn_sums = (bsum, tuple(seg_rows))
    elif packed:
        # K-packed products straight from the [3,3,Cin,Cout] variable (csrc/packed_conv.hip); with concat the label channels ride along
        _call('tg_conv3x3_packed_fwd_f32', x.ptr, x.ld, c_in, _p(kernel), _p(bias), ACT[act], alpha, _p(concat[0]) if fuse_cat else None,
              concat[1] if fuse_cat else 0, y.ptr, ld_out, x.n, x.h, x.w, c_out, cx.stream)
        if fuse_cat:
            y.labels = (concat[0].data_ptr(), concat[1])
    elif fuse_cat:
        _call('tg_igemm_labels_f32', d, x.ptr, _p(w_oti), _p(bias), _p(concat[0]), concat[1], y.ptr, cx.stream)
        y.labels = (concat[0].data_ptr(), concat[1])
    else:
        _call('tg_igemm_f32', d, x.ptr, _p(w_oti), (_p(bias) if mobn is None else None), y.ptr, cx.stream)
    if mobn is not None and not fused:
        b, b_grad, pop = mobn
        sums = None
        if train:
            sums, _ = colstats(0, y.t, y.ld, None, 0, y.rows, c_out, seg_rows)
        shift = cx.scratch('shift', len(seg_rows) * c_out)
        _call('tg_mobn_finalize_f32', _p(sums), seg_array(seg_rows), len(seg_rows), y.rows, c_out, _p(b), _p(pop), 0.9,
              1 if train else 0, _p(shift), cx.stream)
        _call('tg_seg_scale_shift_act_f32', y.ptr, y.ld, y.ptr, y.ld, y.rows, c_out, c_out, seg_array(seg_rows), len(seg_rows),
              None, _p(shift), ACT[act], alpha, cx.stream)

    if not (needs_w or needs_x):
        return _pool_after(cx, y, pooled, pool)
    if mobn is not None and train and act is not None and c_out == co_p and c_out <= 512 and len(seg_rows) <= 8 and ld_out == co_p:
        y.grad_sink = (act, alpha, tuple(seg_rows))      # see Act.grad_sink
    if mobn is None and act in ('relu', 'lrelu') and needs_w and bias_grad is not None and c_out == co_p == ld_out:
        y.bias_sink = (act, alpha, bias_grad)            # a batch norm behind this layer may fold act' and the bias gradient into its backward

    def bwd():
        gy = y.grad
        assert gy is not None, "conv2d backward: no gradient reached the output"
        _single_consumer(y, gy)
        if mobn is None and act is None and gy.ld == co_p:
            dpre = gy.t                                   # the loss head already wrote a padded dlogits
            if needs_w and bias_grad is not None:
                colstats(0, dpre, co_p, None, 0, y.rows, c_out, [y.rows], s1=bias_grad)
        elif mobn is None and y.grad_is_dpre and gy.ld == co_p:
            dpre = gy.t                                   # the batch norm behind this layer already applied act' and summed the bias gradient
        else:
            dpre = cx.scratch('dpre', y.rows * co_p)
        if mobn is None and dpre is gy.t:
            pass
        elif mobn is not None and y.grad_fused is not None:
            # the consumer's input-gradient launch already stored t = dy*act'(y) in y.grad and summed its columns per application
            db = mobn[1] if needs_w else cx.scratch('db', c_out)
            _call('tg_mobn_center_f32', gy.ptr, gy.ld, _p(dpre), co_p, y.rows, c_out, seg_array(seg_rows), len(seg_rows), _p(y.grad_fused[0]),
                  y.grad_fused[1], _p(db), cx.stream)
        elif mobn is not None and c_out == co_p and c_out <= 512 and len(seg_rows) <= 8:
            db = mobn[1] if needs_w else cx.scratch('db', c_out)
            sums64, zd = cx.zscratch('bs64', 16 * len(seg_rows) * c_out)     # 8 replicas x nseg x c doubles
            _call('tg_mobn_bwd_f32', gy.ptr, gy.ld, y.ptr, y.ld, _p(dpre), co_p, y.rows, c_out, seg_array(seg_rows), len(seg_rows),
                  ACT[act], alpha, _p(sums64), zd, _p(db), cx.stream)
        elif mobn is not None:
            sums, _ = colstats(2, gy.t, gy.ld, y.t, y.ld, y.rows, c_out, seg_rows, act, alpha)
            sh = cx.scratch('bshift', len(seg_rows) * c_out)
            db = mobn[1] if needs_w else cx.scratch('db', c_out)
            _call('tg_mobn_bwd_finalize_f32', _p(sums), seg_array(seg_rows), len(seg_rows), y.rows, c_out, _p(sh), _p(db), cx.stream)
            _call('tg_seg_actgrad_shift_f32', gy.ptr, gy.ld, y.ptr, y.ld, _p(dpre), co_p, y.rows, c_out, seg_array(seg_rows),
                  len(seg_rows), _p(sh), ACT[act], alpha, cx.stream)
        elif needs_w and bias_grad is not None and co_p <= 1024:
            zs, zd = cx.zscratch('ab64', 16 * c_out)
            _call('tg_actgrad_bias_f32', gy.ptr, gy.ld, y.ptr if act else None, y.ld, _p(dpre), co_p, y.rows, c_out, ACT[act], alpha,
                  _p(zs), zd, _p(bias_grad), cx.stream)
        else:
            _call('tg_actgrad_f32', gy.ptr, gy.ld, y.ptr if act else None, y.ld, None, 0, 1.0, _p(dpre), co_p, y.rows, c_out,
                  ACT[act], alpha, cx.stream)
            if needs_w and bias_grad is not None:
                colstats(0, dpre, co_p, None, 0, y.rows, c_out, [y.rows], s1=bias_grad)
        if needs_w and packed:
            pws = cx.scratch('pkws', _call('tg_conv3x3_packed_wgrad_workspace_bytes', x.n, x.h, x.w, c_in, c_out) // 4)
            with cx.wgrad_on_side():                      # beside the input-gradient chain; joined in flush_tails
                _call('tg_conv3x3_packed_wgrad_f32', x.ptr, x.ld, c_in, _p(dpre), co_p, x.n, x.h, x.w, c_out, _p(pws), _p(kernel_grad), cx.stream)
        elif needs_w:
            dw_desc = geom.conv_wgrad(x.n, x.h, x.w, ci_p, co_p, k, stride, padding)
            if wn is None:
                filter_grad(dw_desc, x.t, dpre, t, c_in, c_out, kernel_grad)
            else:
                dw = cx.scratch('dw', t * c_in * c_out)
                filter_grad(dw_desc, x.t, dpre, t, c_in, c_out, dw, wn=(kernel, wn[0], kernel_grad, wn[1]))
        if needs_x:
            fresh = x.grad is None
            gx = cx.grad_of(x)
            dlist = geom.conv_dgrad(x.n, x.h, x.w, ci_p, co_p, k, stride, padding, ld_out=gx.ld, n_store=ci_p)
            sink = x.grad_sink
            if (_ACTSUM and sink is not None and fresh and len(dlist) == 1 and x.c == ci_p == gx.ld == x.ld and sum(sink[2]) == x.rows
                    and geom.colsum_supported(dlist[0], sink[2])):
                # x is the output of a mean-only-BN layer: this launch also applies that layer's activation derivative and sums the
                # columns per application, so its backward pass needs no statistics pass of its own (tg_mobn_center_f32)
                nsg = len(sink[2])
                gsum, zd = cx.zscratch('gs64', 2 * nsg * ci_p)
                _call('tg_igemm_actsum_f32', dlist[0], _p(dpre), _p(w_hwio), x.ptr, ACT[sink[0]], sink[1], gx.ptr, seg_array(sink[2]), nsg,
                      _p(gsum), zd, cx.stream)
                x.grad_fused = (gsum, 1)
            elif (_BNBWDSTAT and x.bn_bwd_sink is not None and fresh and len(dlist) == 1 and x.c == ci_p == gx.ld == x.ld
                  and x.bn_bwd_sink[0].ld == gx.ld and len(x.bn_bwd_sink[1]) <= 8 and sum(x.bn_bwd_sink[1]) == x.rows
                  and geom.colsum_supported(dlist[0], x.bn_bwd_sink[1])):
                # x is a training-mode batch norm's output and this launch is the first to write its gradient: the batch norm's backward
                # statistics (sum dy, sum dy * its input) are taken in this epilogue; batch_norm_train's backward checks that nothing else
                # contributed before it trusts them
                bn_in, segs = x.bn_bwd_sink
                bsums, zdb = cx.zscratch('bnb64', 32 * len(segs) * ci_p)
                _call('tg_igemm_bnbwdstat_f32', dlist[0], _p(dpre), _p(w_hwio), bn_in.ptr, gx.ptr, seg_array(segs), len(segs), _p(bsums), zdb, cx.stream)
                x.bn_bwd_sums = (bsums, gx)
            else:
                if x.bn_bwd_sums is not None:
                    x.bn_bwd_sums = None                  # a second contribution to that gradient: the sums of the first alone are not the statistics
                dds = lib.desc_array(dlist)
                _call('tg_igemm_multi_f32', dds, len(dds), _p(dpre), _p(w_hwio), None, gx.ptr, cx.stream)

    cx.record(bwd)
    return _pool_after(cx, y, pooled, pool)


def _pool_after(cx, y, pooled, pool):
    """what conv2d returns: y, or — conv2d(pool=...) — the max-pooled, dropped-out activation: made by the fused apply launch (`pooled`) or by the
    stand-alone pooling op; its backward closure goes on the tape behind the convolution's."""
    if pool is None:
        return y
    if pooled is None:
        return maxpool2_dropout(y, pool[0], pool[1])
    _record_maxpool_bwd(cx, y, pooled, pool[0], pool[1])
    return pooled


def deconv2d(x, kernel, bias, c_out, act=None, kernel_grad=None, bias_grad=None, narrow_out=False, wn=None):
    """tf.layers.conv2d_transpose 5x5 s2 'same' + bias + act (Model/modle_base.py:246-259);
    kernel [5,5,c_out,c_in].  narrow_out: store only the logical channels (generator output).
    wn=(g, g_grad): W = g * l2_normalize(V,[0,1,3]) (NN_Base._WN_deconv2d, Model/modle_base.py:130-155)."""
    cx = ctx()
    assert x.ld % 32 == 0
    c_in, ci_p, co_p = x.c, x.ld, pad32(c_out)
    needs_w = cx.trains() and kernel_grad is not None
    needs_x = cx.tape is not None and x.requires_grad
    scale_a = None
    ld_out = c_out if narrow_out else co_p
    # Narrow outputs (the 3-channel image layer) run the forward as ONE 3x3 problem over (output parity, channel) columns
    # (geom.deconv_fwd_merged): the four parities share one 32-column tile instead of padding 3 -> 32 four times (0.118 -> 0.050 ms).
    # Measured on the 128-channel layer the merged form is slower (0.19 -> 0.22 ms: 36/25 of the arithmetic outweighs the balance).
    merged = pad32(4 * c_out) < 4 * co_p and pad32(4 * c_out) <= 128
    with cx.on_side(forward=True):
        if wn is not None:
            scale_a = cx.scratch('wnsa', c_out)
            _call('tg_wn_scale_tab_f32', _p(kernel), _p(wn[0]), 25, c_out, c_in, _p(scale_a), cx.stream)
        # the 3-channel image layer's backward runs as K-packed products straight from the [5,5,Cout,Cin] variable (csrc/narrow.hip): no transposed copy
        narrow = _NARROW and x.ld == ci_p and bool(lib.call('tg_deconv5x5s2_narrow_supported', x.n, x.h, x.w, c_out, ci_p))
        w_tr = cx.scratch('wtr', 25 * ci_p * co_p) if (needs_x and not narrow) else None
        if merged:
            d, ng, tapmap = geom.deconv_fwd_merged(x.n, x.h, x.w, ci_p, c_out, ld_out, n_store=c_out, act=act)
            w_m = cx.scratch('wmrg', d.c_out * 9 * ci_p)
            _call('tg_deconv_merge_prep_f32', _p(kernel), _p(scale_a), c_out, c_in, ng, d.c_out, ci_p, (C.c_int32 * 36)(*tapmap), _p(w_m), cx.stream)
            if w_tr is not None:
                _call('tg_filter_prep_f32', _p(kernel), None, _p(scale_a), 25, c_out, c_in, co_p, ci_p, None, _p(w_tr), co_p, ci_p * co_p, cx.stream)
        else:
            w_pad = cx.scratch('wpad', 25 * co_p * ci_p)
            _call('tg_filter_prep_f32', _p(kernel), None, _p(scale_a), 25, c_out, c_in, co_p, ci_p, _p(w_pad), _p(w_tr), co_p, ci_p * co_p, cx.stream)
    cx.main_waits_side()
    y = cx.new_act(x.n, 2 * x.h, 2 * x.w, c_out, ld_out, requires_grad=needs_w or needs_x)
    y.strided_grad_ok = True
    if act in ('relu', 'lrelu') and needs_w and bias_grad is not None and c_out == co_p == ld_out:
        y.bias_sink = (act, 0.2, bias_grad)              # see conv2d
    if merged:
        _call('tg_igemm_f32', d, x.ptr, _p(w_m), _p(bias), y.ptr, cx.stream)
    else:
        dds = lib.desc_array(geom.deconv_fwd(x.n, x.h, x.w, ci_p, co_p, ld_out=ld_out, n_store=c_out, act=act))
        _call('tg_igemm_multi_f32', dds, len(dds), x.ptr, _p(w_pad), _p(bias), y.ptr, cx.stream)
    if not (needs_w or needs_x):
        return y

    def bwd():
        gy = y.grad
        assert gy is not None
        _single_consumer(y, gy)
        if y.grad_is_dpre and gy.ld == co_p:
            dpre = gy.t
        else:
            dpre = cx.scratch('dpre', y.rows * co_p)
        if dpre is gy.t:
            pass
        elif needs_w and bias_grad is not None and co_p <= 1024:
            zs, zd = cx.zscratch('ab64', 16 * c_out)
            _call('tg_actgrad_bias_f32', gy.ptr, gy.ld, y.ptr if act else None, y.ld, _p(dpre), co_p, y.rows, c_out, ACT[act], 0.2,
                  _p(zs), zd, _p(bias_grad), cx.stream)
        else:
            _call('tg_actgrad_f32', gy.ptr, gy.ld, y.ptr if act else None, y.ld, None, 0, 1.0, _p(dpre), co_p, y.rows, c_out,
                  ACT[act], 0.2, cx.stream)
            if needs_w and bias_grad is not None:
                colstats(0, dpre, co_p, None, 0, y.rows, c_out, [y.rows], s1=bias_grad)
        # the 3-channel image layer: both gradients as K-packed products (csrc/narrow.hip) — the generic tiles would pad 3 channels to 32
        if needs_w:
            dw = kernel_grad if wn is None else cx.scratch('dw', 25 * c_out * c_in)
            if narrow:
                nws = cx.scratch('nwws', lib.call('tg_deconv5x5s2_narrow_wgrad_workspace_bytes', x.n, x.h, x.w, c_out, ci_p) // 4)
                if wn is None:
                    with cx.wgrad_on_side():             # a vector-ALU kernel: beside the input-gradient chain like the other filter gradients
                        _call('tg_deconv5x5s2_narrow_wgrad_f32', _p(dpre), co_p, x.ptr, x.ld, x.n, x.h, x.w, c_out, c_in, ci_p, _p(nws), _p(dw), cx.stream)
                else:
                    _call('tg_deconv5x5s2_narrow_wgrad_f32', _p(dpre), co_p, x.ptr, x.ld, x.n, x.h, x.w, c_out, c_in, ci_p, _p(nws), _p(dw), cx.stream)
            elif wn is None:
                filter_grad(geom.deconv_wgrad(x.n, x.h, x.w, co_p, ci_p), dpre, x.t, 25, c_out, c_in, kernel_grad)
            else:
                filter_grad(geom.deconv_wgrad(x.n, x.h, x.w, co_p, ci_p), dpre, x.t, 25, c_out, c_in, dw, defer=False)   # consumed right below
            if wn is not None:
                with cx.on_side():
                    _call('tg_wn_bwd_tab_f32', _p(dw), _p(kernel), _p(wn[0]), 25, c_out, c_in, _p(kernel_grad), _p(wn[1]), cx.stream)
        if needs_x:
            gx = cx.grad_of(x)
            if narrow and gx.ld >= ci_p:
                _call('tg_deconv5x5s2_narrow_dgrad_f32', _p(dpre), co_p, _p(kernel), _p(scale_a), x.n, x.h, x.w, c_out, c_in, ci_p, gx.ptr, gx.ld, cx.stream)
            else:
                assert w_tr is not None, "deconv2d backward: the transposed filter was not prepared (gradient buffer narrower than the input's channel stride)"
                _call('tg_igemm_f32', geom.deconv_dgrad(x.n, x.h, x.w, ci_p, co_p, ld_out=gx.ld, n_store=ci_p), _p(dpre), _p(w_tr), None,
                      gx.ptr, cx.stream)

    cx.record(bwd)
    return y


def mean_only_batch_norm(x, pop_mean, b, b_grad=None, train=True, decay=0.9, segments=None):
    """x - mean + b (training, pop_mean updated) or x - pop_mean + b (Model/nn.py:147-187) as a stand-alone op on an activation
    (the layers of the models use the version fused into the convolution, conv2d(mobn=...))."""
    cx = ctx()
    c = x.c
    seg_rows = _segs(x, segments)
    nseg = len(seg_rows)
    needs = cx.tape is not None and (x.requires_grad or (cx.trains() and b_grad is not None))
    y = cx.new_act(x.n, x.h, x.w, c, x.ld, requires_grad=needs)
    y.strided_grad_ok = True
    sums = None
    if train:
        sums, _ = colstats(0, x.t, x.ld, None, 0, x.rows, c, seg_rows)
    shift = cx.scratch('shift', nseg * c)
    _call('tg_mobn_finalize_f32', _p(sums), seg_array(seg_rows), nseg, x.rows, c, _p(b), _p(pop_mean), decay, 1 if train else 0, _p(shift), cx.stream)
    _call('tg_seg_scale_shift_act_f32', x.ptr, x.ld, y.ptr, y.ld, x.rows, c, c, seg_array(seg_rows), nseg, None, _p(shift), 0, 0.0, cx.stream)
    if not needs:
        return y
    want_b = cx.trains() and b_grad is not None

    def bwd():
        gy = y.grad
        assert gy is not None
        gx = cx.grad_of(x)
        if train:
            s1, _ = colstats(0, gy.t, gy.ld, None, 0, x.rows, c, seg_rows)
            sh = cx.scratch('bshift', nseg * c)
            db = b_grad if want_b else cx.scratch('db', c)
            _call('tg_mobn_bwd_finalize_f32', _p(s1), seg_array(seg_rows), nseg, x.rows, c, _p(sh), _p(db), cx.stream)
            _call('tg_seg_actgrad_shift_f32', gy.ptr, gy.ld, gy.ptr, gy.ld, gx.ptr, gx.ld, x.rows, c, seg_array(seg_rows), nseg, _p(sh), 0, 0.0, cx.stream)
        else:
            _call('tg_actgrad_f32', gy.ptr, gy.ld, None, 0, None, 0, 1.0, gx.ptr, gx.ld, x.rows, c, 0, 0.0, cx.stream)
            if want_b:
                colstats(0, gy.t, gy.ld, None, 0, x.rows, c, [x.rows], s1=b_grad)

    cx.record(bwd)
    return y


# ------------------------------------------------------------------ batch norm (tf.contrib.layers.batch_norm, training mode)

def batch_norm_eval(x, gamma, beta, mm, mv, eps):
    """contrib batch_norm with is_training=False: y = gamma*(x-moving_mean)/sqrt(moving_var+eps)+beta (no tape: evaluation only)."""
    cx = ctx()
    c = x.c
    scale, shift = cx.scratch('bnsc', c), cx.scratch('bnsh', c)
    _call('tg_bn_eval_finalize_f32', c, _p(gamma), _p(beta), _p(mm), _p(mv), eps, _p(scale), _p(shift), cx.stream)
    y = cx.new_act(x.n, x.h, x.w, c, x.ld)
    _call('tg_seg_scale_shift_act_f32', x.ptr, x.ld, y.ptr, y.ld, x.rows, c, c, seg_array([x.rows]), 1, _p(scale), _p(shift), 0, 0.0,
          cx.stream)
    return y


def batch_norm_train(x, gamma, beta, mm, mv, eps, decay, gamma_grad=None, beta_grad=None, relu_input=False, segments=None):
    """y = gamma*(x-mu)/sqrt(var+eps)+beta over all rows (Model/modle_base.py:229-237); with `segments` (image counts of the
    applications batched into x) the statistics are per application and the moving statistics are updated application by
    application.  Fused: one statistics launch (fp64 atomics) + one apply launch per direction.
    relu_input: x is the output of a fused ReLU; the backward then also masks by x > 0 and the
    gradient it produces is wrt the PRE-ReLU value (consumed by the producing conv's backward)."""
    cx = ctx()
    c = x.c
    trains = cx.trains()
    needs = cx.tape is not None and (x.requires_grad or trains)
    seg_rows = _segs(x, segments)
    nseg = len(seg_rows)
    mean_inv = cx.scratch('bnmi', 2 * nseg * c)
    y = cx.new_act(x.n, x.h, x.w, c, x.ld, requires_grad=needs)
    if x.bn_sums is not None and x.bn_sums[1] == tuple(seg_rows):
        sums = x.bn_sums[0]                                   # the producing convolution took the statistics in its epilogue (conv2d(bn_stats=True))
        _call('tg_bn_train_apply_f32', x.ptr, x.ld, y.ptr, y.ld, x.rows, c, seg_array(seg_rows), nseg, _p(gamma), _p(beta), eps, decay, _p(mm), _p(mv),
              _p(sums), _p(mean_inv), cx.stream)
    else:
        sums, zd = cx.zscratch('bn64', 32 * nseg * c)         # 8 replicas x 2 x nseg x c doubles
        _call('tg_bn_train_f32', x.ptr, x.ld, y.ptr, y.ld, x.rows, c, seg_array(seg_rows), nseg, _p(gamma), _p(beta), eps, decay, _p(mm), _p(mv),
              _p(sums), zd, _p(mean_inv), cx.stream)
    if cx.state_replay is not None and mm is not None:
        # this forward pass is being KEPT for a later solver run that TensorFlow would re-execute (Context.sub_tape(replay=...)): the
        # re-execution's only lasting effect is one more moving-statistics update from the same batch sums
        rows_total = x.rows
        cx.state_replay.append(lambda: _call('tg_bn_moving_update_f32', _p(sums), rows_total, c, seg_array(seg_rows), nseg, decay, _p(mm), _p(mv),
                                             cx.stream))
    if not needs:
        return y
    if x.ld == y.ld and nseg <= 8:
        y.bn_bwd_sink = (x, tuple(seg_rows))              # see Act.bn_bwd_sink

    def bwd():
        gy = y.grad
        assert gy is not None
        want = trains and gamma_grad is not None
        gx = cx.grad_of(x)
        if y.bn_bwd_sums is not None and y.bn_bwd_sums[1] is gy and gy.contribs == 1:
            bsums, zdb = y.bn_bwd_sums[0], 2                # the launch that produced gy took the statistics in its epilogue (tg_igemm_bnbwdstat_*)
        else:
            bsums, zdb = cx.zscratch('bnb64', 32 * nseg * c)
        sink = x.bias_sink if _BNACT else None
        if sink is not None and x.ld == gx.ld and c % 4 == 0 and (c <= 256 and 256 % (c // 4) == 0 or c % 256 == 0):
            # x = act(conv + bias) of the layer in front (Act.bias_sink): this pass also multiplies by act'(x) and sums the columns — gx IS
            # that layer's pre-activation gradient and its bias gradient is done (no tg_actgrad_bias_f32 pass over the activation)
            act_, alpha_, bias_grad_ = sink
            dsum, zds = cx.zscratch('bnd64', 16 * c)                # 8 replicas x c doubles
            _call('tg_bn_train_bwd_act_f32', gy.ptr, gy.ld, x.ptr, x.ld, gx.ptr, gx.ld, x.rows, c, seg_array(seg_rows), nseg, _p(gamma), _p(mean_inv),
                  ACT[act_], alpha_, _p(bsums), zdb, _p(gamma_grad) if want else None, _p(beta_grad) if want else None, _p(dsum), zds,
                  _p(bias_grad_), cx.stream)
            x.grad_is_dpre = True
            return
        _call('tg_bn_train_bwd_f32', gy.ptr, gy.ld, x.ptr, x.ld, gx.ptr, gx.ld, x.rows, c, seg_array(seg_rows), nseg, _p(gamma), _p(mean_inv),
              1 if relu_input else 0, _p(bsums), zdb, _p(gamma_grad) if want else None, _p(beta_grad) if want else None, cx.stream)

    cx.record(bwd)
    return y


# ------------------------------------------------------------------ pointwise / pooling / concat

def scale_mask(x, mask_t, mscale, defer=False):
    """y = x*mask*mscale (inverted dropout, Model/modle_base.py:190-191).  defer: return a handle WITHOUT storage whose dropout the next
    op applies inside its own launch — only cond_concat does (the discriminator's dropout -> concat pairs, Good_GAN_cifar10.py:63-65,
    73-75); anything else touching the handle fails loudly on its missing buffer."""
    cx = ctx()
    if defer:
        y = Act(None, x.n, x.h, x.w, x.c, x.ld, requires_grad=x.requires_grad)
        y.pending = (x, mask_t, mscale)
        return y
    y = cx.new_act(x.n, x.h, x.w, x.c, x.ld, requires_grad=x.requires_grad)
    _call('tg_actgrad_f32', x.ptr, x.ld, None, 0, _p(mask_t), x.c, mscale, y.ptr, y.ld, x.rows, x.c, 0, 0.0, cx.stream)
    if cx.tape is not None and x.requires_grad:
        def bwd():
            gx = cx.grad_of(x)
            _call('tg_actgrad_f32', y.grad.ptr, y.grad.ld, None, 0, _p(mask_t), x.c, mscale, gx.ptr, gx.ld, x.rows, x.c, 0, 0.0, cx.stream)
        cx.record(bwd)
    return y


def cond_concat(x, y_onehot_t, ncls):
    """concat([x, y*ones], 3), output channel-padded to 32 (Model/modle_base.py:239-244)."""
    cx = ctx()
    ld = pad32(x.c + ncls)
    if x.labels is not None and x.labels == (y_onehot_t.data_ptr(), ncls) and x.ld == ld and x.pending is None:
        # the producing convolution already wrote this concatenation into its own (wide) buffer: the same storage, seen with the label channels
        out = Act(x.t, x.n, x.h, x.w, x.c + ncls, ld, requires_grad=x.requires_grad)
        if cx.tape is not None and x.requires_grad:
            def bwd_alias():                     # the convolution's backward reads its gradient through (pointer, channel stride): no copy
                g = out.grad
                x.grad = Act(g.t, x.n, x.h, x.w, x.c, g.ld)
                x.grad.contribs = 1
            cx.record(bwd_alias)
        return out
    mask_t, mscale = None, 1.0
    if x.pending is not None:                    # a deferred dropout in front (scale_mask(defer=True)): one launch for both
        x, mask_t, mscale = x.pending
    out = cx.new_act(x.n, x.h, x.w, x.c + ncls, ld, requires_grad=x.requires_grad)
    _call('tg_cond_concat_f32', x.ptr, x.ld, x.c, _p(mask_t), x.c, mscale, _p(y_onehot_t), ncls, out.ptr, ld, x.n, x.h * x.w, cx.stream)
    if cx.tape is not None and x.requires_grad:
        def bwd():   # gradient of the first x.c channels; the label channels are constants
            g = out.grad
            if mask_t is not None:
                gx = cx.grad_of(x)
                _call('tg_actgrad_f32', g.ptr, g.ld, None, 0, _p(mask_t), x.c, mscale, gx.ptr, gx.ld, x.rows, x.c, 0, 0.0, cx.stream)
            elif x.grad is None and x.strided_grad_ok:
                # no copy: x's gradient IS the leading channels of the concatenated gradient (every consumer of an activation
                # gradient reads it through (pointer, channel stride))
                x.grad = Act(g.t, x.n, x.h, x.w, x.c, g.ld)
                x.grad.contribs = 1
            else:
                gx = cx.grad_of(x)
                _call('tg_actgrad_f32', g.ptr, g.ld, None, 0, None, 0, 1.0, gx.ptr, gx.ld, x.rows, x.c, 0, 0.0, cx.stream)
        cx.record(bwd)
    return out


def pad_add(x, add_t, ld_out):
    """x + noise, channel-padded (classifier input; no gradient is ever needed upstream)."""
    cx = ctx()
    out = cx.new_act(x.n, x.h, x.w, x.c, ld_out)
    _call('tg_pad_add_f32', x.ptr, x.ld, x.c, _p(add_t), x.c, out.ptr, ld_out, x.rows, cx.stream)
    return out


def im2col3x3_add(x, add_t):
    """3x3 SAME patches of (x + noise) as a [n,h,w,9*c] activation (channel stride padded to 32): the classifier's
    first conv then runs as a 1x1 product over K = 27 instead of K = 9*32 (no gradient is needed upstream)."""
    cx = ctx()
    assert x.ld == x.c
    out = cx.new_act(x.n, x.h, x.w, 9 * x.c, pad32(9 * x.c))
    _call('tg_im2col3x3_add_f32', x.ptr, _p(add_t), x.n, x.h, x.w, x.c, out.ptr, out.ld, cx.stream)
    return out


def maxpool2_dropout(y, mask_t, mscale):
    """tf.nn.max_pool 2x2 + tf.layers.dropout (Model/Good_GAN_cifar10.py:123-124)."""
    cx = ctx()
    out = cx.new_act(y.n, y.h // 2, y.w // 2, y.c, y.ld, requires_grad=y.requires_grad)
    _call('tg_maxpool2_fwd_f32', y.ptr, y.ld, out.ptr, out.ld, _p(mask_t), y.c, mscale, y.n, y.h, y.w, y.c, cx.stream)
    _record_maxpool_bwd(cx, y, out, mask_t, mscale)
    return out


def _record_maxpool_bwd(cx, y, out, mask_t, mscale):
    """backward closure of max-pool 2x2 + dropout from y to out (shared by maxpool2_dropout and the fused apply + pool launch of conv2d)."""
    if cx.tape is not None and y.requires_grad:
        def bwd():
            fresh = y.grad is None
            gy = cx.grad_of(y)
            sink = y.grad_sink
            if (_ACTSUM and sink is not None and fresh and y.c % 4 == 0 and y.c <= 512 and sum(sink[2]) == y.rows
                    and all(r % (y.h * y.w) == 0 for r in sink[2])):
                # y is the output of a mean-only-BN layer: route, multiply by its activation derivative and sum the columns in one pass
                nsg = len(sink[2])
                gsum, zd = cx.zscratch('ps64', 16 * nsg * y.c)
                _call('tg_maxpool2_bwd_actsum_f32', out.grad.ptr, out.grad.ld, _p(mask_t), y.c, mscale, y.ptr, y.ld, gy.ptr, gy.ld, y.n, y.h, y.w,
                      y.c, seg_array(sink[2]), nsg, ACT[sink[0]], sink[1], _p(gsum), zd, cx.stream)
                y.grad_fused = (gsum, 8)
            else:
                _call('tg_maxpool2_bwd_f32', out.grad.ptr, out.grad.ld, _p(mask_t), y.c, mscale, y.ptr, y.ld, gy.ptr, gy.ld, y.n, y.h, y.w,
                      y.c, cx.stream)
        cx.record(bwd)


def global_maxpool(x):
    cx = ctx()
    out = cx.new_act(x.n, 1, 1, x.c, pad32(x.c), requires_grad=x.requires_grad)
    _call('tg_gmaxpool_fwd_f32', x.ptr, x.ld, out.ptr, out.ld, x.n, x.h * x.w, x.c, cx.stream)
    if cx.tape is not None and x.requires_grad:
        def bwd():
            gx = cx.grad_of(x)
            _call('tg_gmaxpool_bwd_f32', out.grad.ptr, out.grad.ld, x.ptr, x.ld, gx.ptr, gx.ld, x.n, x.h * x.w, x.c, cx.stream)
        cx.record(bwd)
    return out


def global_avgpool_concat(x, y_onehot_t, ncls):
    """average_pooling2d over the whole map + squeeze + concat([h, y], 1) (Good_GAN_cifar10.py:94-96)."""
    cx = ctx()
    out = cx.new_act(x.n, 1, 1, x.c + ncls, pad32(x.c + ncls), requires_grad=x.requires_grad)
    _call('tg_gavgpool_concat_f32', x.ptr, x.ld, x.c, _p(y_onehot_t), ncls, out.ptr, out.ld, x.n, x.h * x.w, cx.stream)
    if cx.tape is not None and x.requires_grad:
        def bwd():
            gx = cx.grad_of(x)
            _call('tg_gavgpool_bwd_f32', out.grad.ptr, out.grad.ld, x.ptr, x.ld, gx.ptr, gx.ld, x.n, x.h * x.w, x.c, 0, 0.0, cx.stream)
        cx.record(bwd)
    return out


def minibatch_discrimination(x, w, b, num_kernels, dim, w_grad=None, b_grad=None, concat_input=False):
    """NN_Base._minibatch_discrimination (Model/modle_base.py:110-128) on a dense [n, c] activation: A = x @ W (a 1-tap MFMA product),
    f[i,k] = sum_j exp(-|A[i,k,:] - A[j,k,:]|_1) + b[k].  Returns f, or concat([x, f], 1) when concat_input (what the SVHN
    discriminator does with it, Model/Good_GAN.py:160-161)."""
    cx = ctx()
    holder = {}
    skip = concat_input and cx.tape is not None and x.requires_grad
    if skip:
        def bwd_skip():      # recorded BEFORE the product, so it runs after the product's input gradient has been written to x.grad
            g, gx = holder['g'], cx.grad_of(x)
            _call('tg_pad_add_f32', gx.ptr, gx.ld, x.c, g.ptr, g.ld, gx.ptr, gx.ld, x.rows, cx.stream)
        cx.record(bwd_skip)
    a = conv2d(x, w, None, num_kernels * dim, 1, 1, 'SAME', kernel_grad=w_grad)
    c0 = x.c if concat_input else 0
    out = cx.new_act(x.n, 1, 1, c0 + num_kernels, pad32(c0 + num_kernels), requires_grad=a.requires_grad)
    _call('tg_minibatch_disc_fwd_f32', a.ptr, a.ld, x.ptr, x.ld, c0, _p(b), out.ptr, out.ld, x.n, num_kernels, dim, cx.stream)
    if cx.tape is not None and a.requires_grad:
        def bwd():
            g = out.grad
            holder['g'] = g
            ga = cx.grad_of(a)
            _call('tg_minibatch_disc_bwd_f32', a.ptr, a.ld, C.c_void_p(g.t.data_ptr() + 4 * c0), g.ld, ga.ptr, ga.ld, _p(b_grad), x.n, num_kernels,
                  dim, cx.stream)
        cx.record(bwd)
    return out


def argmax_onehot(logits, k):
    cx = ctx()
    out = cx.scratch('oh', logits.n * k)
    _call('tg_argmax_onehot_f32', logits.ptr, logits.ld, logits.n, k, _p(out), cx.stream)
    return out


def copy2d(dst_t, ld_d, dst_col, src_t, ld_s, rows, c):
    """dst[r][dst_col : dst_col+c] = src[r][:c] (strided device copy)."""
    cx = ctx()
    _call('tg_copy2d_f32', C.c_void_p(src_t.data_ptr()), ld_s, C.c_void_p(dst_t.data_ptr() + 4 * dst_col), ld_d, rows, c, cx.stream)


def copy_rows(dst_t, dst_off, src_t, numel):
    """contiguous device copy (batch concatenation along N)."""
    cx = ctx()
    numel = int(numel)
    _call('tg_copy2d_f32', C.c_void_p(src_t.data_ptr()), numel, C.c_void_p(dst_t.data_ptr() + 4 * int(dst_off)), numel, 1, numel, cx.stream)


def copy_many(jobs):
    """[(dst tensor, dst element offset, src tensor, numel)] as contiguous device copies, 16 per launch (tg_copy_multi_f32)."""
    cx = ctx()
    jobs = [j for j in jobs if j[3] > 0]
    for k in range(0, len(jobs), 16):
        part = jobs[k:k + 16]
        arr = (lib.CopyJob * len(part))(*[lib.CopyJob(s.data_ptr(), d.data_ptr() + 4 * int(off), int(n)) for d, off, s, n in part])
        _call('tg_copy_multi_f32', arr, len(part), cx.stream)


def reshape(x, n, h, w, c):
    """view change of a dense (ld == c) buffer, e.g. [N,8192] -> [N,4,4,512]; gradients share storage."""
    cx = ctx()
    assert x.ld == x.c and n * h * w * c == x.rows * x.c
    y = Act(x.t, n, h, w, c, c, x.requires_grad)
    if cx.tape is not None and x.requires_grad:
        def bwd():
            g = y.grad
            x.grad = Act(g.t, x.n, x.h, x.w, x.c, x.c)
        cx.record(bwd)
    return y


def add_noise(x, noise_t):
    """x + noise on a dense activation (NN_Base._add_noise, Model/modle_base.py:193-202); the gradient passes through."""
    cx = ctx()
    y = cx.new_act(x.n, x.h, x.w, x.c, x.ld, requires_grad=x.requires_grad)
    _call('tg_pad_add_f32', x.ptr, x.ld, x.c, _p(noise_t), x.c, y.ptr, y.ld, x.rows, cx.stream)
    if cx.tape is not None and x.requires_grad:
        def bwd():
            x.grad = y.grad
        cx.record(bwd)
    return y


def global_avgpool(x):
    """tf.reduce_mean(x, axis=[1,2]) (Model/Good_GAN.py:198,346) -> [N,C]."""
    cx = ctx()
    out = cx.new_act(x.n, 1, 1, x.c, pad32(x.c), requires_grad=x.requires_grad)
    _call('tg_gavgpool_concat_f32', x.ptr, x.ld, x.c, None, 0, out.ptr, out.ld, x.n, x.h * x.w, cx.stream)
    if cx.tape is not None and x.requires_grad:
        def bwd():
            gx = cx.grad_of(x)
            _call('tg_gavgpool_bwd_f32', out.grad.ptr, out.grad.ld, x.ptr, x.ld, gx.ptr, gx.ld, x.n, x.h * x.w, x.c, 0, 0.0, cx.stream)
        cx.record(bwd)
    return out


def view(x, h, w, c):
    """reinterpret a dense activation's per-image shape (tf.reshape), gradients share storage."""
    return reshape(x, x.n, h, w, c)


def concat_batch(acts):
    """tf.concat(acts, 0) of network OUTPUTS: the parts receive views of the concatenated gradient."""
    from .batching import concat_acts
    cx = ctx()
    out = concat_acts(acts)
    out.requires_grad = any(a.requires_grad for a in acts)
    if cx.tape is not None and out.requires_grad:
        def bwd():
            if out.grad is None:          # e.g. the concatenated features: no loss term of Train_goodGAN.py uses them
                return
            off = 0
            for a in acts:
                a.grad = out.grad.view_rows(off, off + a.n)
                off += a.n
        cx.record(bwd)
    return out


# ------------------------------------------------------------------ stand-alone activation / batch norm of the reference's free functions

def activation(x, act, alpha=0.2):
    """y = act(x) as its own launch (an activation CALLED on a tensor, e.g. Good_GAN_cifar10.leakyReLu(x), NN_Base._relu(x)); the models
    pass their activations as `nonlinearity=` / `activation=` instead and get them fused into the producing kernel's epilogue."""
    cx = ctx()
    y = cx.new_act(x.n, x.h, x.w, x.c, x.ld, requires_grad=x.requires_grad)
    y.strided_grad_ok = True
    _call('tg_act_f32', x.ptr, x.ld, y.ptr, y.ld, x.rows, x.c, ACT[act], alpha, cx.stream)
    if cx.tape is not None and x.requires_grad:
        def bwd():
            gx = cx.grad_of(x)
            _call('tg_actgrad_f32', y.grad.ptr, y.grad.ld, y.ptr, y.ld, None, 0, 1.0, gx.ptr, gx.ld, x.rows, x.c, ACT[act], alpha, cx.stream)
        cx.record(bwd)
    return y


def batch_norm_moments(x, scale, beta, pop_mean, pop_var, eps, decay, train, scale_grad=None, beta_grad=None):
    """nn.batch_norm_impl (Model/nn.py:192-217): tf.nn.moments + tf.nn.batch_normalization with the running statistics updated
    as pop <- pop*decay + batch*(1-decay) from the BIASED batch variance (tf.nn.moments; contrib's fused batch_norm — ops.batch_norm_train —
    feeds the unbiased one), or, train=False, normalisation by the running statistics.  Two-pass centred variance as tf.nn.moments."""
    cx = ctx()
    c = x.c
    if not train:
        return batch_norm_eval(x, scale, beta, pop_mean, pop_var, eps)
    trains = cx.trains()
    needs = cx.tape is not None and (x.requires_grad or trains)
    s1, _ = colstats(0, x.t, x.ld, None, 0, x.rows, c, [x.rows])
    s2, _ = colstats(4, x.t, x.ld, s1, 0, x.rows, c, [x.rows], alpha=1.0 / x.rows)
    sc, sh, mean_inv = cx.scratch('bnsc', c), cx.scratch('bnsh', c), cx.scratch('bnmi', 2 * c)
    _call('tg_bn_finalize_f32', _p(s1), _p(s2), x.rows, c, _p(scale), _p(beta), eps, _p(sc), _p(sh), _p(mean_inv), _p(pop_mean), _p(pop_var), decay, 0,
          cx.stream)
    y = cx.new_act(x.n, x.h, x.w, c, x.ld, requires_grad=needs)
    _call('tg_seg_scale_shift_act_f32', x.ptr, x.ld, y.ptr, y.ld, x.rows, c, c, seg_array([x.rows]), 1, _p(sc), _p(sh), 0, 0.0, cx.stream)
    if not needs:
        return y

    def bwd():
        gy = y.grad
        assert gy is not None
        gx = cx.grad_of(x)
        d1, d2 = colstats(3, gy.t, gy.ld, x.t, x.ld, x.rows, c, [x.rows])
        abc = cx.scratch('bnabc', 3 * c)
        want = trains and scale_grad is not None
        dg = scale_grad if want else cx.scratch('bndg', c)
        db = beta_grad if want else cx.scratch('bndb', c)
        _call('tg_bn_bwd_finalize_f32', _p(d1), _p(d2), x.rows, c, _p(scale), _p(mean_inv), _p(abc), _p(dg), _p(db), cx.stream)
        _call('tg_bn_bwd_apply_f32', gy.ptr, gy.ld, x.ptr, x.ld, gx.ptr, gx.ld, x.rows, c, _p(abc), 0, cx.stream)

    cx.record(bwd)
    return y


def moments_normalize(x, eps, init_scale=1.0):
    """scale_init * (x - m_init) with m, v = tf.nn.moments(x, all axes but the last), scale_init = init_scale / sqrt(v + eps): the value
    the data-dependent-initialisation branch (init=True) of the Salimans layers returns (Model/nn.py:230-243,263-277,302-316).
    Forward only (the reference never differentiates that branch: its models call it once to create variables)."""
    cx = ctx()
    c = x.c
    s1, _ = colstats(0, x.t, x.ld, None, 0, x.rows, c, [x.rows])
    s2, _ = colstats(4, x.t, x.ld, s1, 0, x.rows, c, [x.rows], alpha=1.0 / x.rows)
    sc, sh, mi = cx.scratch('bnsc', c), cx.scratch('bnsh', c), cx.scratch('bnmi', 2 * c)
    g, b = cx.ws('const:init_scale%g' % init_scale, c), cx.ws('const:zeros', max(c, 1024))
    _call('tg_fill_f32', _p(g), float(init_scale), c, cx.stream)
    _call('tg_bn_finalize_f32', _p(s1), _p(s2), x.rows, c, _p(g), _p(b), eps, _p(sc), _p(sh), _p(mi), None, None, 0.0, 0, cx.stream)
    y = cx.new_act(x.n, x.h, x.w, c, x.ld)
    _call('tg_seg_scale_shift_act_f32', x.ptr, x.ld, y.ptr, y.ld, x.rows, c, c, seg_array([x.rows]), 1, _p(sc), _p(sh), 0, 0.0, cx.stream)
    return y
