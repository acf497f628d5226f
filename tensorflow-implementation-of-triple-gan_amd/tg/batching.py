"""Batch-axis concatenation of activation buffers (device-to-device copies through the C ABI)."""
from . import ops
from .runtime import ctx


def concat_acts(acts, tag='cat'):
    """tf.concat(acts, axis=0) for Acts of identical [h,w,c,ld]; no gradient flows through (inputs only)."""
    cx = ctx()
    a0 = acts[0]
    for a in acts:
        assert (a.h, a.w, a.c, a.ld) == (a0.h, a0.w, a0.c, a0.ld), "concat_acts: layout mismatch"
    n = sum(a.n for a in acts)
    out = cx.new_act(n, a0.h, a0.w, a0.c, a0.ld, tag=tag)
    jobs, off = [], 0
    for a in acts:
        numel = a.rows * a.ld
        jobs.append((out.t, off, a.t, numel))
        off += numel
    ops.copy_many(jobs)                      # one launch for all parts
    return out
