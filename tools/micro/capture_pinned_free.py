#!/usr/bin/env python3
"""Mechanism behind Train._capture's garbage-collection guard (round 4): freeing a PINNED host tensor that was used for an asynchronous copy on
stream S makes torch's caching host allocator record an event on S; if S is capturing, that event is a captured one and the allocator's next
query of it (at the next pinned allocation, from any owner) fails with hipErrorCapturedEvent and invalidates the capture — every later launch
of the capture reports "operation failed due to a previous error during capture".  Measured: the free alone is harmless, free + allocation
inside the window is not.  torch hands out streams from a pool of 32 per device, so a long-lived process can meet an earlier owner's
stream again.  Prints what happens with and without the free inside the window."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
import torch  # noqa: E402
from tg import lib  # noqa: E402

lib.load()
for free_inside in (0, 1, 2):                     # 0: nothing; 1: free only; 2: free + another pinned allocation
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        pin = torch.empty(1 << 16).pin_memory()
        dev = pin.cuda(non_blocking=True)
        buf = torch.empty(1 << 16, device='cuda')
        torch.cuda.synchronize()
        lib.call('tg_graph_begin_capture', C.c_void_p(s.cuda_stream))
        msg = 'ok'
        try:
            lib.call('tg_fill_f32', lib.ptr(buf), 1.0, buf.numel(), C.c_void_p(s.cuda_stream))
            if free_inside:
                del pin                                     # host allocator: event recorded on the capturing stream
            if free_inside == 2:
                again = torch.empty(1 << 16).pin_memory()   # ... and queried here
            lib.call('tg_fill_f32', lib.ptr(buf), 2.0, buf.numel(), C.c_void_p(s.cuda_stream))
        except Exception as e:                              # noqa: BLE001
            msg = '%s: %s' % (type(e).__name__, str(e)[:160])
        h = C.c_void_p()
        try:
            lib.call('tg_graph_end_capture', C.c_void_p(s.cuda_stream), C.byref(h))
        except Exception as e:                              # noqa: BLE001
            msg += ' | end_capture: %s' % str(e)[:160]
    print("inside the capture window: %-34s -> %s" % (("nothing", "pinned tensor freed", "pinned tensor freed + pinned allocation")[free_inside], msg), flush=True)
