#!/usr/bin/env python3
"""Sustained clocks and power under the dominant launch: runs tg_igemm_f32 on conv1_2 (250 images, 32x32, 128 -> 128; the halo kernel) back to
back for TG_CLOCK_SECONDS (default 6) while a child process samples `rocm-smi --showclocks --showpower` twice a second; prints the samples and
the TFLOP/s of the loop.  The fp32 MFMA peak of MI355X_MICROARCH.md (157.3 TFLOP/s) is quoted at 2.4 GHz: a launch that issues MFMAs in 92 %
of its cycles (stamps) reaches 0.92 x (sustained clock / 2.4) of it."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
import torch  # noqa: E402
from tg import geom, lib  # noqa: E402

lib.load()
SECONDS = float(os.environ.get('TG_CLOCK_SECONDS', '6'))
PREC = sys.argv[1] if len(sys.argv) > 1 else 'f32'
N, hw, ci, co = 250, 32, 128, 128
x = torch.randn(N, hw, hw, ci, device='cuda')
w = torch.randn(co, 9, ci, device='cuda') * 0.05
y = torch.empty(N, hw, hw, co, device='cuda')
d = geom.conv_fwd(N, hw, hw, ci, co, 3, 1, 'SAME')
need = lib.call('tg_igemm_workspace_bytes', C.byref(d), 1, None, 0, 1 if PREC == 'bf16' else 0)
ws = torch.empty(max(need, 16) // 4, device='cuda')
st = lib.cur_stream()


def burst(n):
    for _ in range(n):
        lib.call('tg_igemm_' + PREC, d, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), lib.ptr(ws), ws.numel() * 4, st)


burst(20)
torch.cuda.synchronize()
idle = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
print("---- idle\n" + "\n".join(l for l in idle.splitlines() if 'sclk' in l or 'Power' in l or 'mclk' in l))
sampler = subprocess.Popen([sys.executable, '-c', '''
import subprocess, time
for i in range(%d):
    time.sleep(0.5)
    o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    print("---- sample", i)
    print("\\n".join(l for l in o.splitlines() if "sclk" in l or "Power" in l or "mclk" in l), flush=True)
''' % int(SECONDS * 2 - 2)])
t0 = time.time()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
n = 0
while time.time() - t0 < SECONDS:
    burst(200)
    n += 200
    torch.cuda.synchronize()
b.record()
torch.cuda.synchronize()
sampler.wait()
ms = a.elapsed_time(b) / n
fl = 2.0 * N * hw * hw * co * 9 * ci
print("loop: %d launches, %.4f ms per launch, %.1f TFLOP/s (%s operands)" % (n, ms, fl / ms / 1e9, PREC))
