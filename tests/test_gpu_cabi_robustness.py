"""Empty inputs through the C ABI: every compute entry point of include/tg_kernels.h called with ALL sizes zero (valid buffers,
zeroed descriptors, empty job lists) must come back with a status code — TG_OK for an empty batch or TG_ERR_INVALID with a message
— never a host fault (division by a zero size), a HIP launch error (empty grid) or a sticky device error; the library must stay
usable afterwards.  Runs in a child process so that a crash is reported as this test's failure, not the session's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, os, re, sys
sys.path.insert(0, os.path.join({root!r}, "tensorflow-implementation-of-triple-gan_amd"))
import torch
from tg import lib
L = lib.load()
text = re.sub(r"/\*.*?\*/", "", open(lib.HEADER_PATH).read(), flags=re.S)
text = text[text.index('extern "C"'):]
SKIP = re.compile(r"tg_(version|last_error_string|device_count|graph_|prof_|colstats_workspace_floats)")
dev = torch.zeros(1 << 22, dtype=torch.float32, device="cuda")           # 16 MB every pointer argument may point into
host = (C.c_int32 * 64)()
desc = lib.IgemmDesc()
descs = (lib.IgemmDesc * 4)()
jobs = C.create_string_buffer(4096)                                       # zeroed host memory for job lists
stream = lib.cur_stream()
for m in re.finditer(r"(?:const char\*|int64_t|int)\s+(tg_\w+)\s*\(([^;{{]*?)\)\s*;", text, flags=re.S):
    name, args = m.group(1), " ".join(m.group(2).split())
    if SKIP.match(name):
        continue
    restype, argtypes = lib.parse_header()[name]
    names = [a.strip().split()[-1].lstrip("*") for a in args.split(",")]
    vals = []
    for t, n in zip(argtypes, names):
        if n == "stream":
            vals.append(stream)
        elif t is C.c_void_p:
            vals.append(C.cast(descs, C.c_void_p) if n == "descs" else (C.cast(jobs, C.c_void_p) if n == "jobs" else lib.ptr(dev)))
        elif t == C.POINTER(lib.IgemmDesc):
            vals.append(C.byref(desc))
        elif t == C.POINTER(C.c_int32):
            vals.append(host)
        elif t == C.POINTER(C.c_float):
            vals.append((C.c_float * 8)())
        elif t in (C.c_float,):
            vals.append(0.0)
        else:
            vals.append(0)
    print("CALL", name, flush=True)
    rc = getattr(L, name)(*vals)
    print("RC", name, rc, L.tg_last_error_string().decode().replace("\n", " ")[:100] if rc else "", flush=True)
torch.cuda.synchronize()                                                  # no sticky device error
lib.call("tg_fill_f32", lib.ptr(dev), 2.5, 1024, stream)
torch.cuda.synchronize()
assert float(dev[:1024].min()) == 2.5 and float(dev[1024]) == 0.0
print("DONE", flush=True)
'''


def test_every_entry_point_survives_empty_inputs(tmp_path):
    script = tmp_path / "sweep.py"
    script.write_text(WORKER.format(root=ROOT))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    lines = r.stdout.splitlines()
    calls = [l.split()[1] for l in lines if l.startswith("CALL")]
    rcs = {l.split()[1]: (int(l.split()[2]), " ".join(l.split()[3:])) for l in lines if l.startswith("RC")}
    assert r.returncode == 0 and lines and lines[-1] == "DONE", (calls[-1] if calls else None, r.stderr[-1500:])
    assert len(rcs) >= 55, len(rcs)
    bad = {k: v for k, v in rcs.items() if v[0] not in (0, -1)}
    assert not bad, bad                                                  # -2 = a HIP call failed (e.g. an empty grid was launched)
    no_msg = [k for k, v in rcs.items() if v[0] == -1 and not v[1]]
    assert not no_msg, no_msg                                            # every rejection explains itself


def test_bf16_launch_without_its_scratch_is_an_error_not_another_kernel():
    """include/tg_kernels.h: the bf16 3x3 kernel packs its filter into CALLER-OWNED scratch (size: tg_igemm_workspace_bytes).  A
    launch that needs it and gets none / too little fails with a message naming the size — round 2 silently took a slower kernel (and
    allocated device memory inside the library)."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
    import torch
    from tg import geom, lib
    lib.load()
    n, hw, ci, co = 16, 16, 128, 128
    d = geom.conv_fwd(n, hw, hw, ci, co, 3, 1, 'SAME')
    x = torch.randn(n, hw, hw, ci, device='cuda')
    w = torch.randn(co, 9, ci, device='cuda') * 0.05
    y = torch.zeros(n, hw, hw, co, device='cuda')
    st = lib.cur_stream()
    was = lib.call('tg_conv3x3_policy', 1)                    # "wherever the layer applies": this small launch takes the halo kernel
    try:
        need = lib.call('tg_igemm_workspace_bytes', C.byref(d), 1, None, 0, 1)
        assert need == 1 * 2 * 9 * 16384
        before = lib.call('tg_conv3x3_launches')
        for scratch, nbytes in ((None, 0), (torch.empty(need // 4, device='cuda'), need - 16)):
            with pytest.raises(lib.TgError, match=r'bytes of scratch.*tg_igemm_workspace_bytes\(\.\.\., bf16 = 1\)'):
                lib.call('tg_igemm_bf16', d, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), lib.ptr(scratch), nbytes, st)
        torch.cuda.synchronize()
        assert float(y.abs().max()) == 0.0 and lib.call('tg_conv3x3_launches') == before          # nothing ran
        wpk = torch.empty(need // 4, device='cuda')
        lib.call('tg_igemm_bf16', d, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), lib.ptr(wpk), need, st)
        torch.cuda.synchronize()
        assert lib.call('tg_conv3x3_launches') == before + 1 and float(y.abs().max()) > 0.0
    finally:
        lib.call('tg_conv3x3_policy', was)
