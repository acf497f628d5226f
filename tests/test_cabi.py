"""CPU checks of the drop-in boundary: libtg_hip.so loads without a GPU and exports every symbol that
include/tg_kernels.h declares; error reporting works; the product path refuses to run without a GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tg_kernels.h")
IO_HEADER = os.path.join(ROOT, "include", "tg_io.h")
PLAN_HEADER = os.path.join(ROOT, "include", "tg_plan.h")


def _declared(path):
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    return set(re.findall(r"\b(tg_\w+)\s*\(", text)) - {"tg_status", "tg_igemm_desc", "tg_plan_word"}


def test_header_is_plain_c():
    for h in (HEADER, IO_HEADER, PLAN_HEADER):
        subprocess.check_call(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", h])


def test_library_exports_every_declared_symbol():
    from tg import lib
    sigs = lib.parse_header()
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(tg_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
    declared -= {"tg_status", "tg_igemm_desc"}
    assert declared == set(sigs), declared ^ set(sigs)
    assert len(sigs) >= 50
    handle = lib.load()
    for name in sigs:
        assert hasattr(handle, name), name
    exported = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (tg_\w+)", exported))
    io_syms = _declared(IO_HEADER)                           # host-side input pipeline (include/tg_io.h)
    assert len(io_syms) == 11 and not (io_syms & set(sigs))
    plan_syms = _declared(PLAN_HEADER)                       # launch plans (include/tg_plan.h)
    assert len(plan_syms) == 10 and not (plan_syms & set(sigs))
    assert exported == set(sigs) | io_syms | plan_syms, exported ^ (set(sigs) | io_syms | plan_syms)       # nothing undeclared is exported either


def test_no_torch_types_in_signatures():
    for h in (HEADER, IO_HEADER, PLAN_HEADER):
        text = open(h).read()
        assert "torch" not in text and "at::" not in text and "#include <hip" not in text


def test_launch_plan_recorder_without_gpu():
    """include/tg_plan.h on the host: every launch entry point of tg_kernels.h (last parameter `void* stream`) has a generated trampoline
    whose signature string matches the header, the generated file is the one tools/gen_plan_thunks.py renders today, and the recorder
    rejects what it cannot replay (unknown entry point, wrong arity, a stream that is not the plan's) — no launch is made."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_plan_thunks as gen
    from tg import lib, plan
    inc = os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd", "csrc", "plan_thunks.inc")
    assert open(inc).read() == gen.render()
    sigs = lib.parse_header()
    launches = gen.launches()
    assert len(launches) >= 80
    for name, params in launches:
        assert name in sigs and len(sigs[name][1]) == len(params) + 1
        kinds = plan.signature(name)
        assert kinds == "".join("p" if t.endswith("*") else ("f" if t == "float" else "i") for t, _ in params), name
    for name in ("tg_version", "tg_igemm_workspace_bytes", "tg_graph_launch", "tg_wgrad_splits", "no_such_entry"):
        assert plan.signature(name) is None
    p = plan.Plan([0x1000, 0x2000])
    seg = (C.c_int32 * 2)(5, 7)
    p.add_launch("tg_fill_f32", (C.c_void_p(0x10), 1.5, 64, C.c_void_p(0x1000)))
    p.add_launch("tg_fill_f32", (None, C.c_float(2.0), C.c_int64(64), 0x2000))
    assert len(p) == 2 and p.launches == 2
    with pytest.raises(lib.TgError, match="not one of the plan's streams"):
        p.add_launch("tg_fill_f32", (None, 0.0, 1, C.c_void_p(0x3000)))
    with pytest.raises(lib.TgError, match="is not a launch entry point"):
        p.add_launch("tg_version", ())
    with pytest.raises(lib.TgError, match="called with 2 arguments"):
        p.add_launch("tg_fill_f32", (None, C.c_void_p(0x1000)))
    h = lib.load()
    words = (plan.PlanWord * 3)()
    assert h.tg_plan_add_launch(p.p, b"tg_fill_f32", words, 2, 0) == -1 and b"takes 3 arguments" in h.tg_last_error_string()
    assert h.tg_plan_add_launch(p.p, b"tg_nope", words, 3, 0) == -1 and b"not a launch entry point" in h.tg_last_error_string()
    assert h.tg_plan_add_launch(p.p, b"tg_fill_f32", words, 3, 99) == -1 and b"slot" in h.tg_last_error_string()
    held = p._hold(seg)                                       # host data is copied: the plan's address, the caller's bytes
    assert held != C.addressof(seg) and list((C.c_int32 * 2).from_address(held)) == [5, 7] and held % 16 == 0
    assert len(p) == 2
    one = (C.c_void_p * 1)(C.c_void_p(0x1000))
    assert h.tg_plan_replay(p.p, one, 1) == -1 and b"stream slot 1" in h.tg_last_error_string()      # refused before anything is issued


def test_version_and_error_string_without_gpu(has_gpu):
    from tg import lib
    assert lib.call("tg_version") >= 100
    n = lib.call("tg_device_count")
    if not has_gpu:
        assert n < 0 and b"hipGetDeviceCount" in lib.load().tg_last_error_string()
    else:
        assert n >= 1


def test_desc_struct_matches_header_layout():
    """sizeof / offsets of the ctypes mirror against the C compiler's view of tg_igemm_desc."""
    from tg import lib
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu\\n", sizeof(tg_igemm_desc), ' \
          'offsetof(tg_igemm_desc, dy), offsetof(tg_igemm_desc, tapw), offsetof(tg_igemm_desc, w_sn), offsetof(tg_igemm_desc, alpha));return 0;}' % HEADER
    exe = "/tmp/tg_desc_layout"
    subprocess.run(["gcc", "-x", "c", "-", "-o", exe], input=src.encode(), check=True)
    got = [int(v) for v in subprocess.check_output([exe]).split()]
    D = lib.IgemmDesc
    assert got == [C.sizeof(D), D.dy.offset, D.tapw.offset, D.w_sn.offset, D.alpha.offset]


def test_product_path_fails_loudly_without_gpu(has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    from tg import lib, runtime
    with pytest.raises(lib.TgError, match="no MI355X"):
        runtime.Context()
    with pytest.raises(lib.TgError, match="no tg Context"):
        runtime.set_context(None)
        runtime.ctx()


def test_missing_extension_is_an_error(monkeypatch):
    from tg import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libtg_hip.so")
    with pytest.raises(lib.TgError, match="HIP extension missing"):
        lib.load()


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under the package may import, link or execute it."""
    pkg = os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle|oracle[/.](tf_ops|nets_|step_)|['\"]oracle['\"]", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), os.path.join(dirpath, f)


def test_library_never_allocates_device_memory():
    """SURVEY §8b / include/tg_kernels.h: the caller owns every buffer — scratch is sized by a query (tg_*_workspace_bytes) and handed
    in.  No allocation call may appear in the kernel library's sources, and the bf16 3x3 path's scratch query is exported and answers
    on the host (no GPU needed): the packed filter of a 256 -> 256 layer is 2 x 4 x 9 images of 16 KB, a layer of another shape needs none."""
    csrc = os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd", "csrc")
    pat = re.compile(r"\bhip(Malloc|MallocAsync|MallocManaged|HostMalloc|Free|FreeAsync|ExtMallocWithFlags)\b")
    for dirpath, _, files in os.walk(csrc):
        for f in files:
            if f.endswith((".hip", ".cpp", ".h")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), os.path.join(dirpath, f)
    from tg import geom, lib
    d = geom.conv_fwd(250, 16, 16, 256, 256, 3, 1, 'SAME')            # 500 tiles: two rounds of the halo kernel
    assert lib.call('tg_igemm_workspace_bytes', C.byref(d), 1, None, 0, 1) == 2 * 4 * 9 * 16384
    assert lib.call('tg_igemm_workspace_bytes', C.byref(d), 1, None, 0, 0) == 0              # exact-fp32 operands: no packed filter
    # generic kernel: 1 000 tiles of 64 x 64 fill two rounds of 512 slots -> nothing is cut; 225 tiles (the generator's first input gradient)
    # are cut in two and their partial sums need 450 x 64 x 64 floats
    assert lib.call('tg_igemm_workspace_bytes', C.byref(geom.conv_fwd(250, 16, 16, 64, 64, 3, 1, 'SAME')), 1, None, 0, 0) == 0
    dg = geom.deconv_dgrad(100, 4, 4, 544, 256)
    assert lib.call('tg_igemm_workspace_bytes', C.byref(dg), 1, None, 0, 0) == 450 * 64 * 64 * 4
    bad = lib.IgemmDesc()
    with pytest.raises(lib.TgError, match='bad descriptor'):
        lib.call('tg_igemm_workspace_bytes', C.byref(bad), 1, None, 0, 0)


def test_comm_library_exports_every_declared_symbol():
    """include/tg_comm.h <-> libtg_comm.so (no collective is issued: loading needs RCCL in the process, not a GPU)."""
    from tg import comm, lib
    header = os.path.join(ROOT, "include", "tg_comm.h")
    subprocess.check_call(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", header])
    names = _declared(header)
    assert len(names) == 8
    handle = comm.load()
    for n in sorted(names):
        assert hasattr(handle, n), n
    with pytest.raises(lib.TgError, match="null"):
        comm.call('tg_comm_count', None, None, None)
