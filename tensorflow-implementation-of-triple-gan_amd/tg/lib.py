"""ctypes binding of libtg_hip.so (include/tg_kernels.h).  There is NO fallback: if the
HIP library is missing or a call fails, the product path raises."""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", os.environ.get("TG_LIB", "libtg_hip.so"))

TG_MAX_TAPS = 25
ACT = {None: 0, 'none': 0, 'lrelu': 1, 'relu': 2, 'tanh': 3, 'sigmoid': 4, 'softplus': 5}


class IgemmDesc(C.Structure):
    _fields_ = [
        ("n_img", C.c_int32),
        ("h_in", C.c_int32), ("w_in", C.c_int32), ("ld_in", C.c_int32),
        ("h_v", C.c_int32), ("w_v", C.c_int32),
        ("s_y", C.c_int32), ("s_x", C.c_int32),
        ("h_out", C.c_int32), ("w_out", C.c_int32), ("ld_out", C.c_int32),
        ("os_y", C.c_int32), ("os_x", C.c_int32), ("oo_y", C.c_int32), ("oo_x", C.c_int32),
        ("c_out", C.c_int32), ("n_store", C.c_int32), ("n_taps", C.c_int32),
        ("dy", C.c_int8 * TG_MAX_TAPS), ("dx", C.c_int8 * TG_MAX_TAPS),
        ("tapw", C.c_int16 * TG_MAX_TAPS),
        ("w_sn", C.c_int64), ("w_st", C.c_int64),
        ("act", C.c_int32), ("alpha", C.c_float),
        ("n_group", C.c_int32),
    ]


class WnJob(C.Structure):
    """tg_wn_job: one filter-gradient tail (slab reduction [+ weight-norm gradient]) of tg_filter_grad_tail_multi_f32."""
    _fields_ = [("slab", C.c_void_p), ("dw", C.c_void_p), ("v", C.c_void_p), ("g", C.c_void_p), ("dv", C.c_void_p), ("dg", C.c_void_p),
                ("coef", C.c_void_p), ("n_split", C.c_int32), ("t", C.c_int32), ("c_pad", C.c_int32), ("n_pad", C.c_int32),
                ("c_in", C.c_int32), ("c_out", C.c_int32)]


class PrepJob(C.Structure):
    """tg_prep_job: one layer's weight-norm scale + filter re-layout of tg_filter_prep_multi_f32."""
    _fields_ = [("src", C.c_void_p), ("g", C.c_void_p), ("scale", C.c_void_p), ("dst_same", C.c_void_p), ("dst_tr", C.c_void_p),
                ("tr_sb", C.c_int64), ("tr_st", C.c_int64), ("t", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("a_pad", C.c_int32),
                ("b_pad", C.c_int32)]


class CopyJob(C.Structure):
    """tg_copy_job: one contiguous copy of tg_copy_multi_f32."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_int64)]


class RngJob(C.Structure):
    """tg_rng_job: one random draw of tg_rng_multi_f32."""
    _fields_ = [("out", C.c_void_p), ("n", C.c_int64), ("mode", C.c_int32), ("a", C.c_float), ("b", C.c_float), ("stream_id", C.c_uint32)]


HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "tg_kernels.h")

_SCALARS = {"int": C.c_int, "int32_t": C.c_int32, "int64_t": C.c_int64, "uint32_t": C.c_uint32, "float": C.c_float}
_RET = {"int": C.c_int, "int64_t": C.c_int64, "const char*": C.c_char_p}
_NOCHECK = {"tg_version", "tg_last_error_string", "tg_device_count", "tg_prof_num_classes", "tg_prof_class_name",
            "tg_colstats_workspace_floats", "tg_igemm_colsum_supported", "tg_conv3x3_policy", "tg_conv3x3_launches",
            "tg_deconv5x5s2_narrow_supported", "tg_conv3x3_packed_supported"}
_NEGATIVE_IS_ERROR = {"tg_wgrad_splits", "tg_wgrad_splits_bf16", "tg_wgrad_workspace_bytes", "tg_filter_workspace_bytes",
                      "tg_igemm_workspace_bytes", "tg_deconv5x5s2_narrow_wgrad_workspace_bytes",
                      "tg_conv3x3_packed_wgrad_workspace_bytes"}     # return a count / size, < 0 on error
HOST_INT_ARRAYS = {"seg_rows", "tapmap"}          # pointer arguments that are HOST arrays


def _ctype_of(decl):
    decl = re.sub(r"/\*.*?\*/", "", decl).strip()
    name = decl.split()[-1].lstrip("*")
    typ = decl[: decl.rindex(name)].strip()
    if typ.endswith("*"):
        base = typ.replace("const", "").replace("*", "").strip()
        if base == "tg_igemm_desc":
            return C.c_void_p if name == "descs" else C.POINTER(IgemmDesc)
        if typ.count("*") == 2:
            return C.POINTER(C.c_void_p)
        if name in HOST_INT_ARRAYS or (name.endswith("_out") and base == "int32_t"):
            return C.POINTER(C.c_int32)
        if name == "state":
            return C.c_void_p
        if name == "weights":                      # HOST float array (tg_c_loss_terms_f32)
            return C.POINTER(C.c_float)
        if name in ("ms", "flops", "bytes"):
            return C.POINTER(C.c_double)
        if name == "launches":
            return C.POINTER(C.c_int64)
        if name == "path":
            return C.c_char_p
        return C.c_void_p
    return _SCALARS[typ]


def parse_header(path=HEADER_PATH):
    """{symbol: (restype, [argtypes])} for every function include/tg_kernels.h declares."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = text[text.index('extern "C"'):]
    out = {}
    for m in re.finditer(r"(const char\*|int64_t|int)\s+(tg_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        argtypes = [] if args in ("void", "") else [_ctype_of(a) for a in args.split(",")]
        out[name] = (_RET[ret], argtypes)
    return out


_lib = None
_recorder = None          # a tg.plan.Plan while Plan.recording() is open: every launch that succeeds is also appended to it


class TgError(RuntimeError):
    pass


def load():
    """dlopen libtg_hip.so and type every symbol of include/tg_kernels.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TgError("HIP extension missing: %s (run __graft_entry__.build() / make -C csrc)" % LIB_PATH)
    # torch first: it bundles its own libamdhip64; loading ours afterwards binds libtg_hip.so to that SAME runtime
    # (device buffers and streams are torch's).  The other order leaves two HIP runtimes in the process.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(lib, name)          # AttributeError if the header and the .so drift apart
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def call(name, *args):
    """Invoke a tg_* entry point; raise TgError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if name in _NOCHECK:
        return rc
    if name in _NEGATIVE_IS_ERROR:
        if rc < 0:
            raise TgError("%s failed (%d): %s" % (name, rc, lib.tg_last_error_string().decode()))
        return rc
    if rc != 0:
        raise TgError("%s failed (%d): %s" % (name, rc, lib.tg_last_error_string().decode()))
    if _recorder is not None:
        _recorder.on_call(name, args)
    return rc


def igemm_workspace_bytes(name, args):
    """tg_igemm_workspace_bytes for a call `name(*args)` of a tg_igemm_* entry point written WITHOUT its (scratch, scratch_bytes)
    arguments: descriptors, segments and operand type are read off the argument list."""
    n_desc = args[1] if '_multi_' in name else 1
    d0 = C.byref(args[0]) if isinstance(args[0], IgemmDesc) else args[0]
    seg, nseg = None, 0
    if '_colsum_' in name:
        seg, nseg = args[4], args[5]
    elif '_bnstat_' in name or '_bnbwdstat_' in name:
        seg, nseg = args[5], args[6]
    elif '_actsum_' in name:
        seg, nseg = args[7], args[8]
    return call('tg_igemm_workspace_bytes', d0, n_desc, seg, nseg, 1 if name.endswith('bf16') else 0)


def call_igemm(name, *args, scratch=True):
    """call a tg_igemm_* entry point with the arguments of include/tg_kernels.h minus (scratch, scratch_bytes): a torch buffer of the size
    the library asks for is allocated and handed in (tools and tests; the package's own launches use per-call-site workspaces, tg/ops.py).
    scratch=False: NULL scratch (the one-launch schedule; an error for a bf16 launch of the halo kernel)."""
    import torch
    need = igemm_workspace_bytes(name, args) if scratch else 0
    buf = torch.empty(max(need // 4, 4), dtype=torch.float32, device='cuda') if need > 0 else None
    return call(name, *(args[:-1] + (ptr(buf), need, args[-1])))


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def desc_array(descs):
    """contiguous ctypes array of descriptors for tg_igemm_multi_f32."""
    arr = (IgemmDesc * len(descs))(*descs)
    return arr
