"""ctypes binding of libtg_hip.so (include/tg_kernels.h).  There is NO fallback: if the
HIP library is missing or a call fails, the product path raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libtg_hip.so")

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
    ]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_D = C.POINTER(IgemmDesc)

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "tg_version": [],
    "tg_last_error_string": [],
    "tg_device_count": [],
    "tg_graph_begin_capture": [_P],
    "tg_graph_end_capture": [_P, C.POINTER(_P)],
    "tg_graph_launch": [_P, _P],
    "tg_graph_destroy": [_P],
    "tg_prof_enable": [_I],
    "tg_prof_reset": [],
    "tg_prof_num_classes": [],
    "tg_prof_class_name": [_I],
    "tg_prof_collect": [_I, C.POINTER(C.c_double), C.POINTER(_L), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "tg_igemm_f32": [_D, _P, _P, _P, _P, _P],
    "tg_wgrad_f32": [_D, _P, _P, _P, _I, _P],
}
_RESTYPES = {"tg_last_error_string": C.c_char_p, "tg_prof_class_name": C.c_char_p}
_NOCHECK = {"tg_version", "tg_last_error_string", "tg_device_count", "tg_prof_num_classes", "tg_prof_class_name"}

_lib = None


class TgError(RuntimeError):
    pass


def load():
    """dlopen libtg_hip.so and type every symbol of include/tg_kernels.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TgError("HIP extension missing: %s (run __graft_entry__.build() / make -C csrc)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the header and the .so drift apart
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def call(name, *args):
    """Invoke a tg_* entry point; raise TgError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if name in _NOCHECK:
        return rc
    if rc != 0:
        raise TgError("%s failed (%d): %s" % (name, rc, lib.tg_last_error_string().decode()))
    return rc


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
