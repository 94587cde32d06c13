"""ctypes binding of libodise_hip.so (the C ABI declared in include/odise_hip.h).

The library is the product; there is NO CPU fallback.  `load()` raises if the shared object is missing or
does not export every symbol of the header, and every wrapper raises RuntimeError with
`odise_hip_last_error()` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ODISE_HIP_LIB") or os.path.join(HERE, "lib", "libodise_hip.so")  # env: developer A/B builds
HEADER_PATH = os.path.join(HERE, "..", "include", "odise_hip.h")
TOOLS_HEADER_PATH = os.path.join(HERE, "..", "include", "odise_hip_tools.h")   # developer hooks: not part of the boundary
LAB_HEADER_PATH = os.path.join(HERE, "..", "include", "odise_hip_lab.h")       # measurement build only (libodise_hip_tools.so)

F16, F32 = 0, 1
ACT_NONE, ACT_SILU, ACT_RELU, ACT_GELU, ACT_QUICKGELU = 0, 1, 2, 3, 4

c_void_p, c_int, c_int64, c_float, c_size_t = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("A", c_void_p), ("lda", c_int64),
        ("W", c_void_p), ("ldw", c_int64),
        ("C", c_void_p), ("ldc", c_int64),
        ("c_dtype", c_int),
        ("bias_n", c_void_p), ("bias_m", c_void_p), ("scale_m", c_void_p),
        ("residual", c_void_p), ("ldr", c_int64),
        ("rowgroup_add", c_void_p), ("rows_per_group", c_int), ("ldg", c_int64),
        ("act", c_int), ("geglu", c_int), ("alpha", c_float),
        ("batch", c_int),
        ("strideA", c_int64), ("strideW", c_int64), ("strideC", c_int64), ("strideR", c_int64),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int),
        ("Cout", c_int), ("KH", c_int), ("KW", c_int), ("stride", c_int),
        ("pad_t", c_int), ("pad_l", c_int), ("OH", c_int), ("OW", c_int),
        ("upsample2x", c_int),
        ("X", c_void_p), ("Wt", c_void_p), ("Y", c_void_p), ("y_dtype", c_int),
        ("bias", c_void_p), ("residual", c_void_p), ("per_image_add", c_void_p), ("per_image_add_ld", c_int64),
        ("act", c_int),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("B", c_int), ("H", c_int), ("Lq", c_int), ("Lk", c_int), ("D", c_int),
        ("Q", c_void_p), ("ldq", c_int64), ("strideQ", c_int64),
        ("K", c_void_p), ("ldk", c_int64), ("strideK", c_int64),
        ("Vt", c_void_p), ("ldvt", c_int64), ("strideVt", c_int64),
        ("O", c_void_p), ("ldo", c_int64), ("strideO", c_int64),
        ("mask", c_void_p), ("ldmask", c_int64), ("strideMask", c_int64),
        ("scale", c_float),
    ]


class JpegInfo(C.Structure):
    """odise_jpeg_info (include/odise_hip.h)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("components", C.c_int32), ("h_samp", C.c_int32), ("v_samp", C.c_int32),
                ("orientation", C.c_int32), ("restart_interval", C.c_int32), ("blocks_x", C.c_int32 * 3), ("blocks_y", C.c_int32 * 3),
                ("coef_count", C.c_int64)]


class PostDesc(C.Structure):
    """odise_post_desc (include/odise_hip.h)."""
    _fields_ = [("B", c_int), ("pad_h", c_int), ("pad_w", c_int), ("img_hw", c_void_p), ("out_hw", c_void_p), ("mask_cls", c_void_p),
                ("isthing", c_void_p), ("semantic_on", c_int), ("panoptic_on", c_int), ("instance_on", c_int),
                ("object_mask_threshold", c_float), ("overlap_threshold", C.c_double), ("topk", c_int),
                ("sem_seg", c_void_p), ("sem_argmax", c_void_p), ("panoptic", c_void_p), ("inst_masks", c_void_p),
                ("inst_table", c_void_p), ("inst_scores", c_void_p)]


class InferDesc(C.Structure):
    """odise_infer_desc (include/odise_hip.h)."""
    _fields_ = [("B", c_int), ("images", c_void_p), ("image_layout", c_int), ("img_hw", c_void_p), ("mask_cls_out", c_void_p),
                ("post", PostDesc)]


MAX_SEGMENTS = 100          # ODISE_MAX_SEGMENTS
COMM_ID_BYTES = 128         # ODISE_COMM_ID_BYTES


class UnsupportedInput(RuntimeError):
    """ODISE_ERR_UNSUPPORTED: a valid input the library does not handle (e.g. a progressive JPEG); nothing was computed."""


def _header_text(path: str = HEADER_PATH) -> str:
    with open(path) as f:
        text = f.read()
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def header_symbols(path: str = HEADER_PATH) -> list[str]:
    """Every function name declared in include/odise_hip.h (or the given header)."""
    return sorted(set(re.findall(r"\b(odise_hip_[a-z0-9_]+)\s*\(", _header_text(path))))


# ---- argument marshalling derived from the header's prototypes ----------------------------------------------------------------------
# Scalars are converted to the width the C prototype declares whatever the call site passed (a Python int, numpy integer or any ctypes
# scalar): without this ctypes would push a bare Python int as a 32-bit C int and silently truncate an int64_t / size_t argument.
def _scalar(ctype, conv):
    class _S(ctype):
        @classmethod
        def from_param(cls, v):
            return ctype(conv(getattr(v, "value", v)))
    _S.__name__ = "arg_" + ctype.__name__
    return _S


class _Ptr(c_void_p):
    @classmethod
    def from_param(cls, v):
        if v is None or isinstance(v, int):
            return c_void_p(v)
        if hasattr(v, "ptr") and not isinstance(v, (C._SimpleCData, C.Array, C.Structure)):   # runtime.DeviceArray
            return c_void_p(v.ptr)
        if isinstance(v, (bytes, bytearray)):
            return C.c_char_p(bytes(v))
        return v            # ctypes pointers, arrays, byref(...) results, c_void_p / c_char_p instances


_ARG = {"int": _scalar(c_int, int), "int64_t": _scalar(c_int64, int), "size_t": _scalar(c_size_t, int), "float": _scalar(c_float, float),
        "double": _scalar(C.c_double, float)}


def header_prototypes(path: str = HEADER_PATH) -> dict:
    """name -> list of (C type text, ctypes marshaller) per parameter, parsed from include/odise_hip.h (or the given header)."""
    protos = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(odise_hip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", _header_text(path), flags=re.S):
        name, params = m.group(1), " ".join(m.group(2).split())
        args = []
        if params not in ("", "void"):
            for prm in params.split(","):
                prm = prm.strip()
                if "*" in prm:
                    args.append((prm, _Ptr))
                else:
                    base = prm.replace("const ", "").split()[0]
                    if base not in _ARG:
                        raise RuntimeError(f"{name}: cannot marshal parameter '{prm}'")
                    args.append((prm, _ARG[base]))
        protos[name] = args
    return protos


_lib = None


def load() -> C.CDLL:
    """Load the shared library (building is `python -m odise_amd.build`); fail loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -m odise_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the ODISE hot path.")
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in header_symbols() + header_symbols(TOOLS_HEADER_PATH) if not hasattr(lib, s)]
    if missing:
        raise RuntimeError(f"libodise_hip.so does not export: {missing}")
    lib.odise_hip_last_error.restype = C.c_char_p
    protos = {**header_prototypes(), **header_prototypes(TOOLS_HEADER_PATH)}
    lab = header_prototypes(LAB_HEADER_PATH)
    if all(hasattr(lib, n) for n in lab):   # the measurement build (ODISE_HIP_LIB=.../libodise_hip_tools.so) also carries the lab hooks
        protos.update(lab)
    for name, args in protos.items():
        fn = getattr(lib, name)
        if name != "odise_hip_last_error":
            fn.restype = c_int
        fn.argtypes = [a for _, a in args]
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().odise_hip_last_error().decode("utf-8", "replace")
        if rc == -5:
            raise UnsupportedInput(f"libodise_hip {what}: {msg}")
        raise RuntimeError(f"libodise_hip {what} failed (code {rc}): {msg}")
