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


class UnsupportedInput(RuntimeError):
    """ODISE_ERR_UNSUPPORTED: a valid input the library does not handle (e.g. a progressive JPEG); nothing was computed."""


def header_symbols() -> list[str]:
    """Every function name declared in include/odise_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(odise_hip_[a-z0-9_]+)\s*\(", text)))


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
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    if missing:
        raise RuntimeError(f"libodise_hip.so does not export: {missing}")
    lib.odise_hip_last_error.restype = C.c_char_p
    for name in header_symbols():
        if name != "odise_hip_last_error":
            getattr(lib, name).restype = c_int
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().odise_hip_last_error().decode("utf-8", "replace")
        if rc == -5:
            raise UnsupportedInput(f"libodise_hip {what}: {msg}")
        raise RuntimeError(f"libodise_hip {what} failed (code {rc}): {msg}")
