"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol declared in include/odise_hip.h.
No compute is attempted without a GPU; creating a context must fail loudly (no CPU fallback)."""
import ctypes as C
import os

import pytest
import torch

import __graft_entry__ as entry
from odise_amd import _lib


@pytest.fixture(scope="module")
def lib():
    entry.build()
    return _lib.load()


def test_exports_every_header_symbol(lib):
    names = _lib.header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n


def test_version_and_error_string(lib):
    assert lib.odise_hip_version() >= 100
    assert isinstance(lib.odise_hip_last_error(), bytes)


@pytest.mark.skipif(torch.cuda.is_available(), reason="GPU present: covered by the gpu suite")
def test_no_cpu_fallback_without_gpu(lib):
    h = C.c_void_p()
    rc = lib.odise_hip_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert b"no HIP device" in lib.odise_hip_last_error() or rc == -2
    from odise_amd.runtime import Context
    with pytest.raises(RuntimeError):
        Context(0)


def test_struct_sizes_match_header(lib):
    # the ctypes mirrors must have the C layout (checked against sizeof reported by the library)
    assert lib.odise_hip_sizeof_gemm_desc() == C.sizeof(_lib.GemmDesc)
    assert lib.odise_hip_sizeof_conv_desc() == C.sizeof(_lib.ConvDesc)
    assert lib.odise_hip_sizeof_attn_desc() == C.sizeof(_lib.AttnDesc)
    assert lib.odise_hip_sizeof_post_desc() == C.sizeof(_lib.PostDesc)
    assert lib.odise_hip_sizeof_infer_desc() == C.sizeof(_lib.InferDesc)


def test_every_export_is_declared_and_every_prototype_has_argtypes(lib):
    """The boundary header and the tools header together declare exactly what the library exports (no undeclared hooks), and every
    function carries argtypes derived from its prototype, so 64-bit arguments cannot be truncated by a call site passing a bare int."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("odise_hip_")}
    declared = set(_lib.header_symbols()) | set(_lib.header_symbols(_lib.TOOLS_HEADER_PATH))
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    # the measurement-only hooks (rate probes, yardstick kernels: include/odise_hip_lab.h) are NOT in the product library
    lab = set(_lib.header_symbols(_lib.LAB_HEADER_PATH))
    assert lab and not (lab & exported), sorted(lab & exported)
    protos = {**_lib.header_prototypes(), **_lib.header_prototypes(_lib.TOOLS_HEADER_PATH)}
    assert set(protos) == declared
    for name, args in protos.items():
        fn = getattr(lib, name)
        assert fn.argtypes is not None and len(fn.argtypes) == len(args), name


def test_wide_scalars_survive_marshalling(lib):
    # size_t / int64_t parameters take values beyond 32 bits from plain Python ints (null context: the call fails before using them)
    p = C.c_void_p()
    assert lib.odise_hip_malloc(None, 1 << 40, C.byref(p)) == -1
    conv = lib.odise_hip_malloc.argtypes[1].from_param(1 << 40)
    assert conv.value == 1 << 40
    conv = lib.odise_hip_allgather_predictions.argtypes[2].from_param(C.c_int64(1 << 35))
    assert conv.value == 1 << 35
