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
