"""GPU: verify the MFMA 32x32x16 f16 fragment layout every kernel in libodise_hip.so assumes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_mfma_32x32x16_layout(ctx):
    out = np.zeros((3, 64, 16), dtype=np.float32)
    rc = ctx.lib.odise_hip_mfma_probe(ctx.h, out.ctypes.data_as(C.c_void_p))
    assert rc == 0, ctx.lib.odise_hip_last_error()
    lane = np.arange(64)[:, None]
    r = np.arange(16)[None, :]
    row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    col = np.broadcast_to(lane & 31, (64, 16))
    print("rows seen by lane 0:", out[0, 0], "lane 32:", out[0, 32])
    print("cols seen by lane 5:", out[1, 5])
    np.testing.assert_array_equal(out[0], row + 1)
    np.testing.assert_array_equal(out[1], col + 1)
    # k pairing: sum over both lane groups of sum_e 2^e * (8*hi+e+1)
    expect = sum((1 << e) * (8 * hi + e + 1) for hi in (0, 1) for e in range(8))
    np.testing.assert_array_equal(out[2], np.full((64, 16), expect, dtype=np.float32))
