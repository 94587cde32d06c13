"""CPU: the MSDeformAttn oracle restatements vs the golden vectors produced by the reference's own
ms_deform_attn_core_pytorch (tests/golden/make_golden.py), incl. the reference's ops/test.py case."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle.msda import make_inputs, msda_forward_loops, msda_forward_torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "msda_*.npz")))


def test_golden_files_present():
    names = {os.path.basename(f) for f in FILES}
    assert {"msda_ops_test_f64.npz", "msda_ops_test_f32.npz", "msda_oob.npz", "msda_odd_d.npz", "msda_d32_3lvl.npz"} <= names


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_torch_restatement_matches_reference(path):
    g = np.load(path)
    out = msda_forward_torch(torch.from_numpy(g["value"]).double(), g["shapes"], g["start"], torch.from_numpy(g["loc"]).double(),
                             torch.from_numpy(g["w"]).double()).numpy()
    # reference ops/test.py:41 uses torch.allclose defaults in fp64; fp32 golden gets its fp32 tolerance (:57)
    tol = dict(rtol=1e-2, atol=1e-3) if "f32" in path else dict(rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(out, g["out"].astype(np.float64), **tol)


@pytest.mark.parametrize("name", ["msda_ops_test_f64.npz", "msda_oob.npz", "msda_odd_d.npz"])
def test_loop_restatement_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, name))
    out = msda_forward_loops(g["value"], g["shapes"], g["start"], g["loc"], g["w"])
    np.testing.assert_allclose(out, g["out"].astype(np.float64), rtol=1e-5, atol=1e-8)


def test_two_restatements_agree_on_production_like_shape():
    value, shp, start, loc, w = make_inputs(1, 8, 32, 64, [(8, 8), (16, 16), (32, 32)], 4, seed=5, loc_range=(-0.2, 1.2))
    a = msda_forward_torch(value.double(), shp, start, loc, w).numpy()
    b = msda_forward_loops(value.numpy(), shp.numpy(), start.numpy(), loc.numpy(), w.numpy())
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)


def test_linearity_in_value():
    # size-independent property used at full size on the GPU: the op is linear in `value`
    value, shp, start, loc, w = make_inputs(1, 4, 8, 33, [(7, 9), (4, 5)], 4, seed=6)
    value, v2 = value.double(), torch.rand_like(value).double()
    f = lambda v: msda_forward_torch(v, shp, start, loc, w)
    np.testing.assert_allclose(f(value + 3 * v2).numpy(), (f(value) + 3 * f(v2)).numpy(), rtol=1e-10, atol=1e-12)


def test_constant_value_gives_weight_sum_inside():
    # with value == 1 and every sample strictly inside, out = sum of weights = 1 (weights normalised over L*P)
    value, shp, start, loc, w = make_inputs(1, 2, 4, 10, [(8, 8)], 4, seed=7, loc_range=(0.2, 0.8))
    out = msda_forward_torch(torch.ones_like(value).double(), shp, start, loc, w)
    np.testing.assert_allclose(out.numpy(), 1.0, rtol=1e-6)
