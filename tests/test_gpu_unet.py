"""GPU parity of the SD-v1 UNet single-step tap extraction (odise_hip_unet_features) against the CPU oracle
(oracle/sd_unet.py, which follows LdmExtractor.unet_forward, odise/modeling/meta_arch/ldm.py:469-491).

Tolerance (SURVEY.md §8c tolerance contract, per-stage): fp16 MFMA path vs fp32 oracle on identical fp32 inputs and
weights: |err| <= 2e-2 * max|ref| element-wise and cosine similarity >= 0.9995 per tap."""
import numpy as np
import pytest
import torch

from odise_amd.unet import HipUNet
from oracle.sd_unet import UNetModel, config2_inputs, init_synthetic_, unet_forward

pytestmark = pytest.mark.gpu


def _check_taps(got, ref, what):
    assert len(got) == len(ref) == 4
    for i, (g, r) in enumerate(zip(got, ref)):
        r = r.numpy().astype(np.float64)
        g = g.astype(np.float64)
        assert g.shape == r.shape, (what, i, g.shape, r.shape)
        assert np.isfinite(g).all(), f"{what} tap {i}: non-finite values"
        scale = np.abs(r).max()
        err = np.abs(g - r).max() / scale
        cos = float((g * r).sum() / (np.linalg.norm(g) * np.linalg.norm(r)))
        print(f"{what} tap{i} shape {g.shape} max|ref| {scale:.3f} max-err/scale {err:.3e} cos {cos:.6f}")
        assert err <= 2e-2, f"{what} tap {i}: normalised max error {err:.3e}"
        assert cos >= 0.9995, f"{what} tap {i}: cosine {cos:.6f}"


@pytest.fixture(scope="module")
def small_unet(ctx):
    torch.manual_seed(0)
    model = init_synthetic_(UNetModel(width_div=5), seed=1234).eval()
    hip = HipUNet(ctx, {k: v for k, v in model.state_dict().items()})
    return model, hip


@pytest.mark.parametrize("batch,latent", [(1, 16), (2, 32), (3, 8)])
def test_small_unet_taps(small_unet, batch, latent):
    model, hip = small_unet
    x, context, cond_emb = config2_inputs(batch, latent, width_div=5)
    _, ref = unet_forward(model, x, torch.zeros(batch, dtype=torch.long), context, cond_emb)
    got = hip.features(x.numpy(), context.numpy(), cond_emb.numpy(), t=0)
    _check_taps(got, ref, f"small B{batch} L{latent}")
    assert hip.last_macs() > 0


def test_small_unet_without_cond_emb_and_nonzero_t(small_unet):
    model, hip = small_unet
    x, context, _ = config2_inputs(2, 16, width_div=5)
    _, ref = unet_forward(model, x, torch.full((2,), 37, dtype=torch.long), context, None)
    got = hip.features(x.numpy(), context.numpy(), None, t=37)
    _check_taps(got, ref, "small t=37 no cond_emb")


def test_small_unet_graph_replay_matches_eager(small_unet, ctx):
    _, hip = small_unet
    x, context, cond_emb = config2_inputs(2, 16, width_div=5)
    dx, dc, de = ctx.to_device(x.numpy()), ctx.to_device(context.numpy()), ctx.to_device(cond_emb.numpy())
    eager = [o.numpy() for o in hip.features_device(dx, dc, de)]
    hip.use_graph(True)
    try:
        for _ in range(3):
            replay = [o.numpy() for o in hip.features_device(dx, dc, de)]
            for a, b in zip(eager, replay):
                np.testing.assert_array_equal(a, b)
    finally:
        hip.use_graph(False)


def test_missing_weight_is_reported(ctx):
    model = UNetModel(width_div=5)
    sd = {k: v for k, v in model.state_dict().items() if k != "middle_block.1.norm.weight"}
    with pytest.raises(RuntimeError, match="middle_block.1.norm"):
        HipUNet(ctx, sd)


def test_full_width_unet_config2(ctx):
    """BASELINE configs[1]: SD-UNet single-step, bs=1, 512x512 crop (64x64 latent), full 859.5 M-parameter shapes."""
    model = init_synthetic_(UNetModel(width_div=1), seed=1234).eval()
    hip = HipUNet(ctx, model.state_dict())
    x, context, cond_emb = config2_inputs(1, 64)
    _, ref = unet_forward(model, x, torch.zeros(1, dtype=torch.long), context, cond_emb)
    got = hip.features(x.numpy(), context.numpy(), cond_emb.numpy(), t=0)
    _check_taps(got, ref, "full B1 L64")
    macs = hip.last_macs()
    # analytic live work of SURVEY.md §8d: 370.06 GMAC (401.63 total minus the dead output_blocks[11] + out)
    assert abs(macs / 370.06e9 - 1.0) < 0.02, macs
