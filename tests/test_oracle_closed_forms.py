"""CPU: closed-form / known-answer checks of the oracle restatements (SURVEY.md §8c "golden vectors available without weights")."""
import math

import numpy as np
import torch

from oracle import odise_model as om
from oracle.backbone import crop_boxes
from oracle.ldm_extractor import ldm_linear_alphas_cumprod, q_sample_coeffs, shared_noise
from oracle.m2f import EncoderOnly, position_embedding_sine
from oracle.sd_unet import timestep_embedding


def test_timestep_embedding_at_zero_is_ones_then_zeros():
    e = timestep_embedding(torch.zeros(2, dtype=torch.long), 320)
    assert e.shape == (2, 320)
    assert torch.equal(e[:, :160], torch.ones(2, 160)) and torch.equal(e[:, 160:], torch.zeros(2, 160))


def test_ldm_linear_schedule_and_q_sample_coefficients():
    ac = ldm_linear_alphas_cumprod()
    assert abs(ac[0] - (1 - 0.00085)) < 1e-12 and ac.shape == (1000,) and np.all(np.diff(ac) < 0)
    a, b = q_sample_coeffs(0)
    assert abs(a - math.sqrt(1 - 0.00085)) < 1e-7 and abs(b - math.sqrt(0.00085)) < 1e-7


def test_shared_noise_is_reproducible_cpu_generator_seed_42():
    n1, n2 = shared_noise(), shared_noise()
    assert n1.shape == (1, 4, 64, 64) and torch.equal(n1, n2)
    assert torch.equal(n1, torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(42)))


def test_slide_window_boxes():
    assert crop_boxes(1024, 1024) == [(0, 0, 512, 512), (0, 512, 512, 1024), (512, 0, 1024, 512), (512, 512, 1024, 1024)]
    b = crop_boxes(1280, 1280)
    assert len(b) == 9 and sorted({y for y, _, _, _ in b}) == [0, 512, 768]
    assert crop_boxes(512, 512) == [(0, 0, 512, 512)]


def test_position_embedding_sine_properties():
    pe = position_embedding_sine(4, 6, 8)
    assert pe.shape == (16, 4, 6)
    # y half depends only on the row, x half only on the column; first channel is sin(2*pi*(i+1)/(n+eps))
    assert torch.allclose(pe[:8, :, 0], pe[:8, :, 5]) and torch.allclose(pe[8:, 0, :], pe[8:, 3, :])
    assert torch.allclose(pe[0, :, 0], torch.sin(2 * math.pi * torch.arange(1, 5) / (4 + 1e-6)), atol=1e-6)


def test_reference_points_are_cell_centres():
    ref = EncoderOnly.get_reference_points(torch.tensor([[2, 4], [1, 2]]))
    assert ref.shape == (1, 10, 2, 2)
    np.testing.assert_allclose(ref[0, 0, 0].numpy(), [0.125, 0.25])      # (x, y) of the first cell of a 2x4 map
    np.testing.assert_allclose(ref[0, 9, 1].numpy(), [0.75, 0.5])        # last cell of the 1x2 map
    assert torch.equal(ref[..., 0, :], ref[..., 1, :])                   # same point for every sampled level (valid ratios = 1)


def test_ensemble_logits_max_over_synonyms():
    logits = torch.tensor([[1.0, 5.0, 2.0, -1.0, 7.0, 0.0]])
    out = om.ensemble_logits_with_labels(logits, [2, 1, 3])
    assert out.tolist() == [[5.0, 2.0, 7.0]]


def test_merge_with_null_sums_to_one():
    g = torch.Generator().manual_seed(0)
    pred = torch.randn(2, 5, 7, generator=g)
    merged = om.merge_with_null(pred, pred[..., :-1])
    np.testing.assert_allclose(torch.exp(merged).sum(-1).numpy(), 1.0, atol=1e-5)


def test_panoptic_inference_on_handmade_masks():
    # 3 queries on a 4x4 map: q0 'thing' class 0 on the left half, q1 'stuff' class 2 on the right half, q2 predicts null
    K = 3
    mask_cls = torch.full((3, K + 1), -10.0)
    mask_cls[0, 0], mask_cls[1, 2], mask_cls[2, K] = 10.0, 10.0, 10.0
    mask_pred = torch.full((3, 4, 4), -10.0)
    mask_pred[0, :, :2], mask_pred[1, :, 2:], mask_pred[2] = 10.0, 10.0, 10.0
    seg, info = om.panoptic_inference(mask_cls, mask_pred, K, thing_ids={0}, overlap_threshold=0.8)
    assert info == [{"id": 1, "isthing": True, "category_id": 0}, {"id": 2, "isthing": False, "category_id": 2}]
    assert (seg[:, :2] == 1).all() and (seg[:, 2:] == 2).all() and seg.dtype == torch.int32
    sem = om.semantic_inference(mask_cls, mask_pred)
    assert sem.shape == (K, 4, 4) and sem.argmax(0)[0, 0] == 0 and sem.argmax(0)[0, 3] == 2


def test_attention_mask_of_a_layer_from_resized_mask_features():
    """forward_prediction_heads (odise.py:756-765) resizes the prediction [Q, H/4, W/4] = mask_embed . mask_features to the next layer's level
    and thresholds it.  Both steps before the threshold are linear: resize(me . mf) = me . resize(mf) - what the library uses to compute the
    nine intermediate predictions at 1/4 .. 1/64 of the pixels (csrc/maskgen.cpp predictor_forward).  In fp64 the two orders agree to rounding,
    and so do the thresholded masks wherever the logit is not within rounding of zero; for exact 2x / 4x / 8x reductions the bilinear resize is
    the mean of the four centre pixels of each cell."""
    g = torch.Generator().manual_seed(7)
    me = torch.randn(2, 9, 16, generator=g, dtype=torch.float64)
    mf = torch.randn(2, 16, 32, 48, generator=g, dtype=torch.float64)
    full = torch.einsum("bqc,bchw->bqhw", me, mf)
    for f in (2, 4, 8):
        size = (32 // f, 48 // f)
        a = torch.nn.functional.interpolate(full, size=size, mode="bilinear", align_corners=False)
        mfl = torch.nn.functional.interpolate(mf, size=size, mode="bilinear", align_corners=False)
        b = torch.einsum("bqc,bchw->bqhw", me, mfl)
        assert (a - b).abs().max().item() < 1e-12 * full.abs().max().item()
        sure = a.abs() > 1e-9
        assert torch.equal((a.sigmoid() < 0.5)[sure], (b.sigmoid() < 0.5)[sure])
        c = f // 2 - 1   # the four centre pixels of an f x f cell
        mean4 = (mf[..., c::f, c::f] + mf[..., c + 1::f, c::f] + mf[..., c::f, c + 1::f] + mf[..., c + 1::f, c + 1::f]) / 4
        assert (mean4 - mfl).abs().max().item() < 1e-12
