"""The margin analysis the full-size GPU tests lean on (tests/margins.py, tests/contracts.py: test infrastructure) on small synthetic cases,
on the CPU: a segment table far from every threshold is declared decided, one whose `mask_area / original_area` sits next to the overlap
threshold (maskformer_model.py:321-327) is not; the launch-log diff matches launches by per-crop shape."""
import numpy as np
import torch

from contracts import launch_choice_diff
from margins import decided_labels, segments_decided


def _case(overlap_ratio):
    """Two queries on a 32x32 map: query 0 owns the left half outright; query 1 (lower score) claims columns 12..31, so it wins only where
    query 0 is absent: its mask_area / original_area = 16 / 20 = 0.8 for overlap_ratio = 0.8, adjustable through its extent."""
    H = W = 32
    K = 3
    logits = torch.full((3, H, W), -8.0)
    logits[0, :, :16] = 8.0
    start = 32 - int(round(16 / overlap_ratio))            # query 1 covers [start, 32): 16 columns of it are uncontested
    logits[1, :, start:] = 8.0
    probs = torch.full((3, K + 1), 0.01)
    probs[0, 0], probs[1, 2], probs[2, K] = 0.9, 0.7, 0.95  # query 0: thing class 0; query 1: stuff class 2; query 2: null
    probs = probs / probs.sum(-1, keepdim=True)
    return probs.log(), logits, K, {0, 1}


def test_a_table_far_from_the_thresholds_is_decided():
    lp, logits, K, things = _case(overlap_ratio=0.95)        # 16 / 17 columns kept: ratio 0.94 against the 0.8 threshold
    decided, differ = segments_decided(lp.numpy(), logits, K, things, np.full(3, 1e-2), np.full(3, 0.2), 0.8)
    assert decided and differ == 0


def test_a_table_next_to_the_overlap_threshold_is_not():
    lp, logits, K, things = _case(overlap_ratio=0.8)         # exactly at the threshold: any flipped boundary pixel changes the decision
    logits = logits.clone()
    logits[1, :, 11] = -0.05                                 # ... and the next column of query 1 sits inside the logit error: where noise switches it on, original_area grows past 16 / 0.8
    decided, differ = segments_decided(lp.numpy(), logits, K, things, np.full(3, 1e-2), np.full(3, 0.2), 0.8)
    assert not decided and differ > 0


def test_decided_labels_follow_the_per_query_error():
    p = np.array([[0.50, 0.45, 0.05], [0.90, 0.05, 0.05]])
    assert decided_labels(p, np.array([0.01, 0.01])).tolist() == [True, True]
    assert decided_labels(p, np.array([0.03, 0.03])).tolist() == [False, True]       # margin 0.05 <= 2 x 0.03


def test_launch_choice_diff_matches_per_crop_shapes(capsys):
    # (conv, M, N, K, tile, split): the same operator at 16 and 4 crops has M / crops in common; a per-image operator does not scale
    batch = np.array([[0, 16 * 584, 1024, 1024, 4, 1], [1, 16 * 4096, 320, 2880, 9, 1], [0, 400, 256, 256, 2, 1]], np.int32)
    alone = np.array([[0, 4 * 584, 1024, 1024, 2, 1], [1, 4 * 4096, 320, 2880, 9, 1], [0, 100, 256, 256, 2, 1]], np.int32)
    diff = launch_choice_diff(batch, 16, alone, 4)
    assert [(d[0][0], d[0][1]) for d in diff] == [(0, 584.0)] and diff[0][1] == [(4, 1)] and diff[0][2] == [(2, 1)]
    assert "differs on 1 shapes" in capsys.readouterr().out
