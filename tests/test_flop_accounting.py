"""bench.py's roofline numerators are algorithmic FLOPs of the live path: re-derive them with torch's FLOP counter over the full-size
oracle on the meta device (shapes only) - UNet single step (configs[1]) and the extractor per 512^2 crop, the bulk of configs[2]."""
import importlib.util
import os

import torch
from torch.utils.flop_counter import FlopCounterMode

from oracle.ldm_extractor import ImplicitCaptionerExtractor
from oracle.sd_unet import UNetModel, unet_forward

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_flop_constants_match_the_counted_graph():
    bench = _bench()
    with torch.device("meta"):
        unet = UNetModel()
        x, context, cond_emb, t = torch.empty(1, 4, 64, 64), torch.empty(1, 77, 768), torch.empty(1, 1280), torch.zeros(1, dtype=torch.long)
        with FlopCounterMode(display=False) as live, torch.no_grad():
            unet_forward(unet, x, t, context, cond_emb)
        with FlopCounterMode(display=False) as full, torch.no_grad():
            unet_forward(unet, x, t, context, cond_emb, run_dead_code=True)
        ext = ImplicitCaptionerExtractor()
        with FlopCounterMode(display=False) as crop, torch.no_grad():
            ext(torch.empty(1, 3, 512, 512))
    assert abs(live.get_total_flops() / bench.UNET_FLOPS_LIVE - 1) < 1e-3                     # 0.7401 TFLOP; the untapped tail is not counted
    assert full.get_total_flops() > live.get_total_flops() * 1.08                              # ... and it is 8.5 % of the launched graph
    per_crop = crop.get_total_flops()
    assert abs(per_crop / 2.8625e12 - 1) < 1e-3                                                # the oracle's extractor as the reference runs it ...
    dead = 2 * (576 * (1024 * 1024 + 2 * 1024 * 4096) + 576 * 577 * 64 * 16 * 2)              # ... incl. the patch-token rows of the last CLIP block (clip.py:196-206 reads x[:, 0] only)
    assert abs(dead / bench.CLIP_LAST_BLOCK_DEAD_FLOPS - 1) < 1e-3 and abs((per_crop - dead) / bench.CROP_FLOPS - 1) < 1e-3
    per_crop = per_crop - dead
    rest = 0.125e12 + 0.391e12 + 0.410e12 + 0.028e12                                            # projections, mask generator, classification, einsum (measured once, see bench.py)
    assert abs((4 * per_crop + rest) / bench.FLOPS_PER_IMAGE_1024 - 1) < 5e-3
    assert bench.MFMA_F16_PEAK == 2.5e15
