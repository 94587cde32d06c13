"""Golden fixtures for the mask generator, produced by the REFERENCE's own modules (build container only).

Runs `MaskFormerHead` (third_party/Mask2Former/mask2former/modeling/meta_arch/mask_former_head.py) holding the real
`MSDeformAttnPixelDecoder` (pixel_decoder/msdeformattn.py, with `MSDeformAttn` on its pure-PyTorch core) and ODISE's
`ODISEMultiScaleMaskedTransformerDecoder` + `PooledMaskEmbed` + `MaskPooling` + `PseudoClassEmbed` (odise/modeling/meta_arch/odise.py:620-1015),
built with the keyword arguments of configs/common/models/mask_generator_with_label.py at reduced widths, loaded with the weights of
the oracle's seeded `SemSegHead(small=True)` (the oracle keeps the reference's parameter names, so `load_state_dict(strict=True)`
checks the structure too), on seeded s2..s5 feature maps.  Inputs and every output the hot path consumes are written to
tests/golden/m2f_head_*.npz; tests/test_oracle_golden.py replays them through oracle/m2f.py.  Third-party imports are handled by
tests/golden/ref_stubs.py (nothing but the helpers restated there can execute).

    python tests/golden/make_golden_m2f.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_stubs  # noqa: E402

ref_stubs.install()
from mask2former.modeling.meta_arch.mask_former_head import MaskFormerHead  # noqa: E402
from mask2former.modeling.pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder  # noqa: E402
from odise.modeling.meta_arch.odise import ODISEMultiScaleMaskedTransformerDecoder, PooledMaskEmbed, PseudoClassEmbed  # noqa: E402

ref_stubs.seal()
from oracle.m2f import SemSegHead, init_synthetic_  # noqa: E402


def reference_head(num_classes, in_channels, conv_dim, mask_dim, enc_layers, d_ffn_enc, hidden, queries, d_ffn_dec, dec_layers, learned_class_embed=False):
    shapes = {f"s{i}": ref_stubs.ShapeSpec(channels=in_channels, stride=2 ** i) for i in (2, 3, 4, 5)}
    pixel_decoder = MSDeformAttnPixelDecoder(shapes, conv_dim=conv_dim, mask_dim=mask_dim, norm="GN", transformer_dropout=0.0, transformer_nheads=8,
                                             transformer_dim_feedforward=d_ffn_enc, transformer_enc_layers=enc_layers,
                                             transformer_in_features=["s3", "s4", "s5"], common_stride=4)
    predictor = ODISEMultiScaleMaskedTransformerDecoder(
        class_embed=None if learned_class_embed else PseudoClassEmbed(num_classes=num_classes), hidden_dim=hidden,   # None: the decoder's own Linear(hidden, num_classes + 1)
        post_mask_embed=PooledMaskEmbed(hidden_dim=hidden, mask_dim=mask_dim, projection_dim=mask_dim), in_channels=conv_dim,
        mask_classification=True, num_classes=num_classes, num_queries=queries, nheads=8, dim_feedforward=d_ffn_dec, dec_layers=dec_layers,
        pre_norm=False, enforce_input_project=False, mask_dim=mask_dim)
    return MaskFormerHead(shapes, ignore_value=255, num_classes=num_classes, pixel_decoder=pixel_decoder, loss_weight=1.0,
                          transformer_in_feature="multi_scale_pixel_decoder", transformer_predictor=predictor).eval()


def case(name, seed, B, H4, W4, num_classes=8, in_channels=96):
    oracle = init_synthetic_(SemSegHead(small=True, num_classes=num_classes, in_channels=in_channels), seed=seed)
    ref = reference_head(num_classes, in_channels, 64, 64, 2, 128, 64, 20, 128, 3)
    missing = ref.load_state_dict(oracle.state_dict(), strict=True)
    g = torch.Generator().manual_seed(seed + 1)
    feats = {f"s{i}": torch.randn(B, in_channels, H4 >> (i - 2), W4 >> (i - 2), generator=g) for i in (2, 3, 4, 5)}
    with torch.no_grad():
        mask_features, enc, multi_scale = ref.pixel_decoder.forward_features(feats)
        out = ref(feats)
    keep = ("pred_logits", "pred_masks", "mask_embed", "mask_pred_logits") if "mask_embed" in out else tuple(k for k in out if torch.is_tensor(out[k]))
    print(name, "reference outputs:", {k: tuple(v.shape) for k, v in out.items() if torch.is_tensor(v)}, missing)
    arrays = {"in_" + k: v.numpy() for k, v in feats.items()}
    arrays.update({"out_" + k: out[k].numpy() for k in out if torch.is_tensor(out[k])})
    arrays.update(out_mask_features=mask_features.numpy(), **{f"out_multi_scale_{i}": m.numpy() for i, m in enumerate(multi_scale)})
    arrays.update(seed=np.int64(seed), num_classes=np.int64(num_classes), in_channels=np.int64(in_channels))
    np.savez_compressed(os.path.join(HERE, f"m2f_head_{name}.npz"), **arrays)


if __name__ == "__main__":
    case("a", seed=1234, B=1, H4=32, W4=32, in_channels=32)
    case("b", seed=99, B=2, H4=32, W4=40, num_classes=5, in_channels=16)
