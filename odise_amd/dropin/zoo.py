"""The released label model spelled out with the overlay's classes, argument by argument as configs/common/models/odise_with_label.py and
mask_generator_with_label.py instantiate it (tests/test_dropin_cpu.py checks that the reference's own LazyConfig files produce exactly this
through `instantiate_odise`).  For callers without the reference checkout / detectron2's LazyConfig: the GPU tests and tools/replay.py."""
from __future__ import annotations

import sys

from . import OVERLAY_DIR


def category_odise_with_label(labels, thing_ids, overlapping=None, overlap_threshold: float = 0.8, alpha: float = 0.3, beta: float = 0.7):
    """-> odise.modeling.meta_arch.odise.CategoryODISE (overlay).  `labels`: list of synonym lists; `overlapping[k]`: class k is a training
    class (PoolingCLIPHead.train_labels); frozen weights come from `dropin.set_frozen_state` / the checkpoint files."""
    if OVERLAY_DIR not in sys.path:
        sys.path.insert(0, OVERLAY_DIR)
    from mask2former.modeling.meta_arch.mask_former_head import MaskFormerHead
    from mask2former.modeling.pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
    from odise.modeling.backbone.feature_extractor import FeatureExtractorBackbone
    from odise.modeling.meta_arch.ldm import LdmImplicitCaptionerExtractor
    from odise.modeling.meta_arch.odise import (CategoryEmbed, CategoryODISE, ODISEMultiScaleMaskedTransformerDecoder, PooledMaskEmbed, PoolingCLIPHead,
                                                PseudoClassEmbed)
    backbone = FeatureExtractorBackbone(
        feature_extractor=LdmImplicitCaptionerExtractor(encoder_block_indices=(5, 7), unet_block_indices=(2, 5, 8, 11), decoder_block_indices=(2, 5), steps=(0,),
                                                        learnable_time_embed=True, num_timesteps=1, clip_model_name="ViT-L-14-336"),
        out_features=["s2", "s3", "s4", "s5"], use_checkpoint=True, slide_training=True)
    shape = backbone.output_shape()
    k = len(labels)
    train_labels = list(labels) if overlapping is None else [l for l, o in zip(labels, overlapping) if o]
    return CategoryODISE(
        backbone=backbone,
        sem_seg_head=MaskFormerHead(
            shape, ignore_value=255, num_classes=k,
            pixel_decoder=MSDeformAttnPixelDecoder(shape, conv_dim=256, mask_dim=256, norm="GN", transformer_dropout=0.0, transformer_nheads=8,
                                                   transformer_dim_feedforward=1024, transformer_enc_layers=6, transformer_in_features=["s3", "s4", "s5"],
                                                   common_stride=4),
            loss_weight=1.0, transformer_in_feature="multi_scale_pixel_decoder",
            transformer_predictor=ODISEMultiScaleMaskedTransformerDecoder(
                class_embed=PseudoClassEmbed(num_classes=k), hidden_dim=256,
                post_mask_embed=PooledMaskEmbed(hidden_dim=256, mask_dim=256, projection_dim=256), in_channels=256, mask_classification=True,
                num_classes=k, num_queries=100, nheads=8, dim_feedforward=2048, dec_layers=9, pre_norm=False, enforce_input_project=False, mask_dim=256)),
        criterion=None,
        category_head=CategoryEmbed(clip_model_name="ViT-L-14-336", labels=labels, projection_dim=256),
        clip_head=PoolingCLIPHead(alpha=alpha, beta=beta, train_labels=train_labels),
        num_queries=100, object_mask_threshold=0.0, overlap_threshold=overlap_threshold, metadata={"thing_ids": list(thing_ids)}, size_divisibility=64,
        sem_seg_postprocess_before_inference=True, pixel_mean=[0.0, 0.0, 0.0], pixel_std=[255.0, 255.0, 255.0], semantic_on=True, instance_on=True,
        panoptic_on=True, test_topk_per_image=100)


OWN_PREFIXES = ("backbone.feature_projections.", "backbone.feature_extractor.clip_project", "backbone.feature_extractor.alpha",
                "backbone.feature_extractor.time_embed_project", "sem_seg_head.", "category_head.")
FROZEN_PREFIXES = ("model.diffusion_model.", "first_stage_model.", "clip.", "backbone.feature_extractor.ldm_extractor.")


def load_flat_state(model, state) -> None:
    """Split a flat {key: array} state (odise_amd.synthetic.synthetic_state / tests/fullsize.export_state) the way the real files are split: the
    frozen Stable-Diffusion / CLIP towers go to `dropin.set_frozen_state`, the trainable part through `load_state_dict(strict=True)`."""
    import numpy as np
    import torch
    from . import set_frozen_state
    set_frozen_state({k: v for k, v in state.items() if k.startswith(FROZEN_PREFIXES)})
    own = {k: torch.as_tensor(np.asarray(v)) for k, v in state.items() if k.startswith(OWN_PREFIXES)}
    model.load_state_dict(own, strict=True)
