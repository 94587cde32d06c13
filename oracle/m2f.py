"""Oracle: Mask2Former pixel decoder + ODISE masked transformer decoder (TEST INFRASTRUCTURE — see oracle/__init__.py).

Mask2Former IS vendored in the reference, so these follow its files line by line (module / parameter names identical to
the checkpoint keys `sem_seg_head.pixel_decoder.*`, `sem_seg_head.predictor.*`):
  PositionEmbeddingSine.forward                 M2F/modeling/transformer_decoder/position_encoding.py:29-52
  MSDeformAttn.forward                          M2F/modeling/pixel_decoder/ops/modules/ms_deform_attn.py:82-125
  MSDeformAttnTransformerEncoder(-Layer/-Only)  M2F/modeling/pixel_decoder/msdeformattn.py:61-158
  MSDeformAttnPixelDecoder.forward_features     M2F/modeling/pixel_decoder/msdeformattn.py:314-358
  Self/Cross attention, FFN, MLP layers         M2F/modeling/transformer_decoder/mask2former_transformer_decoder.py:17-204
  ODISEMultiScaleMaskedTransformerDecoder       odise/modeling/meta_arch/odise.py:642-776
  PseudoClassEmbed / MaskPooling / PooledMaskEmbed   odise/modeling/meta_arch/odise.py:910-1015
(M2F = third_party/Mask2Former/mask2former).  The sampling core is oracle.msda.msda_forward_torch, which is pinned to the
reference's ms_deform_attn_core_pytorch by tests/golden.  PINNED as a whole: tests/test_oracle_golden.py replays golden vectors
written by the reference's own MaskFormerHead (tests/golden/make_golden_m2f.py).  Hyper-parameters: configs/common/models/mask_generator_with_label.py.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .msda import msda_forward_torch


def position_embedding_sine(h: int, w: int, num_pos_feats: int = 128, temperature: float = 10000.0) -> torch.Tensor:
    """position_encoding.py:29-52 with normalize=True, scale=2*pi, mask=None -> [2*num_pos_feats, h, w]."""
    scale = 2 * math.pi
    y_embed = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).expand(h, w)
    x_embed = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).expand(h, w)
    eps = 1e-6
    y_embed = y_embed / (y_embed[-1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    pos_x = x_embed[:, :, None] / dim_t
    pos_y = y_embed[:, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).flatten(2)
    pos_y = torch.stack((pos_y[:, :, 0::2].sin(), pos_y[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((pos_y, pos_x), dim=2).permute(2, 0, 1)


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=3, n_heads=8, n_points=4):
        super().__init__()
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)

    def forward(self, query, reference_points, input_flatten, spatial_shapes, level_start_index):
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        value = self.value_proj(input_flatten).view(N, Len_in, self.n_heads, self.d_model // self.n_heads)       # :99-102
        so = self.sampling_offsets(query).view(N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)          # :103
        aw = self.attention_weights(query).view(N, Len_q, self.n_heads, self.n_levels * self.n_points)           # :104
        aw = F.softmax(aw, -1).view(N, Len_q, self.n_heads, self.n_levels, self.n_points)                        # :105
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1).to(query.dtype)           # :108
        loc = reference_points[:, :, None, :, None, :] + so / normalizer[None, None, None, :, None, :]           # :109-110
        out = msda_forward_torch(value, spatial_shapes, level_start_index, loc, aw)                               # :114-121
        return self.output_proj(out)                                                                              # :123


class EncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, n_levels=3, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index):
        src2 = self.self_attn(src + pos, reference_points, src, spatial_shapes, level_start_index)               # msdeformattn.py:121
        src = self.norm1(src + src2)
        src2 = self.linear2(F.relu(self.linear1(src)))
        return self.norm2(src + src2)


class _Encoder(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)


class EncoderOnly(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_layers=6, d_ffn=1024, n_levels=3, n_points=4):
        super().__init__()
        self.encoder = _Encoder([EncoderLayer(d_model, d_ffn, n_levels, nhead, n_points) for _ in range(num_layers)])
        self.level_embed = nn.Parameter(torch.zeros(n_levels, d_model))

    @staticmethod
    def get_reference_points(spatial_shapes):                                                                     # :141-153, valid_ratios == 1
        refs = []
        for (H_, W_) in spatial_shapes.tolist():
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_), torch.linspace(0.5, W_ - 0.5, W_), indexing="ij")
            refs.append(torch.stack((ref_x.reshape(-1) / W_, ref_y.reshape(-1) / H_), -1))
        ref = torch.cat(refs, 0)[None]                      # [1, Lq, 2]
        return ref[:, :, None].repeat(1, 1, len(spatial_shapes), 1)

    def forward(self, srcs: List[torch.Tensor], pos_embeds: List[torch.Tensor]):
        src_flatten, pos_flatten, shapes = [], [], []
        for lvl, (src, pos) in enumerate(zip(srcs, pos_embeds)):                                                 # :64-79
            shapes.append(src.shape[-2:])
            src_flatten.append(src.flatten(2).transpose(1, 2))
            pos_flatten.append(pos.flatten(2).transpose(1, 2) + self.level_embed[lvl].view(1, 1, -1))
        src = torch.cat(src_flatten, 1)
        pos = torch.cat(pos_flatten, 1)
        spatial_shapes = torch.as_tensor([list(s) for s in shapes], dtype=torch.long)
        level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
        reference_points = self.get_reference_points(spatial_shapes).expand(src.shape[0], -1, -1, -1)
        out = src
        for layer in self.encoder.layers:
            out = layer(out, pos, reference_points, spatial_shapes, level_start_index)
        return out, spatial_shapes, level_start_index


class _ConvGN(nn.Conv2d):
    def __init__(self, cin, cout, k, padding=0, relu=False):
        super().__init__(cin, cout, k, padding=padding, bias=False)
        self.norm = nn.GroupNorm(32, cout)
        self._relu = relu

    def forward(self, x):
        x = self.norm(super().forward(x))
        return F.relu(x) if self._relu else x


class PixelDecoder(nn.Module):
    """MSDeformAttnPixelDecoder with transformer_in_features [s3,s4,s5], FPN level on s2, conv_dim = mask_dim = 256, norm GN."""

    def __init__(self, in_channels=512, conv_dim=256, mask_dim=256, enc_layers=6, d_ffn=1024, nheads=8):
        super().__init__()
        self.input_proj = nn.ModuleList([nn.Sequential(nn.Conv2d(in_channels, conv_dim, 1), nn.GroupNorm(32, conv_dim)) for _ in range(3)])
        self.transformer = EncoderOnly(conv_dim, nheads, enc_layers, d_ffn, 3, 4)
        self.mask_features = nn.Conv2d(conv_dim, mask_dim, 1)
        self.adapter_1 = _ConvGN(in_channels, conv_dim, 1)
        self.layer_1 = _ConvGN(conv_dim, conv_dim, 3, padding=1, relu=True)
        self.conv_dim = conv_dim

    @torch.no_grad()
    def forward_features(self, features: Dict[str, torch.Tensor]):                                                # :314-358
        srcs, pos = [], []
        for idx, f in enumerate(["s5", "s4", "s3"]):                    # transformer_in_features reversed (low -> high resolution)
            x = features[f].float()
            srcs.append(self.input_proj[idx](x))
            pos.append(position_embedding_sine(x.shape[-2], x.shape[-1], self.conv_dim // 2)[None].expand(x.shape[0], -1, -1, -1))
        y, spatial_shapes, level_start_index = self.transformer(srcs, pos)
        bs = y.shape[0]
        sizes = [int(h * w) for h, w in spatial_shapes.tolist()]
        out = [z.transpose(1, 2).reshape(bs, -1, int(spatial_shapes[i][0]), int(spatial_shapes[i][1])) for i, z in enumerate(torch.split(y, sizes, dim=1))]
        x = features["s2"].float()
        cur_fpn = self.adapter_1(x)
        y2 = cur_fpn + F.interpolate(out[-1], size=cur_fpn.shape[-2:], mode="bilinear", align_corners=False)
        out.append(self.layer_1(y2))
        return self.mask_features(out[-1]), out[0], out[:3]


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = F.relu(layer(x)) if i < self.num_layers - 1 else layer(x)
        return x


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=0.0)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt, query_pos):
        q = k = tgt + query_pos
        return self.norm(tgt + self.self_attn(q, k, value=tgt)[0])


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=0.0)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt, memory, memory_mask, pos, query_pos):
        tgt2 = self.multihead_attn(query=tgt + query_pos, key=memory + pos, value=memory, attn_mask=memory_mask)[0]
        return self.norm(tgt + tgt2)


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt):
        return self.norm(tgt + self.linear2(F.relu(self.linear1(tgt))))


class MaskPooling(nn.Module):
    def forward(self, x, mask):                                                                                   # odise.py:937-963
        assert x.shape[-2:] == mask.shape[-2:]
        mask = mask.detach().sigmoid()
        mask = (mask > 0.5).to(mask.dtype)
        denorm = mask.sum(dim=(-1, -2), keepdim=True) + 1e-8
        return {"mask_pooled_features": torch.einsum("bchw,bqhw->bqc", x, mask / denorm)}


class PooledMaskEmbed(nn.Module):
    def __init__(self, hidden_dim=256, mask_dim=256, projection_dim=256, temperature=0.07):
        super().__init__()
        self.pool_proj = nn.Sequential(nn.LayerNorm(hidden_dim), nn.Linear(hidden_dim, hidden_dim))
        self.mask_embed = nn.Sequential(nn.LayerNorm(mask_dim), MLP(mask_dim, hidden_dim, projection_dim, 3))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / temperature))
        self.mask_pooling = MaskPooling()

    def forward(self, decoder_output, input_mask_embed, mask_features, pred_logits, pred_masks):                  # odise.py:984-1015
        x = self.mask_pooling(mask_features, pred_masks)["mask_pooled_features"]
        x = self.pool_proj(x)
        x = x + decoder_output                                                                                    # :1000 (in-place += on a fresh tensor)
        return {"mask_embed": self.mask_embed(x), "mask_pooled_features": x, "logit_scale": torch.clamp(self.logit_scale.exp(), max=100)}


class MaskedTransformerDecoder(nn.Module):
    def __init__(self, hidden_dim=256, num_queries=100, nheads=8, dim_feedforward=2048, dec_layers=9, mask_dim=256, num_classes=133,
                 learned_class_embed=False):
        super().__init__()
        self.num_heads, self.num_layers, self.num_feature_levels, self.num_classes = nheads, dec_layers, 3, num_classes
        # CaptionODISE keeps Mask2Former's own nn.Linear(hidden_dim, num_classes + 1) with num_classes = 1 (object / no-object,
        # mask2former_transformer_decoder.py:333, configs/common/models/mask_generator_with_caption.py:4); CategoryODISE replaces it
        # with the parameter-free PseudoClassEmbed
        self.class_embed = nn.Linear(hidden_dim, num_classes + 1) if learned_class_embed else None
        self.hidden_dim = hidden_dim
        self.transformer_self_attention_layers = nn.ModuleList(SelfAttentionLayer(hidden_dim, nheads) for _ in range(dec_layers))
        self.transformer_cross_attention_layers = nn.ModuleList(CrossAttentionLayer(hidden_dim, nheads) for _ in range(dec_layers))
        self.transformer_ffn_layers = nn.ModuleList(FFNLayer(hidden_dim, dim_feedforward) for _ in range(dec_layers))
        self.decoder_norm = nn.LayerNorm(hidden_dim)
        self.query_feat = nn.Embedding(num_queries, hidden_dim)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.level_embed = nn.Embedding(3, hidden_dim)
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)
        self.post_mask_embed = PooledMaskEmbed(hidden_dim, mask_dim, mask_dim)

    def forward_prediction_heads(self, output, mask_features, attn_mask_target_size):                             # odise.py:729-776
        decoder_output = self.decoder_norm(output).transpose(0, 1)
        if self.class_embed is not None:
            outputs_class = self.class_embed(decoder_output)                                                      # :734
        else:
            fg = torch.ones((*decoder_output.shape[:-1], self.num_classes))
            outputs_class = torch.cat([fg, torch.zeros((*decoder_output.shape[:-1], 1))], dim=-1)                 # PseudoClassEmbed :910-920
        mask_embed = self.mask_embed(decoder_output)
        outputs_mask = torch.einsum("bqc,bchw->bqhw", mask_embed, mask_features)
        extra = self.post_mask_embed(decoder_output, mask_embed, mask_features, outputs_class, outputs_mask)
        attn_mask = F.interpolate(outputs_mask, size=attn_mask_target_size, mode="bilinear", align_corners=False)
        attn_mask = (attn_mask.sigmoid().flatten(2).unsqueeze(1).repeat(1, self.num_heads, 1, 1).flatten(0, 1) < 0.5).bool()
        return outputs_class, outputs_mask, attn_mask, extra

    @torch.no_grad()
    def forward(self, x: List[torch.Tensor], mask_features: torch.Tensor):                                        # odise.py:642-727
        src, pos, size_list = [], [], []
        for i in range(3):
            size_list.append(x[i].shape[-2:])
            p = position_embedding_sine(x[i].shape[-2], x[i].shape[-1], self.hidden_dim // 2)[None].expand(x[i].shape[0], -1, -1, -1)
            pos.append(p.flatten(2).permute(2, 0, 1))
            src.append((x[i].flatten(2) + self.level_embed.weight[i][None, :, None]).permute(2, 0, 1))            # input_proj = identity
        bs = src[0].shape[1]
        query_embed = self.query_embed.weight.unsqueeze(1).repeat(1, bs, 1)
        output = self.query_feat.weight.unsqueeze(1).repeat(1, bs, 1)
        outputs_class, outputs_mask, attn_mask, extra = self.forward_prediction_heads(output, mask_features, size_list[0])
        for i in range(self.num_layers):
            lvl = i % 3
            attn_mask[torch.where(attn_mask.sum(-1) == attn_mask.shape[-1])] = False                              # :683
            output = self.transformer_cross_attention_layers[i](output, src[lvl], attn_mask, pos[lvl], query_embed)
            output = self.transformer_self_attention_layers[i](output, query_embed)
            output = self.transformer_ffn_layers[i](output)
            outputs_class, outputs_mask, attn_mask, extra = self.forward_prediction_heads(output, mask_features, size_list[(i + 1) % 3])
        out = {"pred_logits": outputs_class, "pred_masks": outputs_mask}
        out.update(extra)
        return out


class SemSegHead(nn.Module):
    """MaskFormerHead.layers (M2F/modeling/meta_arch/mask_former_head.py:118-132): pixel decoder -> predictor."""

    def __init__(self, num_classes=133, in_channels=512, small=False, learned_class_embed=False):
        super().__init__()
        if small:
            self.pixel_decoder = PixelDecoder(in_channels, conv_dim=64, mask_dim=64, enc_layers=2, d_ffn=128, nheads=8)
            self.predictor = MaskedTransformerDecoder(64, 20, 8, 128, 3, 64, num_classes, learned_class_embed=learned_class_embed)
        else:
            self.pixel_decoder = PixelDecoder(in_channels)
            self.predictor = MaskedTransformerDecoder(num_classes=num_classes, learned_class_embed=learned_class_embed)

    @torch.no_grad()
    def forward(self, features):
        mask_features, _enc, multi_scale = self.pixel_decoder.forward_features(features)
        return self.predictor(multi_scale, mask_features)


def init_synthetic_(model: nn.Module, seed: int = 777, branch_gain: float = 1.0, qk_gain: float = 1.0, level_gain: float = 1.0) -> nn.Module:
    """Seeded weights.  With the defaults the 100 queries of the full-size decoder collapse onto one mask: every attention sublayer adds
    (nearly) the same vector to all queries - W_v . mean(src), as large as the query itself - and the post-norm halves their distinct part,
    18 times.  `branch_gain` < 1 scales the residual branches of the masked decoder (attention out_proj and FFN linear2 weights), the
    regime of a trained network whose branches refine the query instead of replacing it: queries stay distinct (tests/fullsize.py uses
    0.3).  `qk_gain` > 1 sharpens the decoder's attention instead and `level_gain` shrinks its level embedding; that also separates the
    queries but makes the fp32 model itself chaotic - rounding weights and inputs to fp16 moves its mask logits by 22 % at qk_gain 4
    (tools/oracle_sensitivity.py) - so it is not used for parity.  The random draws are the same for every gain."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            if name.endswith("logit_scale"):
                continue
            if "sampling_offsets.bias" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 2.0)       # the reference initialises a spread of offsets (ms_deform_attn.py:67-75)
            elif "sampling_offsets.weight" in name:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / math.sqrt(p.shape[1])))
            elif p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / math.sqrt(p[0].numel())))
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        # embeddings O(1)
        for emb in (model.predictor.query_feat, model.predictor.query_embed, model.predictor.level_embed):
            emb.weight.copy_(torch.randn(emb.weight.shape, generator=g))
        model.pixel_decoder.transformer.level_embed.copy_(torch.randn(model.pixel_decoder.transformer.level_embed.shape, generator=g))
        for name, p in model.predictor.named_parameters():
            if qk_gain != 1.0 and name.endswith("in_proj_weight"):
                p[: 2 * p.shape[1]].mul_(qk_gain)
            if branch_gain != 1.0 and (name.endswith("out_proj.weight") or name.endswith("linear2.weight")):
                p.mul_(branch_gain)
        if level_gain != 1.0:
            model.predictor.level_embed.weight.mul_(level_gain)
    return model.eval()
