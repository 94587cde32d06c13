"""What does fp16 STORAGE alone cost the backbone features, stage by stage - and what would an fp32 residual stream buy?  

CPU only (no GPU, no library): the fp32 oracle of one 512 x 512 crop of the benchmarked picture (tests/fullsize.py weights) is run beside
copies of itself that round tensors to fp16 at the points where the device stores fp16 - weights of every GEMM / convolution, the input and
output of every Conv2d / Linear / GroupNorm / LayerNorm / attention, and the residual stream after every residual add - while keeping the
oracle's fp32 arithmetic inside each layer (the device accumulates in fp32).  Summation order aside, that is the device's number format; the
figures below are therefore the FLOOR of the fp16-storage design, and the policies say which storage decision each part of the error
belongs to:

    weights       only the GEMM / conv weights are fp16 (nothing any fp16-MFMA implementation can avoid)
    device        + every layer input / output and the residual stream in fp16 (what csrc/ does)
    stream32      as `device`, but the residual stream (block outputs, the convolutions that write it, the norm inputs that read it) stays fp32
    operands      only what an fp16 MFMA cannot avoid: weights and the A operand of every GEMM / convolution / attention rounded on the way in,
                  every tensor stored in fp32 (the floor of ANY fp16-MFMA implementation, at twice the activation bytes of `device`)

Per policy: every stage ALONE on the oracle's inputs (its own error), every stage alone with everything downstream in fp32 (its share of the
s2..s5 error), and the whole chain.  Errors are max |x - ref| / max |ref| (the figure the parity tests and VERDICT quote) and rms / rms.

    python tools/fp16_floor.py [seed=0] [crop=0]            (~6 min on 8 cores, ~20 GB)
    python tools/fp16_floor.py iou [seed=0]                 what the north star's "mask IoU within 1e-3" can be held to: the whole 1024 x 1024
                                                            picture's backbone under each policy -> the fp32 ORACLE head (an ideal head) -> per-query
                                                            raw IoU of the binary masks against the all-fp32 reference (~10 min)"""
import copy
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fullsize import FEATURE_DIMS, image_u8  # noqa: E402
from oracle import clip_vit, sd_unet, sd_vae  # noqa: E402
from oracle.backbone import BottleneckBlock, FeatureExtractorBackbone  # noqa: E402
from oracle.ldm_extractor import ImplicitCaptionerExtractor, q_sample_coeffs  # noqa: E402

LEAVES = (nn.Conv2d, nn.Linear, nn.GroupNorm, nn.LayerNorm, nn.MultiheadAttention)
NORMS = (nn.GroupNorm, nn.LayerNorm)
BLOCKS = (sd_unet.ResBlock, sd_unet.SpatialTransformer, sd_vae.ResnetBlock, sd_vae.AttnBlock, BottleneckBlock)
# layers whose output IS the residual stream (or is added to it in the same epilogue): with an fp32 stream they would write fp32
STREAM_WRITERS = ("out_layers.3", "skip_connection", "proj_out", "to_out.0", "ff.net.2", ".op", "upsample.conv", "input_blocks.0.0", ".conv2", "nin_shortcut",
                  "downsample.conv", "conv_in", "mlp.c_proj", ".attn", "conv3", "shortcut")


class Policy:
    leaf = False      # layer inputs / outputs fp16
    stream = False    # residual stream fp16
    operands = False  # only the operands of GEMM-like layers are rounded (on the way in)


POL = Policy()


def r16(t):
    return t.half().float() if torch.is_tensor(t) and t.dtype == torch.float32 else t


def install(root: nn.Module):
    """Hooks on a COPY of the oracle whose matrix weights were rounded to fp16; POL switches them at run time."""
    with torch.no_grad():
        for p in root.parameters():
            if p.ndim >= 2:
                p.copy_(r16(p))
    for name, m in root.named_modules():
        if isinstance(m, LEAVES):
            writer = any(s in "." + name for s in STREAM_WRITERS) and not isinstance(m, NORMS)
            is_norm = isinstance(m, NORMS)

            def pre(mod, args, is_norm=is_norm):
                if POL.operands:
                    return None if is_norm else tuple(r16(a) for a in args)
                if not POL.leaf or (is_norm and not POL.stream):
                    return None
                return tuple(r16(a) for a in args)

            def post(mod, args, out, writer=writer):
                if not POL.leaf or (writer and not POL.stream):
                    return None
                return tuple(r16(o) for o in out) if isinstance(out, tuple) else r16(out)

            m.register_forward_pre_hook(pre)
            m.register_forward_hook(post)
        elif isinstance(m, BLOCKS):
            m.register_forward_hook(lambda mod, args, out: r16(out) if POL.stream else None)

    def S(t):
        return r16(t) if POL.stream else t

    def btb(self, x, context=None):                      # sd_unet.BasicTransformerBlock.forward with the stream rounded after every add
        x = S(self.attn1(self.norm1(x)) + x)
        x = S(self.attn2(self.norm2(x), context=context) + x)
        return S(self.ff(self.norm3(x)) + x)

    def rab(self, x, attn_mask=None):                    # clip_vit.ResidualAttentionBlock.forward likewise
        x = S(x + self.attention(self.ln_1(x), attn_mask=attn_mask))
        return S(x + self.mlp(self.ln_2(x)))

    for m in root.modules():
        if isinstance(m, sd_unet.BasicTransformerBlock):
            m.forward = btb.__get__(m)
        elif isinstance(m, clip_vit.ResidualAttentionBlock):
            m.forward = rab.__get__(m)
    return root


def err(x, ref):
    x, ref = x.double(), ref.double()
    d = (x - ref)
    return float(d.abs().max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


def fmt(e):
    return f"{e[0]:.2e} (rms {e[1]:.2e})"


class Chain:
    """The backbone of one crop as five stages with explicit hand-over tensors."""

    def __init__(self, bb):
        self.bb, self.ext = bb, bb.feature_extractor

    def cond(self, img):
        return self.ext.conditioning(img)

    def enc(self, img):
        latent, feats = sd_vae.encode_to_latent(self.ext.vae, (img - 0.5) / 0.5)
        return latent, feats

    def unet(self, latent, cond_inputs, cond_emb):
        a, b = q_sample_coeffs(0)
        x_t = a * latent + b * self.ext.shared_noise.expand_as(latent)
        return sd_unet.unet_forward(self.ext.unet, x_t, torch.zeros(latent.shape[0], dtype=torch.long), cond_inputs, cond_emb[:, 0])[1]

    def dec(self, latent):
        return sd_vae.decode_to_image(self.ext.vae, latent)[1]

    def proj(self, taps):
        return self.bb.forward_features(taps, (512, 512))


def run(chains, img, emulated):
    """chains = (ref, emu); stage names in `emulated` run on the emulating copy, the others on the fp32 oracle."""
    ref, emu = chains
    pick = lambda s: emu if s in emulated else ref   # noqa: E731
    ci, ce = pick("cond").cond(img)
    latent, encf = pick("enc").enc(img)
    uf = pick("unet").unet(latent, ci, ce)
    df = pick("dec").dec(latent)
    feats = pick("proj").proj([*encf, *uf, *df])
    return dict(cond_inputs=ci, cond_emb=ce, latent=latent, enc5=encf[0], enc7=encf[1], u2=uf[0], u5=uf[1], u8=uf[2], u11=uf[3], dec2=df[0], dec5=df[1], **feats)


def iou_main(seed):
    """Per-query raw mask IoU an IDEAL (fp32) head reaches on backbone features computed under each storage policy."""
    from fullsize import build_models, features
    torch.set_num_threads(os.cpu_count())
    ext, bb, head = build_models()
    img, feats_ref = features(ext, bb, 1024, seed)
    emu_bb = install(copy.deepcopy(bb))
    img01 = img.float()[None] / 255.0
    with torch.no_grad():
        pm_ref = head(feats_ref)["pred_masks"][0]
        rb = pm_ref > 0
        print(f"picture {seed}: fp32 oracle head on backbone features computed under each policy, against the all-fp32 reference "
              f"(mask areas {float(rb.float().mean((1, 2)).min()):.3f} .. {float(rb.float().mean((1, 2)).max()):.3f} of the image)", flush=True)
        for pol, leaf, stream, operands in (("weights", False, False, False), ("operands", False, False, True), ("stream32", True, False, False),
                                            ("device", True, True, False)):
            POL.leaf, POL.stream, POL.operands = leaf, stream, operands
            f = emu_bb(img01)
            ferr = "  ".join(f"{k} {err(f[k], feats_ref[k])[0]:.2e}" for k in ("s2", "s3", "s4", "s5"))
            pm = head(f)["pred_masks"][0]
            gb = pm > 0
            iou = ((gb & rb).sum((1, 2)).double() / (gb | rb).sum((1, 2)).clamp(min=1).double()).numpy()
            qerr = ((pm - pm_ref).abs().amax((1, 2)) / pm_ref.abs().max()).numpy()
            print(f"  policy {pol:9s} features {ferr} | mask logits: worst query {qerr.max():.2e} median query {np.median(qerr):.2e} | raw IoU min {iou.min():.4f} "
                  f"median {np.median(iou):.5f} mean {iou.mean():.5f}  queries >= 1-1e-3: {int((iou >= 1 - 1e-3).sum())}/100", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "iou":
        return iou_main(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    crop = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    torch.set_num_threads(os.cpu_count())
    ext = ImplicitCaptionerExtractor()
    bb = FeatureExtractorBackbone(ext, FEATURE_DIMS)
    emu_bb = install(copy.deepcopy(bb))
    chains = (Chain(bb), Chain(emu_bb))
    y, x = (crop // 2) * 512, (crop % 2) * 512
    img = (image_u8(1024, 1024, seed).float()[None] / 255.0)[:, :, y:y + 512, x:x + 512].contiguous()
    with torch.no_grad():
        R = run(chains, img, ())
        print(f"fp32 oracle, picture {seed} crop {crop}; errors are max|x-ref|/max|ref| (rms/rms)", flush=True)
        outs = ("s2", "s3", "s4", "s5")
        stage_outs = {"cond": ("cond_inputs", "cond_emb"), "enc": ("latent", "enc5", "enc7"), "unet": ("u2", "u5", "u8", "u11"), "dec": ("dec2", "dec5"), "proj": outs}
        for pol, leaf, stream, operands in (("weights", False, False, False), ("operands", False, False, True), ("device", True, True, False),
                                            ("stream32", True, False, False)):
            POL.leaf, POL.stream, POL.operands = leaf, stream, operands
            print(f"\n== policy {pol} ==", flush=True)
            E = run(chains, img, ("cond", "enc", "unet", "dec", "proj"))
            print("  whole chain:  " + "  ".join(f"{k} {fmt(err(E[k], R[k]))}" for k in outs), flush=True)
            print("  whole chain, taps:  " + "  ".join(f"{k} {err(E[k], R[k])[0]:.2e}" for k in ("cond_inputs", "cond_emb", "latent", "enc5", "enc7", "u2", "u5", "u8", "u11", "dec2", "dec5")),
                  flush=True)
            for st in ("cond", "enc", "unet", "dec", "proj"):
                E = run(chains, img, (st,))
                own = "  ".join(f"{k} {fmt(err(E[k], R[k]))}" for k in stage_outs[st])
                share = "  ".join(f"{k} {err(E[k], R[k])[0]:.2e}" for k in outs)
                print(f"  stage {st:5s} alone on oracle inputs: {own}\n        its share of the feature error (downstream fp32): {share}", flush=True)


if __name__ == "__main__":
    main()
