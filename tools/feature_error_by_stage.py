"""The DEVICE's backbone error stage by stage, next to the fp16-storage floor tools/fp16_floor.py computes on the CPU.

One 1024 x 1024 picture of the full-size parity set-up (tests/fullsize.py), its four 512 x 512 windows in one call, against the fp32 oracle:
  * the 8 taps of LdmImplicitCaptionerExtractor (odise_hip_extractor_forward): enc5 / enc7 depend on the VAE encoder alone, u2..u11 carry the
    encoder's latent error, the conditioning's and the UNet's own, dec2 / dec5 the encoder's and the decoder's;
  * the UNet ALONE on the oracle's x_t / conditioning (odise_hip_unet_features): its own error;
  * the stitched s2..s5 maps of the whole backbone (odise_hip_backbone_forward).
Figures are max |x - ref| / max |ref| (rms / rms) like the CPU tool's, so the two files can be read side by side: a stage that sits at the
emulated `device` policy's figure is at the floor of fp16 storage - there is no kernel defect to find in it.

    python tools/feature_error_by_stage.py [seed=0]          (GPU; ~2 min of host oracle time)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fullsize import build_models, export_state, features, reference  # noqa: E402
from odise_amd._lib import check  # noqa: E402
from odise_amd.extractor import TAP_NAMES  # noqa: E402
from odise_amd.pipeline import HipCategoryODISE  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402
from oracle import sd_unet, sd_vae  # noqa: E402
from oracle.ldm_extractor import q_sample_coeffs  # noqa: E402


def err(x, ref):
    x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
    d = x - ref
    return np.abs(d).max() / np.abs(ref).max(), np.sqrt((d * d).mean()) / np.sqrt((ref * ref).mean())


def fmt(e):
    return f"{e[0]:.2e} (rms {e[1]:.2e})"


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    torch.set_num_threads(min(32, os.cpu_count()))
    ctx = Context(0)
    ext, bb, head = build_models()
    _, heads, _ = reference(bb, head, ext, 1024, 133, 254)
    hip = HipCategoryODISE(ctx, export_state(ext, bb, head, heads), overlap_threshold=0.8)
    img, feats_ref = features(ext, bb, 1024, seed)
    img01 = img.float()[None] / 255.0
    crops = torch.cat([img01[:, :, y:y + 512, x:x + 512] for y in (0, 512) for x in (0, 512)]).contiguous()
    with torch.no_grad():
        ci, ce = ext.conditioning(crops)
        latent, encf = sd_vae.encode_to_latent(ext.vae, (crops - 0.5) / 0.5)
        a, b = q_sample_coeffs(0)
        x_t = a * latent + b * ext.shared_noise.expand_as(latent)
        uf = sd_unet.unet_forward(ext.unet, x_t, torch.zeros(4, dtype=torch.long), ci, ce[:, 0])[1]
        df = sd_vae.decode_to_image(ext.vae, latent)[1]
    taps_ref = [*encf, *uf, *df]
    print(f"picture {seed}, 4 windows in one call; device against the fp32 oracle, max|x-ref|/max|ref| (rms/rms)", flush=True)
    for fold, tag in ((2, "CLIP LayerNorms as kernels"), (1, "CLIP LayerNorms folded into the GEMMs (what 16 crops run)")):
        ctx.set_option(ctx.OPT_CLIP_LN_FOLD, fold)
        # ---- the 8 taps of the extractor, whole chain on the device
        d_img = ctx.to_device(crops.numpy())
        outs = [ctx.empty(tuple(t.shape), np.float32) for t in taps_ref]
        arr = (C.c_void_p * 8)(*[o.ptr for o in outs])
        check(ctx.lib.odise_hip_extractor_forward(ctx.h, C.c_void_p(d_img.ptr), 4, 512, 512, arr), "extractor_forward")
        print(f"  [{tag}]", flush=True)
        print("  extractor taps (whole chain on the device):  " + "  ".join(f"{n} {fmt(err(o.numpy(), t.numpy()))}" for n, o, t in zip(TAP_NAMES, outs, taps_ref)), flush=True)
        for o in outs:
            o.free()
        # ---- the stitched maps
        got = hip.backbone(img01.numpy())
        print("  backbone maps (whole chain + projections + stitch):  " + "  ".join(f"{k} {fmt(err(got[k], feats_ref[k].numpy()))}" for k in ("s2", "s3", "s4", "s5")), flush=True)
    ctx.set_option(ctx.OPT_CLIP_LN_FOLD, 0)
    # ---- the UNet alone on the oracle's inputs
    dx, dc, de = ctx.to_device(x_t.numpy()), ctx.to_device(ci.numpy()), ctx.to_device(ce[:, 0].contiguous().numpy())
    outs = [ctx.empty(tuple(t.shape), np.float32) for t in uf]
    check(ctx.lib.odise_hip_unet_features(ctx.h, C.c_void_p(dx.ptr), C.c_void_p(dc.ptr), C.c_void_p(de.ptr), 4, 64, 64, 0, *[C.c_void_p(o.ptr) for o in outs]), "unet_features")
    print("  UNet ALONE on the oracle's x_t / conditioning:  " + "  ".join(f"{n} {fmt(err(o.numpy(), t.numpy()))}" for n, o, t in zip(TAP_NAMES[2:6], outs, uf)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
