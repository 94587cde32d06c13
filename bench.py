#!/usr/bin/env python3
"""bench.py — hot-path benchmark of the MI355X-native ODISE inference path.

Default workload (BASELINE.json configs[2], the configuration the metric "panoptic-inference images/sec @1024x1024" is quoted on;
SURVEY.md 8d config 3): full ODISE(label) panoptic inference, B=4 images of 1024x1024 per GPU per step, fp16 MFMA, COCO-133 vocabulary
shape (133 classes / 254 prompt strings), semantic + panoptic + instance outputs on — CategoryODISE.forward eval branch
(odise/modeling/meta_arch/odise.py:236-372).  Inputs are the seeded, 9x9-box-filtered uint8 pictures of SURVEY.md 8d (seeds rank*B ..),
resident in HBM as uint8 [H,W,3] when the timed region starts; one "step" = one `model(batched_inputs)` call = one `odise_hip_infer`
(normalise + pad, backbone, head, classification, the three heads with every decision on the device), the read-back of the segment /
instance tables (the API edge of the reference's model call) and the exchange step: every image's prediction record is all-gathered
over RCCL by the library's own communicator on a second stream (a communicator of ONE rank at --gpus 1: the same code path).
Weights are random-init tensors of the real architectures (SD-v1 UNet 859.5M, AutoencoderKL, CLIP ViT-L/14@336, ODISE heads 28M; no
network for checkpoints).  Random weights as drawn collapse the 100 queries onto one mask and one label, which would leave the decision
kernels an empty table; before the timed region the set-up of the full-size parity tests is applied (odise_amd/synthetic.py: residual-
branch gain 0.3 in the masked decoder, mask logits centred so that ~15 % are positive, text banks of the real shape spread over the
queries' own embeddings as the device computes them on the first image), so that every image yields segments and instances - the run
FAILS if it does not.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--images B] [--size S] [--vocab coco133|ade150|ade847] [--semantic-only]
                    [--stage full|unet] [--no-cpu-baseline] [--no-inclusive]

`--stage unet` runs BASELINE configs[1] (SD-UNet single-step feature extraction, `--images` = crops of 512x512).
`--vocab ade150` = configs[3] shape (K=150 / 403 strings), `--vocab ade847 --size 1280 --semantic-only` = configs[4] (K=847 / 1342
strings, 9 crops, semantic head only with the fused per-pixel argmax instead of the 5.5 GB [847,1280,1280] tensor).
Multi-GPU: one process per GPU; images are independent units sharded across ranks, no collective inside the model forward, one RCCL
all-gather of prediction records per step inside the timed region; torch.distributed (gloo, CPU) only carries the communicator id, the
barriers and the max-over-ranks of the timings.  `--gpus N` outside a launcher starts its N ranks itself (odise_amd/launch.py, like
tools/train_net.py:390-399 does through detectron2's launch) and exits non-zero if any rank fails; under the driver's own
`torch.distributed.run` the world size must equal N.  The line reports the RCCL communicator's size (`config.rccl_ranks`).  Weak scaling:
fixed images per GPU.  ONE JSON line on rank 0's stdout (everything else - RCCL's own prints included - goes to stderr).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# Algorithmic FLOPs (2 x MAC) of the live path, counted with torch.utils.flop_counter over the full-size oracle (tests/test_flop_accounting.py
# re-derives the extractor / UNet part on the meta device): per 512^2 crop CLIP 0.382 + VAE encoder 1.117 + UNet 0.740 + truncated VAE decoder
# 0.623 = 2.8625 TFLOP; per 1024^2 image 4 crops = 11.450, tap projections 0.125, mask generator 0.391, classification (MaskCLIP with 100
# mask tokens, text logits) 0.410, post-processing einsum 0.028 (K = 133).
# Not counted since round 4: the last CLIP block's out-proj / MLP / attention rows of the 576 patch tokens - the reference computes them and reads
# only ln_post(x[:, 0]) (clip.py:196-206); the device path runs that block on the class-token rows only.  Per crop 576 rows x (1024x1024 +
# 2 x 1024x4096) MACs + 576 x 577 x 64 x 16 heads x 2 attention MACs = 6.12 GMAC.
CLIP_LAST_BLOCK_DEAD_FLOPS = 0.01224e12
CROP_FLOPS = 2.8625e12 - CLIP_LAST_BLOCK_DEAD_FLOPS
FLOPS_PER_IMAGE_1024 = 12.40e12 - 4 * CLIP_LAST_BLOCK_DEAD_FLOPS
UNET_FLOPS_LIVE = 0.7401e12
MFMA_F16_PEAK = 2.5e15            # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
VOCABS = {"coco133": (133, 254, 80), "ade150": (150, 403, 100), "ade847": (847, 1342, 0)}   # classes, prompt strings, "thing" classes


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--images", type=int, default=None, help="images per step per GPU (default 4 = configs[2]); with --stage unet: 512^2 crops (default 1 = configs[1])")
    p.add_argument("--size", type=int, default=1024)
    p.add_argument("--vocab", choices=sorted(VOCABS), default="coco133")
    p.add_argument("--semantic-only", action="store_true", help="semantic head only, fused argmax output (configs[4])")
    p.add_argument("--stage", choices=["full", "unet"], default="full")
    p.add_argument("--in-flight", type=int, default=1, help="batches in flight per GPU: independent model instances (context, streams, activation arena; "
                   "one host thread each) whose steps interleave, so the launch-bound serial tail of one batch runs under the chip-filling "
                   "convolutions of the next.  Every step is still one synchronous model call on one batch of --images pictures")
    p.add_argument("--clip-ln-fold", type=int, default=0, choices=[0, 1, 2], help="A/B: the CLIP towers' LayerNorm fold, 0 = the library's rule, 1 = always, 2 = never")
    p.add_argument("--vae-chunk-mb", type=float, default=None, help="A/B: ODISE_OPT_VAE_CHUNK_BYTES in MiB (0 = all crops per launch; default: the library's)")
    p.add_argument("--attn-kvres", type=int, default=1, choices=[0, 1], help="A/B: 0 = the CLIP towers' attention on the tiled kernel instead of the K/V-resident one")
    p.add_argument("--attn-sa", type=int, default=1, choices=[0, 1], help="A/B: 0 = the UNet's self-attention on the tiled kernel instead of the software-pipelined one")
    p.add_argument("--maskclip-passes", type=int, default=0, choices=[0, 1, 2, 3], help="A/B: ODISE_OPT_MASKCLIP_PASSES (0 = MaskCLIP's image tokens ride in the crops' CLIP "
                   "tower, 1 = two passes in place, 2 = one pass over image + mask tokens, 3 = image tokens as a tower of their own behind the UNet)")
    p.add_argument("--pipeline", action="store_true", help="encoder prefetch (odise_hip_infer_prefetch): the steps alternate between two resident sets of "
                   "--images pictures, and every model call enqueues the OTHER set's input side + VAE encoder behind its own VAE lane on a low-priority "
                   "stream; the next call starts from that latent.  Outputs are bit-identical to the plain call; every step is still one synchronous call")
    p.add_argument("--prefetch-cus", type=int, default=0, help="with --pipeline: ODISE_OPT_PREFETCH_CU_EIGHTHS (1..7 of every 8 CUs for the prefetch stream; 0 = no mask)")
    p.add_argument("--prefetch-start", type=int, default=1, choices=[0, 1], help="with --pipeline: ODISE_OPT_PREFETCH_START (0 = behind the VAE lane, 1 = behind the backbone)")
    p.add_argument("--gemm-flags", type=int, default=0, help="A/B: kernel-selection switches of the GEMM / convolution library (csrc/gemm.hip launch_gemm_select; "
                   "4096 = never the 8-phase kernels, 8192 = the 8-phase kernels on 32x32x16 MFMAs), process-wide")
    p.add_argument("--picture-rank", type=int, default=None, help="rehearsal: use the pictures (and the vocabulary calibration) rank R of a multi-GPU run would "
                   "use, on one GPU - checks that every rank's decisions are non-degenerate without eight GPUs")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-inclusive", action="store_true", help="skip the PCIe- / JPEG-inclusive legs")
    return p.parse_args()


def _threads():
    """Host threads of THIS rank: the host's cores are shared by the ranks of the node (8 ranks x 16 threads of set-up work - 1.1 G synthetic
    parameters each, packed to fp16 - would oversubscribe a 128-thread host), and torch's intra-op pool stops scaling far below a 256-thread host."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    world = max(1, int(os.environ.get("WORLD_SIZE", "1")))
    return max(1, min(avail // world if world > 1 else avail, 16))


def _peak_rss_gb():
    import resource
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / (1 << 20)   # Linux: kilobytes


def image_u8(size, seed):
    """SURVEY.md 8d config 1/3 input: uint8 noise from numpy.random.default_rng(seed), smoothed by a 9x9 box filter (reflect border),
    stretched to 0..255; [size, size, 3]."""
    x = np.random.default_rng(seed).integers(0, 256, size=(3, size, size)).astype(np.float32)
    xp = np.pad(x, ((0, 0), (4, 4), (4, 4)), mode="reflect")
    c = np.cumsum(np.cumsum(np.pad(xp, ((0, 0), (1, 0), (1, 0))), axis=1, dtype=np.float64), axis=2)
    box = (c[:, 9:, 9:] - c[:, :-9, 9:] - c[:, 9:, :-9] + c[:, :-9, :-9]) / 81.0
    box = (box - box.min()) / (box.max() - box.min())
    return np.ascontiguousarray(np.round(box * 255.0).astype(np.uint8).transpose(1, 2, 0))


def cpu_baseline_full():
    """Oracle restatement (kind='port') on the host cores over WHOLE 1024x1024 images, heads included: FeatureExtractorBackbone
    (4 crops through CLIP + VAE encoder + UNet + truncated VAE decoder, projections, stitching) -> MaskFormerHead -> category logits +
    MaskCLIP + ensemble -> semantic / panoptic / instance inference, fp32 torch.  One warm-up crop (allocator, thread pool), then two timed
    whole-image passes of the LIVE path (the work whose results are used, what the device path executes); the LITERAL variant - the
    reference's forward as written, including the discarded UNet output block / VAE decoder tail (ldm.py:485-491, 515-516) - is one timed
    literal crop, scaled: literal image = live image + 4 x (literal crop - live crop)."""
    import torch
    from odise_amd.synthetic import synthetic_vocabulary
    from oracle import odise_model as om
    from oracle.backbone import FeatureExtractorBackbone
    from oracle.ldm_extractor import ImplicitCaptionerExtractor
    from oracle.m2f import SemSegHead, init_synthetic_
    cores = _threads()
    torch.set_num_threads(cores)
    ext = ImplicitCaptionerExtractor()
    bb = FeatureExtractorBackbone(ext, [512, 512, 2560, 1920, 960, 640, 512, 512])
    head = init_synthetic_(SemSegHead(num_classes=133), branch_gain=0.3)
    _, _, sizes, overlap = synthetic_vocabulary(133, 254, 768)
    heads = om.OpenVocabHeads(ext.clip, [int(v) for v in sizes], projection_dim=256, overlap=torch.from_numpy(overlap.astype(bool)))
    things = set(range(80))
    crop = torch.from_numpy(image_u8(512, 0).transpose(2, 0, 1).astype(np.float32) / 255.0)[None]

    def whole_image(seed):
        img = torch.from_numpy(image_u8(1024, seed).transpose(2, 0, 1).astype(np.float32) / 255.0)[None]
        t0 = time.perf_counter()
        out = head(bb(img))
        mask_cls = heads.classify(out, img)
        res = om.postprocess(mask_cls, out["pred_masks"], (1024, 1024), [(1024, 1024)], [(1024, 1024)], 133, things, 0.8)[0]
        return time.perf_counter() - t0, len(res["panoptic_seg"][1])

    with torch.no_grad():
        ext(crop)                                    # warm-up (allocator, thread pool)
        t0 = time.perf_counter()
        ext(crop)                                    # the live crop time the literal variant is scaled from
        t_crop_live = time.perf_counter() - t0
        t0 = time.perf_counter()
        ext(crop, run_dead_code=True)
        t_crop_lit = time.perf_counter() - t0
        t_img = [whole_image(seed)[0] for seed in (0, 1)]
    live = float(np.mean(t_img))
    literal = live + 4.0 * max(0.0, t_crop_lit - t_crop_live)
    return {"value": 1.0 / live, "unit": "images/s", "cores": cores, "kind": "port", "image_seconds_live": t_img, "crop_seconds_live": t_crop_live,
            "crop_seconds_literal": t_crop_lit, "value_literal": 1.0 / literal,
            "sample": "two whole 1024x1024 images (4 crops each), heads and post-processing included, through the fp32 torch CPU oracle of the "
                      "live path (mean); one warm-up crop pair before; literal variant (incl. the reference's dead code) = live image + 4 x "
                      "(literal crop - live crop) from one timed literal crop"}


def cpu_baseline_unet():
    import torch
    from oracle.sd_unet import UNetModel, config2_inputs, init_synthetic_, unet_forward
    model = init_synthetic_(UNetModel(width_div=1), seed=1234).eval()
    cores = _threads()
    torch.set_num_threads(cores)
    x, context, cond_emb = config2_inputs(1, 64)
    t = torch.zeros(1, dtype=torch.long)
    unet_forward(model, x, t, context, cond_emb)   # warm-up
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        unet_forward(model, x, t, context, cond_emb)
        ts.append(time.perf_counter() - t0)
    return {"value": 1.0 / float(np.median(ts)), "unit": "crops/s", "cores": cores, "kind": "port", "seconds": ts,
            "sample": "UNet single-step forward (bs=1, 64x64 latent, fp32 torch CPU oracle, live path): 1 warm-up + 3 timed passes (median)"}


DOM = {"hw": 128, "cin": 512, "cout": 512, "launches_per_step": 7}   # the dominant kernel's shape (dominant_shape / dominant_kernel_isolated)


def dominant_shape(crops):
    """The kernel with the largest share of the step (profiles/r04_bench_full_by_shape.txt): the 3x3 convolution 512->512 at 128x128 over all
    crops of a step (VAE encoder level 2 / decoder level 2; 7 launches per step).  -> (M, N, K) of its implicit GEMM, algorithmic FLOPs per
    launch = 2 * pixels * Cout * 9 * Cin.  With ODISE_OPT_VAE_CHUNK_BYTES the library runs these levels over a few crops per launch: `crops`
    is then the chunk, and a step has 7 * (crops of the step / chunk) launches."""
    M = crops * DOM["hw"] * DOM["hw"]
    return (M, DOM["cout"], 9 * DOM["cin"]), 2.0 * M * DOM["cout"] * 9 * DOM["cin"]


def dominant_kernel_isolated(ctx, n):
    """The same launch ALONE on an otherwise idle chip (after the timed region): the figure earlier rounds reported as `roofline`; kept as the
    secondary `isolated` key.  The in-step figure (launch probe, HIP events on the lane the library launches the kernel on, over the timed
    region, beside the other lane's kernels and at the clock the whole step sustains) is the `roofline` proper."""
    hw, cin, cout = DOM["hw"], DOM["cin"], DOM["cout"]
    rng = np.random.default_rng(0)
    X = ctx.to_device(rng.standard_normal((n, hw, hw, cin), dtype=np.float32).astype(np.float16))
    Wt = ctx.to_device((rng.standard_normal((cout, 3, 3, cin), dtype=np.float32) * (9 * cin) ** -0.5).astype(np.float16))
    O = ctx.empty((n, hw, hw, cout), np.float16)
    for _ in range(3):
        ctx.conv2d(X, Wt, out=O)
    ctx.sync()
    ctx.timer_start()
    it = 10
    for _ in range(it):
        ctx.conv2d(X, Wt, out=O)
    us = ctx.timer_stop() / it * 1e3
    tile = ctx.lib.odise_hip_last_tile() & 255            # what the library's cost model ran this shape on (gemm.hip kTileBM / kTileBN)
    kernel = {4: "gemm_pp2_kernel<256,256,2,2,conv>", 7: "conv3_halo_kernel<256,2>", 8: "conv3_halo_kernel<128,1>", 9: "conv3_halo4_kernel<128>"}.get(tile, f"tile {tile}")
    flops = 2.0 * n * hw * hw * cout * 9 * cin
    for a in (X, Wt, O):
        a.free()
    return {"kernel": f"{kernel} (3x3 conv 512->512 @128x128, {n} crops per launch)", "launch_us": us, "flops": flops,
            "achieved": flops / (us * 1e-6) / 1e12}


def profiled_traffic():
    """HBM bytes per launch of the dominant kernel from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes committed under profiles/
    (tools/one_conv.py; FETCH doubled as the guide's gfx950 correction prescribes).  NOT measured in this run: the line carries it as a pointer
    (`traffic_profiled`), `roofline.traffic` itself stays null."""
    for name in ("r06_dominant_conv_traffic.json", "r05_dominant_conv_traffic.json", "r04_dominant_conv_traffic.json", "r03_dominant_conv_traffic.json"):
        tp = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tp):
            with open(tp) as f:
                d = json.load(f)
            return {"file": "profiles/" + name, "hbm_bytes_per_launch": d.get("hbm_bytes_per_launch"), "crops_per_launch": d.get("crops_per_launch", 16)}
    return None


def inclusive_rates(ctx, hip, u8, S, B, sizes, steps):
    """images/s of the same batch when the boundary hands over (a) uint8 HWC host arrays (upload + conversion on the device) and
    (b) JPEG files (odise_amd.ingest.HipDatasetMapper: Huffman decoding on host threads, the rest on the device)."""
    import io

    def timed(fn):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / steps

    t_host = timed(lambda: hip.forward([{"image": ctx.to_device(u), "height": S, "width": S} for u in u8], to_host=False))
    out = {"host_u8": {"value": B / t_host, "unit": "images/s", "ms_per_step": t_host * 1e3,
                       "what": f"{B} uint8 [{S},{S},3] host arrays uploaded over PCIe and converted on the device every step"}}
    try:
        from PIL import Image
        from odise_amd.ingest import HipDatasetMapper
        jpegs = []
        for u in u8:
            buf = io.BytesIO()
            Image.fromarray(u).save(buf, "JPEG", quality=90, subsampling=2)
            jpegs.append(buf.getvalue())
        mapper = HipDatasetMapper(ctx)
        t_jpeg = timed(lambda: hip.forward(list(mapper.map_many([{"jpeg": j} for j in jpegs], workers=4)), to_host=False))
        out["jpeg"] = {"value": B / t_jpeg, "unit": "images/s", "ms_per_step": t_jpeg * 1e3,
                       "what": f"{B} baseline JPEG files ({sum(map(len, jpegs)) // B // 1000} kB each, 4:2:0, quality 90) decoded every step"}
    except ImportError:  # Pillow only writes the test files
        pass
    return out


def unet_in_step(marks):
    """ms of every UNet stage in a list of stage marks [(name, gpu_ms, host_ms)]: "lane 2: latent available, UNet starts" -> the next
    "lane 2: UNet done" (csrc/extractor.cpp unet_on_lane2; both events are recorded on the lane that runs the UNet)."""
    out, start = [], None
    for name, gpu_ms, _ in marks:
        if name.startswith("lane 2: latent available"):
            start = gpu_ms
        elif name.startswith("lane 2: UNet done") and start is not None:
            out.append(gpu_ms - start)
            start = None
    return out


def unet_isolated(ctx, crops, reps=3):
    """The UNet stage of `crops` 512^2 crops ALONE on the idle chip (the extractor's own UNet weights, t = 0, seeded inputs of the stage's
    shapes; one warm-up + `reps` timed passes on the context's stream) -> {"ms": mean, "ms_all": [...]}."""
    import ctypes as C
    from odise_amd._lib import check
    r = np.random.default_rng(1)
    x = ctx.to_device(r.standard_normal((crops, 4, 64, 64), dtype=np.float32))
    cx = ctx.to_device((r.standard_normal((1, 77, 768), dtype=np.float32) + 0.1 * r.standard_normal((crops, 77, 768), dtype=np.float32)).astype(np.float32))
    ce = ctx.to_device((0.02 * r.standard_normal((crops, 1280), dtype=np.float32)).astype(np.float32))
    taps = (C.c_void_p * 4)()

    def run():
        check(ctx.lib.odise_hip_unet_features_nhwc(ctx.h, C.c_void_p(x.ptr), C.c_void_p(cx.ptr), C.c_void_p(ce.ptr), crops, 64, 64, 0, taps), "unet_features_nhwc")

    run()
    ctx.sync()
    ms = []
    for _ in range(reps):
        ctx.timer_start()
        run()
        ms.append(ctx.timer_stop())
    for a in (x, cx, ce):
        a.free()
    return {"ms": float(np.mean(ms)), "ms_all": [float(v) for v in ms], "where": "alone on the idle chip after the timed region (one stream, eager launches)"}


MASK_POSITIVE = 0.01   # (0.15, what the parity tests use, leaves every picture ONE panoptic segment at overlap threshold 0.8; 0.01 keeps
                       # 2-18 per picture, 5-7 of them stuff - tools/segments_calib.py, profiles/r05_bench_segments_calibration.txt)  # fraction of the mask logits that is positive after calibration (odise_amd/synthetic.py mask_bias_shift)


def calibrated_model(ctx, first_image_u8, S, K, K_TOT, things, anchored, positive_fraction=None, anchor_images=None, vocab_image_u8=None):
    """HipCategoryODISE on synthetic weights with NON-DEGENERATE decisions (module docstring; odise_amd/synthetic.py): branch gain, mask
    logits centred from the device's own head outputs on `first_image_u8` (two rounds, each a reload of the 28 M-parameter head), text
    banks spread over the device's own mask / MaskCLIP embeddings.  Everything here happens before the timed region."""
    from odise_amd import synthetic as syn
    from odise_amd.pipeline import HipCategoryODISE
    # several ranks on one host (--gpus N): ONE fp32 copy of the 1.28 G synthetic parameters in /dev/shm, mapped by every rank, instead of N
    # private 5.1 GB copies generated N times (the 8-rank set-up had never run anywhere; tests/test_launch_cpu.py rehearses it)
    shared = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        state, shared = syn.synthetic_state_shared()
        state = dict(state)
    else:
        state = syn.synthetic_state()
    syn.apply_branch_gain(state)
    hip = HipCategoryODISE(ctx, state, overlap_threshold=0.8)
    state = {k: np.array(v) for k, v in state.items() if k.startswith(("sem_seg_head.", "category_head."))}   # the frozen towers are on the device now
    if shared is not None:   # every rank has mapped the file and handed the weights to its library: the local leader removes it
        import torch.distributed as dist
        syn.release_shared(shared, unlink=False)
        dist.barrier()
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            syn.release_shared(shared, unlink=True)
    cat, clp, sizes, overlap = syn.synthetic_vocabulary(K, K_TOT, 768)
    hip.set_vocabulary(cat, clp, sizes, overlap, things, 0.3, 0.7)              # provisional (random) banks: the calibration pass needs one
    img01 = ctx.to_device(np.ascontiguousarray(first_image_u8.transpose(2, 0, 1)[None].astype(np.float32) / 255.0))

    feats = hip.backbone_device(img01)                                          # s2..s5 fp32 NCHW: the head is rebuilt below, the features stay

    def head_pass():
        mf = hip.mask_features_device(feats, 1, S // 4, S // 4)
        pm, me, _, _ = hip.head_device(feats, 1, S // 4, S // 4)
        pm, me, mf_h = pm.numpy()[0], me.numpy()[0], mf.numpy()[0]
        mf.free()
        return pm, me, mf_h

    for _ in range(2):
        pm, _, mf_h = head_pass()
        shift = syn.mask_bias_shift(syn.mask_embeddings_from(pm, mf_h), pm, MASK_POSITIVE if positive_fraction is None else positive_fraction)
        state["sem_seg_head.pixel_decoder.mask_features.bias"] = (state["sem_seg_head.pixel_decoder.mask_features.bias"] + shift).astype(np.float32)
        hip.reload_head(state)
    pm, me, _ = head_pass()
    for f in feats:
        f.free()
    if vocab_image_u8 is not None:   # the vocabulary follows another picture than the mask centring: that picture's embeddings
        img01.free()
        img01 = ctx.to_device(np.ascontiguousarray(vocab_image_u8.transpose(2, 0, 1)[None].astype(np.float32) / 255.0))
        feats = hip.backbone_device(img01)
        _, me_d, _, _ = hip.head_device(feats, 1, S // 4, S // 4)
        me = me_d.numpy()[0]
        for f in feats:
            f.free()
    _, ce = hip.classify_device(img01, want_clip_embed=True)
    me_all, ce_all = [me], [ce.numpy()[0]]
    # `anchor_images` (optional): the text banks' anchor queries drawn from SEVERAL pictures' embeddings.  Measured (tools/segments_calib.py): the tables
    # get more even but smaller ([2, 4, 6, 3] segments against [18, 4, 2, 12] with picture 0 alone at 1 % positive logits), so the bench does not use it
    for u in (anchor_images or []):
        x = ctx.to_device(np.ascontiguousarray(u.transpose(2, 0, 1)[None].astype(np.float32) / 255.0))
        fx = hip.backbone_device(x)
        _, me_x, _, _ = hip.head_device(fx, 1, S // 4, S // 4)
        _, ce_x = hip.classify_device(x, want_clip_embed=True)
        me_all.append(me_x.numpy()[0])
        ce_all.append(ce_x.numpy()[0])
        for f in fx:
            f.free()
        x.free()
    t1, t2, null = syn.spread_vocabulary(np.concatenate(me_all), np.concatenate(ce_all), sizes, state["category_head.text_proj.weight"],
                                         state["category_head.text_proj.bias"], anchored=anchored, null_queries=6 * len(me_all))
    state["category_head.null_embed"] = null
    hip.load_category_head(state)
    hip.set_vocabulary(t1, t2, sizes, overlap, things, 0.3, 0.7)
    img01.free()
    return hip, float((pm > 0).mean())


def main():
    args = parse()
    from odise_amd import launch
    launch.ensure_world(args.gpus)             # --gpus N outside a launcher: re-executes itself as N ranks and exits with their status
    out_stream = launch.protect_stdout()       # fd 1 -> stderr from here on (RCCL prints through C stdio); the JSON line goes to out_stream
    os.environ["NCCL_DEBUG"] = os.environ.get("ODISE_NCCL_DEBUG", "WARN")
    rank, world, local_rank = launch.world_from_env()

    t_setup0 = time.perf_counter()
    import torch
    torch.set_num_threads(_threads())
    dist = None
    if world > 1:   # CPU rendezvous only: communicator id, barriers, max over ranks of the timings
        import torch.distributed as dist
        dist.init_process_group("gloo")

    from odise_amd.runtime import Context
    ctx = Context(local_rank)
    ctx.set_option(ctx.OPT_CLIP_LN_FOLD, args.clip_ln_fold)
    ctx.set_option(ctx.OPT_MASKCLIP_PASSES, args.maskclip_passes)
    if args.gemm_flags:
        ctx.lib.odise_hip_gemm_debug(args.gemm_flags << 4)
    if args.pipeline:
        ctx.set_option(ctx.OPT_PREFETCH_CU_EIGHTHS, args.prefetch_cus)
        ctx.set_option(ctx.OPT_PREFETCH_START, args.prefetch_start)
    if not args.attn_kvres or not args.attn_sa:
        ctx.set_option(ctx.OPT_ATTN_KV_RESIDENT, (0 if args.attn_kvres else 2) | (0 if args.attn_sa else 4))
    if args.vae_chunk_mb is not None:
        ctx.set_option(ctx.OPT_VAE_CHUNK_BYTES, int(args.vae_chunk_mb * (1 << 20)))
    B = args.images if args.images is not None else (4 if args.stage == "full" else 1)

    # Weights: random tensors of the real architecture's shapes (odise_amd/synthetic.py; no checkpoints, no network).  The oracle is
    # imported only by the cpu_baseline leg below.
    from odise_amd.synthetic import synthetic_state
    check_exchange = None
    rccl_ranks, positive_fraction = None, None
    if args.stage == "unet":
        from odise_amd.unet import HipUNet
        hip = HipUNet(ctx, synthetic_state(["model.diffusion_model."], strip="model.diffusion_model."), use_graph=True)
        r = np.random.default_rng(1)
        x = r.standard_normal((B, 4, 64, 64), dtype=np.float32)                                   # x_t
        context = (r.standard_normal((1, 77, 768), dtype=np.float32) + 0.1 * r.standard_normal((B, 77, 768), dtype=np.float32))
        cond_emb = 0.02 * r.standard_normal((B, 1280), dtype=np.float32)                          # implicit-captioner time-embedding term
        dx, dc, de = ctx.to_device(x), ctx.to_device(context.astype(np.float32)), ctx.to_device(cond_emb.astype(np.float32))

        def step():
            hip.run_nhwc(dx, dc, de, 0)
            ctx.sync()
        flops_per_unit, unit, metric = UNET_FLOPS_LIVE, "crops/s", "SD-UNet single-step feature extraction crops/sec (stage of panoptic-inference images/sec @1024x1024; UNet MFMA %peak)"
        workload = (f"BASELINE configs[1]: SD-UNet single-step feature extraction, bs={B} x 512x512 crop (64x64 latent) per GPU, t=0, "
                    "taps u2/u5/u8/u11; synthetic SD-v1-shaped weights (859.5M params); hipGraph replay")
        baseline = cpu_baseline_unet
        gather = None
    else:
        from odise_amd import distributed as D
        from odise_amd.pipeline import HipCategoryODISE
        S = args.size
        K, K_TOT, N_THINGS = VOCABS[args.vocab]
        prank = rank if args.picture_rank is None else args.picture_rank   # whose pictures this process holds (--picture-rank: a rehearsal on one GPU)
        u8 = [image_u8(S, prank * B + b) for b in range(B)]              # images shard across ranks: each rank has its own
        # every rank centres the mask logits on the SAME picture (seed 0): identical network weights on all ranks, like a loaded checkpoint;
        # the text banks (inputs of the path) are spread over the rank's OWN first picture, so every rank times non-empty decision tables
        hip, positive_fraction = calibrated_model(ctx, u8[0] if prank == 0 else image_u8(S, 0), S, K, K_TOT, set(range(N_THINGS)),
                                                  None if K <= 200 else 188, vocab_image_u8=None if prank == 0 else u8[0])
        if args.semantic_only:   # configs[4]: pano_open_d2_eval.py:127-133 switches the other heads off; the evaluator keeps argmax(0)
            hip.panoptic_on = hip.instance_on = False
            hip.semantic_argmax = True
        d_img = [ctx.to_device(u) for u in u8]
        hw = [(S, S)] * B
        rec = D.record_size(S, S)
        exchange = D.Exchange(ctx, rank, world, D.gloo_broadcast if world > 1 else None)
        # further batches in flight: a model instance of their own (same weights, same calibration -> identical state), their own pictures
        slots = [{"ctx": ctx, "hip": hip, "img": d_img}]
        if args.pipeline:   # a second resident set of pictures: the steps alternate, each call prefetching the other set's encoder
            assert args.in_flight == 1, "--pipeline is the one-context form; --in-flight runs whole contexts side by side"
            slots[0]["img_alt"] = [ctx.to_device(image_u8(S, (world + rank) * B + b)) for b in range(B)]
            slots[0]["turn"] = 0
        for k in range(1, max(1, args.in_flight)):
            c2 = Context(local_rank)
            h2, _ = calibrated_model(c2, image_u8(S, 0), S, K, K_TOT, set(range(N_THINGS)), None if K <= 200 else 188)
            h2.panoptic_on, h2.instance_on, h2.semantic_argmax = hip.panoptic_on, hip.instance_on, hip.semantic_argmax
            slots.append({"ctx": c2, "hip": h2, "img": [c2.to_device(image_u8(S, (k * world + rank) * B + b)) for b in range(B)]})
        import ctypes
        cw = ctypes.c_int(0)
        ctx.lib.odise_hip_comm_info(ctx.h, None, ctypes.byref(cw))
        if cw.value != args.gpus:                                        # the line must never report more GPUs than the communicator spans
            raise SystemExit(f"RCCL communicator spans {cw.value} ranks, --gpus {args.gpus}")
        rccl_ranks = cw.value
        for sl in slots:
            sl["local"] = sl["ctx"].zeros((B, rec), np.int32)            # this rank's prediction records, written by the kernels
            sl["allrec"] = sl["ctx"].zeros((world * B, rec), np.int32)
            sl["pan_out"] = [sl["local"].ptr + b * rec * 4 for b in range(B)] if hip.panoptic_on else None
        import threading
        turn = {"next": 0, "cv": threading.Condition()}

        def step(sl=slots[0], ticket=None):
            # one model call (synchronous at its API edge: the tables are read back) + the one exchange step of the path, which then
            # runs on the library's exchange stream while the next call's kernels execute.  With several batches in flight the one
            # communicator is used in global step order (`ticket`), the same order on every rank.
            imgs = sl["img"]
            if args.pipeline:
                cur, nxt = (sl["img"], sl["img_alt"]) if sl["turn"] == 0 else (sl["img_alt"], sl["img"])
                sl["turn"] ^= 1
                sl["left"] = sl.get("left", 1 << 30) - 1
                # `left` is set to the number of steps before the warm-up loop AND before the timed loop: the last step of either prepares nothing,
                # so no encoder of a timed batch runs outside the timed region and none of the timed region's work is left undone - K timed steps
                # run exactly K encoders inside the region (the first timed step has no prefetched latent to start from).  (with the
                # counter armed only for the timed loop, the last warm-up step ran the first timed batch's encoder before the clock started.)
                if sl["left"] > 0:
                    sl["hip"].prefetch_device(nxt, 0, hw)    # registered before the call that will enqueue it behind its own VAE lane
                imgs = cur
            sl["res"] = sl["hip"].infer_device(imgs, 0, hw, hw, to_host=False, pan_out=sl["pan_out"])
            if hip.panoptic_on:
                if ticket is None:
                    exchange.allgather(sl["local"], sl["allrec"])
                else:
                    with turn["cv"]:
                        turn["cv"].wait_for(lambda: turn["next"] == ticket)
                        exchange.allgather(sl["local"], sl["allrec"])
                        turn["next"] += 1
                        turn["cv"].notify_all()

        def check_exchange():
            """After the timed region: this rank's slice of the gathered buffer must hold its own records, and EVERY image's record a
            well-formed, NON-EMPTY segment table (an empty one means the timed decision kernels ran on nothing)."""
            report = check_slot(slots[0])
            for k, sl in enumerate(slots[1:], 1):
                report[f"in_flight_{k}"] = check_slot(sl)
            return report

        def check_slot(sl):
            res, local, allrec = sl["res"], sl["local"], sl["allrec"]
            report = {}
            # Every rank spreads the vocabulary over its own first picture (calibrated_model), so EVERY rank asserts that its decision kernels were
            # timed on non-empty tables; the stricter floor of the headline configuration (>= 2 segments on every picture,
            # 5 per picture on average) is what was measured for rank 0's pictures and is asserted there.
            strict = True
            headline = rank == 0 and args.picture_rank in (None, 0)
            if hip.instance_on:
                report["instances_per_image"] = [int(len(r["instances"]["scores"])) for r in res]
                assert not strict or min(report["instances_per_image"]) > 0, f"an image has no instances: {report['instances_per_image']}"
            if hip.semantic_argmax:
                lab = res[0]["sem_seg_argmax"].numpy()
                report["semantic_labels_image0"] = int(len(np.unique(lab)))
                assert not strict or report["semantic_labels_image0"] > 1, "the semantic arg-max is one constant label"
            if not hip.panoptic_on:
                return report
            exchange.wait(True)
            mine = allrec.view((B, rec), np.int32, rank * B * rec * 4).numpy()
            own = local.numpy()
            assert np.array_equal(mine, own), "all-gather: this rank's slice differs from its local records"
            counts = []
            for b in range(B):
                seg, info = D.unpack_record(torch.from_numpy(own[b]), S, S)
                ids = sorted({s["id"] for s in info})
                assert ids == list(range(1, len(info) + 1)) and set(np.unique(seg)) <= set([0] + ids), "bad prediction record"
                counts.append(len(info))
            # the vocabulary is spread over the FIRST image's queries: that image must produce segments (the other pictures of the batch are
            # reported; at overlap threshold 0.8 a picture with every mask contested can legitimately keep none)
            assert not strict or (counts[0] > 0 and sum(counts) > 0), f"empty segment tables: {counts} (degenerate decisions)"
            if headline and S == 1024 and args.vocab == "coco133" and B >= 4:   # the headline configuration: the decision kernels are timed on real tables
                assert min(counts) >= 2 and sum(counts) >= 5 * B, f"near-degenerate segment tables: {counts}"
            if not headline:
                print(f"[bench] rank {rank} (pictures of rank {prank}): segments per image {counts}, instances {report.get('instances_per_image')}", file=sys.stderr, flush=True)
            report.update({"segments_per_image": counts, "segments_image0": counts[0], "records_bytes_per_rank": int(B * rec * 4)})
            return report
        ncrops = (-(-S // 512)) ** 2                                       # slide windows of 512 (feature_extractor.py:197-222): 4 at 1024, 9 at 1280
        flops_per_unit = ncrops * CROP_FLOPS + (FLOPS_PER_IMAGE_1024 - 4 * CROP_FLOPS) * (S / 1024.0) ** 2
        unit, metric = "images/s", f"panoptic-inference images/sec @{S}x{S}"
        cfg = {"coco133": "BASELINE configs[2]", "ade150": "BASELINE configs[3] shapes", "ade847": "BASELINE configs[4] shapes"}[args.vocab]
        workload = (f"{cfg}: full ODISE(label) inference (CategoryODISE eval forward: {ncrops} crops/image through "
                    f"CLIP+VAE+UNet, projections, MSDeformAttn pixel decoder, 9-layer masked decoder, MaskCLIP, "
                    f"{'semantic head with fused argmax' if args.semantic_only else 'semantic+panoptic+instance heads decided on the device'}), "
                    f"bs={B} x {S}x{S} uint8 per GPU resident in HBM, vocabulary {K} classes/{K_TOT} strings; synthetic weights of the real shapes "
                    f"calibrated for non-degenerate decisions (branch gain 0.3, {positive_fraction:.0%} positive mask logits, text banks spread over "
                    f"the queries); seeded box-filtered uint8 inputs (SURVEY 8d)")
        baseline = cpu_baseline_full
        gather = True

    def barrier():
        ctx.sync()
        if args.stage == "full":
            for sl in slots[1:]:
                sl["ctx"].sync()
            exchange.wait(True)
        if dist is not None:
            dist.barrier()

    # set-up is over (weights generated, packed, uploaded and calibrated on every rank): one line per rank, so that a multi-GPU run that dies or crawls
    # in set-up leaves its cause behind (never exercised at 8 concurrent ranks on one host)
    print(f"[bench rank {rank}/{world}] set-up {time.perf_counter() - t_setup0:.1f} s, peak host RSS {_peak_rss_gb():.1f} GB, {_threads()} host threads",
          file=sys.stderr, flush=True)
    n_fly = len(slots) if args.stage == "full" else 1
    single_ms = None
    probe = None
    if args.stage == "full":
        # launch probe on the dominant kernel's shape: the library records HIP events around each of its launches in the timed region, on the
        # lane it launches them on.  The VAE levels may run in crop chunks (ODISE_OPT_VAE_CHUNK_BYTES): the chunk follows from the option.
        total_crops = B * ncrops
        chunk_bytes = ctx.get_option(ctx.OPT_VAE_CHUNK_BYTES)
        per_crop = DOM["hw"] * DOM["hw"] * DOM["cout"] * 2
        chunk = total_crops if chunk_bytes <= 0 else max(1, min(total_crops, chunk_bytes // per_crop))
        probe = {"crops": chunk, "shape": dominant_shape(chunk)[0], "flops": dominant_shape(chunk)[1]}
    marks = None
    if n_fly == 1:
        if args.stage == "full":
            slots[0]["left"] = args.warmup
        for _ in range(args.warmup):
            step()
        barrier()
        if probe is not None:
            ctx.probe_arm(True, *probe["shape"], max_launches=4096)
            ctx.stage_timeline(True)   # an event per stage boundary on the lane that runs the stage (~22 per step): where the UNet's share of the step comes from
        t0 = time.perf_counter()
        if args.stage == "full":
            slots[0]["left"] = args.steps
        ctx.timer_start()
        for _ in range(args.steps):
            step()
        ev_ms = ctx.timer_stop()  # HIP events on the library's stream (synchronises)
        barrier()
        wall = time.perf_counter() - t0
        if probe is not None:
            probe["us"] = ctx.probe_read()
            marks = ctx.stage_timeline_read()
            ctx.stage_timeline(False)
    else:
        # K steps in all, dealt round-robin to the instances; each instance's steps run on its own host thread (a step is one C call that
        # releases the interpreter lock).  The timed region opens after every instance has warmed up and drained, and closes after all K
        # steps and their exchanges are complete on every rank.
        gate = threading.Barrier(n_fly + 1)
        errors = []
        queue = {"next": 0, "lock": threading.Lock()}
        stagger = [0.0]

        def worker(k):
            try:
                sl = slots[k]
                gate.wait()      # timed region open
                time.sleep(k * stagger[0])   # instances start a fraction of a step apart (inside the timed region), so that they sit in different phases
                while True:
                    with queue["lock"]:
                        i = queue["next"]
                        queue["next"] += 1
                    if i >= args.steps:
                        break
                    step(sl, i)
                sl["ctx"].sync()
            except BaseException as exc:   # a failed instance must not leave the others waiting at the gate
                errors.append(exc)
                gate.abort()
                raise
            gate.wait()          # done

        threads = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(n_fly)]
        for t in threads:
            t.start()
        for sl in slots:
            for _ in range(max(1, args.warmup)):
                tw = time.perf_counter()
                step(sl)
                sl["ctx"].sync()
                single_ms = (time.perf_counter() - tw) * 1e3    # one batch alone on the chip (the last one: warm)
        stagger[0] = single_ms * 1e-3 / n_fly
        barrier()
        try:
            t0 = time.perf_counter()
            gate.wait()
            gate.wait()
        except threading.BrokenBarrierError:
            raise SystemExit(f"an in-flight instance failed: {errors}")
        barrier()
        wall = time.perf_counter() - t0
        ev_ms = wall * 1e3   # no single stream spans the region: the host clock (which includes the final synchronisation) stands in
        for t in threads:
            t.join()

    if dist is not None:
        tt = torch.tensor([wall, ev_ms], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tt[0]), float(tt[1])

    exch = check_exchange() if check_exchange is not None else None

    # Boundary variants, reported beside `value` (never as it): the same step fed from host memory over PCIe, and from JPEG bytes
    # (host Huffman decoding in loader threads + device IDCT / resize).  Outputs stay on the device as in the reference.
    inclusive = None
    if rank == 0 and world == 1 and args.stage == "full" and not args.no_inclusive:
        try:
            inclusive = inclusive_rates(ctx, hip, u8, S, B, hw, max(2, min(args.steps, 3)))
        except Exception as exc:  # side legs must never take the headline measurement down with them
            inclusive = {"error": f"{type(exc).__name__}: {exc}"}

    # every rank: one diagnostic line on stderr (which device, how many ranks the RCCL communicator spans, this rank's own step time), so that a
    # multi-GPU run that goes wrong leaves more than one number behind
    print(f"[bench rank {rank}/{world}] device {local_rank} ({ctx.device_info()[0]}), rccl_ranks {rccl_ranks}, {args.steps} steps, "
          f"{wall * 1e3 / args.steps:.2f} ms/step (max over ranks), in-step dominant-kernel launches recorded "
          f"{0 if probe is None or 'us' not in probe else len(probe['us'])}", file=sys.stderr, flush=True)
    dom = None
    if rank == 0 and args.stage == "full" and probe is not None:
        try:
            dom = dominant_kernel_isolated(ctx, probe["crops"])
        except Exception as exc:  # the whole-step roofline below still goes out
            print(f"[bench] dominant-kernel measurement failed: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = world * B * args.steps / wall
        dev_name, cus, _ = ctx.device_info()
        step_ms_ev = ev_ms / args.steps
        achieved = flops_per_unit * B / (step_ms_ev * 1e-3) / 1e12  # TFLOP/s per GPU of live algorithmic work
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": workload, "units_per_step_per_gpu": B, "device": dev_name, "compute_units": cus,
                       "parallelism": (f"dp{world} (independent images, one RCCL all-gather of prediction records per step on the library's "
                                       f"exchange stream)") if gather else f"dp{world}",
                       "rccl_ranks": rccl_ranks, "batches_in_flight": n_fly, "one_batch_alone_ms": single_ms,
                       "vae_chunk_bytes": ctx.get_option(ctx.OPT_VAE_CHUNK_BYTES), "clip_ln_fold": ctx.get_option(ctx.OPT_CLIP_LN_FOLD),
                       "gemm_flags": args.gemm_flags, "maskclip_passes": ctx.get_option(ctx.OPT_MASKCLIP_PASSES),
                       "pipeline": (f"encoder-prefetch (start {args.prefetch_start}, {args.prefetch_cus or 8}/8 CUs)" if args.pipeline else None)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_F16_PEAK / 1e12, "unit": "TFLOP/s", "frac": achieved * 1e12 / MFMA_F16_PEAK,
                         "traffic": None,
                         "kernel": "whole step (all kernels; per-kernel times in profiles/)",
                         "algorithmic_flops_per_unit": flops_per_unit, "event_ms_per_step": step_ms_ev},
        }
        in_step = probe.get("us") if probe is not None else None
        if in_step is not None and len(in_step) > 0:
            # The contract's roofline object describes the DOMINANT kernel AS IT RUNS IN THE TIMED STEP: the mean over every one of its launches
            # in the timed region (HIP events recorded by the library on the lane that launches it), i.e. beside the other lane's kernels and at
            # the clock the step sustains.  `isolated` = the same launch alone on the idle chip; the whole-step figure is in the step_* keys.
            us = float(np.mean(in_step))
            ach = probe["flops"] / (us * 1e-6) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_F16_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach * 1e12 / MFMA_F16_PEAK,
                               "traffic": None, "traffic_profiled": profiled_traffic(),
                               "kernel": (dom["kernel"] if dom is not None else f"3x3 conv 512->512 @128x128, {probe['crops']} crops per launch"),
                               "where": "in the timed step (launch probe: HIP events on the launching lane, all launches of the timed region)",
                               "launch_us": us, "launch_us_min": float(np.min(in_step)), "launch_us_max": float(np.max(in_step)),
                               "launches_timed": int(len(in_step)), "launches_per_step": len(in_step) / args.steps,
                               "crops_per_launch": probe["crops"], "algorithmic_flops_per_launch": probe["flops"],
                               "isolated": None if dom is None else {"launch_us": dom["launch_us"], "achieved": dom["achieved"],
                                                                     "frac": dom["achieved"] * 1e12 / MFMA_F16_PEAK},
                               "step_achieved": achieved, "step_frac": achieved * 1e12 / MFMA_F16_PEAK,
                               "algorithmic_flops_per_unit": flops_per_unit, "event_ms_per_step": step_ms_ev}
        if args.stage == "full" and marks:
            # The metric's second half ("UNet MFMA %peak"): the UNet stage AS IT RUNS IN THE TIMED STEP, from the stage-boundary events the library
            # records on the lane that runs it ("latent available, UNet starts" -> "UNet done": beside the VAE decoder on the other lane), and the
            # same stage ALONE on the idle chip at the step's crop count, measured after the timed region.
            total_crops = B * ncrops
            un = unet_in_step(marks)
            if un:
                ms_u = float(np.mean(un))
                out["roofline"]["unet"] = {"ms": ms_u, "ms_min": float(np.min(un)), "ms_max": float(np.max(un)), "steps": len(un), "crops": total_crops,
                                           "algorithmic_flops_per_crop": UNET_FLOPS_LIVE, "achieved": total_crops * UNET_FLOPS_LIVE / (ms_u * 1e-3) / 1e12,
                                           "frac": total_crops * UNET_FLOPS_LIVE / (ms_u * 1e-3) / MFMA_F16_PEAK,
                                           "where": "in the timed step, stage marks (HIP events on the CLIP -> UNet lane; the VAE decoder runs beside it)"}
            try:
                iso = unet_isolated(ctx, total_crops)
                out["roofline"]["unet_isolated"] = dict(iso, crops=total_crops, achieved=total_crops * UNET_FLOPS_LIVE / (iso["ms"] * 1e-3) / 1e12,
                                                        frac=total_crops * UNET_FLOPS_LIVE / (iso["ms"] * 1e-3) / MFMA_F16_PEAK)
            except Exception as exc:
                print(f"[bench] isolated UNet measurement failed: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
        if exch is not None:
            out["exchange"] = exch
        if inclusive is not None:
            out["inclusive"] = inclusive
            if "host_u8" in inclusive:   # the evaluator-faithful figure (pictures arrive from host memory every step), next to `value` - never as it
                out["value_host_fed"] = inclusive["host_u8"]["value"]
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = baseline()
            except Exception as exc:
                out["cpu_baseline"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(out), file=out_stream, flush=True)
    if dist is not None:
        dist.barrier()
    if args.stage == "full":
        exchange.close()
        for sl in slots[1:]:
            sl["ctx"].close()
    if dist is not None:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
