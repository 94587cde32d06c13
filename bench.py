#!/usr/bin/env python3
"""bench.py — hot-path benchmark of the MI355X-native ODISE inference path.

Default workload (BASELINE.json configs[2], the configuration the metric "panoptic-inference images/sec @1024x1024" is quoted on):
full ODISE(label) panoptic inference, B=4 images of 1024x1024 per GPU per step, fp16 MFMA, COCO-133 vocabulary shape
(133 classes / 254 prompt strings), semantic + panoptic + instance outputs on — CategoryODISE.forward eval branch
(odise/modeling/meta_arch/odise.py:236-372).  One "step" = one pass of the whole path over the batch with the images already
resident in HBM; results stay on the device like the reference's outputs.  Weights are random-init tensors of the real
architectures (SD-v1 UNet 859.5M, AutoencoderKL, CLIP ViT-L/14@336, ODISE heads 28M; no network for checkpoints) and the
vocabulary is a seeded random text bank of the real shape (the CLIP text tower is a later row of SURVEY.md §8f).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--images B] [--size S] [--stage full|unet] [--no-cpu-baseline]

`--stage unet` runs BASELINE configs[1] instead (SD-UNet single-step feature extraction, `--images` = crops of 512x512).
Multi-GPU: one process per GPU (torch.distributed.run); images are independent units sharded across ranks with no collective
inside the model forward; after the timed steps of the full path every rank all-gathers its panoptic prediction records over
RCCL (one collective per step, inside the timed region).  Weak scaling: fixed images per GPU.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# analytic live work (SURVEY.md §8d / BASELINE.md §2)
# Algorithmic FLOPs (2 x MAC) of the live path, counted with torch.utils.flop_counter over the full-size oracle (tests/test_flop_accounting.py
# re-derives the extractor / UNet part on the meta device): per 512^2 crop CLIP 0.382 + VAE encoder 1.117 + UNet 0.740 + truncated VAE decoder
# 0.623 = 2.8625 TFLOP; per 1024^2 image 4 crops = 11.450, tap projections 0.125, mask generator 0.391, classification (MaskCLIP with 100
# mask tokens, text logits) 0.410, post-processing einsum 0.028.
FLOPS_PER_IMAGE_1024 = 12.40e12
UNET_FLOPS_LIVE = 0.7401e12
MFMA_F16_PEAK = 2.5e15            # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--images", type=int, default=None, help="images per step per GPU (default 4 = configs[2]); with --stage unet: 512^2 crops (default 1 = configs[1])")
    p.add_argument("--size", type=int, default=1024)
    p.add_argument("--stage", choices=["full", "unet"], default="full")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-inclusive", action="store_true", help="skip the PCIe- / JPEG-inclusive legs")
    return p.parse_args()


def _threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))  # torch's intra-op pool stops scaling (and thrashes) far below a 256-thread host


def cpu_baseline_full():
    """Oracle restatement (kind='port') on the host cores.  Bounded sample: the 4 crops of ONE 1024x1024 image through the feature
    extractor (CLIP + VAE encoder + UNet + truncated VAE decoder = 2.86 of the 3.13 TFLOP a crop costs end to end, i.e. 92 % of an
    image's work), one warm-up pass and one timed pass (~10-20 s of CPU work); images/s = 1 / pass time, an upper bound of the CPU rate."""
    import torch
    from oracle.ldm_extractor import ImplicitCaptionerExtractor
    cores = _threads()
    torch.set_num_threads(cores)
    ext = ImplicitCaptionerExtractor()
    img = torch.rand(4, 3, 512, 512, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ext(img[:1])  # warm-up (allocator, thread pool)
        t0 = time.perf_counter()
        ext(img)
        dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "the 4 crops (512x512) of one 1024x1024 image through the fp32 torch CPU oracle of LdmImplicitCaptionerExtractor "
                      "(92% of an image's work), 1 timed pass after a 1-crop warm-up; heads and post-processing excluded"}


def cpu_baseline_unet():
    import torch
    from oracle.sd_unet import UNetModel, config2_inputs, init_synthetic_, unet_forward
    model = init_synthetic_(UNetModel(width_div=1), seed=1234).eval()
    cores = _threads()
    torch.set_num_threads(cores)
    x, context, cond_emb = config2_inputs(1, 64)
    t = torch.zeros(1, dtype=torch.long)
    t0 = time.perf_counter()
    unet_forward(model, x, t, context, cond_emb)
    dt = time.perf_counter() - t0
    if dt < 8.0:
        t0 = time.perf_counter()
        unet_forward(model, x, t, context, cond_emb)
        dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": "1 timed UNet single-step forward (bs=1, 64x64 latent, fp32 torch CPU oracle, live path)"}


def dominant_kernel(ctx):
    """The kernel with the largest share of the step (profiles/): conv3_halo_kernel<256,2> on the VAE 512->512 3x3 convolutions at
    128x128 for the 16 crops of a 4-image step (7 launches per step, 18 % of the step's FLOPs).  Timed live with HIP events on the
    library's stream; algorithmic FLOPs per launch = 2 * pixels * Cout * 9 * Cin.  `traffic` = HBM bytes per launch from the PMC
    passes of the same launch (tools/one_conv.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled as the guide's gfx950
    correction prescribes), recorded in profiles/r01_dominant_conv_traffic.json; null when that file is absent."""
    n, hw, cin, cout = 16, 128, 512, 512
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
    flops = 2.0 * n * hw * hw * cout * 9 * cin
    traffic = None
    tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_dominant_conv_traffic.json")
    if os.path.exists(tp):
        with open(tp) as f:
            traffic = json.load(f).get("hbm_bytes_per_launch")
    for a in (X, Wt, O):
        a.free()
    return {"kernel": "conv3_halo_kernel<256,2> (3x3 conv 512->512 @128x128, 16 crops)", "launch_us": us, "flops": flops,
            "achieved": flops / (us * 1e-6) / 1e12, "traffic": traffic, "launches_per_step": 7, "ceiling": _safe(mfma_ceiling, ctx)}


def _safe(fn, *args):
    try:
        return fn(*args)
    except Exception:
        return None


def mfma_ceiling(ctx):
    """What the matrix pipes of THIS device sustain on random fp16 operands held in registers (no memory traffic at all; the ping-pong
    barrier skeleton of the GEMM kernels): odise_amd/csrc/probe.hip, tools/mfma_rate.py.  The part is power-limited - the shader clock
    drops from ~2.4 GHz (constant operands: 2.4-2.5 PFLOP/s) to ~1.6 GHz - so this, not the spec sheet, is what a GEMM can approach."""
    import ctypes as C
    ms, fl, mhz = C.c_float(0), C.c_double(0), C.c_double(0)
    cus = ctx.device_info()[1]
    if ctx.lib.odise_hip_mfma_rate(ctx.h, 8, 1000, cus * 4, 8, C.byref(ms), C.byref(fl), C.byref(mhz)) != 0:
        return None
    return {"tflops": fl.value / (ms.value * 1e-3) / 1e12, "shader_clock_mhz": mhz.value,
            "what": "v_mfma_f32_32x32x16_f16 on register-resident random operands, 2 waves/SIMD, barrier skeleton of the GEMM kernels, 8 launches"}


def inclusive_rates(ctx, hip, img, S, B, steps):
    """images/s of the same batch when the boundary hands over (a) uint8 HWC host arrays (upload + conversion on the device) and
    (b) JPEG files (odise_amd.ingest.HipDatasetMapper: Huffman decoding on host threads, the rest on the device)."""
    import io
    u8 = [np.ascontiguousarray((img[b].transpose(1, 2, 0) * 255.0).astype(np.uint8)) for b in range(B)]

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
            yy, xx = np.mgrid[0:S, 0:S]
            pic = np.stack([128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 29.0), 128 + 90 * np.cos(xx / 23.0 - yy / 41.0), (xx + yy) % 256], -1)
            pic = np.clip(pic + (u.astype(np.float32) - 128) * 0.08, 0, 255).astype(np.uint8)   # photo-like statistics (~2 bits/pixel)
            buf = io.BytesIO()
            Image.fromarray(pic).save(buf, "JPEG", quality=90, subsampling=2)
            jpegs.append(buf.getvalue())
        mapper = HipDatasetMapper(ctx)
        t_jpeg = timed(lambda: hip.forward(list(mapper.map_many([{"jpeg": j} for j in jpegs], workers=4)), to_host=False))
        out["jpeg"] = {"value": B / t_jpeg, "unit": "images/s", "ms_per_step": t_jpeg * 1e3,
                       "what": f"{B} baseline JPEG files ({sum(map(len, jpegs)) // B // 1000} kB each, 4:2:0, quality 90) decoded every step"}
    except ImportError:  # Pillow only writes the test files
        pass
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    torch.set_num_threads(_threads())
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from odise_amd.runtime import Context
    ctx = Context(local_rank)
    B = args.images if args.images is not None else (4 if args.stage == "full" else 1)

    # Weights: random tensors of the real architecture's shapes (odise_amd/synthetic.py; no checkpoints, no network).  The oracle is
    # imported only by the cpu_baseline leg below.
    from odise_amd.synthetic import synthetic_state, synthetic_vocabulary
    if args.stage == "unet":
        from odise_amd.unet import HipUNet
        hip = HipUNet(ctx, synthetic_state(["model.diffusion_model."], strip="model.diffusion_model."), use_graph=True)
        r = np.random.default_rng(1)
        x = r.standard_normal((B, 4, 64, 64), dtype=np.float32)                                   # x_t
        context = (r.standard_normal((1, 77, 768), dtype=np.float32) + 0.1 * r.standard_normal((B, 77, 768), dtype=np.float32))
        cond_emb = 0.02 * r.standard_normal((B, 1280), dtype=np.float32)                          # implicit-captioner time-embedding term
        dx, dc, de = ctx.to_device(x), ctx.to_device(context.astype(np.float32)), ctx.to_device(cond_emb.astype(np.float32))
        step = lambda: hip.run_nhwc(dx, dc, de, 0)
        flops_per_unit, unit, metric = UNET_FLOPS_LIVE, "crops/s", "SD-UNet single-step feature extraction crops/sec (stage of panoptic-inference images/sec @1024x1024; UNet MFMA %peak)"
        workload = (f"BASELINE configs[1]: SD-UNet single-step feature extraction, bs={B} x 512x512 crop (64x64 latent) per GPU, t=0, "
                    "taps u2/u5/u8/u11; synthetic SD-v1-shaped weights (859.5M params); hipGraph replay")
        baseline = cpu_baseline_unet
        gather = None
    else:
        from odise_amd import distributed as D
        from odise_amd.pipeline import HipCategoryODISE
        S = args.size
        K, K_TOT = 133, 254   # COCO panoptic: 133 classes, 254 prompt-engineered strings
        state = synthetic_state()
        hip = HipCategoryODISE(ctx, state, overlap_threshold=0.8)
        cat, clp, sizes, overlap = synthetic_vocabulary(K, K_TOT, 768)
        hip.set_vocabulary(cat, clp, sizes, overlap, set(range(80)), 0.3, 0.7)
        del state
        img = np.random.default_rng(rank).random((B, 3, S, S), dtype=np.float32)  # images shard across ranks: each rank has its own
        d_img = ctx.to_device(img)
        out_sizes = [(S, S)] * B
        records = torch.zeros((B, D.record_size(S, S)), dtype=torch.int32, device="cuda" if world > 1 else "cpu")

        hw = S * S

        def step():
            # with several ranks the panoptic maps are written straight into this rank's slice of the gather buffer
            pan_out = [records[b].data_ptr() for b in range(B)] if dist is not None else None
            res = hip.forward_device(d_img, d_img, out_sizes, to_host=False, pan_out=pan_out)
            if dist is not None:  # the one exchange step of the path: every rank ends up with every image's panoptic record
                for b, r in enumerate(res):
                    table = torch.zeros(1 + D.MAX_SEGMENTS * 3, dtype=torch.int32)
                    info = r["panoptic_seg"][1][: D.MAX_SEGMENTS]
                    table[0] = len(info)
                    for i, sgm in enumerate(info):
                        table[1 + 3 * i: 4 + 3 * i] = torch.tensor([sgm["id"], int(sgm["isthing"]), sgm["category_id"]], dtype=torch.int32)
                    records[b, hw:] = table.to(records.device)
                ctx.sync()
                D.allgather_records(records)
            return res
        flops_per_unit, unit, metric = FLOPS_PER_IMAGE_1024 * (S / 1024.0) ** 2, "images/s", "panoptic-inference images/sec @1024x1024"
        workload = (f"BASELINE configs[2]: full ODISE(label) panoptic inference (CategoryODISE eval forward: 4 crops/image through CLIP+VAE+UNet, "
                    f"projections, MSDeformAttn pixel decoder, 9-layer masked decoder, MaskCLIP, semantic+panoptic+instance post-processing), "
                    f"bs={B} x {S}x{S} per GPU, vocabulary {K} classes/{K_TOT} strings; synthetic weights of the real shapes, random text bank")
        baseline = cpu_baseline_full
        gather = True

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_stop()  # HIP events on the library's stream (synchronises)
    barrier()
    wall = time.perf_counter() - t0

    if dist is not None:
        tt = torch.tensor([wall, ev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tt[0]), float(tt[1])

    # Boundary variants, reported beside `value` (never as it): the same step fed from host memory over PCIe, and from JPEG bytes
    # (host Huffman decoding in loader threads + device IDCT / resize).  Outputs stay on the device as in the reference.
    inclusive = None
    if rank == 0 and world == 1 and args.stage == "full" and not args.no_inclusive:
        try:
            inclusive = inclusive_rates(ctx, hip, img, S, B, max(2, min(args.steps, 3)))
        except Exception as exc:  # side legs must never take the headline measurement down with them
            inclusive = {"error": f"{type(exc).__name__}: {exc}"}

    dom = None
    if rank == 0 and args.stage == "full":
        try:
            dom = dominant_kernel(ctx)
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
                       "parallelism": f"dp{world} (independent images, one RCCL all-gather of predictions per step)" if gather else f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_F16_PEAK / 1e12, "unit": "TFLOP/s", "frac": achieved * 1e12 / MFMA_F16_PEAK,
                         "traffic": None,
                         "kernel": "whole step (all kernels; per-kernel times in profiles/)",
                         "algorithmic_flops_per_unit": flops_per_unit, "event_ms_per_step": step_ms_ev},
        }
        if dom is not None:
            # the contract's roofline object describes the DOMINANT kernel; the whole-step figure moves to step_* keys
            out["roofline"] = {"bound": "mfma", "achieved": dom["achieved"], "peak": MFMA_F16_PEAK / 1e12, "unit": "TFLOP/s",
                               "frac": dom["achieved"] * 1e12 / MFMA_F16_PEAK, "traffic": dom["traffic"], "kernel": dom["kernel"],
                               "launch_us": dom["launch_us"], "algorithmic_flops_per_launch": dom["flops"], "launches_per_step": dom["launches_per_step"],
                               "step_achieved": achieved, "step_frac": achieved * 1e12 / MFMA_F16_PEAK,
                               "measured_ceiling": dom["ceiling"],
                               "frac_of_measured_ceiling": (dom["achieved"] / dom["ceiling"]["tflops"]) if dom["ceiling"] else None,
                               "algorithmic_flops_per_unit": flops_per_unit, "event_ms_per_step": step_ms_ev}
        if inclusive is not None:
            out["inclusive"] = inclusive
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = baseline()
            except Exception as exc:
                out["cpu_baseline"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
