"""Same-process A/B of the two GEMM / conv epilogues (tools build: ODISE_HIP_LIB=.../libodise_hip_tools.so): the fp32-staged multi-pass form
(`old`, odise_hip_gemm_debug bit 1 << 24) against the math-first fp16-staged form (`new`, the product default).  Every case prints both times
(min over interleaved rounds) and whether the OUTPUT BYTES are identical - they must be: same arithmetic per element in the same order.
Part 1 walks every tile / kernel family with every epilogue feature on ragged shapes (correctness); part 2 times the shapes of the step.
usage: epi16_ab.py [quick]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd._lib import ACT_GELU, ACT_NONE, ACT_QUICKGELU, ACT_RELU, ACT_SILU  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)
OLD = 1 << 24
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
bad = []


def f16(shape, s=1.0):
    return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))


def f32(shape, s=1.0):
    return ctx.to_device(rng.standard_normal(shape, dtype=np.float32) * s)


def ab(label, fn, outs, flop, it=10, rounds=2, extra_flags=0):
    """fn() launches; outs = list of DeviceArrays to compare.  Returns (old_us, new_us)."""
    res, best = {}, {}
    for r in range(rounds + 1):
        for name, flags in (("old", OLD | extra_flags), ("new", extra_flags)):
            ctx.lib.odise_hip_gemm_debug(flags)
            if r == 0:
                for o in outs:   # a stale result of the other variant must not pass for this one's
                    ctx.lib.odise_hip_memset(ctx.h, C.c_void_p(o.ptr), 0xA5, C.c_size_t(o.nbytes))
            fn()
            ctx.sync()
            if r == 0:
                res[name] = [o.numpy().tobytes() for o in outs]
                continue
            ctx.timer_start()
            for _ in range(it):
                fn()
            best[name] = min(best.get(name, 1e9), ctx.timer_stop() / it)
    ctx.lib.odise_hip_gemm_debug(0)
    same = res["old"] == res["new"]
    if not same:
        a = np.frombuffer(res["old"][0], np.float16).astype(np.float32)
        b = np.frombuffer(res["new"][0], np.float16).astype(np.float32)
        bad.append(label)
        diff = f"DIFFERENT ({int((a != b).sum())} of {a.size} elements, max |d| {np.abs(a - b).max():.3g})"
    else:
        diff = "identical"
    print(f"{label:84s} old {best['old']*1e3:8.1f} us  new {best['new']*1e3:8.1f} us  {best['old']/best['new']:5.2f}x  {flop/(best['new']*1e-3)/1e12:7.1f} TF/s  {diff}", flush=True)
    return best["old"], best["new"]


# ---- part 1: every tile, every feature, ragged shapes -------------------------------------------------------------------------------
print("# part 1: correctness (bit identity) on ragged shapes, forced tiles")
M, N, K = 1000, 648, 192          # M not a multiple of any tile, N % 8 == 0 but ragged against every BN
A, W = f16((M, K)), f16((N, K), K ** -0.5)
feat = {"plain": {}, "bias+SiLU+res": dict(bias_n=f32((N,)), act=ACT_SILU, residual=f16((M, N))), "QuickGELU": dict(bias_n=f32((N,)), act=ACT_QUICKGELU),
        "GELU+res": dict(act=ACT_GELU, residual=f16((M, N))), "ReLU+bias_m+scale_m": dict(act=ACT_RELU, bias_m=f32((M,)), scale_m=f32((M,))),
        "rowgroup+bias": dict(bias_n=f32((N,)), rowgroup_add=f32((-(-M // 77), N)), rows_per_group=77), "GEGLU": dict(bias_n=f32((N,)), geglu=True)}
for tile in (0, 1, 2, 3, 4, 5, 6):
    for name, kw in feat.items():
        if quick and name not in ("bias+SiLU+res", "GEGLU"):
            continue
        O = ctx.empty((M, N // 2 if kw.get("geglu") else N), np.float16)
        ab(f"gemm tile {tile} {name} M={M} N={N} K={K}", lambda: ctx.gemm(A, W, force_tile=tile, out=O, **kw), [O], 2.0 * M * N * K, it=3, rounds=1)
        O.free()
for tile in (3, 4, 6, 7, 8, 5, 0):
    B, H, Wd, Cin, Cout = 2, 37, 41, 128, 264
    X, Wt, O = f16((B, H, Wd, Cin)), f16((Cout, 3, 3, Cin), (9 * Cin) ** -0.5), ctx.empty((B, H, Wd, Cout), np.float16)
    b, pia, r = f32((Cout,)), f32((B, Cout)), f16((B, H, Wd, Cout))
    ab(f"conv3x3 tile {tile} {B}x{H}x{Wd} {Cin}->{Cout} bias+per-image+SiLU+res", lambda: ctx.conv2d(X, Wt, bias=b, per_image_add=pia, residual=r, act=ACT_SILU, force_tile=tile, out=O),
       [O], 2.0 * B * H * Wd * Cout * 9 * Cin, it=3, rounds=1)
    for a in (X, Wt, O, r):
        a.free()
# conv + fused GroupNorm statistics (the VAE pair): conv output, normalised output
for tile, (B, H, Wd, Cin, Cout) in ((7, (2, 64, 64, 256, 256)), (8, (2, 64, 64, 128, 128)), (6, (2, 64, 64, 128, 128)), (4, (2, 64, 64, 256, 256))):
    X, Wt = f16((B, H, Wd, Cin)), f16((Cout, 3, 3, Cin), (9 * Cin) ** -0.5)
    gam, bet, b = f32((Cout,)), f32((Cout,)), f32((Cout,))
    keep = {}

    def run():
        y, yn, blocks = ctx.conv2d_gn(X, Wt, gam, bet, bias=b, act=ACT_SILU, force_tile=tile)
        for k in ("y", "yn"):
            if k in keep:
                keep[k].free()
        keep["y"], keep["yn"], keep["blocks"] = y, yn, blocks

    outs = {}
    for name, flags in (("old", OLD), ("new", 0)):
        ctx.lib.odise_hip_gemm_debug(flags)
        run()
        outs[name] = (keep["y"].numpy().tobytes(), keep["yn"].numpy().tobytes(), keep["blocks"])
    ctx.lib.odise_hip_gemm_debug(0)
    ok = outs["old"] == outs["new"]
    if not ok:
        bad.append(f"conv+gn tile {tile}")
    print(f"conv3x3 + fused GroupNorm statistics tile {tile} {B}x{H}x{Wd} {Cin}->{Cout}: stats blocks {outs['new'][2]}  {'identical' if ok else 'DIFFERENT'}", flush=True)

# ---- part 2: the shapes of the step --------------------------------------------------------------------------------------------------
print("# part 2: shapes of the benchmarked step (cost-model tile choice)")
cases = [("CLIP out-proj +res", 9232, 1024, 1024, dict(res=True)), ("CLIP fc1 QuickGELU", 9232, 4096, 1024, dict(act=ACT_QUICKGELU)),
         ("CLIP fc2 +res", 9232, 1024, 4096, dict(res=True)), ("CLIP q,k", 9232, 2048, 1024, {}), ("MaskCLIP fc1", 2708, 4096, 1024, dict(act=ACT_QUICKGELU)),
         ("UNet 64^2 proj", 65536, 320, 320, {}), ("UNet 64^2 GEGLU", 65536, 2560, 320, dict(geglu=True)), ("UNet 64^2 ff2 +res", 65536, 320, 1280, dict(res=True)),
         ("UNet 32^2 proj", 16384, 640, 640, {}), ("UNet 32^2 GEGLU", 16384, 5120, 640, dict(geglu=True)), ("UNet 16^2 proj +res", 4096, 1280, 1280, dict(res=True)),
         ("pixel decoder proj", 86016, 256, 256, {}), ("pixel decoder ffn1 ReLU", 86016, 1024, 256, dict(act=ACT_RELU)), ("pixel decoder ffn2 +res", 86016, 256, 1024, dict(res=True)),
         ("VAE 1x1 shortcut", 262144, 512, 256, {}), ("large reference", 65536, 1024, 4096, {})]
tot_old = tot_new = 0.0
for name, M, N, K, kw in cases:
    A, W, b = f16((M, K)), f16((N, K), K ** -0.5), f32((N,))
    O = ctx.empty((M, N // 2 if kw.get("geglu") else N), np.float16)
    r = f16((M, N)) if kw.get("res") else None
    o, n = ab(f"{name} M={M} N={N} K={K}", lambda: ctx.gemm(A, W, bias_n=b, residual=r, act=kw.get("act", ACT_NONE), geglu=bool(kw.get("geglu")), out=O), [O], 2.0 * M * N * K,
              it=5 if quick else 10, rounds=1 if quick else 3)
    tot_old, tot_new = tot_old + o, tot_new + n
    for a in (A, W, O, r):
        if a is not None:
            a.free()
convs = [("VAE 512->512 @128^2 x16 (dominant)", 16, 128, 128, 512, 512, True), ("VAE 128->128 @512^2 x16 +res", 16, 512, 512, 128, 128, True),
         ("VAE 256->256 @256^2 x16", 16, 256, 256, 256, 256, False), ("VAE 512->512 @64^2 x16 +res", 16, 64, 64, 512, 512, True),
         ("UNet 320->320 @64^2 x16 per-image", 16, 64, 64, 320, 320, False), ("UNet 640->640 @32^2 x16", 16, 32, 32, 640, 640, False),
         ("UNet 1280->1280 @16^2 x16", 16, 16, 16, 1280, 1280, False)]
for name, B, H, Wd, Cin, Cout, res in convs:
    X, Wt, O = f16((B, H, Wd, Cin)), f16((Cout, 3, 3, Cin), (9 * Cin) ** -0.5), ctx.empty((B, H, Wd, Cout), np.float16)
    b, pia = f32((Cout,)), f32((B, Cout))
    r = f16((B, H, Wd, Cout)) if res else None
    o, n = ab(f"conv3x3 {name}", lambda: ctx.conv2d(X, Wt, bias=b, per_image_add=None if res else pia, residual=r, out=O), [O], 2.0 * B * H * Wd * Cout * 9 * Cin,
              it=3 if quick else 5, rounds=1 if quick else 3)
    tot_old, tot_new = tot_old + o, tot_new + n
    for a in (X, Wt, O, r):
        if a is not None:
            a.free()
print(f"# sum over part 2 (one launch each): old {tot_old:.3f} ms  new {tot_new:.3f} ms")
print("# ALL IDENTICAL" if not bad else f"# MISMATCHES: {bad}")
sys.exit(1 if bad else 0)
