"""The 8-phase 256x256 kernel (gemm8_kernel, round 5) against what the cost model ran before it, on the step's heaviest GEMM / convolution shapes
(profiles/r04_gemm_efficiency_by_shape.txt, 16 crops / 4 pictures), same process, interleaved rounds, random operands:

    old    odise_hip_gemm_debug(32768 << 4): the current selection with the block-wide epilogues everywhere
    new    the current selection (wave-private epilogue where it applies)
    g8     tile 4 forced (8-phase 256x256 kernel, v_mfma_f32_16x16x32_f16);  g8m32: the same on v_mfma_f32_32x32x16_f16
    t6     tile 6 forced (8-phase 512x128 kernel), for N <= 256

Prints time (min over rounds), TFLOP/s, the tile | split-K each variant ran on, and max |difference| of `new` / `g8` against `old`.
    python tools/g8_shapes.py [gemm|conv|all]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd._lib import ACT_QUICKGELU, ACT_SILU  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
ctx = Context(0)
rng = np.random.default_rng(0)
NO_G8 = 32768 << 4   # "old" = the block-wide epilogues everywhere
G8 = 16384 << 4


def rand(shape, s=1.0):
    return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))


def bench(label, flop, fns, it, rounds=4):
    best, outs, tiles = {}, {}, {}
    for r in range(rounds + 1):
        for name, fn in fns:
            out = fn()
            tiles[name] = ctx.lib.odise_hip_last_tile()
            ctx.sync()
            ctx.timer_start()
            for _ in range(it):
                fn()
            ms = ctx.timer_stop() / it
            if r > 0:
                best[name] = min(best.get(name, 1e9), ms)
            if r == rounds:
                outs[name] = out.numpy().astype(np.float32)
    ctx.lib.odise_hip_gemm_debug(0)
    ref = outs[fns[0][0]]
    line = f"{label:52s}"
    for name, _ in fns:
        t = tiles[name]
        line += f" | {name} {best[name]*1e3:8.1f} us {flop/(best[name]*1e-3)/1e12:6.0f} TF t{t & 255}/s{t >> 8}"
        if name != fns[0][0]:
            line += f" d={np.abs(outs[name]-ref).max():.2g}"
    print(line, flush=True)


def gemm_case(M, N, K, batch=1, act=0, bias=True, residual=False, geglu=False):
    A = rand((batch, M, K) if batch > 1 else (M, K))
    W = rand((batch, N, K) if batch > 1 else (N, K), K ** -0.5)
    No = N // 2 if geglu else N
    O = ctx.empty((batch, M, No) if batch > 1 else (M, No), np.float16)
    b = ctx.to_device(rng.standard_normal(N).astype(np.float32)) if bias else None
    R = rand((M, No)) if residual else None
    kw = dict(bias_n=b, act=act, residual=R, geglu=geglu, out=O)

    def run(flags, tile):
        ctx.lib.odise_hip_gemm_debug(flags)
        return ctx.gemm(A, W, force_tile=tile, **kw)
    fns = [("old", lambda: run(NO_G8, -1)), ("new", lambda: run(0, -1)), ("g8", lambda: run(G8, 4)), ("g8m32", lambda: run(G8 | (8192 << 4), 4))]
    if N <= 256:
        fns.append(("t6", lambda: run(0, 6)))
    flop = 2.0 * batch * M * N * K
    bench(f"gemm {M}x{N}x{K} b{batch}{' act' if act else ''}{' res' if residual else ''}{' geglu' if geglu else ''}", flop, fns, max(3, int(4e12 / flop)))
    for a in (A, W, O, b, R):
        if a is not None:
            a.free()


def conv_case(B, H, W_, Cin, Cout, stride=1, act=0, residual=False):
    X = rand((B, H, W_, Cin))
    Wt = rand((Cout, 3, 3, Cin), (9 * Cin) ** -0.5)
    OH, OW = H // stride, W_ // stride
    O = ctx.empty((B, OH, OW, Cout), np.float16)
    b = ctx.to_device(rng.standard_normal(Cout).astype(np.float32))
    R = rand((B, OH, OW, Cout)) if residual else None
    kw = dict(bias=b, act=act, residual=R, out=O, stride=stride)
    if stride == 2:
        kw.update(pad=0, pad_tl=(0, 0), out_hw=(OH, OW))   # the VAE's F.pad(0, 1, 0, 1) + stride-2 conv

    def run(flags, tile):
        ctx.lib.odise_hip_gemm_debug(flags)
        return ctx.conv2d(X, Wt, force_tile=tile, **kw)
    fns = [("old", lambda: run(NO_G8, -1)), ("new", lambda: run(0, -1)), ("g8", lambda: run(G8, 4)), ("t4", lambda: run(0, 4))]
    if Cout <= 256:
        fns.append(("t6", lambda: run(0, 6)))
    if stride == 1 and Cin % 64 == 0:
        fns += [("t7", lambda: run(0, 7)), ("t9", lambda: run(0, 9))]
    flop = 2.0 * B * OH * OW * Cout * 9 * Cin
    bench(f"conv {B}x{H}x{W_} {Cin}->{Cout} s{stride}{' res' if residual else ''}", flop, fns, max(3, int(6e12 / flop)))
    for a in (X, Wt, O, b, R):
        if a is not None:
            a.free()


if what == "epi":      # what the epilogue's terms cost on the CLIP c_fc shape (the yardstick kernel of tools/gemm8p_bench.py: 83 us without any)
    gemm_case(9344, 4096, 1024, bias=False)
    gemm_case(9344, 4096, 1024)
    gemm_case(9344, 4096, 1024, act=ACT_QUICKGELU)
    gemm_case(9344, 4096, 1024, residual=True)
    gemm_case(9344, 1024, 4096, bias=False)
    gemm_case(9344, 1024, 4096, residual=True)
if what in ("gemm", "all"):
    gemm_case(9344, 4096, 1024, act=ACT_QUICKGELU)         # CLIP c_fc
    gemm_case(9344, 2048, 1024)                            # CLIP q|k
    gemm_case(9344, 1024, 4096, residual=True)             # CLIP c_proj
    gemm_case(9344, 1024, 1024, residual=True)             # CLIP out-proj
    gemm_case(1024, 9344, 1024)                            # CLIP V^T
    gemm_case(2720, 4096, 1024, act=ACT_QUICKGELU)         # MaskCLIP c_fc
    gemm_case(2720, 1024, 4096, residual=True)
    gemm_case(65536, 2560, 320, geglu=True)                # UNet 64^2 feed-forward in
    gemm_case(65536, 320, 1280, residual=True)             # UNet 64^2 feed-forward out
    gemm_case(16384, 5120, 640, geglu=True)
    gemm_case(16384, 640, 2560, residual=True)
    gemm_case(4096, 10240, 1280, geglu=True)
    gemm_case(4096, 1280, 5120, residual=True)
    gemm_case(65536, 1024, 512)
    gemm_case(4096, 4096, 512, batch=16, bias=False)       # VAE mid-block attention scores
    gemm_case(4096, 512, 4096, batch=16, bias=False)
    gemm_case(86016, 256, 1024, residual=True)             # pixel decoder FFN out
    gemm_case(65536, 512, 512)
if what in ("conv", "all"):
    conv_case(16, 128, 128, 512, 512)                      # the dominant launch
    conv_case(16, 128, 128, 512, 512, residual=True)
    conv_case(16, 64, 64, 512, 512, residual=True)
    conv_case(16, 256, 256, 256, 256, residual=True)
    conv_case(16, 256, 256, 128, 256)
    conv_case(16, 128, 128, 256, 512)
    conv_case(16, 512, 512, 128, 128)                      # N = 128: half of the 256-wide tile is padding
    conv_case(16, 256, 256, 256, 256, stride=2)            # VAE downsample (asymmetric padding)
    conv_case(16, 64, 64, 320, 320)                        # UNet levels
    conv_case(16, 32, 32, 640, 640)
    conv_case(16, 16, 16, 1280, 1280)
    conv_case(16, 32, 32, 1280, 640)
    conv_case(16, 64, 64, 640, 320)
