"""The five GEMMs of one CLIP block with folded LayerNorms (extractor.cpp clip_tower) at the benchmarked 16-crop shape (M = 16 x 584 rows), each
under the two epilogue forms of the same binary: the default wave-private item loop (round 6: the producers' row statistics ride in its lean form) and
the block-wide math-first form (`odise_hip_gemm_debug` 32768 << 4); plus the same shapes with a plain epilogue (no LayerNorm terms) as the yardstick.  Interleaved rounds, HIP-event timing on the context's stream.

    python tools/clip_gemm_bench.py [rounds=5] [M=9344]          (GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from odise_amd import _lib  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402

FORMS = (("wave-private", 0), ("block-wide", 32768))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 9344
    Wd = 1024
    ctx = Context(0)
    rng = np.random.default_rng(0)
    f16 = lambda *s, sc=1.0: ctx.to_device((rng.standard_normal(s, dtype=np.float32) * sc).astype(np.float16))   # noqa: E731
    f32 = lambda *s, sc=1.0: ctx.to_device((rng.standard_normal(s, dtype=np.float32) * sc).astype(np.float32))   # noqa: E731
    x, att, hid = f16(M, Wd), f16(M, Wd), f16(M, 4 * Wd)
    Wqk, Wv, Wo, Wfc, Wpr = f16(2 * Wd, Wd, sc=Wd ** -0.5), f16(Wd, Wd, sc=Wd ** -0.5), f16(Wd, Wd, sc=Wd ** -0.5), f16(4 * Wd, Wd, sc=Wd ** -0.5), f16(Wd, 4 * Wd, sc=(4 * Wd) ** -0.5)
    bqk, bv, bo, bfc, bpr = f32(2 * Wd, sc=0.1), f32(Wd, sc=0.1), f32(Wd, sc=0.1), f32(4 * Wd, sc=0.1), f32(Wd, sc=0.1)
    csqk, csv, csfc = f32(2 * Wd), f32(Wd), f32(4 * Wd)
    P = Wd // 64
    part = ctx.to_device(np.abs(rng.standard_normal((M, P, 2), dtype=np.float32)) * 64 + np.array([0.0, 200.0], np.float32))
    fin = ctx.empty((M, 2), np.float32)
    stats = ctx.empty((M, P, 2), np.float32)
    ln_c = dict(part=part, parts=P, inv_c=1.0 / Wd, eps=1e-5)
    outs = {"qk": ctx.empty((M, 2 * Wd)), "vt": ctx.empty((Wd, M)), "o": ctx.empty((M, Wd)), "fc": ctx.empty((M, 4 * Wd)), "pr": ctx.empty((M, Wd))}
    cases = [
        ("q|k  = LN(x) Wqk^T + b        (rows: part + colsum + final_out)", 2.0 * M * 2 * Wd * Wd,
         lambda ln: ctx.gemm(x, Wqk, bias_n=bqk, out=outs["qk"], ln=dict(ln_c, colsum=csqk, final_out=fin) if ln else None)),
        ("V^T  = Wv LN(x)^T + b         (swapped: fin + rowsum + bias_m)", 2.0 * M * Wd * Wd,
         lambda ln: ctx.gemm(Wv, x, bias_m=bv, out=outs["vt"], ln=dict(fin=fin, rowsum=csv) if ln else None)),
        ("x2   = x + att Wo^T + b       (residual + stats_out)", 2.0 * M * Wd * Wd,
         lambda ln: ctx.gemm(att, Wo, bias_n=bo, residual=x, out=outs["o"], ln=dict(stats_out=stats) if ln else None)),
        ("hid  = qgelu(LN(x2) Wfc^T + b) (rows: part + colsum, QuickGELU)", 2.0 * M * 4 * Wd * Wd,
         lambda ln: ctx.gemm(x, Wfc, bias_n=bfc, act=_lib.ACT_QUICKGELU, out=outs["fc"], ln=dict(ln_c, colsum=csfc) if ln else None)),
        ("x    = x2 + hid Wpr^T + b     (residual + stats_out)", 2.0 * M * Wd * 4 * Wd,
         lambda ln: ctx.gemm(hid, Wpr, bias_n=bpr, residual=x, out=outs["pr"], ln=dict(stats_out=stats) if ln else None)),
    ]

    def timed(fn, it=10):
        fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(it):
            fn()
        return ctx.timer_stop() / it * 1e3   # us

    res = {}
    for _ in range(rounds):
        for ci, (name, flops, fn) in enumerate(cases):
            for fname, flag in FORMS:
                ctx.lib.odise_hip_gemm_debug(flag << 4)
                res.setdefault((ci, fname), []).append(timed(lambda: fn(True)))
            ctx.lib.odise_hip_gemm_debug(0)
            res.setdefault((ci, "plain"), []).append(timed(lambda: fn(False)))
    print(f"M = {M} token rows, width {Wd}; median of {rounds} interleaved rounds of 10 launches; us (TFLOP/s)")
    tot = {f: 0.0 for f, _ in FORMS}
    tot["plain"] = 0.0
    for ci, (name, flops, _) in enumerate(cases):
        cells = []
        for fname in [f for f, _ in FORMS] + ["plain"]:
            us = float(np.median(res[(ci, fname)]))
            tot[fname] += us
            cells.append(f"{fname} {us:7.1f} ({flops / us / 1e6:6.0f})")
        print(f"  {name:66s} " + "   ".join(cells))
    print("  sum of the five (one block):" + "".join(f"   {k} {v:7.1f} us" for k, v in tot.items()) + f"   x 23 folded blocks: {tot['wave-private'] * 23 / 1e3:.2f} ms")
    ctx.close()


if __name__ == "__main__":
    main()
