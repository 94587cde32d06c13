import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from odise_amd.runtime import Context
ctx = Context(0)
rng = np.random.default_rng(0)
def uni(shape): return ctx.to_device(rng.uniform(-1.0, 1.0, size=shape).astype(np.float16))
for (M, N, K) in [(9472, 4096, 1024), (9472, 1024, 4096), (4096, 4096, 4096)]:
    A, W = uni((M, K)), uni((N, K)); O = ctx.empty((M, N), np.float16)
    flop = 2.0 * M * N * K
    def t(fn, it=20):
        fn(); ctx.sync(); ctx.timer_start()
        for _ in range(it): fn()
        return ctx.timer_stop() / it * 1e3
    res = {}
    for r in range(3):
        for name, fn in [("tool8p", lambda: ctx.lib.odise_hip_gemm8p(ctx.h, A, W, O, M, N, K, 0)),
                         ("g8 full", lambda: (ctx.lib.odise_hip_gemm_debug(0), ctx.gemm(A, W, force_tile=4, force_split=1, out=O))),
                         ("g8 main loop only", lambda: (ctx.lib.odise_hip_gemm_debug(4), ctx.gemm(A, W, force_tile=4, force_split=1, out=O))),
                         ("pp full", lambda: (ctx.lib.odise_hip_gemm_debug(1024 << 4), ctx.gemm(A, W, force_tile=4, force_split=1, out=O))),
                         ("pp main loop only", lambda: (ctx.lib.odise_hip_gemm_debug((1024 << 4) | 4), ctx.gemm(A, W, force_tile=4, force_split=1, out=O)))]:
            res[name] = min(res.get(name, 1e9), t(fn))
    ctx.lib.odise_hip_gemm_debug(0)
    print(f"{M}x{N}x{K}: " + " | ".join(f"{k} {v:7.1f} us {flop/v/1e6:6.0f} TF" for k, v in res.items()), flush=True)
