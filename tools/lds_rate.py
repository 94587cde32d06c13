"""EXPERIMENT: the LDS port under the traffic of one GEMM K-tile (odise_amd/csrc/probe.hip: lds_rate_kernel) - clocks per round for the
64 KiB of LDS-DMA alone, the 24 fragment reads per thread alone, and both; 2048 clocks is what the MFMAs of that K-tile need."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
NAMES = {0: "8 x global_load_lds_dwordx4 per thread (64 KiB per round)", 1: "24 x ds_read_b128 per thread (192 KiB per round)", 2: "both",
         3: "4 loads (32 KiB, halo-like) + 24 reads", 4: "16 loads per thread (128 KiB per round)"}
cus = ctx.device_info()[1]
for blocks, what in ((cus, "one workgroup per CU"), (32, "32 workgroups")):
    for v in range(5):
        clk, ms = C.c_double(0), C.c_float(0)
        rc = ctx.lib.odise_hip_lds_rate(ctx.h, v, 2000, blocks, C.byref(clk), C.byref(ms))
        assert rc == 0, rc
        dma_bytes = {0: 65536, 2: 65536, 3: 32768, 4: 131072}.get(v, 0)
        extra = f"  DMA landing {dma_bytes / clk.value:6.1f} B/clk/CU" if dma_bytes else ""
        print(f"{what:22s} {NAMES[v]:58s}: {clk.value:8.1f} clk/round  ({ms.value*1e3/2000:6.2f} us){extra}", flush=True)
