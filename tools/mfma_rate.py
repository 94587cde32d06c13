"""What the MFMA pipes sustain without operand traffic, under the synchronisation skeletons of the GEMM kernels
(odise_amd/csrc/probe.hip: mfma_rate_kernel).  Puts the 'pipe occupancy' of the real kernels in DESIGN.md into perspective."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
lib = ctx.lib
NAMES = {0: "free running, 2 waves/SIMD, 4 chains", 1: "free running, 1 wave/SIMD, 4 chains", 2: "ping-pong skeleton (2 barriers / 16 MFMAs, staggered)",
         3: "workgroup barrier every 16 MFMAs", 4: "workgroup barrier every 32 MFMAs", 5: "free running, 2 waves/SIMD, 8 chains",
         6: "free running, 1 wave/SIMD, 8 chains", 7: "RANDOM operands, free running, 2 waves/SIMD, 4 chains",
         8: "RANDOM operands, ping-pong skeleton", 9: "RANDOM operands, free running, 1 wave/SIMD, 8 chains",
         10: "16x16x32 MFMA, RANDOM operands, free running, 2 waves/SIMD", 11: "16x16x32 MFMA, RANDOM operands, ping-pong skeleton",
         12: "16x16x32 MFMA, RANDOM operands, free running, 1 wave/SIMD"}
cus = ctx.device_info()[1]
for rnd in range(2):
    for v in (0, 2, 6, 7, 8, 9, 10, 11, 12):
        ms, fl, mhz = C.c_float(0), C.c_double(0), C.c_double(0)
        rc = lib.odise_hip_mfma_rate(ctx.h, v, 2000, cus * 4, 10, C.byref(ms), C.byref(fl), C.byref(mhz))
        assert rc == 0, rc
        print(f"round {rnd} {NAMES[v]:58s}: {ms.value*1e3:8.1f} us/launch  {fl.value/(ms.value*1e-3)/1e12:7.1f} TFLOP/s  shader clock {mhz.value:6.0f} MHz",
              flush=True)
