"""Time the UNet step at several crop batch sizes in one process (saves model init on the GPU box).
usage: python tools/unet_sweep.py 1,4,16 [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402
from odise_amd.unet import HipUNet  # noqa: E402
from oracle.sd_unet import UNetModel, config2_inputs, init_synthetic_  # noqa: E402

crops = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1,4,16").split(",")]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = Context(0)
hip = HipUNet(ctx, init_synthetic_(UNetModel(), seed=1234).state_dict())
for graph in (False, True):
    hip.use_graph(graph)
    for B in crops:
        x, c, e = config2_inputs(B, 64)
        dx, dc, de = ctx.to_device(x.numpy()), ctx.to_device(c.numpy()), ctx.to_device(e.numpy())
        for _ in range(3):
            hip.run_nhwc(dx, dc, de)
        ctx.sync()
        ctx.timer_start()
        for _ in range(steps):
            hip.run_nhwc(dx, dc, de)
        ms = ctx.timer_stop() / steps
        print(f"graph={int(graph)} crops={B:3d} {ms:8.3f} ms/step {B / ms * 1e3:8.1f} crops/s  {0.7401e12 * B / (ms * 1e-3) / 1e12:7.1f} TF/s "
              f"({0.7401e12 * B / (ms * 1e-3) / 2.5e15 * 100:5.2f}% of 2.5 PF)", flush=True)
