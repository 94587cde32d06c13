"""One eager (graph off) UNet step at crops=16 with the GEMM launch log on, preceded/followed by untimed steps, for joining the
log with a rocprofv3 kernel trace (tools/join_gemm_log.py).  Run as:
  ODISE_GEMM_FLAGS=32 rocprofv3 --kernel-trace --output-format csv -d out -o u -- python tools/unet_shape_prof.py 2> gemmlog.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402
from odise_amd.unet import HipUNet  # noqa: E402
from oracle.sd_unet import UNetModel, config2_inputs, init_synthetic_  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = Context(0)
hip = HipUNet(ctx, init_synthetic_(UNetModel(), seed=1234).state_dict())
hip.use_graph(False)
x, c, e = config2_inputs(B, 64)
dx, dc, de = ctx.to_device(x.numpy()), ctx.to_device(c.numpy()), ctx.to_device(e.numpy())
for i in range(3):
    sys.stderr.write(f"STEP {i}\n")
    sys.stderr.flush()
    hip.run_nhwc(dx, dc, de)
    ctx.sync()
