"""Generate committed golden fixtures from the REFERENCE itself (runs only in the build container).

The only piece of the reference's hot path that imports as-is here is
`ms_deform_attn_core_pytorch` (third_party/Mask2Former/mask2former/modeling/pixel_decoder/ops/functions/
ms_deform_attn_func.py:52-72; it needs nothing but torch).  It is loaded by file path from /root/reference with
the failing `import MultiScaleDeformableAttention` guarded by the reference's own try/except, run on

  * the exact inputs of the reference's ops/test.py:24-39 (torch.manual_seed(3), N,M,D=1,2,2, Lq,L,P=2,2,2,
    shapes (6,4),(3,2)) in float64 and float32, and
  * a few extra seeded cases (odd channel counts, locations outside [0,1], 3 levels x 4 points, fp32),

and the inputs + outputs are written to tests/golden/msda_*.npz.  /root/reference does not exist on the GPU box;
the tests only read the .npz files.

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/third_party/Mask2Former/mask2former/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py"
sys.path.insert(0, os.path.join(HERE, "..", ".."))


def load_reference_core():
    spec = importlib.util.spec_from_file_location("ref_ms_deform_attn_func", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # MSDA import fails -> the file's own except branch sets MSDA=None
    return mod.ms_deform_attn_core_pytorch


def case_ops_test(core, dtype):
    # mirrors ops/test.py:24-47 on CPU (the reference draws value, loc, weights in this order after manual_seed(3))
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    level_start_index = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    w = torch.rand(N, Lq, M, L, P) + 1e-5
    w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
    out = core(value.to(dtype), shapes, loc.to(dtype), w.to(dtype))
    return dict(value=value.numpy(), shapes=shapes.numpy(), start=level_start_index.numpy(), loc=loc.numpy(), w=w.numpy(),
                out=out.numpy())


def case_seeded(core, B, M, D, Lq, shapes, P, seed, loc_range, dtype=torch.float32):
    from oracle.msda import make_inputs
    value, shp, start, loc, w = make_inputs(B, M, D, Lq, shapes, P, seed, loc_range)
    out = core(value.to(dtype), shp, loc.to(dtype), w.to(dtype))
    return dict(value=value.numpy(), shapes=shp.numpy(), start=start.numpy(), loc=loc.numpy(), w=w.numpy(), out=out.numpy())


def main():
    core = load_reference_core()
    np.savez_compressed(os.path.join(HERE, "msda_ops_test_f64.npz"), **case_ops_test(core, torch.float64))
    np.savez_compressed(os.path.join(HERE, "msda_ops_test_f32.npz"), **case_ops_test(core, torch.float32))
    cases = {
        "msda_d32_3lvl": dict(B=1, M=8, D=32, Lq=37, shapes=[(6, 5), (12, 10), (24, 20)], P=4, seed=11, loc_range=(0.0, 1.0)),
        "msda_oob": dict(B=1, M=3, D=4, Lq=50, shapes=[(5, 7), (3, 2)], P=3, seed=12, loc_range=(-0.4, 1.4)),
        "msda_odd_d": dict(B=2, M=2, D=7, Lq=9, shapes=[(4, 6)], P=2, seed=13, loc_range=(-0.1, 1.1)),
    }
    for name, kw in cases.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **case_seeded(core, dtype=torch.float64, **kw))
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
