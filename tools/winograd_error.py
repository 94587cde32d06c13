"""Feasibility of a Winograd F(2x2, 3x3) form of the VAE's 3x3 convolutions (queued item a) - NUMERICS ONLY, on the CPU.
The device computes convolutions with fp16 operands and fp32 accumulation (MFMA).  A Winograd kernel would feed the MFMA with TRANSFORMED
operands: U = G g G^T (per filter, offline, rounded to fp16) and V = B^T d B (per 4x4 input tile, computed in fp32 from the fp16 activations,
rounded to fp16 for the matrix cores), accumulate the 16 element-wise products over Cin in fp32 and apply A^T . A in fp32.  This script measures,
on a seeded 512 -> 512 convolution at 64x64 with activations shaped like the VAE's (GroupNorm + SiLU output), the error of
  (1) the direct form with fp16 operands / fp32 accumulation (what the halo kernel does), and
  (2) the Winograd form as described,
both against the fp64 convolution of the SAME fp16-rounded inputs and weights, as a fraction of max |reference|.
usage: winograd_error.py [channels=256] [size=64]"""
import sys

import torch
import torch.nn.functional as F

torch.manual_seed(0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64

G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def h(t):
    return t.half().double()


def winograd(x16, w16, round_operands=True):
    """x16 [1,C,S,S], w16 [O,C,3,3] (values already fp16-representable, dtype float64); pad 1, stride 1."""
    O = w16.shape[0]
    U = torch.einsum("ij,ocjk,lk->ocil", G, w16, G)                       # [O,C,4,4]
    if round_operands:
        U = h(U)
    xp = F.pad(x16, (1, 1, 1, 1))
    T = S // 2
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                            # [1,C,T,T,4,4]
    V = torch.einsum("ij,nctujk,lk->nctuil", BT, tiles, BT)               # B^T d B
    if round_operands:
        V = h(V)
    M = torch.einsum("ocil,nctuil->notuil", U.float().double(), V)        # fp32-accumulated on the device; fp64 here bounds it from below
    Y = torch.einsum("ij,notujk,lk->notuil", AT, M, AT)                   # [1,O,T,T,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(1, O, S, S)


def main():
    # activations: GroupNorm + SiLU of a smooth field plus noise (zero-mean, heavy right tail, like the VAE's); weights: N(0, 1 / fan_in)
    x = torch.randn(1, C, S, S, dtype=torch.float64)
    x = F.avg_pool2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1) * 1.5 + 0.3 * torch.randn(1, C, S, S, dtype=torch.float64)
    x = F.silu(F.group_norm(x, 32))
    w = torch.randn(C, C, 3, 3, dtype=torch.float64) / (9 * C) ** 0.5
    x16, w16 = h(x), h(w)
    ref = F.conv2d(x16, w16, padding=1)                                   # fp64 on the fp16-rounded operands
    direct = F.conv2d(x16.float(), w16.float(), padding=1).double()       # fp32 accumulation
    wino = winograd(x16, w16, True)
    wino_exact = winograd(x16, w16, False)                                # transform algebra only (no operand rounding): must be ~1e-15
    scale = ref.abs().max()
    out16 = lambda t: (h(t) - ref).abs().max() / scale                    # incl. the fp16 rounding of the stored output
    print(f"conv {C}->{C} @ {S}x{S}, max|ref| {scale:.3f}; errors as a fraction of max|ref|:")
    print(f"  direct, fp16 operands / fp32 accumulate          {((direct - ref).abs().max() / scale):.3e}   (stored as fp16: {out16(direct):.3e})")
    print(f"  Winograd F(2x2,3x3), exact operands (algebra)     {((wino_exact - ref).abs().max() / scale):.3e}")
    print(f"  Winograd F(2x2,3x3), U and V rounded to fp16      {((wino - ref).abs().max() / scale):.3e}   (stored as fp16: {out16(wino):.3e})")
    print(f"  rms: direct {((direct - ref).pow(2).mean().sqrt() / scale):.3e}, Winograd {((wino - ref).pow(2).mean().sqrt() / scale):.3e}; the fp16 rounding of the output alone: "
          f"{((h(ref) - ref).abs().max() / scale):.3e} max, {((h(ref) - ref).pow(2).mean().sqrt() / scale):.3e} rms")


if __name__ == "__main__":
    main()
