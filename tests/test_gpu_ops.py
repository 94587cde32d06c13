"""GPU parity tests of the op-level C ABI (libodise_hip.so) against the CPU oracle / plain torch fp32.

Tolerances: fp32 ops (MSDeformAttn fp32) 1e-5 relative; fp16-input MFMA ops are compared with an fp32
computation on the SAME fp16-rounded inputs, so the only differences are accumulation order and the final fp16
rounding of the output: rtol 2e-3 + atol scaled to the output magnitude (SURVEY.md §8c allows 2e-2/2e-3 per stage).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from odise_amd import _lib
from oracle.msda import make_inputs, msda_forward_torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def h(x):  # fp16-round a torch tensor, keep fp32
    return x.half().float()


def close(got, ref, rtol=2e-3, atol=None, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = np.abs(ref).max() + 1e-12
    atol = atol if atol is not None else 2e-3 * scale
    err = np.abs(got - ref)
    bad = err > atol + rtol * np.abs(ref)
    if bad.any() or not np.isfinite(got).all():
        idx = np.argwhere(bad | ~np.isfinite(got))[:5]
        raise AssertionError(f"{what}: {bad.sum()}/{bad.size} mismatches, max err {err.max():.4g} (scale {scale:.4g}); first {idx.tolist()} "
                             f"got {[got[tuple(i)] for i in idx]} ref {[ref[tuple(i)] for i in idx]}")


# ---------------------------------------------------------------------------------------------------------------
# MSDeformAttn
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["msda_ops_test_f64", "msda_ops_test_f32", "msda_oob", "msda_odd_d", "msda_d32_3lvl"])
def test_msda_golden(ctx, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = ctx.ms_deform_attn_forward(ctx.to_device(g["value"], np.float32), g["shapes"], g["start"], ctx.to_device(g["loc"], np.float32),
                                     ctx.to_device(g["w"], np.float32), im2col_step=2 if "ops_test" in name else 128).numpy()
    # the reference's own fp32 tolerance (ops/test.py:57) is rtol 1e-2 / atol 1e-3; we hold fp32 round-off
    np.testing.assert_allclose(out, g["out"], rtol=1e-4, atol=1e-7)


def test_msda_production_shape_fp32_and_fp16(ctx):
    shapes = [(32, 32), (64, 64), (128, 128)]
    Lq = sum(a * b for a, b in shapes)  # 21504 queries (1024^2 image)
    value, shp, start, loc, w = make_inputs(1, 8, 32, Lq, shapes, 4, seed=21, loc_range=(-0.05, 1.05), value_scale=1.0)
    ref = msda_forward_torch(value.double(), shp, start, loc, w).numpy()
    out = ctx.ms_deform_attn_forward(ctx.to_device(value), shp.numpy(), start.numpy(), ctx.to_device(loc), ctx.to_device(w)).numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5)
    v16 = value.half()
    ref16 = msda_forward_torch(v16.double(), shp, start, loc, w).numpy()
    out16 = ctx.ms_deform_attn_forward(ctx.to_device(v16.numpy()), shp.numpy(), start.numpy(), ctx.to_device(loc), ctx.to_device(w)).numpy()
    close(out16, ref16, rtol=2e-3, what="msda fp16")


def test_msda_linearity_and_empty(ctx):
    value, shp, start, loc, w = make_inputs(2, 8, 32, 300, [(16, 16), (8, 8)], 4, seed=22, value_scale=1.0)
    v2 = torch.rand_like(value)
    f = lambda v: ctx.ms_deform_attn_forward(ctx.to_device(v), shp.numpy(), start.numpy(), ctx.to_device(loc), ctx.to_device(w)).numpy()
    np.testing.assert_allclose(f(value + 3 * v2), f(value) + 3 * f(v2), rtol=1e-4, atol=1e-5)
    # empty query set
    e = ctx.ms_deform_attn_forward(ctx.to_device(value), shp.numpy(), start.numpy(), ctx.empty((2, 0, 8, 2, 4, 2), np.float32),
                                   ctx.empty((2, 0, 8, 2, 4), np.float32))
    assert e.shape == (2, 0, 256)


def test_maskclip_visibility_rows_two_forms_and_torch(ctx):
    """MaskCLIP's attention-mask rows (clip.py:288-318): mask logits resized bilinearly to the 336^2 CLIP input, max-pooled over the 14 x 14
    patches, visible iff sigmoid >= 0.5.  `maskclip_token_mask_kernel` walks columns and reuses a source row's horizontal interpolant for the
    sample rows that share it; `plain` interpolates anew for every sample row: the same expression evaluated once or twice - the rows must agree
    to the bit.  Against torch's interpolate + max_pool2d: every entry whose pooled logit is not within rounding of zero."""
    g = torch.Generator().manual_seed(5)
    B, Q, hh, ww, S, patch = 2, 7, 64, 48, 336, 14
    T = (S // patch) ** 2 + 1
    ldm = (T + 7) // 8 * 8
    coarse = torch.randn(B, Q, 8, 6, generator=g) * 3.0
    logits = h(F.interpolate(coarse, size=(hh, ww), mode="bicubic", align_corners=False) + 0.3 * torch.randn(B, Q, hh, ww, generator=g))
    dl = ctx.to_device(logits.half().numpy())
    rows = {}
    for plain in (0, 1):
        out = ctx.empty((B, T + Q, ldm), np.uint8)
        _lib.check(ctx.lib.odise_hip_maskclip_token_mask(ctx.h, dl, out, B, Q, hh, ww, S, patch, T, ldm, plain), "maskclip_token_mask")
        rows[plain] = out.numpy()
    assert np.array_equal(rows[0], rows[1]), "reusing a source row's interpolant changed a visibility bit"
    got = rows[0]
    assert (got[:, :T, :T] == 0).all() and (got[:, :, T:] == 1).all() and (got[:, T:, 0] == 0).all()
    up = F.interpolate(logits.float(), size=(S, S), mode="bilinear", align_corners=False)
    pooled = F.max_pool2d(up, patch, patch).reshape(B, Q, -1)
    ref = (torch.sigmoid(pooled) < 0.5).numpy().astype(np.uint8)
    sure = (pooled.abs() > 1e-3).numpy()
    mism = (got[:, T:, 1:T] != ref) & sure
    print(f"visibility rows: {int(ref.sum())} of {ref.size} patches hidden; mismatches outside |logit| < 1e-3: {int(mism.sum())}")
    assert mism.sum() == 0 and 0.05 < ref.mean() < 0.95


def test_msda_fused_gather_against_prepare_plus_native_op(ctx):
    """msda_fused_kernel (round 6: softmax + sampling locations + branch-free gather, XCD-aware block order) against msda_prepare_kernel + the
    native-op kernel on the same raw projections, offsets large enough that a fifth of the points leave the maps (zero padding outside
    (-1, H) x (-1, W), ms_deform_im2col_cuda.cuh:38-89) - and both against the fp64 restatement of the reference op."""
    import ctypes as C
    from oracle.msda import msda_forward_torch
    g = torch.Generator().manual_seed(11)
    B, M, hs, ws = 2, 8, (8, 16, 32), (8, 16, 32)
    Lq = sum(h_ * w_ for h_, w_ in zip(hs, ws))
    value = h(torch.randn(B, Lq, M, 32, generator=g))
    off = torch.randn(B * Lq, M * 12 * 2, generator=g) * 6.0
    aw = torch.randn(B * Lq, M * 12, generator=g) * 2.0
    dv, do, da = ctx.to_device(value.half().numpy()), ctx.to_device(off.numpy()), ctx.to_device(aw.numpy())
    loc_s, w_s = ctx.empty((B * Lq * M * 24,), np.float32), ctx.empty((B * Lq * M * 12,), np.float32)
    hs3, ws3 = (C.c_int * 3)(*hs), (C.c_int * 3)(*ws)
    outs = {}
    for fused in (2, 1, 0):
        o = ctx.empty((B * Lq, M * 32), np.float16)
        _lib.check(ctx.lib.odise_hip_msda_fused_forward(ctx.h, dv, do, da, hs3, ws3, B, M, fused, o, loc_s, w_s), "msda_fused_forward")
        outs[fused] = o.numpy().astype(np.float64)
    # the reference op on the same locations / weights (fp64)
    starts = np.cumsum([0] + [a * b for a, b in zip(hs, ws)])[:3]
    q = torch.arange(Lq)
    lvl = (q[:, None] >= torch.as_tensor(starts)[None]).sum(1) - 1
    local = q - torch.as_tensor(starts)[lvl]
    Wl, Hl = torch.as_tensor(ws)[lvl], torch.as_tensor(hs)[lvl]
    ref_xy = torch.stack([((local % Wl).double() + 0.5) / Wl, ((local // Wl).double() + 0.5) / Hl], -1)               # [Lq, 2]
    norm = torch.tensor([[w_, h_] for h_, w_ in zip(hs, ws)], dtype=torch.float64)                                    # [3, 2] = (W, H)
    loc = ref_xy[None, :, None, None, None, :] + off.double().view(B, Lq, M, 3, 4, 2) / norm[None, None, None, :, None, :]
    wts = torch.softmax(aw.double().view(B, Lq, M, 12), -1).view(B, Lq, M, 3, 4)
    shp = torch.tensor([[h_, w_] for h_, w_ in zip(hs, ws)])
    ref = msda_forward_torch(value.double(), shp, torch.as_tensor(starts), loc, wts).numpy().reshape(B * Lq, M * 32)
    outside = float(((loc < 0) | (loc > 1)).any(-1).double().mean())
    d = max(np.abs(outs[2] - outs[0]).max(), np.abs(outs[1] - outs[0]).max())
    print(f"msda fused vs two-kernel: max diff {d:.3e}; vs fp64 reference: fused {np.abs(outs[2] - ref).max():.3e} two-kernel {np.abs(outs[0] - ref).max():.3e}; "
          f"points outside [0, 1]: {outside:.2f}")
    assert outside > 0.1
    for k in (2, 1, 0):
        close(outs[k], ref, rtol=2e-3, atol=2e-3, what=f"msda form {k} vs fp64 reference")
    assert np.array_equal(outs[2], outs[1])      # the two lane widths of the fused kernel: the same sums
    assert d <= 2e-3, d      # fused and two-kernel forms round their fp32 sums to fp16 once: they may differ in the last fp16 bit of an output, not more


def test_msda_bad_im2col_step_raises(ctx):
    value, shp, start, loc, w = make_inputs(3, 2, 4, 5, [(4, 4)], 2, seed=23)
    with pytest.raises(RuntimeError):
        ctx.ms_deform_attn_forward(ctx.to_device(value), shp.numpy(), start.numpy(), ctx.to_device(loc), ctx.to_device(w), im2col_step=2)


# ---------------------------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------------------------
GEMM_CASES = [
    # M, N, K, tile, split
    (128, 128, 64, 0, 0), (256, 384, 320, 0, 0), (200, 136, 72, 0, 0),
    (64, 128, 128, 1, 0), (100, 100, 256, 1, 0), (64, 64, 64, 2, 0), (37, 24, 40, 2, 0),
    (64, 1280, 2304, 2, 6), (77, 320, 768, 1, 3), (4096, 320, 320, -1, 0), (1, 1280, 320, -1, 0),
    (300, 250, 1000, -1, 0),
    # 8-wave 256-row tiles (3: 256x320, 4: 256x256, 5: 256x128), incl. ragged edges and split-K
    (512, 640, 320, 3, 0), (300, 330, 128, 3, 0), (512, 512, 256, 4, 0), (260, 130, 72, 5, 0), (1024, 320, 2880, 3, 3),
    (256, 320, 64, 3, 0), (4096, 1280, 320, -1, 0), (700, 300, 200, 4, 2),
]


@pytest.mark.parametrize("M,N,K,tile,split", GEMM_CASES)
def test_gemm_plain(ctx, M, N, K, tile, split):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = h(torch.randn(M, K, generator=g))
    W = h(torch.randn(N, K, generator=g) / K ** 0.5)
    ref = A @ W.t()
    out = ctx.gemm(ctx.to_device(A.half().numpy()), ctx.to_device(W.half().numpy()), force_tile=tile, force_split=split).numpy()
    close(out, ref.numpy(), what=f"gemm {M}x{N}x{K} tile{tile} split{split}")
    out32 = ctx.gemm(ctx.to_device(A.half().numpy()), ctx.to_device(W.half().numpy()), out_dtype=np.float32, force_tile=tile,
                     force_split=split).numpy()
    close(out32, ref.numpy(), rtol=1e-4, atol=1e-4 * float(ref.abs().max()), what="gemm f32 out")


# ---- the 8-phase kernels (gemm8_kernel, round 5: tiles 4 = 256x256 and 6 = 512x128 wherever K % 64 == 0) against the ping-pong kernels
# that run those tiles by default (the 8-phase kernels are opt-in: odise_hip_gemm_debug 16384 << 4).  Every main loop of csrc/gemm.hip multiplies with
# v_mfma_f32_16x16x32_f16 and takes the k-steps of a K-tile in the same order: the same bits.  The 8-phase 256x256 kernel also exists on
# v_mfma_f32_32x32x16_f16 (8192 << 4, the A/B form of tools/g8_shapes.py: another rounding sequence, the same error bound): held to the
# fp32 reference and to the default form within one fp16 step of the largest output.
# Ragged M / N (zero-line rows), one to many K-tiles with odd and even counts (the tail of one to three K-tiles), split-K, batched,
# every epilogue family.
G8, G8_M32, PP = 16384 << 4, (16384 | 8192) << 4, 0


@pytest.mark.parametrize("M,N,K,split,batch,tile", [(256, 256, 64, 1, 1, 4), (256, 256, 128, 1, 1, 4), (512, 512, 192, 1, 1, 4), (300, 330, 256, 1, 1, 4),
                                                    (1000, 264, 320, 1, 1, 4), (2720, 1024, 1024, 1, 1, 4), (512, 256, 1536, 3, 1, 4), (700, 300, 448, 2, 1, 4),
                                                    (256, 512, 512, 1, 3, 4), (9344, 1024, 1024, 1, 1, 4),
                                                    (512, 128, 64, 1, 1, 6), (1100, 136, 320, 1, 1, 6), (4096, 128, 1152, 1, 1, 6), (1024, 256, 512, 2, 1, 6)])
def test_gemm8_against_pingpong_and_reference(ctx, M, N, K, split, batch, tile):
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + split)
    shp = (lambda r, c: (batch, r, c)) if batch > 1 else (lambda r, c: (r, c))
    A = h(torch.randn(*shp(M, K), generator=g))
    W = h(torch.randn(*shp(N, K), generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    res = h(torch.randn(M, N, generator=g))
    dA, dW, db = ctx.to_device(A.half().numpy()), ctx.to_device(W.half().numpy()), ctx.to_device(bias)
    dr = ctx.to_device(res.half().numpy()) if batch == 1 else None
    outs = {}
    try:
        for name, flags in (("g8", G8), ("g8m32", G8_M32), ("pp", PP)):
            ctx.lib.odise_hip_gemm_debug(flags)
            outs[name] = [ctx.gemm(dA, dW, bias_n=db, act=_lib.ACT_SILU, residual=dr, force_tile=tile, force_split=split).numpy(),
                          ctx.gemm(dA, dW, out_dtype=np.float32, force_tile=tile, force_split=split).numpy()]
            if batch == 1 and N % 16 == 0:
                outs[name].append(ctx.gemm(dA, dW, bias_n=db, geglu=True, force_tile=tile, force_split=split).numpy())
    finally:
        ctx.lib.odise_hip_gemm_debug(0)
    ref = (A @ W.transpose(-1, -2)).numpy()
    close(outs["g8"][1], ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()), what=f"gemm8 {M}x{N}x{K} tile {tile} f32 out")
    for a, b in zip(outs["g8"], outs["pp"]):
        assert np.array_equal(a, b), f"gemm8 differs bitwise from the ping-pong kernel at {M}x{N}x{K} tile {tile} split {split} batch {batch}"
    for a, b in zip(outs["g8m32"], outs["g8"]):     # (tile 6 has no 32x32x16 form: the flag leaves it as it is)
        a, b = a.astype(np.float32), b.astype(np.float32)
        assert np.abs(a - b).max() <= 2.0 ** -10 * max(np.abs(b).max(), 1.0), f"gemm8 on 32x32x16 vs 16x16x32 at {M}x{N}x{K} tile {tile}: {np.abs(a - b).max()}"


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,split,tile", [(2, 16, 16, 128, 256, 3, 1, 1, 4), (1, 33, 17, 64, 300, 3, 1, 1, 4), (2, 32, 32, 192, 256, 3, 1, 2, 4),
                                                                (1, 32, 32, 128, 256, 1, 1, 1, 4), (2, 32, 32, 64, 512, 3, 2, 1, 4), (1, 24, 40, 320, 320, 3, 1, 1, 4),
                                                                (2, 32, 32, 128, 128, 3, 1, 1, 6), (1, 33, 17, 64, 136, 3, 1, 1, 6), (1, 64, 64, 128, 128, 3, 2, 1, 6)])
def test_conv_gemm8_against_pingpong_and_reference(ctx, N, H, W, Cin, Cout, k, stride, split, tile):
    g = torch.Generator().manual_seed(N + H + Cin + Cout + k)
    x = h(torch.randn(N, H, W, Cin, generator=g))
    w = h(torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5)
    b = torch.randn(Cout, generator=g)
    dx, dw, db = ctx.to_device(x.half().numpy()), ctx.to_device(w.half().numpy()), ctx.to_device(b)
    outs = {}
    try:
        for name, flags in (("g8", G8), ("g8m32", G8_M32), ("pp", PP)):
            ctx.lib.odise_hip_gemm_debug(flags)
            outs[name] = ctx.conv2d(dx, dw, bias=db, stride=stride, act=_lib.ACT_SILU, force_tile=tile, force_split=split).numpy()
    finally:
        ctx.lib.odise_hip_gemm_debug(0)
    ref = F.silu(_conv_ref(x, w, stride, k // 2, b)).numpy()
    close(outs["g8"], ref, what=f"conv gemm8 {N}x{H}x{W}x{Cin}->{Cout} k{k} s{stride} tile {tile}")
    assert np.array_equal(outs["g8"], outs["pp"]), "gemm8 conv differs bitwise from the ping-pong kernel"
    assert np.abs(outs["g8m32"].astype(np.float32) - outs["g8"].astype(np.float32)).max() <= 2.0 ** -10 * max(np.abs(ref).max(), 1.0)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6])
def test_gemm_repeated_launches_agree(ctx, tile):
    """Every tile on a many-block problem (several residency rounds, several blocks per CU for the small tiles), launched 12 times into a
    buffer pre-filled with a sentinel: every launch must write every element and reproduce the first launch bit for bit (a race in an epilogue's
    staging shows up as a few elements that differ from launch to launch), and the first must match the fp32 reference."""
    M, N, K = 4096, 1280, 320
    g = torch.Generator().manual_seed(11 + tile)
    A = h(torch.randn(M, K, generator=g))
    W = h(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    res = h(torch.randn(M, N, generator=g))
    dA, dW, db, dr = ctx.to_device(A.half().numpy()), ctx.to_device(W.half().numpy()), ctx.to_device(bias), ctx.to_device(res.half().numpy())
    ref = (A @ W.t() + bias + res).numpy()
    first = None
    O = ctx.empty((M, N), np.float16)
    for rep in range(12):
        O.copy_from(np.full((M, N), 777.0, np.float16))
        got = ctx.gemm(dA, dW, bias_n=db, residual=dr, force_tile=tile, force_split=1, out=O).numpy()
        if first is None:
            first = got
            close(first, ref, what=f"tile {tile} at {M}x{N}x{K}")
        else:
            diff = np.argwhere(got != first)
            assert diff.size == 0, f"tile {tile}: launch {rep} differs from the first in {len(diff)} elements, e.g. {diff[:4].tolist()} -> {got[tuple(diff[0])]} vs {first[tuple(diff[0])]}"


@pytest.mark.parametrize("tile,M,N,K", [(4, 1000, 768, 256), (4, 2304, 1024, 1024), (5, 1536, 384, 320), (6, 2048, 256, 512), (3, 1024, 640, 320)])
def test_wave_private_epilogue_stress_against_the_block_wide_forms(ctx, tile, M, N, K):
    """the wave-private epilogue (every 8-wave GEMM kernel's default since round 5) against the block-wide forms it replaced
    (`odise_hip_gemm_debug` 32768 << 4), bit for bit, over random epilogue configurations and repeated launches - the 4-wave kernels showed rare
    zeros in a staged dword with this code (they keep the block-wide form); nothing of the kind was ever seen with 8 waves, and this is the test
    that would see it: 5 tiles x 6 configurations x 8 launches x up to 2.4 M elements each."""
    g = torch.Generator().manual_seed(tile * 1000 + M)
    A = ctx.to_device(h(torch.randn(M, K, generator=g)).half().numpy())
    W = ctx.to_device(h(torch.randn(N, K, generator=g) / K ** 0.5).half().numpy())
    bias = ctx.to_device(torch.randn(N, generator=g).numpy())
    res = ctx.to_device(h(torch.randn(M, N, generator=g)).half().numpy())
    configs = [dict(), dict(bias_n=bias), dict(bias_n=bias, act=_lib.ACT_SILU), dict(residual=res), dict(bias_n=bias, residual=res, act=_lib.ACT_QUICKGELU),
               dict(bias_n=bias, out_dtype=np.float32)]
    try:
        for kw in configs:
            ctx.lib.odise_hip_gemm_debug(32768 << 4)
            ref = ctx.gemm(A, W, force_tile=tile, force_split=1, **kw).numpy()
            ctx.lib.odise_hip_gemm_debug(0)
            for rep in range(8):
                out = ctx.gemm(A, W, force_tile=tile, force_split=1, **kw).numpy()
                assert (ctx.lib.odise_hip_last_tile() & 255) == tile
                bad = np.flatnonzero(out.reshape(-1) != ref.reshape(-1))
                assert bad.size == 0, (tile, sorted(kw), rep, bad[:8].tolist(), out.reshape(-1)[bad[:8]].tolist(), ref.reshape(-1)[bad[:8]].tolist())
    finally:
        ctx.lib.odise_hip_gemm_debug(0)


def test_gemm_asymmetric_identity(ctx):
    # transpose-detecting check (guide rule 16): A = I, W asymmetric -> C = W^T
    n = 128
    A = torch.eye(n)
    W = torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97 - 48 + torch.arange(n).view(n, 1) * 0.25
    W = h(W)
    out = ctx.gemm(ctx.to_device(A.half().numpy()), ctx.to_device(W.half().numpy()), out_dtype=np.float32).numpy()
    np.testing.assert_allclose(out, W.t().numpy(), rtol=0, atol=1e-3)


@pytest.mark.parametrize("tile,split", [(0, 0), (2, 0), (2, 4), (3, 0), (4, 2)])
def test_gemm_epilogues(ctx, tile, split):
    g = torch.Generator().manual_seed(5)
    M, N, K = 192, 160, 320
    A = h(torch.randn(M, K, generator=g))
    W = h(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    bias_m = torch.randn(M, generator=g)
    res = h(torch.randn(M, N, generator=g))
    rga = torch.randn(3, N, generator=g)  # rows_per_group = 64
    dA, dW = ctx.to_device(A.half().numpy()), ctx.to_device(W.half().numpy())
    base = A @ W.t()
    kw = dict(force_tile=tile, force_split=split)
    # bias + silu + residual
    out = ctx.gemm(dA, dW, bias_n=ctx.to_device(bias), act=_lib.ACT_SILU, residual=ctx.to_device(res.half().numpy()), **kw).numpy()
    close(out, (F.silu(base + bias) + res).numpy(), what="bias+silu+res")
    # time-embedding style broadcast + bias_m, alpha
    out = ctx.gemm(dA, dW, bias_m=ctx.to_device(bias_m), rowgroup_add=ctx.to_device(rga), rows_per_group=64, alpha=0.5, **kw).numpy()
    ref = 0.5 * base + bias_m[:, None] + rga.repeat_interleave(64, 0)
    close(out, ref.numpy(), what="rowgroup+bias_m+alpha")
    # GEGLU: interleaved (a, gate) columns
    out = ctx.gemm(dA, dW, bias_n=ctx.to_device(bias), geglu=True, **kw).numpy()
    t = base + bias
    close(out, (t[:, 0::2] * F.gelu(t[:, 1::2])).numpy(), what="geglu")
    # quickgelu / gelu / relu + scale_m
    sm = torch.rand(M, generator=g) + 0.5
    for act, fn in ((_lib.ACT_QUICKGELU, lambda x: x * torch.sigmoid(1.702 * x)), (_lib.ACT_GELU, F.gelu), (_lib.ACT_RELU, F.relu)):
        out = ctx.gemm(dA, dW, act=act, scale_m=ctx.to_device(sm), **kw).numpy()
        close(out, fn(base * sm[:, None]).numpy(), what=f"act{act}")


@pytest.mark.parametrize("M", [584, 2336, 33280])   # 33280 rows = 520 tiles > 2 x 256 CUs: the first-generation ping-pong kernel carries the fold there (MaskCLIP from 13 pictures, 57 crops)
def test_gemm_chain_with_folded_layer_norm(ctx, M):
    """The CLIP tower's GEMM chain without LayerNorm kernels (extractor.cpp clip_tower, common.h LnEpi): the GEMM that writes the residual stream
    leaves per-row partial sums, the next GEMMs read the raw stream with the affine folded into their weights - against LayerNorm + GEMM in fp32
    on the same fp16 stream (the fold itself is exact algebra; what differs is one fp16 rounding of LN(x) that the folded form does not make)."""
    g = torch.Generator().manual_seed(M)
    Cw, N2 = 1024, 2048
    att = h(torch.randn(M, Cw, generator=g))
    Wo = h(torch.randn(Cw, Cw, generator=g) / Cw ** 0.5)
    bo = torch.randn(Cw, generator=g) * 0.1
    x = h(torch.randn(M, Cw, generator=g) * 2 + 0.7)
    x[:, 5] += 40.0        # a massive-activation channel, as the CLIP residual stream has
    x = h(x)
    gamma = torch.rand(Cw, generator=g) + 0.5
    beta = torch.randn(Cw, generator=g) * 0.2
    W1 = torch.randn(N2, Cw, generator=g) / Cw ** 0.5
    b1 = torch.randn(N2, generator=g) * 0.1
    Wv = torch.randn(Cw, Cw, generator=g) / Cw ** 0.5
    bv = torch.randn(Cw, generator=g) * 0.1
    parts = Cw // 64    # csrc/common.h kLnPartCols
    # producer: x2 = x + att Wo^T + bo, with the row statistics of the rounded x2
    stats = ctx.empty((M, parts, 2), np.float32)
    x2 = ctx.gemm(ctx.to_device(att.half().numpy()), ctx.to_device(Wo.half().numpy()), bias_n=ctx.to_device(bo),
                  residual=ctx.to_device(x.half().numpy()), ln=dict(stats_out=stats))
    x2h = torch.from_numpy(x2.numpy().astype(np.float32))
    close(x2h.numpy(), (x + att @ Wo.t() + bo).numpy(), what="producer output")
    st = stats.numpy().astype(np.float64)
    blocks = x2h.double().view(M, parts, 64)
    np.testing.assert_allclose(st[..., 0], blocks.sum(-1).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(st[..., 1], (blocks ** 2).sum(-1).numpy(), rtol=1e-5, atol=1e-3)
    # consumer: quick_gelu(LN(x2) W1^T + b1) from the raw x2
    W1f = (W1 * gamma).half()
    cs1 = W1f.float().sum(1)
    b1f = b1 + W1 @ beta
    fin = ctx.empty((M, 2), np.float32)
    ln = dict(part=stats, parts=parts, inv_c=1.0 / Cw, eps=1e-5, colsum=ctx.to_device(cs1), final_out=fin)
    y = ctx.gemm(x2, ctx.to_device(W1f.numpy()), bias_n=ctx.to_device(b1f), act=_lib.ACT_QUICKGELU, ln=ln).numpy()
    nrm = F.layer_norm(x2h, (Cw,), gamma, beta, 1e-5)
    t = nrm @ W1.t() + b1
    close(y, (t * torch.sigmoid(1.702 * t)).numpy(), rtol=4e-3, what="folded LN consumer")
    mean, var = x2h.double().mean(1), x2h.double().var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    f = fin.numpy().astype(np.float64)
    np.testing.assert_allclose(f[:, 1], rstd.numpy(), rtol=2e-4)
    np.testing.assert_allclose(f[:, 0], (-mean * rstd).numpy(), rtol=2e-4, atol=2e-4)
    # swapped consumer: V^T = Wv LN(x2)^T + bv[:, None] (the normalised tokens are the rows of the W operand)
    Wvf = (Wv * gamma).half()
    csv = Wvf.float().sum(1)
    bvf = bv + Wv @ beta
    vt = ctx.gemm(ctx.to_device(Wvf.numpy()), x2, bias_m=ctx.to_device(bvf), ln=dict(fin=fin, rowsum=ctx.to_device(csv))).numpy()
    close(vt, (Wv @ nrm.t() + bv[:, None]).numpy(), rtol=4e-3, what="folded LN swapped consumer")


def test_gemm_batched_and_swapped_vt(ctx):
    # the V^T production trick: Vt[b] = Wv @ X[b]^T  ==  gemm(A=Wv, W=X[b])
    g = torch.Generator().manual_seed(9)
    B, L, Cc = 3, 77, 320
    X = h(torch.randn(B, L, 768, generator=g))
    Wv = h(torch.randn(Cc, 768, generator=g) / 768 ** 0.5)
    out = ctx.gemm(ctx.to_device(Wv.half().numpy()), ctx.to_device(X.half().numpy())).numpy()
    ref = torch.einsum("ck,blk->bcl", Wv, X)
    close(out, ref.numpy(), what="batched swapped")


def test_gemm_rejects_bad_k(ctx):
    with pytest.raises(RuntimeError):
        ctx.gemm(ctx.zeros((8, 12)), ctx.zeros((8, 12)))


# ---------------------------------------------------------------------------------------------------------------
# Convolution (NHWC implicit GEMM)
# ---------------------------------------------------------------------------------------------------------------
def _conv_ref(x_nhwc, w_okkc, stride, padding, bias=None):
    x = x_nhwc.permute(0, 3, 1, 2)
    w = w_okkc.permute(0, 3, 1, 2)
    return F.conv2d(x, w, bias, stride=stride, padding=padding).permute(0, 2, 3, 1)


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,tile,split", [
    (2, 16, 16, 32, 64, 3, -1, 0), (1, 64, 64, 320, 320, 3, -1, 0), (1, 8, 8, 1280, 640, 3, -1, 0), (3, 9, 7, 8, 24, 3, 2, 0),
    (1, 32, 32, 64, 128, 1, -1, 0), (2, 8, 8, 640, 320, 3, 2, 5), (1, 24, 40, 128, 136, 3, 0, 0),
    (2, 32, 32, 64, 320, 3, 3, 0), (1, 17, 19, 32, 300, 3, 3, 2), (2, 16, 16, 128, 256, 3, 4, 0), (1, 20, 20, 64, 128, 3, 5, 0),
    (4, 64, 64, 320, 320, 3, -1, 0),
])
def test_conv3x3_and_1x1(ctx, N, H, W, Cin, Cout, k, tile, split):
    g = torch.Generator().manual_seed(N + H + Cin + Cout)
    x = h(torch.randn(N, H, W, Cin, generator=g))
    w = h(torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5)
    b = torch.randn(Cout, generator=g)
    ref = _conv_ref(x, w, 1, k // 2, b)
    out = ctx.conv2d(ctx.to_device(x.half().numpy()), ctx.to_device(w.half().numpy()), bias=ctx.to_device(b), force_tile=tile,
                     force_split=split).numpy()
    close(out, ref.numpy(), what=f"conv{k}x{k} {N}x{H}x{W}x{Cin}->{Cout}")


# the halo kernels (3x3 / stride 1 / pad 1, Cin % 64 == 0; tile ids 7: 16x16 pixels x 256 channels, 8 / 9: 16x16 x 128), forced: whole patches,
# ragged image sizes (partial patches in both directions), several 64-channel chunks, split-K over whole chunks, a Cout that is not a
# multiple of the column tile
@pytest.mark.parametrize("N,H,W,Cin,Cout,tile,split", [
    (2, 32, 32, 128, 128, 8, 0), (1, 64, 48, 128, 128, 8, 0), (2, 40, 24, 128, 128, 8, 0), (1, 33, 17, 64, 128, 8, 0), (1, 16, 16, 192, 256, 8, 0),
    (1, 32, 32, 256, 128, 8, 2), (1, 8, 8, 128, 72, 8, 0), (3, 96, 32, 128, 128, 8, 0), (1, 33, 17, 192, 128, 8, 3),
    (1, 40, 24, 128, 256, 7, 0), (1, 16, 48, 256, 512, 7, 2), (2, 32, 32, 64, 320, 7, 0),
    # tile 9: the 128-channel tile as four waves, two co-resident blocks per CU (one halo buffer, refilled at the chunk boundaries)
    (2, 32, 32, 128, 128, 9, 0), (2, 40, 24, 128, 128, 9, 0), (1, 33, 17, 64, 128, 9, 0), (1, 16, 16, 192, 256, 9, 0), (1, 32, 32, 256, 128, 9, 2),
    (1, 8, 8, 128, 72, 9, 0), (3, 96, 32, 128, 128, 9, 0), (1, 33, 17, 192, 128, 9, 3), (4, 128, 128, 128, 128, 9, 0),
])
def test_conv3x3_halo_tiles(ctx, N, H, W, Cin, Cout, tile, split):
    g = torch.Generator().manual_seed(N + H + W + Cin + Cout + tile)
    x = h(torch.randn(N, H, W, Cin, generator=g))
    w = h(torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5)
    b = torch.randn(Cout, generator=g)
    res = h(torch.randn(N, H, W, Cout, generator=g))
    ref = _conv_ref(x, w, 1, 1, b) + res
    dx, dw, db, dr = ctx.to_device(x.half().numpy()), ctx.to_device(w.half().numpy()), ctx.to_device(b), ctx.to_device(res.half().numpy())
    out = ctx.conv2d(dx, dw, bias=db, residual=dr, force_tile=tile, force_split=split).numpy()   # split 0: the cost model's own choice
    close(out, ref.numpy(), what=f"halo tile {tile} conv {N}x{H}x{W}x{Cin}->{Cout} split {split}")
    # same fp32 summation order (chunk-major) in both halo tiles: without split-K they are bit-identical
    one = ctx.conv2d(dx, dw, bias=db, residual=dr, force_tile=tile, force_split=1).numpy()
    twin = 8 if tile == 9 else 15 - tile
    other = ctx.conv2d(dx, dw, bias=db, residual=dr, force_tile=twin, force_split=1).numpy()
    assert np.array_equal(one, other), f"tile {tile} differs bitwise from tile {twin}"


@pytest.mark.parametrize("N,H,W,Cin,Cout,tile", [(2, 64, 64, 128, 128, 8), (1, 40, 24, 128, 128, 8), (1, 48, 32, 128, 256, 7), (1, 33, 20, 64, 512, 7),
                                                 (1, 64, 32, 128, 128, 6), (1, 32, 32, 128, 256, 4), (2, 64, 64, 128, 128, 9), (1, 40, 24, 128, 128, 9)])
def test_conv_fused_groupnorm_statistics(ctx, N, H, W, Cin, Cout, tile):
    """The conv -> GroupNorm pair of the ResBlocks (ldm ResnetBlock: conv1 -> norm2 -> swish): the conv epilogue reduces the statistics,
    the GroupNorm only finalises and applies.  Compared with torch conv2d -> group_norm -> silu."""
    g = torch.Generator().manual_seed(H + W + Cout + tile)
    x = h(torch.randn(N, H, W, Cin, generator=g))
    w = h(torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5 * 2.0)
    b = torch.randn(Cout, generator=g)
    gamma, beta = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    y, yn, blocks = ctx.conv2d_gn(ctx.to_device(x.half().numpy()), ctx.to_device(w.half().numpy()), ctx.to_device(gamma), ctx.to_device(beta),
                                  bias=ctx.to_device(b), eps=1e-6, act=1, force_tile=tile, force_split=1)
    assert blocks > 0, "the forced kernel declined the statistics fusion"
    ref = _conv_ref(x, w, 1, 1, b)
    close(y.numpy(), ref.numpy(), what=f"conv (tile {tile})")
    y16 = torch.from_numpy(y.numpy().astype(np.float32))   # GroupNorm of the f16 tensor the conv actually wrote
    refn = F.silu(F.group_norm(y16.permute(0, 3, 1, 2), 32, gamma, beta, eps=1e-6)).permute(0, 2, 3, 1)
    close(yn.numpy(), refn.numpy(), rtol=3e-3, what=f"group_norm from fused statistics (tile {tile}, {blocks} row blocks)")


@pytest.mark.parametrize("N,H,W", [(2, 64, 64), (1, 16, 48), (3, 32, 8), (1, 512, 512)])
def test_conv_in_kernel_for_8_channel_images(ctx, N, H, W):
    """AutoencoderKL's encoder.conv_in (3 -> 128 channels, input padded to 8; ldm.py:556-560) runs `conv3_c8_kernel` (csrc/conv_c8.hip): MFMA
    operands loaded straight from the NHWC image, GroupNorm statistics reduced in the epilogue.  Against torch, against the implicit GEMM of the
    same library (forced tile: sums of 27 products in another order - fp32 rounding before the fp16 store), and the GroupNorm that reads the
    statistics against torch's on the tensor the kernel wrote.  Borders (zero padding), several row blocks, a one-block image."""
    g = torch.Generator().manual_seed(N + H + W)
    x = torch.zeros(N, H, W, 8)
    x[..., :3] = h(torch.randn(N, H, W, 3, generator=g))
    w = torch.zeros(128, 3, 3, 8)
    w[..., :3] = h(torch.randn(128, 3, 3, 3, generator=g) / 27 ** 0.5 * 2.0)
    b = torch.randn(128, generator=g)
    gamma, beta = torch.randn(128, generator=g), torch.randn(128, generator=g)
    dx, dw, db = ctx.to_device(x.half().numpy()), ctx.to_device(w.half().numpy()), ctx.to_device(b)
    ref = _conv_ref(x, w, 1, 1, b)
    out = ctx.conv2d(dx, dw, bias=db).numpy()
    close(out, ref.numpy(), what=f"conv_in kernel {N}x{H}x{W}")
    gen = ctx.conv2d(dx, dw, bias=db, force_tile=1).numpy()
    ndiff = int((out != gen).sum())
    print(f"conv_in kernel vs the implicit GEMM (64 x 128 tile): {ndiff} of {out.size} outputs differ")
    assert ndiff == 0, "the conv_in kernel feeds the MFMAs the same k groups in the same order as the implicit GEMM: the outputs must agree to the bit"
    y, yn, blocks = ctx.conv2d_gn(dx, dw, ctx.to_device(gamma), ctx.to_device(beta), bias=db, eps=1e-6, act=1)
    assert blocks == H * W // 256, "the conv_in kernel did not leave its statistics"
    assert np.array_equal(y.numpy(), out), "the statistics epilogue changed the convolution's output"
    y16 = torch.from_numpy(y.numpy().astype(np.float32))
    refn = F.silu(F.group_norm(y16.permute(0, 3, 1, 2), 32, gamma, beta, eps=1e-6)).permute(0, 2, 3, 1)
    close(yn.numpy(), refn.numpy(), rtol=3e-3, what=f"group_norm from the conv_in kernel's statistics ({blocks} row blocks)")


def test_conv_stride2_variants(ctx):
    g = torch.Generator().manual_seed(3)
    x = h(torch.randn(2, 16, 16, 64, generator=g))
    w = h(torch.randn(96, 3, 3, 64, generator=g) / 24.0)
    dx, dw = ctx.to_device(x.half().numpy()), ctx.to_device(w.half().numpy())
    # UNet Downsample: conv3x3 stride 2 pad 1
    close(ctx.conv2d(dx, dw, stride=2, pad=1).numpy(), _conv_ref(x, w, 2, 1).numpy(), what="stride2 pad1")
    # VAE encoder Downsample: F.pad (0,1,0,1) then conv3x3 stride 2 pad 0
    xp = F.pad(x.permute(0, 3, 1, 2), (0, 1, 0, 1)).permute(0, 2, 3, 1)
    ref = _conv_ref(xp, w, 2, 0)
    close(ctx.conv2d(dx, dw, stride=2, pad_tl=(0, 0), out_hw=(8, 8)).numpy(), ref.numpy(), what="stride2 asym pad")


def test_conv_fused_upsample_residual_timeemb(ctx):
    g = torch.Generator().manual_seed(4)
    x = h(torch.randn(2, 8, 8, 64, generator=g))
    w = h(torch.randn(64, 3, 3, 64, generator=g) / 24.0)
    b = torch.randn(64, generator=g)
    dx, dw = ctx.to_device(x.half().numpy()), ctx.to_device(w.half().numpy())
    up = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
    close(ctx.conv2d(dx, dw, upsample2x=True, bias=ctx.to_device(b)).numpy(), _conv_ref(up, w, 1, 1, b).numpy(), what="fused upsample")
    res = h(torch.randn(2, 8, 8, 64, generator=g))
    temb = torch.randn(2, 64, generator=g)
    out = ctx.conv2d(dx, dw, bias=ctx.to_device(b), residual=ctx.to_device(res.half().numpy()), per_image_add=ctx.to_device(temb)).numpy()
    ref = _conv_ref(x, w, 1, 1, b) + temb[:, None, None, :] + res
    close(out, ref.numpy(), what="conv + temb + residual")


@pytest.mark.parametrize("N,H,W,Cin,Cout,tile", [(2, 16, 16, 128, 256, 7), (1, 20, 12, 64, 128, 8), (2, 24, 40, 128, 128, 9), (1, 32, 32, 512, 512, -1)])
def test_conv_fused_upsample_on_the_halo_tiles(ctx, N, H, W, Cin, Cout, tile):
    """Upsample = nearest 2x + 3x3 conv (ldm Upsample, ldm.py:493-533 / 469-491) on the halo kernels (round 6): the 18 x 18 patch of the UPSAMPLED
    input is gathered from the half-size source (pixel (y, x) <- (y / 2, x / 2)), everything after is the plain halo convolution.  Odd patch
    counts, borders, and the cost model's own choice (tile -1: at 512 -> 512 it must now be a halo tile, not the un-pipelined implicit GEMM)."""
    g = torch.Generator().manual_seed(N + H + W + Cin + Cout)
    x = h(torch.randn(N, H, W, Cin, generator=g))
    w = h(torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5)
    b = torch.randn(Cout, generator=g)
    up = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
    ref = _conv_ref(up, w, 1, 1, b).numpy()
    dx, dw, db = ctx.to_device(x.half().numpy()), ctx.to_device(w.half().numpy()), ctx.to_device(b)
    out = ctx.conv2d(dx, dw, upsample2x=True, bias=db, force_tile=tile).numpy()
    ran = ctx.lib.odise_hip_last_tile() & 255
    assert ran in (7, 8, 9) and (tile < 0 or ran == tile), (tile, ran)
    close(out, ref, what=f"fused-upsample halo tile {ran}")
    plain = ctx.conv2d(dx, dw, upsample2x=True, bias=db, force_tile=0).numpy()      # the implicit-GEMM form of the same convolution
    assert (ctx.lib.odise_hip_last_tile() & 255) == 0
    close(out, plain, rtol=2e-3, what="halo vs implicit GEMM with the fused upsample")


# ---------------------------------------------------------------------------------------------------------------
# Norms
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,HW,C,act", [(2, 64, 320, 1), (1, 4096, 320, 0), (3, 256, 128, 1), (1, 64, 2560, 1), (2, 1024, 960, 2),
                                        (1, 100, 512, 0), (2, 32768, 128, 1), (1, 16384, 512, 0), (16, 1024, 640, 1)])
def test_group_norm(ctx, N, HW, C, act):
    g = torch.Generator().manual_seed(C + HW)
    x = h(torch.randn(N, HW, C, generator=g) * 2 + 0.5)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, eps=1e-5).permute(0, 2, 1)
    ref = {0: lambda t: t, 1: F.silu, 2: F.relu}[act](ref)
    out = ctx.group_norm(ctx.to_device(x.half().numpy()), ctx.to_device(gamma), ctx.to_device(beta), 32, 1e-5, act).numpy()
    close(out, ref.numpy(), rtol=3e-3, what=f"group_norm C={C}")


@pytest.mark.parametrize("HW,C", [(256, 256), (40000, 256)])
def test_group_norm_residual_and_accumulate(ctx, HW, C):
    """odise_hip_group_norm_ex: y = relu(GroupNorm(x) + residual) + accum (the BottleneckBlock tails of the tap projections)."""
    import ctypes as C_
    g = torch.Generator().manual_seed(HW)
    N = 2
    x, r, a = (h(torch.randn(N, HW, C, generator=g)) for _ in range(3))
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.relu(F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, eps=1e-5).permute(0, 2, 1) + r) + a
    dx, dr, da = (ctx.to_device(t.half().numpy()) for t in (x, r, a))
    dg, db = ctx.to_device(gamma), ctx.to_device(beta)
    y = ctx.empty((N, HW, C), np.float16)
    rc = ctx.lib.odise_hip_group_norm_ex(ctx.h, dx.ptr, y.ptr, dg.ptr, db.ptr, N, HW, C, 32, C_.c_float(1e-5), 2, dr.ptr, da.ptr)
    assert rc == 0, ctx.lib.odise_hip_last_error()
    close(y.numpy(), ref.numpy(), rtol=3e-3, atol=3e-3, what=f"group_norm_ex HW={HW}")


@pytest.mark.parametrize("rows,C", [(77, 768), (4096, 320), (5, 1280), (577, 1024), (100, 256), (20003, 256), (301, 128), (9, 64)])   # C <= 256: two rows per wavefront
def test_layer_norm(ctx, rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = h(torch.randn(rows, C, generator=g) * 3 + 1)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    out = ctx.layer_norm(ctx.to_device(x.half().numpy()), ctx.to_device(gamma), ctx.to_device(beta), 1e-5).numpy()
    close(out, ref.numpy(), rtol=3e-3, what=f"layer_norm C={C}")


# ---------------------------------------------------------------------------------------------------------------
# Attention
# ---------------------------------------------------------------------------------------------------------------
def _vt(V, ldvt):  # [B,Lk,HD] -> [B,HD,ldvt] with NaN-poisoned padding (the kernel must ignore it)
    B, Lk, HD = V.shape
    out = np.full((B, HD, ldvt), np.nan, dtype=np.float16)
    out[:, :, :Lk] = V.transpose(0, 2, 1)
    return out


@pytest.mark.parametrize("B,H,Lq,Lk,D", [(1, 8, 256, 256, 40), (2, 8, 64, 64, 160), (1, 8, 1024, 1024, 80), (2, 8, 300, 77, 40),
                                         (1, 16, 577, 577, 64), (2, 8, 100, 1000, 32), (1, 2, 130, 65, 64), (1, 8, 4096, 4096, 40)])
def test_attention(ctx, B, H, Lq, Lk, D):
    g = torch.Generator().manual_seed(B + H + Lq + Lk + D)
    HD = H * D
    Q, K, V = (h(torch.randn(B, L, HD, generator=g)) for L in (Lq, Lk, Lk))
    scale = D ** -0.5
    q = Q.view(B, Lq, H, D).transpose(1, 2)
    k = K.view(B, Lk, H, D).transpose(1, 2)
    v = V.view(B, Lk, H, D).transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(B, Lq, HD)
    ldvt = (Lk + 7) // 8 * 8
    out = ctx.attention(ctx.to_device(Q.half().numpy()), ctx.to_device(K.half().numpy()), ctx.to_device(_vt(V.half().numpy(), ldvt)), H, scale,
                        Lk=Lk).numpy()
    close(out, ref.numpy(), rtol=5e-3, atol=3e-3, what=f"attention B{B} H{H} Lq{Lq} Lk{Lk} D{D}")


@pytest.mark.parametrize("B,L,D", [(1, 4096, 40), (4, 1024, 80), (2, 2048, 64), (16, 256, 48)])
def test_attention_pipelined_self_attention_is_bit_identical_to_the_tiled_kernel(ctx, B, L, D):
    """attn_sa_kernel (round 6: S^T of tile t+1 issued before the softmax of tile t, K double- / V^T triple-buffered, one barrier per tile) takes
    the unmasked self-attention launches made of whole 128-query / 128-key blocks that fill the chip - the SD UNet's 64^2 / 32^2 levels.  Same
    arithmetic per score in the same order as the tiled kernel: bit-identical outputs, also through the online-softmax rescale (a late key with a
    huge score) and with d_head below the padded MFMA depth; and against fp32 torch."""
    g = torch.Generator().manual_seed(B * 7 + L + D)
    H = 8
    HD = H * D
    Q, K, V = (h(torch.randn(B, L, HD, generator=g)) for _ in range(3))
    K[:, L - 70] = Q[:, 5] * 3         # forces a rescale in the last-but-one tile
    scale = D ** -0.5
    q, k, v = (t.view(B, L, H, D).transpose(1, 2) for t in (Q, K, V))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(B, L, HD)
    dq, dk, dvt = ctx.to_device(Q.half().numpy()), ctx.to_device(K.half().numpy()), ctx.to_device(_vt(V.half().numpy(), L))
    assert ctx.get_option(ctx.OPT_ATTN_KV_RESIDENT) == 0
    out = ctx.attention(dq, dk, dvt, H, scale).numpy()
    ctx.set_option(ctx.OPT_ATTN_KV_RESIDENT, 4)
    try:
        tiled = ctx.attention(dq, dk, dvt, H, scale).numpy()
    finally:
        ctx.set_option(ctx.OPT_ATTN_KV_RESIDENT, 0)
    assert np.array_equal(out, tiled), f"pipelined and tiled attention differ (max {np.abs(out.astype(np.float32) - tiled.astype(np.float32)).max()})"
    close(out, ref.numpy(), rtol=5e-3, atol=3e-3, what=f"pipelined self-attention B{B} L{L} D{D}")


@pytest.mark.parametrize("Lk", [333, 2100])  # 2100 keys: the split-KV path (few query blocks, many key tiles) + combine kernel
def test_attention_masked_and_spiked(ctx, Lk):
    g = torch.Generator().manual_seed(77)
    B, H, Lq, D = 2, 8, 100, 32
    HD = H * D
    Q, K, V = (h(torch.randn(B, L, HD, generator=g)) for L in (Lq, Lk, Lk))
    # force the online-softmax rescale path: one key with a huge score in a late tile
    K[:, 200] = Q[:, 5] * 4
    mask = torch.rand(B, Lq, Lk, generator=g) < 0.5
    mask[:, 3, :] = True      # a fully masked row -> output 0 (the caller un-masks such rows, odise.py:683)
    mask[:, 7, :] = False
    mask[:, 7, 64:] = True    # only the first tile visible
    scale = D ** -0.5
    q = Q.view(B, Lq, H, D).transpose(1, 2)
    k = K.view(B, Lk, H, D).transpose(1, 2)
    v = V.view(B, Lk, H, D).transpose(1, 2)
    s = (q @ k.transpose(-1, -2) * scale).masked_fill(mask[:, None], float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)
    ref = (p @ v).transpose(1, 2).reshape(B, Lq, HD)
    ldm = (Lk + 3) // 4 * 4
    m8 = np.zeros((B, Lq, ldm), dtype=np.uint8)
    m8[:, :, :Lk] = mask.numpy()
    ldvt = (Lk + 7) // 8 * 8
    out = ctx.attention(ctx.to_device(Q.half().numpy()), ctx.to_device(K.half().numpy()), ctx.to_device(_vt(V.half().numpy(), ldvt)), H, scale,
                        mask=ctx.to_device(m8), Lk=Lk).numpy()
    close(out, ref.numpy(), rtol=5e-3, atol=3e-3, what="masked attention")


@pytest.mark.parametrize("B,H,Lq,Lk,masked", [(16, 16, 577, 577, False),     # the crops' CLIP tower of a 4-picture step: one block per (head, image) = 256 blocks
                                              (16, 16, 677, 577, True),      # MaskCLIP's shape on 16 pictures: 100 mask tokens after the image tokens, u8 visibility
                                              (32, 8, 300, 608, True), (16, 16, 1200, 257, False),   # the key-count limits of the resident form
                                              (2, 16, 577, 577, False), (4, 16, 677, 577, True)])    # too few (head, image) pairs: these stay on the tiled kernel
def test_attention_kv_resident(ctx, B, H, Lq, Lk, masked):
    """d_head 64 with <= 608 keys and enough (head, image) pairs runs the K/V-resident kernel (attn.hip attn_kvres_kernel): against fp32 torch,
    and against the tiled kernel on the same inputs (per-context option ODISE_OPT_ATTN_KV_RESIDENT = 2)."""
    g = torch.Generator().manual_seed(B * 1000 + Lq + Lk)
    D = 64
    HD = H * D
    Q, K, V = (h(torch.randn(B, L, HD, generator=g)) for L in (Lq, Lk, Lk))
    K[:, min(200, Lk - 1)] = Q[:, 5] * 3        # a spike in a late key tile: the online-softmax rescale path
    scale = D ** -0.5
    q = Q.view(B, Lq, H, D).transpose(1, 2)
    k = K.view(B, Lk, H, D).transpose(1, 2)
    v = V.view(B, Lk, H, D).transpose(1, 2)
    s = q @ k.transpose(-1, -2) * scale
    dm = None
    if masked:
        mask = torch.rand(B, Lq, Lk, generator=g) < 0.4
        mask[:, 3, :] = True                       # fully masked row -> 0
        mask[:, 7, :] = False
        mask[:, 7, 32:] = True                     # only the first key tile visible
        s = s.masked_fill(mask[:, None], float("-inf"))
        ldm = (Lk + 3) // 4 * 4
        m8 = np.zeros((B, Lq, ldm), dtype=np.uint8)
        m8[:, :, :Lk] = mask.numpy()
        dm = ctx.to_device(m8)
    ref = (torch.nan_to_num(torch.softmax(s, -1), nan=0.0) @ v).transpose(1, 2).reshape(B, Lq, HD)
    ldvt = (Lk + 7) // 8 * 8
    dq, dk, dv = ctx.to_device(Q.half().numpy()), ctx.to_device(K.half().numpy()), ctx.to_device(_vt(V.half().numpy(), ldvt))
    assert ctx.get_option(ctx.OPT_ATTN_KV_RESIDENT) == 0
    out = ctx.attention(dq, dk, dv, H, scale, mask=dm, Lk=Lk).numpy()
    close(out, ref.numpy(), rtol=5e-3, atol=3e-3, what=f"kv-resident attention B{B} H{H} Lq{Lq} Lk{Lk} masked={masked}")
    ctx.set_option(ctx.OPT_ATTN_KV_RESIDENT, 2)
    try:
        tiled = ctx.attention(dq, dk, dv, H, scale, mask=dm, Lk=Lk).numpy()
    finally:
        ctx.set_option(ctx.OPT_ATTN_KV_RESIDENT, 0)
    diff = float(np.abs(out.astype(np.float32) - tiled.astype(np.float32)).max())
    print(f"kv-resident vs tiled attention B{B} H{H} Lq{Lq} Lk{Lk}: max abs diff {diff:.2e}")
    assert diff < 2e-3, diff                       # same products, same fp32 softmax; the running max steps per 32 keys instead of 64


# ---------------------------------------------------------------------------------------------------------------
# Layout helpers and MaskPooling
# ---------------------------------------------------------------------------------------------------------------
def test_layout_roundtrip_and_concat(ctx):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 16, 12, generator=g)
    y = ctx.nchw_to_nhwc_f16(ctx.to_device(x), 8)
    yn = y.numpy()
    np.testing.assert_array_equal(yn[..., :4], x.permute(0, 2, 3, 1).half().numpy())
    assert (yn[..., 4:] == 0).all()
    back = ctx.nhwc_to_nchw_f32(y).numpy()
    np.testing.assert_array_equal(back[:, :4], x.half().float().numpy())
    a = torch.randn(3, 5, 16, generator=g).half()
    b = torch.randn(3, 5, 24, generator=g).half()
    c = ctx.concat_channels(ctx.to_device(a.numpy()), ctx.to_device(b.numpy())).numpy()
    np.testing.assert_array_equal(c, torch.cat([a, b], -1).numpy())


@pytest.mark.parametrize("B,C,Q,H,W", [(2, 256, 100, 32, 32), (1, 256, 100, 128, 128), (1, 64, 7, 8, 9)])
def test_mask_pooling(ctx, B, C, Q, H, W):
    # MaskPooling.forward, odise/modeling/meta_arch/odise.py:937-963
    g = torch.Generator().manual_seed(B + C + Q + H)
    x = torch.randn(B, C, H, W, generator=g)
    mask = torch.randn(B, Q, H, W, generator=g)
    mask[:, 0] = -1.0  # empty mask -> zeros (denominator 1e-8)
    m = (mask.sigmoid() > 0.5).to(mask.dtype)
    denorm = m.sum(dim=(-1, -2), keepdim=True) + 1e-8
    ref = torch.einsum("bchw,bqhw->bqc", x, m / denorm)
    if (H * W) % 8:
        with pytest.raises(RuntimeError):
            ctx.mask_pooling(ctx.to_device(x), ctx.to_device(mask))
        return
    out = ctx.mask_pooling(ctx.to_device(x), ctx.to_device(mask)).numpy()
    close(out, ref.numpy(), rtol=2e-3, atol=2e-3, what="mask_pooling")
