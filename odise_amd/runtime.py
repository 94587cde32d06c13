"""Host-side runtime over the C ABI: one Context per process/GPU, device buffers, op wrappers.

One process per GPU (SURVEY.md §8b "threading / process model"): the Context binds to `cuda:<LOCAL_RANK>`,
owns one HIP stream, and is not re-entrant.  numpy arrays are the host-side currency (fp16/fp32); torch is
only used by callers for CPU tensors and `torch.distributed`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import ACT_NONE, F16, F32, AttnDesc, ConvDesc, GemmDesc, check

_NP = {F16: np.float16, F32: np.float32}


class DeviceArray:
    """A hipMalloc'ed buffer with shape/dtype metadata (dtype is a numpy dtype)."""

    def __init__(self, ctx: "Context", shape: Sequence[int], dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        check(ctx.lib.odise_hip_malloc(ctx.h, C.c_size_t(max(self.nbytes, 16)), C.byref(p)), "malloc")
        self.ptr = p.value
        self._owned = True

    def view(self, shape, dtype=None, offset_bytes: int = 0) -> "DeviceArray":
        """Non-owning array over (a slice of) this buffer; keeps the owner alive."""
        v = DeviceArray.__new__(DeviceArray)
        v.ctx = self.ctx
        v.shape = tuple(int(x) for x in shape)
        v.dtype = np.dtype(dtype if dtype is not None else self.dtype)
        v.nbytes = int(np.prod(v.shape, dtype=np.int64)) * v.dtype.itemsize
        assert offset_bytes >= 0 and offset_bytes + v.nbytes <= self.nbytes, "view out of range"
        v.ptr = self.ptr + offset_bytes
        v._owned = False
        v._base = self
        return v

    def free(self):
        if self._owned and self.ptr and self.ctx.h:
            self.ctx.lib.odise_hip_free(self.ctx.h, C.c_void_p(self.ptr))
        self.ptr = None
        self._owned = False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        check(self.ctx.lib.odise_hip_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                                                C.c_size_t(self.nbytes)), "memcpy_d2h")
        return out

    def copy_from(self, host: np.ndarray) -> "DeviceArray":
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.shape == self.shape, (host.shape, self.shape)
        check(self.ctx.lib.odise_hip_memcpy_h2d(self.ctx.h, C.c_void_p(self.ptr), host.ctypes.data_as(C.c_void_p),
                                                C.c_size_t(self.nbytes)), "memcpy_h2d")
        return self


def _p(a: Optional[DeviceArray]):
    return C.c_void_p(a.ptr) if a is not None else C.c_void_p(None)


class Context:
    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.odise_hip_create(C.c_int(device), C.byref(h)), "create")
        self.h = h
        self.device = device
        # A context holds ONE model (the library's weight store is per context): the host wrapper that loaded weights last registers itself
        # here, so long-lived holders of a wrapper (test fixtures, servers swapping models) can tell whether theirs is still the resident one.
        self.model_owner = None

    def close(self):
        if self.h:
            self.lib.odise_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory -----------------------------------------------------------------------------
    def empty(self, shape, dtype=np.float16) -> DeviceArray:
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float16) -> DeviceArray:
        a = DeviceArray(self, shape, dtype)
        check(self.lib.odise_hip_memset(self.h, C.c_void_p(a.ptr), 0, C.c_size_t(a.nbytes)), "memset")
        return a

    def to_device(self, host, dtype=None) -> DeviceArray:
        if hasattr(host, "detach"):  # torch CPU tensor
            host = host.detach().cpu().numpy()
        host = np.ascontiguousarray(host, dtype=dtype if dtype is not None else host.dtype)
        return DeviceArray(self, host.shape, host.dtype).copy_from(host)

    def sync(self):
        check(self.lib.odise_hip_sync(self.h), "sync")

    # ---- per-context execution options (include/odise_hip.h ODISE_OPT_*) ---------------------
    OPT_CLIP_LN_FOLD, OPT_VAE_CHUNK_BYTES, OPT_ATTN_KV_RESIDENT, OPT_PREFETCH_CU_EIGHTHS, OPT_PREFETCH_START, OPT_MASKCLIP_PASSES = 1, 2, 3, 4, 5, 6

    def set_option(self, option: int, value: int) -> None:
        check(self.lib.odise_hip_set_option(self.h, option, value), "set_option")

    def get_option(self, option: int) -> int:
        v = C.c_int64()
        check(self.lib.odise_hip_get_option(self.h, option, C.byref(v)), "get_option")
        return int(v.value)

    def stage_timeline(self, on: bool = True) -> None:
        check(self.lib.odise_hip_stage_timeline(self.h, 1 if on else 0), "stage_timeline")

    def stage_timeline_read(self):
        """[(name, gpu_ms, host_ms)] of the stage boundaries since stage_timeline(True), relative to the first."""
        cap = 8192
        names = C.create_string_buffer(1 << 20)
        g = (C.c_float * cap)()
        h = (C.c_double * cap)()
        n = C.c_int()
        check(self.lib.odise_hip_stage_timeline_read(self.h, names, len(names), g, h, cap, C.byref(n)), "stage_timeline_read")
        nm = names.value.decode().split("\n")
        return [(nm[i], float(g[i]), float(h[i])) for i in range(min(n.value, cap))]

    def launch_log(self, on: bool = True) -> None:
        check(self.lib.odise_hip_launch_log(self.h, 1 if on else 0), "launch_log")

    def launch_log_read(self) -> np.ndarray:
        """[n, 6] int32 records (conv, M, N, K, tile id, split-K) of the GEMM / conv launches since launch_log(True)."""
        n = C.c_int()
        check(self.lib.odise_hip_launch_log_read(self.h, None, 0, C.byref(n)), "launch_log_read")
        out = np.zeros((max(n.value, 1), 6), np.int32)
        check(self.lib.odise_hip_launch_log_read(self.h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)), "launch_log_read")
        return out[: n.value]

    # ---- launch probe: per-launch durations of one GEMM / conv shape as it runs inside a step ------
    def probe_arm(self, conv: bool, M: int, N: int, K: int, max_launches: int = 256) -> None:
        check(self.lib.odise_hip_probe_arm(self.h, 1 if conv else 0, M, N, K, max_launches), "probe_arm")

    def probe_read(self, cap: int = 4096) -> np.ndarray:
        us = np.zeros(cap, np.float32)
        n = C.c_int()
        check(self.lib.odise_hip_probe_read(self.h, us.ctypes.data_as(C.c_void_p), cap, C.byref(n)), "probe_read")
        return us[: min(cap, n.value)].copy()

    def timer_start(self):
        check(self.lib.odise_hip_timer_start(self.h), "timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(self.lib.odise_hip_timer_stop(self.h, C.byref(ms)), "timer_stop")
        return float(ms.value)

    def device_info(self):
        buf = C.create_string_buffer(256)
        cu = C.c_int()
        mem = C.c_size_t()
        check(self.lib.odise_hip_device_info(self.h, buf, 256, C.byref(cu), C.byref(mem)), "device_info")
        return buf.value.decode(), cu.value, mem.value

    # ---- ops --------------------------------------------------------------------------------
    def ms_deform_attn_forward(self, value: DeviceArray, spatial_shapes, level_start_index, sampling_loc: DeviceArray,
                               attn_weight: DeviceArray, im2col_step: int = 128) -> DeviceArray:
        """MSDA.ms_deform_attn_forward (ms_deform_attn.h:25-44): value [B,S,M,D] -> [B,Lq,M*D]."""
        B, S, M, D = value.shape
        _, Lq, M2, L, P, two = sampling_loc.shape
        assert M2 == M and two == 2 and attn_weight.shape == (B, Lq, M, L, P)
        ss = np.ascontiguousarray(np.asarray(spatial_shapes, dtype=np.int64).reshape(L, 2))
        ls = np.ascontiguousarray(np.asarray(level_start_index, dtype=np.int64).reshape(L))
        assert sampling_loc.dtype == np.float32 and attn_weight.dtype == np.float32
        dt = F32 if value.dtype == np.float32 else F16
        out = self.empty((B, Lq, M * D), value.dtype)
        check(self.lib.odise_hip_ms_deform_attn_forward(
            self.h, _p(value), ss.ctypes.data_as(C.POINTER(C.c_int64)), ls.ctypes.data_as(C.POINTER(C.c_int64)),
            _p(sampling_loc), _p(attn_weight), B, S, M, D, Lq, L, P, int(im2col_step), dt, _p(out)),
            "ms_deform_attn_forward")
        return out

    def gemm(self, A: DeviceArray, W: DeviceArray, *, bias_n=None, bias_m=None, scale_m=None, residual=None,
             rowgroup_add=None, rows_per_group=0, act=ACT_NONE, geglu=False, alpha=1.0, out_dtype=np.float16,
             force_tile=-1, force_split=0, out=None, lda=None, ln=None) -> DeviceArray:
        """C[M,N] = epi(alpha * A[M,K] @ W[N,K]^T); 3-D inputs are batched over dim 0.  `lda` overrides the row stride of A
        (tools: overlapping rows make A cache-resident).  `ln` (test hook, odise_hip_gemm_ln): dict with any of part / parts / inv_c / eps /
        colsum / final_out / fin / rowsum / stats_out - a LayerNorm folded into the epilogue the way the CLIP towers chain their GEMMs."""
        batched = len(A.shape) == 3 or len(W.shape) == 3
        batch = (A.shape[0] if len(A.shape) == 3 else W.shape[0]) if batched else 1
        M, K = A.shape[-2:]
        N, K2 = W.shape[-2:]
        assert K == K2
        No = N // 2 if geglu else N
        oshape = (batch, M, No) if batched else (M, No)
        out = out if out is not None else self.empty(oshape, out_dtype)
        d = GemmDesc()
        d.M, d.N, d.K = M, N, K
        d.A, d.lda = A.ptr, (K if lda is None else int(lda))
        d.W, d.ldw = W.ptr, K
        d.C, d.ldc = out.ptr, No
        d.c_dtype = F32 if np.dtype(out_dtype) == np.float32 else F16
        d.bias_n = bias_n.ptr if bias_n is not None else None
        d.bias_m = bias_m.ptr if bias_m is not None else None
        d.scale_m = scale_m.ptr if scale_m is not None else None
        d.residual = residual.ptr if residual is not None else None
        d.ldr = No
        d.rowgroup_add = rowgroup_add.ptr if rowgroup_add is not None else None
        d.rows_per_group = rows_per_group
        d.act, d.geglu, d.alpha = act, int(geglu), float(alpha)
        d.batch = batch
        d.strideA = M * K if len(A.shape) == 3 else 0
        d.strideW = N * K if len(W.shape) == 3 else 0
        d.strideC = M * No
        d.strideR = M * No
        if ln is not None:
            check(self.lib.odise_hip_gemm_ln(self.h, C.byref(d), ln.get("part"), int(ln.get("parts", 0)), float(ln.get("inv_c", 0.0)),
                                             float(ln.get("eps", 0.0)), ln.get("colsum"), ln.get("final_out"), ln.get("fin"), ln.get("rowsum"),
                                             ln.get("stats_out")), "gemm_ln")
        elif force_tile >= 0 or force_split > 0:
            check(self.lib.odise_hip_gemm_forced(self.h, C.byref(d), int(force_tile), int(force_split)), "gemm_forced")
        else:
            check(self.lib.odise_hip_gemm(self.h, C.byref(d)), "gemm")
        return out

    def conv2d(self, X: DeviceArray, Wt: DeviceArray, *, stride=1, pad=None, pad_tl=None, out_hw=None, upsample2x=False,
               bias=None, residual=None, per_image_add=None, act=ACT_NONE, out_dtype=np.float16, force_tile=-1,
               force_split=0, out=None) -> DeviceArray:
        """NHWC conv: X [N,H,W,Cin] f16, Wt [Cout,KH,KW,Cin] f16 -> [N,OH,OW,Cout]."""
        N, H, W, Cin = X.shape
        Cout, KH, KW, Cin2 = Wt.shape
        assert Cin == Cin2
        if pad is None:
            pad = KH // 2
        pt, pl = pad_tl if pad_tl is not None else (pad, pad)
        Hin, Win = (2 * H, 2 * W) if upsample2x else (H, W)
        if out_hw is None:
            OH = (Hin + 2 * pad - KH) // stride + 1
            OW = (Win + 2 * pad - KW) // stride + 1
        else:
            OH, OW = out_hw
        out = out if out is not None else self.empty((N, OH, OW, Cout), out_dtype)
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin = N, H, W, Cin
        d.Cout, d.KH, d.KW, d.stride = Cout, KH, KW, stride
        d.pad_t, d.pad_l, d.OH, d.OW = pt, pl, OH, OW
        d.upsample2x = int(upsample2x)
        d.X, d.Wt, d.Y = X.ptr, Wt.ptr, out.ptr
        d.y_dtype = F32 if np.dtype(out_dtype) == np.float32 else F16
        d.bias = bias.ptr if bias is not None else None
        d.residual = residual.ptr if residual is not None else None
        d.per_image_add = per_image_add.ptr if per_image_add is not None else None
        d.act = act
        if force_tile >= 0 or force_split > 0:
            check(self.lib.odise_hip_conv2d_forced(self.h, C.byref(d), int(force_tile), int(force_split)), "conv2d_forced")
        else:
            check(self.lib.odise_hip_conv2d(self.h, C.byref(d)), "conv2d")
        return out

    def conv2d_gn(self, X: DeviceArray, Wt: DeviceArray, gamma: DeviceArray, beta: DeviceArray, *, bias=None, groups=32, eps=1e-5, act=ACT_NONE,
                  force_tile=-1, force_split=0):
        """3x3 / 1x1 stride-1 conv whose epilogue reduces the GroupNorm statistics, then that GroupNorm (developer hook
        odise_hip_conv2d_gn_forced): returns (conv output, normalised output, row blocks per image; 0 = fusion declined)."""
        N, H, W, Cin = X.shape
        Cout, KH, KW, _ = Wt.shape
        y = self.empty((N, H, W, Cout), np.float16)
        yn = self.empty((N, H, W, Cout), np.float16)
        scratch = self.empty((N * ((H * W + 63) // 64) * Cout * 2,), np.float32)
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin = N, H, W, Cin
        d.Cout, d.KH, d.KW, d.stride = Cout, KH, KW, 1
        d.pad_t, d.pad_l, d.OH, d.OW = KH // 2, KW // 2, H, W
        d.X, d.Wt, d.Y = X.ptr, Wt.ptr, y.ptr
        d.y_dtype = F16
        d.bias = bias.ptr if bias is not None else None
        d.act = ACT_NONE
        blocks = C.c_int(0)
        check(self.lib.odise_hip_conv2d_gn_forced(self.h, C.byref(d), int(force_tile), int(force_split), _p(gamma), _p(beta), int(groups),
                                                  C.c_float(eps), int(act), C.c_void_p(yn.ptr), C.c_void_p(scratch.ptr), C.byref(blocks)),
              "conv2d_gn_forced")
        self.sync()
        scratch.free()
        return y, yn, blocks.value

    def group_norm(self, x: DeviceArray, gamma: Optional[DeviceArray], beta: Optional[DeviceArray], groups=32, eps=1e-5,
                   act=ACT_NONE) -> DeviceArray:
        """x [N, ..., C] f16 channels-last."""
        N, Cc = x.shape[0], x.shape[-1]
        HW = int(np.prod(x.shape[1:-1]))
        y = self.empty(x.shape, np.float16)
        check(self.lib.odise_hip_group_norm(self.h, _p(x), _p(y), _p(gamma), _p(beta), N, HW, Cc, groups, C.c_float(eps), act),
              "group_norm")
        return y

    def layer_norm(self, x: DeviceArray, gamma, beta, eps=1e-5) -> DeviceArray:
        Cc = x.shape[-1]
        rows = int(np.prod(x.shape[:-1]))
        y = self.empty(x.shape, np.float16)
        check(self.lib.odise_hip_layer_norm(self.h, _p(x), _p(y), _p(gamma), _p(beta), rows, Cc, C.c_float(eps)), "layer_norm")
        return y

    def attention(self, Q: DeviceArray, K: DeviceArray, Vt: DeviceArray, heads: int, scale: float,
                  mask: Optional[DeviceArray] = None, Lk: Optional[int] = None, out: Optional[DeviceArray] = None) -> DeviceArray:
        """Q [B,Lq,H*D], K [B,Lk,H*D], Vt [B,H*D,ldvt] (transposed V) -> O [B,Lq,H*D] (all f16)."""
        B, Lq, HD = Q.shape
        Lk = Lk if Lk is not None else K.shape[1]
        D = HD // heads
        O = out if out is not None else self.empty((B, Lq, HD), np.float16)
        d = AttnDesc()
        d.B, d.H, d.Lq, d.Lk, d.D = B, heads, Lq, Lk, D
        d.Q, d.ldq, d.strideQ = Q.ptr, HD, Lq * HD
        d.K, d.ldk, d.strideK = K.ptr, HD, K.shape[1] * HD
        d.Vt, d.ldvt, d.strideVt = Vt.ptr, Vt.shape[2], Vt.shape[1] * Vt.shape[2]
        d.O, d.ldo, d.strideO = O.ptr, HD, Lq * HD
        if mask is not None:
            assert mask.dtype == np.uint8 and mask.shape[0] == B and mask.shape[1] == Lq
            d.mask, d.ldmask, d.strideMask = mask.ptr, mask.shape[2], mask.shape[1] * mask.shape[2]
        d.scale = float(scale)
        check(self.lib.odise_hip_attention(self.h, C.byref(d)), "attention")
        return O

    def nchw_to_nhwc_f16(self, x: DeviceArray, cpad: Optional[int] = None) -> DeviceArray:
        N, Cc, H, W = x.shape
        cpad = cpad or ((Cc + 7) // 8) * 8
        y = self.empty((N, H, W, cpad), np.float16)
        check(self.lib.odise_hip_nchw_f32_to_nhwc_f16(self.h, _p(x), _p(y), N, Cc, H, W, cpad), "nchw_to_nhwc")
        return y

    def nhwc_to_nchw_f32(self, x: DeviceArray) -> DeviceArray:
        N, H, W, Cc = x.shape
        y = self.empty((N, Cc, H, W), np.float32)
        check(self.lib.odise_hip_nhwc_f16_to_nchw_f32(self.h, _p(x), _p(y), N, Cc, H, W), "nhwc_to_nchw")
        return y

    def concat_channels(self, a: DeviceArray, b: DeviceArray) -> DeviceArray:
        assert a.shape[:-1] == b.shape[:-1]
        pixels = int(np.prod(a.shape[:-1]))
        y = self.empty(a.shape[:-1] + (a.shape[-1] + b.shape[-1],), np.float16)
        check(self.lib.odise_hip_concat_channels(self.h, _p(a), _p(b), _p(y), C.c_size_t(pixels), a.shape[-1], b.shape[-1]),
              "concat_channels")
        return y

    def mask_pooling(self, x: DeviceArray, mask: DeviceArray) -> DeviceArray:
        """MaskPooling.forward (odise.py:937-963): x [B,C,H,W] f32, mask [B,Q,H,W] f32 -> [B,Q,C] f32."""
        B, Cc, H, W = x.shape
        Q = mask.shape[1]
        out = self.empty((B, Q, Cc), np.float32)
        check(self.lib.odise_hip_mask_pooling(self.h, _p(x), _p(mask), _p(out), B, Cc, Q, H * W), "mask_pooling")
        return out

    # ---- eval-loop helpers (include/odise_hip.h, last section) ---------------------------------------------------------------------
    def resize_bilinear_u8(self, img: DeviceArray, out_h: int, out_w: int) -> DeviceArray:
        """uint8 [H,W,C] -> uint8 [out_h,out_w,C], bit-identical to PIL.Image.resize((out_w, out_h), BILINEAR)."""
        H, W, Cc = img.shape
        assert img.dtype == np.uint8
        out = self.empty((out_h, out_w, Cc), np.uint8)
        check(self.lib.odise_hip_resize_bilinear_u8(self.h, _p(img), H, W, Cc, _p(out), int(out_h), int(out_w)), "resize_bilinear_u8")
        return out

    def u8_hwc_to_f32_chw(self, img: DeviceArray, scale: float = 1.0) -> DeviceArray:
        H, W, Cc = img.shape
        out = self.empty((Cc, H, W), np.float32)
        check(self.lib.odise_hip_u8_hwc_to_f32_chw(self.h, _p(img), _p(out), H, W, Cc, C.c_float(scale)), "u8_hwc_to_f32_chw")
        return out

    def u8_hwc_to_f32_chw_padded(self, img: DeviceArray, Hp: int, Wp: int, scale: float = 1.0, out: Optional[DeviceArray] = None) -> DeviceArray:
        """uint8 [H,W,C] -> fp32 [C,Hp,Wp] with the image in the top-left corner and zeros elsewhere (ImageList.from_tensors)."""
        H, W, Cc = img.shape
        if out is None:
            out = self.empty((Cc, Hp, Wp), np.float32)
        assert out.nbytes == Cc * Hp * Wp * 4
        check(self.lib.odise_hip_u8_hwc_to_f32_chw_padded(self.h, _p(img), _p(out), H, W, Cc, int(Hp), int(Wp), C.c_float(scale)), "u8_hwc_to_f32_chw_padded")
        return out

    def semantic_confusion(self, sem_seg: DeviceArray, gt: DeviceArray, conf: Optional[DeviceArray] = None) -> DeviceArray:
        """sem_seg f32 [K,H,W], gt int32 [H,W] (ignore label already mapped outside [0,K)) -> int64 [(K+1),(K+1)] accumulated into conf."""
        K = sem_seg.shape[0]
        npix = int(np.prod(sem_seg.shape[1:]))
        if conf is None:
            conf = self.zeros((K + 1, K + 1), np.int64)
        check(self.lib.odise_hip_semantic_confusion(self.h, _p(sem_seg), _p(gt), K, npix, _p(conf)), "semantic_confusion")
        return conf

    def pair_histogram(self, a: DeviceArray, b: DeviceArray, na: int, nb: int, hist: Optional[DeviceArray] = None) -> DeviceArray:
        npix = int(np.prod(a.shape))
        if hist is None:
            hist = self.zeros((na, nb), np.int32)
        check(self.lib.odise_hip_pair_histogram(self.h, _p(a), _p(b), npix, int(na), int(nb), _p(hist)), "pair_histogram")
        return hist


    def jpeg_decode(self, data: bytes, apply_orientation: bool = True, out: Optional[DeviceArray] = None) -> DeviceArray:
        """read_image(file, "RGB") for a baseline JPEG: host Huffman decoding, IDCT / upsampling / colour conversion / EXIF transpose on
        the device -> uint8 [H,W,3], bit-identical to Pillow.  Baseline, extended-sequential and progressive files; raises `UnsupportedInput` for arithmetic-coded / CMYK / RGB-coded ones."""
        info = jpeg_info(data)
        swap = apply_orientation and info["orientation"] >= 5
        oh, ow = (info["width"], info["height"]) if swap else (info["height"], info["width"])
        if out is None or out.nbytes < oh * ow * 3:
            out = self.empty((oh, ow, 3), np.uint8)
        h, w = C.c_int(0), C.c_int(0)
        buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
        check(self.lib.odise_hip_jpeg_decode(self.h, buf, C.c_int64(len(data)), _p(out), C.c_int64(out.nbytes), int(bool(apply_orientation)),
                                             C.byref(h), C.byref(w)), "jpeg_decode")
        return out.view((h.value, w.value, 3)) if out.shape != (h.value, w.value, 3) else out


def _jpeg_info_struct(info: dict):
    from ._lib import JpegInfo
    st = JpegInfo()
    for k in ("width", "height", "components", "h_samp", "v_samp", "orientation", "restart_interval", "coef_count"):
        setattr(st, k, info[k])
    for c in range(info["components"]):
        st.blocks_x[c], st.blocks_y[c] = info["blocks_x"][c], info["blocks_y"][c]
    return st


def jpeg_decode_coefs(ctx: Context, info: dict, coefs, qt, apply_orientation: bool = True) -> DeviceArray:
    """Device half of the decoder on coefficients from `jpeg_entropy_decode` (which may have run in a loader thread)."""
    swap = apply_orientation and info["orientation"] >= 5
    oh, ow = (info["width"], info["height"]) if swap else (info["height"], info["width"])
    out = ctx.empty((oh, ow, 3), np.uint8)
    flat = coefs if isinstance(coefs, np.ndarray) else np.concatenate([np.asarray(c).ravel() for c in coefs])
    flat = np.ascontiguousarray(flat.ravel(), np.int16)
    qt = np.ascontiguousarray(qt, np.uint16)
    st = _jpeg_info_struct(info)
    check(ctx.lib.odise_hip_jpeg_decode_coefs(ctx.h, C.byref(st), flat.ctypes.data_as(C.c_void_p), qt.ctypes.data_as(C.c_void_p), _p(out),
                                              C.c_int64(out.nbytes), int(bool(apply_orientation)), None, None), "jpeg_decode_coefs")
    return out


def jpeg_info(data: bytes) -> dict:
    """Header fields of a JPEG byte stream (host only)."""
    from ._lib import JpegInfo, load
    info = JpegInfo()
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    check(load().odise_hip_jpeg_info(buf, C.c_int64(len(data)), C.byref(info)), "jpeg_info")
    return dict(width=info.width, height=info.height, components=info.components, h_samp=info.h_samp, v_samp=info.v_samp,
                orientation=info.orientation, restart_interval=info.restart_interval, blocks_x=list(info.blocks_x)[:info.components],
                blocks_y=list(info.blocks_y)[:info.components], coef_count=info.coef_count)


def jpeg_entropy_decode(data: bytes, flat: bool = False):
    """Host half of the decoder on its own: (info, [int16 [blocks_y, blocks_x, 64] per component], uint16 [components, 64] tables);
    `flat=True` returns the coefficients as the one int16 array `jpeg_decode_coefs` uploads.  Thread-safe (no context involved)."""
    from ._lib import load
    info = jpeg_info(data)
    coefs = np.zeros(info["coef_count"], np.int16)
    qt = np.zeros((info["components"], 64), np.uint16)
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    check(load().odise_hip_jpeg_entropy_decode(buf, C.c_int64(len(data)), coefs.ctypes.data_as(C.c_void_p), C.c_int64(coefs.size),
                                               qt.ctypes.data_as(C.c_void_p)), "jpeg_entropy_decode")
    if flat:
        return info, coefs, qt
    out, off = [], 0
    for by, bx in zip(info["blocks_y"], info["blocks_x"]):
        out.append(coefs[off:off + by * bx * 64].reshape(by, bx, 64))    # views of one flat array (what jpeg_decode_coefs uploads)
        off += by * bx * 64
    return info, out, qt


_default: Optional[Context] = None


def default_context() -> Context:
    """Process-wide context on cuda:<LOCAL_RANK> (one process per GPU)."""
    global _default
    if _default is None:
        import os
        _default = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default
