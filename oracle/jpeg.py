"""Oracle: baseline JPEG decoding as the reference's input path performs it (TEST INFRASTRUCTURE - see oracle/__init__.py;
SURVEY.md 8f row 4).

The reference reads images with detectron2 `utils.read_image(file, "RGB")` (dataset mapper of
configs/common/data/pano_open_d2_eval.py:74-107; demo/demo.py:399) = `PIL.Image.open(f)` -> EXIF transpose -> `convert("RGB")`,
i.e. libjpeg-turbo behind Pillow with its defaults: JDCT_ISLOW, fancy (triangle) chroma upsampling, JFIF YCbCr -> RGB.  libjpeg is
third-party (absent from /root/reference); its published algorithm is restated here and PINNED bit-exactly against the Pillow
installed in this image (tests/test_oracle_jpeg.py): jdhuff.c (Huffman entropy decoding, restart intervals), jidctint.c (the
"slow-but-accurate" LL&M integer IDCT, CONST_BITS 13 / PASS1_BITS 2), jdsample.c (h2v1 / h2v2 fancy upsampling, plain replication
when a component is <= 2 samples wide), jdcolor.c (16-bit fixed-point YCbCr -> RGB tables).

Scope = what the product path implements: 8-bit baseline / extended-sequential Huffman JPEG (SOF0 / SOF1), one interleaved scan,
1 (grey) or 3 (YCbCr) components, luma sampling 1x1 / 2x1 / 2x2 with 1x1 chroma.  Progressive, arithmetic, CMYK / RGB-coded files
raise `Unsupported`.
"""
from __future__ import annotations

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                   62, 63], np.int32)                                    # zigzag index -> natural (row-major) index


class Unsupported(ValueError):
    pass


def parse(data: bytes) -> dict:
    """Marker segments up to and including the first SOS header."""
    if data[:2] != b"\xff\xd8":
        raise ValueError("not a JPEG (no SOI)")
    p, info = 2, dict(qt={}, dc={}, ac={}, restart=0, jfif=False, adobe=None, orientation=1)
    while True:
        while data[p] != 0xFF:
            p += 1
        while data[p] == 0xFF:
            p += 1
        m = data[p]
        p += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        L = (data[p] << 8) | data[p + 1]
        seg = data[p + 2:p + L]
        p += L
        if m == 0xDB:                                                     # DQT
            q = 0
            while q < len(seg):
                pq, tq = seg[q] >> 4, seg[q] & 15
                n = 128 if pq else 64
                raw = np.frombuffer(seg[q + 1:q + 1 + n], ">u2" if pq else np.uint8).astype(np.int32)
                t = np.zeros(64, np.int32)
                t[ZIGZAG] = raw
                info["qt"][tq] = t
                q += 1 + n
        elif m in (0xC0, 0xC1):                                           # SOF0 / SOF1
            if seg[0] != 8:
                raise Unsupported("only 8-bit samples")
            info["height"], info["width"] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            info["comps"] = [dict(id=seg[6 + 3 * i], h=seg[7 + 3 * i] >> 4, v=seg[7 + 3 * i] & 15, tq=seg[8 + 3 * i]) for i in range(seg[5])]
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported(f"SOF marker 0x{m:02x} (progressive / lossless / arithmetic)")
        elif m == 0xC4:                                                   # DHT
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                counts = list(seg[q + 1:q + 17])
                n = sum(counts)
                info["ac" if tc else "dc"][th] = (counts, list(seg[q + 17:q + 17 + n]))
                q += 17 + n
        elif m == 0xDD:
            info["restart"] = (seg[0] << 8) | seg[1]
        elif m == 0xE0 and seg[:5] == b"JFIF\0":
            info["jfif"] = True
        elif m == 0xEE and seg[:5] == b"Adobe":
            info["adobe"] = seg[11]
        elif m == 0xE1 and seg[:6] == b"Exif\0\0":
            info["orientation"] = _exif_orientation(seg[6:])
        elif m == 0xDA:                                                   # SOS
            ns = seg[0]
            info["scan"] = [dict(id=seg[1 + 2 * i], td=seg[2 + 2 * i] >> 4, ta=seg[2 + 2 * i] & 15) for i in range(ns)]
            info["data_start"] = p
            return info
        elif m == 0xD9:
            raise ValueError("EOI before SOS")


def _exif_orientation(tiff: bytes) -> int:
    if len(tiff) < 8 or tiff[:2] not in (b"II", b"MM"):
        return 1
    e = "<" if tiff[:2] == b"II" else ">"
    u16 = lambda o: int(np.frombuffer(tiff[o:o + 2], e + "u2")[0])
    u32 = lambda o: int(np.frombuffer(tiff[o:o + 4], e + "u4")[0])
    ifd = u32(4)
    if ifd + 2 > len(tiff):
        return 1
    for i in range(u16(ifd)):
        o = ifd + 2 + 12 * i
        if o + 12 > len(tiff):
            break
        if u16(o) == 0x0112:
            v = u16(o + 8)
            return v if 1 <= v <= 8 else 1
    return 1


def _huff_table(counts, symbols):
    """jdhuff.c jpeg_make_d_derived_tbl as a dict (length, code) -> symbol."""
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            table[(length, code)] = symbols[k]
            code += 1
            k += 1
        code <<= 1
    return table


class _Bits:
    def __init__(self, data, pos):
        self.d, self.p, self.acc, self.n = data, pos, 0, 0

    def bit(self):
        if self.n == 0:
            b = self.d[self.p] if self.p < len(self.d) else 0
            self.p += 1
            if b == 0xFF:
                nxt = self.d[self.p] if self.p < len(self.d) else 0xD9
                if nxt == 0:
                    self.p += 1
                else:                                                     # a marker inside entropy data: feed zeros (jdhuff.c)
                    self.p -= 1
                    b = 0
            self.acc, self.n = b, 8
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def symbol(self, table):
        code = 0
        for length in range(1, 17):
            code = (code << 1) | self.bit()
            s = table.get((length, code))
            if s is not None:
                return s
        raise ValueError("bad Huffman code")

    def restart(self):
        self.n = 0
        while not (self.d[self.p] == 0xFF and 0xD0 <= self.d[self.p + 1] <= 0xD7):
            self.p += 1
        self.p += 2


def _extend(v, s):
    return v - (1 << s) + 1 if s and v < (1 << (s - 1)) else v


def entropy_decode(data: bytes):
    """-> (info, [coef_c]) with coef_c int16 [blocks_y, blocks_x, 64] in natural order, NOT dequantised.  Component block grids are
    padded to whole MCUs for interleaved scans (jdcoefct.c)."""
    info = parse(data)
    comps = info["comps"]
    if len(comps) not in (1, 3):
        raise Unsupported(f"{len(comps)} components")
    if len(info["scan"]) != len(comps):
        raise Unsupported("multiple scans")
    if len(comps) == 3:
        ids = bytes(c["id"] for c in comps)
        if info["adobe"] == 0 or (not info["jfif"] and info["adobe"] is None and ids == b"RGB"):
            raise Unsupported("RGB-coded JPEG")
        if any((c["h"], c["v"]) != (1, 1) for c in comps[1:]) or (comps[0]["h"], comps[0]["v"]) not in ((1, 1), (2, 1), (2, 2)):
            raise Unsupported("sampling factors")
    else:
        comps[0]["h"] = comps[0]["v"] = 1                                 # a single-component scan is never interleaved
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    W, H = info["width"], info["height"]
    mx, my = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    info.update(hmax=hmax, vmax=vmax, mcus_x=mx, mcus_y=my)
    coefs = [np.zeros((my * c["v"], mx * c["h"], 64), np.int16) for c in comps]
    tabs = [(_huff_table(*info["dc"][s["td"]]), _huff_table(*info["ac"][s["ta"]])) for s in info["scan"]]
    br, pred, ri = _Bits(data, info["data_start"]), [0] * len(comps), info["restart"]
    for mcu in range(mx * my):
        if ri and mcu and mcu % ri == 0:
            br.restart()
            pred = [0] * len(comps)
        y0, x0 = divmod(mcu, mx)
        for ci, c in enumerate(comps):
            dc, ac = tabs[ci]
            for v in range(c["v"]):
                for h in range(c["h"]):
                    blk = coefs[ci][y0 * c["v"] + v, x0 * c["h"] + h]
                    s = br.symbol(dc)
                    pred[ci] += _extend(br.bits(s), s)
                    blk[0] = np.int16(pred[ci])                           # JCOEF is a short: wraps like libjpeg's store
                    k = 1
                    while k < 64:
                        rs = br.symbol(ac)
                        r, s = rs >> 4, rs & 15
                        if s == 0:
                            if r != 15:
                                break
                            k += 16
                            continue
                        k += r
                        if k > 63:
                            break                                         # corrupt data: libjpeg warns and carries on
                        blk[ZIGZAG[k]] = _extend(br.bits(s), s)
                        k += 1
    return info, coefs


# ---- jidctint.c --------------------------------------------------------------------------------------------------------------------
F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865, F_0_899976223, F_1_175875602 = 2446, 3196, 4433, 6270, 7373, 9633
F_1_501321110, F_1_847759065, F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 12299, 15137, 16069, 16819, 20995, 25172


def _idct_1d(r, shift_const):
    """One LL&M pass over the first axis of r [8, ...] (int64): returns the 8 un-descaled outputs."""
    z2, z3 = r[2], r[6]
    z1 = (z2 + z3) * F_0_541196100
    tmp2 = z1 - z3 * F_1_847759065
    tmp3 = z1 + z2 * F_0_765366865
    tmp0 = (r[0] + r[4]) << shift_const
    tmp1 = (r[0] - r[4]) << shift_const
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = r[7], r[5], r[3], r[1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F_1_175875602
    t0, t1, t2, t3 = t0 * F_0_298631336, t1 * F_2_053119869, t2 * F_3_072711026, t3 * F_1_501321110
    z1, z2, z3, z4 = -z1 * F_0_899976223, -z2 * F_2_562915447, -z3 * F_1_961570560 + z5, -z4 * F_0_390180644 + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    return [tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3]


def idct_islow(coef: np.ndarray, qt: np.ndarray) -> np.ndarray:
    """coef int16 [..., 64] (natural order), qt [64] -> uint8 [..., 8, 8]."""
    x = (coef.astype(np.int64) * qt.astype(np.int64)).reshape(coef.shape[:-1] + (8, 8))
    rows = np.moveaxis(x, -2, 0)                                          # pass 1 works down the columns: axis 0 = row index
    ws = np.stack([(v + (1 << 10)) >> 11 for v in _idct_1d(rows, 13)], 0)     # DESCALE(CONST_BITS - PASS1_BITS); [8(row), ..., 8(col)]
    cols = np.moveaxis(ws, -1, 0)                                         # pass 2 works along the rows: axis 0 = column index
    out = np.stack([(v + (1 << 17)) >> 18 for v in _idct_1d(cols, 13)], 0)    # [8(col), 8(row), ...]
    out = np.moveaxis(np.moveaxis(out, 0, -1), 0, -2)                     # [..., row, col]
    return np.clip(out + 128, 0, 255).astype(np.uint8)


def _plane(blocks_u8: np.ndarray) -> np.ndarray:
    by, bx = blocks_u8.shape[:2]
    return blocks_u8.transpose(0, 2, 1, 3).reshape(by * 8, bx * 8)


# ---- jdsample.c --------------------------------------------------------------------------------------------------------------------
def _h2v1_fancy(p: np.ndarray) -> np.ndarray:
    p = p.astype(np.int32)
    w = p.shape[1]
    out = np.empty((p.shape[0], 2 * w), np.int32)
    left, right = np.concatenate([p[:, :1], p[:, :-1]], 1), np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0], out[:, -1] = p[:, 0], p[:, -1]
    return out


def _h2v2_fancy(p: np.ndarray) -> np.ndarray:
    p = p.astype(np.int32)
    h, w = p.shape
    above, below = np.concatenate([p[:1], p[:-1]], 0), np.concatenate([p[1:], p[-1:]], 0)
    out = np.empty((2 * h, 2 * w), np.int32)
    for v, far in ((0, above), (1, below)):
        s = 3 * p + far                                                   # column sums of the nearer (x3) and further row
        left, right = np.concatenate([s[:, :1], s[:, :-1]], 1), np.concatenate([s[:, 1:], s[:, -1:]], 1)
        o = np.empty((h, 2 * w), np.int32)
        o[:, 0::2] = (3 * s + left + 8) >> 4
        o[:, 1::2] = (3 * s + right + 7) >> 4
        o[:, 0], o[:, -1] = (4 * s[:, 0] + 8) >> 4, (4 * s[:, -1] + 7) >> 4
        out[v::2] = o
    return out


def upsample(p: np.ndarray, hx: int, vx: int) -> np.ndarray:
    """p = the component's real samples [downsampled_height, downsampled_width]; expansion factors (1|2, 1|2)."""
    if (hx, vx) == (1, 1):
        return p
    fancy = p.shape[1] > 2
    if (hx, vx) == (2, 1):
        return _h2v1_fancy(p) if fancy else np.repeat(p, 2, 1)
    if (hx, vx) == (2, 2):
        return _h2v2_fancy(p) if fancy else np.repeat(np.repeat(p, 2, 0), 2, 1)
    raise Unsupported("upsampling factors")


# ---- jdcolor.c ---------------------------------------------------------------------------------------------------------------------
def _fix(x):
    return int(x * 65536 + 0.5)


_X = np.arange(256, dtype=np.int64) - 128
CR_R = ((_fix(1.40200) * _X + 32768) >> 16).astype(np.int32)
CB_B = ((_fix(1.77200) * _X + 32768) >> 16).astype(np.int32)
CR_G = (-_fix(0.71414) * _X).astype(np.int64)
CB_G = (-_fix(0.34414) * _X + 32768).astype(np.int64)


def ycc_to_rgb(y, cb, cr):
    y = y.astype(np.int32)
    r = y + CR_R[cr]
    g = y + ((CB_G[cb] + CR_G[cr]) >> 16).astype(np.int32)
    b = y + CB_B[cb]
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def decode_planes(info, coefs):
    """IDCT + upsampling + colour conversion of entropy-decoded coefficients -> uint8 [H, W, 3] (before any EXIF transpose)."""
    W, H, comps = info["width"], info["height"], info["comps"]
    planes = []
    for c, cf in zip(comps, coefs):
        full = _plane(idct_islow(cf, info["qt"][c["tq"]]))
        dw, dh = -(-W * c["h"] // info["hmax"]), -(-H * c["v"] // info["vmax"])
        planes.append(upsample(full[:dh, :dw], info["hmax"] // c["h"], info["vmax"] // c["v"])[:H, :W])
    if len(planes) == 1:
        return np.repeat(planes[0][..., None], 3, -1).astype(np.uint8)
    return ycc_to_rgb(planes[0], planes[1].astype(np.int64), planes[2].astype(np.int64))


def exif_transpose(img: np.ndarray, orientation: int) -> np.ndarray:
    """detectron2 `_apply_exif_orientation` (= PIL.ImageOps.exif_transpose): orientation 2..8 -> flips / rotations."""
    if orientation == 2:
        return img[:, ::-1]
    if orientation == 3:
        return img[::-1, ::-1]
    if orientation == 4:
        return img[::-1]
    if orientation == 5:
        return img.transpose(1, 0, 2)
    if orientation == 6:
        return img.transpose(1, 0, 2)[:, ::-1]                            # rotate 270 counter-clockwise
    if orientation == 7:
        return img[::-1, ::-1].transpose(1, 0, 2)
    if orientation == 8:
        return img.transpose(1, 0, 2)[::-1]                               # rotate 90 counter-clockwise
    return img


def decode(data: bytes, apply_orientation: bool = True) -> np.ndarray:
    """read_image(file, "RGB") on a JPEG: uint8 [H, W, 3]."""
    info, coefs = entropy_decode(data)
    img = decode_planes(info, coefs)
    return np.ascontiguousarray(exif_transpose(img, info["orientation"]) if apply_orientation else img)
