// jpeg.hip — baseline JPEG decoding for the input side of the eval loop.
//
// The reference reads every image with detectron2 `read_image(file, "RGB")` (DatasetMapper of
// configs/common/data/pano_open_d2_eval.py:74-107, demo/demo.py:399) = Pillow on libjpeg-turbo with its defaults: JDCT_ISLOW, fancy
// chroma upsampling, JFIF YCbCr -> RGB, then the EXIF transpose.  Here the serial part (marker parsing, Huffman entropy decoding) runs
// on the host straight into a pinned staging buffer, and everything per-sample runs on the device, bit-identical to libjpeg:
//   jpeg_idct_kernel   dequantisation + the LL&M "islow" integer IDCT (CONST_BITS 13, PASS1_BITS 2): 8 lanes per 8x8 block - a lane
//                      owns a column in pass 1 and a row in pass 2, the 8x8 int32 workspace is exchanged through LDS
//   jpeg_color_kernel  triangle-filter ("fancy") h2v1 / h2v2 chroma upsampling evaluated per output pixel from the decoded planes,
//                      the 16-bit fixed-point YCbCr -> RGB conversion, the EXIF orientation as an index map, packed RGB store
// Scope: 8-bit Huffman files - baseline / extended sequential (SOF0 / SOF1, one or several scans) and progressive (SOF2: spectral
// selection + successive approximation) - grey or YCbCr, luma sampling 1x1 / 2x1 / 2x2 and 1x1 chroma; arithmetic-coded, lossless,
// CMYK and RGB-coded files return ODISE_ERR_UNSUPPORTED (the caller decides what to do with them;
// there is no CPU decode path in this library).  The input bytes are untrusted: every read is bounds-checked, truncated entropy data
// decodes as zero bits like libjpeg does.
#include <string.h>

#include <vector>

#include "common.h"

namespace odise {

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    uint8_t look_bits[512];  // 9-bit lookahead: code length (0 = longer than 9 bits)
    uint8_t look_sym[512];
    int32_t maxcode[18];     // largest code of each length (-1 = none); [17] is a sentinel that always matches
    int32_t valoff[17];      // symbol index = code + valoff[length]
    uint8_t vals[256];
    // AC shortcut: when code + magnitude bits fit the 9-bit lookahead, the whole (run, value) pair comes from one lookup
    uint8_t fast_len[512];   // code length + magnitude bits (0 = take the general path)
    uint8_t fast_run[512];
    int16_t fast_val[512];
};

static bool build_table(const uint8_t* counts, const uint8_t* symbols, int nsym, HuffTable& t) {
    memset(t.look_bits, 0, sizeof(t.look_bits));
    memcpy(t.vals, symbols, nsym);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
        t.valoff[len] = k - code;
        const int n = counts[len - 1];
        if (code + n > (1 << len)) return false;  // over-subscribed
        if (len <= 9) {
            for (int i = 0; i < n; ++i) {
                const int first = (code + i) << (9 - len);
                for (int f = 0; f < (1 << (9 - len)); ++f) {
                    t.look_bits[first + f] = (uint8_t)len;
                    t.look_sym[first + f] = symbols[k + i];
                }
            }
        }
        code += n;
        k += n;
        t.maxcode[len] = n ? code - 1 : -1;
        code <<= 1;
    }
    t.maxcode[17] = 0x7fffffff;
    for (int i = 0; i < 512; ++i) {
        t.fast_len[i] = 0;
        const int len = t.look_bits[i];
        if (!len) continue;
        const int rs = t.look_sym[i], mag = rs & 15;
        if (mag == 0 || len + mag > 9) continue;
        const int bits = (i >> (9 - len - mag)) & ((1 << mag) - 1);
        t.fast_len[i] = (uint8_t)(len + mag);
        t.fast_run[i] = (uint8_t)(rs >> 4);
        t.fast_val[i] = (int16_t)(bits < (1 << (mag - 1)) ? bits - (1 << mag) + 1 : bits);
    }
    t.present = true;
    return true;
}

struct JpegHeader {
    int width = 0, height = 0, ncomp = 0;
    int id[3] = {0, 0, 0}, h[3] = {1, 1, 1}, v[3] = {1, 1, 1}, tq[3] = {0, 0, 0};
    int hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    int bx[3] = {0, 0, 0}, by[3] = {0, 0, 0};  // component block grids (padded to whole MCUs)
    int64_t coef_off[3] = {0, 0, 0}, coef_count = 0;
    int restart = 0, orientation = 1, adobe = -1;
    bool jfif = false, have_sof = false, progressive = false, scans_started = false;
    int64_t sos_pos = -1;  // offset of the first SOS segment's length field
    bool qt_ok[4] = {false, false, false, false};
    uint16_t qt[4][64];
    HuffTable dc[4], ac[4];
};

static int exif_orientation(const uint8_t* t, int64_t n) {
    if (n < 8) return 1;
    bool le;
    if (t[0] == 'I' && t[1] == 'I') le = true;
    else if (t[0] == 'M' && t[1] == 'M') le = false;
    else return 1;
    auto u16 = [&](int64_t o) -> int { return le ? (t[o] | (t[o + 1] << 8)) : ((t[o] << 8) | t[o + 1]); };
    auto u32 = [&](int64_t o) -> int64_t {
        return le ? ((int64_t)t[o] | ((int64_t)t[o + 1] << 8) | ((int64_t)t[o + 2] << 16) | ((int64_t)t[o + 3] << 24))
                  : (((int64_t)t[o] << 24) | ((int64_t)t[o + 1] << 16) | ((int64_t)t[o + 2] << 8) | (int64_t)t[o + 3]);
    };
    const int64_t ifd = u32(4);
    if (ifd + 2 > n) return 1;
    const int cnt = u16(ifd);
    for (int i = 0; i < cnt; ++i) {
        const int64_t o = ifd + 2 + 12 * (int64_t)i;
        if (o + 12 > n) break;
        if (u16(o) == 0x0112) {
            const int val = u16(o + 8);
            return (val >= 1 && val <= 8) ? val : 1;
        }
    }
    return 1;
}

#define JPEG_FAIL(code, ...)        \
    do {                            \
        set_error(__VA_ARGS__);     \
        return code;                \
    } while (0)

// One marker segment other than SOS (tables, frame header, application data): shared by the header pass and the inter-scan pass.
static int handle_segment(int m, const uint8_t* s, int n, JpegHeader& H) {
    if (m == 0xDB) {
        int q = 0;
        while (q < n) {
            const int pq = s[q] >> 4, tq = s[q] & 15;
            const int bytes = pq ? 128 : 64;
            if (tq > 3 || pq > 1 || q + 1 + bytes > n) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad DQT");
            if (!H.scans_started) {  // the tables a component uses are latched when decoding starts (jdinput.c latch_quant_tables)
                for (int i = 0; i < 64; ++i) H.qt[tq][kZigzag[i]] = pq ? (uint16_t)((s[q + 1 + 2 * i] << 8) | s[q + 2 + 2 * i]) : s[q + 1 + i];
                H.qt_ok[tq] = true;
            }
            q += 1 + bytes;
        }
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
        if (H.have_sof || n < 6) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad SOF");
        if (s[0] != 8) JPEG_FAIL(ODISE_ERR_UNSUPPORTED, "jpeg: %d-bit samples are not supported", (int)s[0]);
        H.progressive = m == 0xC2;
        H.height = (s[1] << 8) | s[2];
        H.width = (s[3] << 8) | s[4];
        H.ncomp = s[5];
        if (H.ncomp != 1 && H.ncomp != 3) JPEG_FAIL(ODISE_ERR_UNSUPPORTED, "jpeg: %d components (CMYK / YCCK) are not supported", H.ncomp);
        if (n < 6 + 3 * H.ncomp || H.width == 0 || H.height == 0) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad SOF");
        for (int i = 0; i < H.ncomp; ++i) {
            H.id[i] = s[6 + 3 * i];
            H.h[i] = s[7 + 3 * i] >> 4;
            H.v[i] = s[7 + 3 * i] & 15;
            H.tq[i] = s[8 + 3 * i];
            if (H.tq[i] > 3 || H.h[i] < 1 || H.v[i] < 1 || H.h[i] > 4 || H.v[i] > 4) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad component spec");
        }
        H.have_sof = true;
    } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xC7) || (m >= 0xC9 && m <= 0xCB) || (m >= 0xCD && m <= 0xCF)) {
        JPEG_FAIL(ODISE_ERR_UNSUPPORTED, "jpeg: SOF marker 0x%02x (lossless / hierarchical / arithmetic coding) is not supported", m);
    } else if (m == 0xC4) {
        int q = 0;
        while (q < n) {
            if (q + 17 > n) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad DHT");
            const int tc = s[q] >> 4, th = s[q] & 15;
            int total = 0;
            for (int i = 0; i < 16; ++i) total += s[q + 1 + i];
            if (tc > 1 || th > 3 || total > 256 || q + 17 + total > n) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad DHT");
            if (!build_table(s + q + 1, s + q + 17, total, tc ? H.ac[th] : H.dc[th])) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad Huffman table");
            q += 17 + total;
        }
    } else if (m == 0xDD) {
        if (n < 2) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad DRI");
        H.restart = (s[0] << 8) | s[1];
    } else if (m == 0xE0 && n >= 5 && memcmp(s, "JFIF\0", 5) == 0) {
        H.jfif = true;
    } else if (m == 0xEE && n >= 12 && memcmp(s, "Adobe", 5) == 0) {
        H.adobe = s[11];
    } else if (m == 0xE1 && n >= 6 && memcmp(s, "Exif\0\0", 6) == 0 && !H.scans_started) {
        H.orientation = exif_orientation(s + 6, n - 6);
    }
    return ODISE_OK;
}

// Walks marker segments from offset p to the next SOS.  Returns ODISE_OK with *sos = offset of that SOS segment's length field, or
// *sos = -1 when the stream ends (EOI / no more data) first.
static int next_sos(const uint8_t* d, int64_t len, int64_t p, JpegHeader& H, int64_t* sos) {
    *sos = -1;
    for (;;) {
        while (p < len && d[p] != 0xFF) ++p;
        while (p < len && d[p] == 0xFF) ++p;
        if (p >= len) return ODISE_OK;
        const int m = d[p++];
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) return ODISE_OK;
        if (p + 2 > len) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: truncated marker segment");
        const int L = (d[p] << 8) | d[p + 1];
        if (L < 2 || p + L > len) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad marker segment length");
        if (m == 0xDA) {
            *sos = p;
            return ODISE_OK;
        }
        const int rc = handle_segment(m, d + p + 2, L - 2, H);
        if (rc != ODISE_OK) return rc;
        p += L;
    }
}

// Header pass: everything up to the first SOS (which stays unconsumed at H.sos_pos); validates what the decoder relies on.
static int parse_header(const uint8_t* d, int64_t len, JpegHeader& H) {
    if (!d || len < 4 || d[0] != 0xFF || d[1] != 0xD8) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: not a JPEG stream (no SOI)");
    int rc = next_sos(d, len, 2, H, &H.sos_pos);
    if (rc != ODISE_OK) return rc;
    if (H.sos_pos < 0) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: no scan in the stream");
    if (!H.have_sof) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: SOS before SOF");
    // the header alone decides how much memory the decoder asks for: refuse absurd frames instead of allocating for them
    if ((int64_t)H.width * H.height > ((int64_t)1 << 28)) JPEG_FAIL(ODISE_ERR_UNSUPPORTED, "jpeg: %dx%d is larger than the 2^28-pixel limit", H.width, H.height);
    for (int i = 0; i < H.ncomp; ++i)
        if (!H.qt_ok[H.tq[i]]) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: component refers to a missing quantisation table");
    if (H.ncomp == 3) {
        const bool rgb_ids = H.id[0] == 'R' && H.id[1] == 'G' && H.id[2] == 'B';
        if (H.adobe == 0 || (!H.jfif && H.adobe < 0 && rgb_ids)) JPEG_FAIL(ODISE_ERR_UNSUPPORTED, "jpeg: RGB-coded files are not supported");
        const bool luma_ok = (H.h[0] == 1 && H.v[0] == 1) || (H.h[0] == 2 && H.v[0] == 1) || (H.h[0] == 2 && H.v[0] == 2);
        if (!luma_ok || H.h[1] != 1 || H.v[1] != 1 || H.h[2] != 1 || H.v[2] != 1)
            JPEG_FAIL(ODISE_ERR_UNSUPPORTED, "jpeg: sampling factors %dx%d,%dx%d,%dx%d are not supported", H.h[0], H.v[0], H.h[1], H.v[1], H.h[2], H.v[2]);
    } else {
        H.h[0] = H.v[0] = 1;  // a single-component scan is never interleaved
    }
    H.hmax = H.h[0];
    H.vmax = H.v[0];
    H.mcus_x = (H.width + 8 * H.hmax - 1) / (8 * H.hmax);
    H.mcus_y = (H.height + 8 * H.vmax - 1) / (8 * H.vmax);
    int64_t off = 0;
    for (int c = 0; c < H.ncomp; ++c) {
        H.bx[c] = H.mcus_x * H.h[c];
        H.by[c] = H.mcus_y * H.v[c];
        H.coef_off[c] = off;
        off += (int64_t)H.bx[c] * H.by[c] * 64;
    }
    H.coef_count = off;
    return ODISE_OK;
}

// The entropy-coded data is first copied out of the file with the FF00 stuffing removed and cut into restart segments; every segment
// is followed by 16 zero bytes, so the bit reader below can load 8 bytes at a time without checks and a truncated or corrupt stream
// decodes as zero bits (what libjpeg supplies after a premature end) instead of running into its neighbour.
struct Segments {
    std::vector<uint8_t> bytes;
    std::vector<size_t> begin;  // per segment: offset of its first byte; its zero padding ends at the next segment's begin
};

// Returns the offset at which the scan's data ends (the 0xFF of the marker that terminates it, or len).
static int64_t unstuff(const uint8_t* d, int64_t len, int64_t start, Segments& S) {
    S.bytes.clear();
    S.begin.clear();
    S.bytes.reserve((size_t)(len - start) + 64);
    S.begin.push_back(0);
    const uint8_t* p = d + start;
    const uint8_t* end = d + len;
    int64_t stop = len;
    auto pad = [&]() { S.bytes.insert(S.bytes.end(), 16, (uint8_t)0); };
    while (p < end) {
        const uint8_t* ff = (const uint8_t*)memchr(p, 0xFF, (size_t)(end - p));
        if (!ff) { S.bytes.insert(S.bytes.end(), p, end); break; }
        S.bytes.insert(S.bytes.end(), p, ff);
        p = ff + 1;
        while (p < end && *p == 0xFF) ++p;  // fill bytes
        if (p >= end) break;
        const unsigned m = *p++;
        if (m == 0) S.bytes.push_back(0xFF);
        else if (m >= 0xD0 && m <= 0xD7) { pad(); S.begin.push_back(S.bytes.size()); }
        else { stop = (p - 2) - d; break; }  // any other marker ends the scan
    }
    pad();
    S.begin.push_back(S.bytes.size());
    return stop;
}

struct BitReader {
    const uint8_t* p = nullptr;
    const uint8_t* limit = nullptr;  // start of the segment's zero padding: loads never begin beyond it
    uint64_t acc = 0;                // valid bits are MSB-aligned
    int n = 0;
    inline void open(const uint8_t* b, const uint8_t* e) { p = b; limit = e; acc = 0; n = 0; }
    inline void fill() {             // afterwards n >= 56
        if (p > limit) p = limit;
        uint64_t w;
        memcpy(&w, p, 8);
        acc |= __builtin_bswap64(w) >> n;
        p += (63 - n) >> 3;
        n |= 56;
    }
    inline unsigned peek(int k) const { return (unsigned)(acc >> (64 - k)); }
    inline void skip(int k) { acc <<= k; n -= k; }
    inline unsigned take(int k) { const unsigned v = (unsigned)(acc >> (64 - k)); acc <<= k; n -= k; return v; }
    inline int symbol(const HuffTable& t) {  // needs >= 16 valid bits
        const unsigned look = peek(9);
        const int nb = t.look_bits[look];
        if (nb) { skip(nb); return t.look_sym[look]; }
        for (int len = 10; len <= 16; ++len) {
            const int code = (int)peek(len);
            if (code <= t.maxcode[len]) { skip(len); return t.vals[(code + t.valoff[len]) & 255]; }
        }
        skip(16);
        return -1;  // not a code of this table (corrupt data)
    }
};

static inline int extend(unsigned v, int s) { return (s && v < (1u << (s - 1))) ? (int)v - (1 << s) + 1 : (int)v; }

struct ScanSpec {
    int ncomp = 0;
    int ci[3] = {0, 0, 0};               // frame component index of every scan component
    int td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    int ss = 0, se = 63, ah = 0, al = 0;  // spectral selection and successive approximation (0, 63, 0, 0 for sequential scans)
};

static int parse_scan_header(const uint8_t* d, int64_t len, int64_t pos, const JpegHeader& H, ScanSpec& sp, int64_t* data_start) {
    const int L = (d[pos] << 8) | d[pos + 1];  // validated by next_sos
    const uint8_t* s = d + pos + 2;
    const int n = L - 2;
    if (n < 1) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad SOS");
    sp.ncomp = s[0];
    if (sp.ncomp < 1 || sp.ncomp > H.ncomp || n < 1 + 2 * sp.ncomp + 3) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad SOS");
    if (sp.ncomp > 1 && sp.ncomp != H.ncomp) JPEG_FAIL(ODISE_ERR_UNSUPPORTED, "jpeg: interleaved scans over a subset of the components are not supported");
    for (int i = 0; i < sp.ncomp; ++i) {
        int c = -1;
        for (int k = 0; k < H.ncomp; ++k)
            if (H.id[k] == s[1 + 2 * i]) c = k;
        if (c < 0 || (i > 0 && c <= sp.ci[i - 1])) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad scan component");
        sp.ci[i] = c;
        sp.td[i] = s[2 + 2 * i] >> 4;
        sp.ta[i] = s[2 + 2 * i] & 15;
        if (sp.td[i] > 3 || sp.ta[i] > 3) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad Huffman table index");
    }
    const uint8_t* t = s + 1 + 2 * sp.ncomp;
    sp.ss = t[0];
    sp.se = t[1];
    sp.ah = t[2] >> 4;
    sp.al = t[2] & 15;
    if (H.progressive) {
        const bool ok = sp.ss <= sp.se && sp.se <= 63 && sp.ah <= 13 && sp.al <= 13 && (sp.ss == 0 ? sp.se == 0 : sp.ncomp == 1) &&
                        (sp.ah == 0 || sp.ah == sp.al + 1);
        if (!ok) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: bad progressive scan parameters");
    } else {
        sp.ss = 0;
        sp.se = 63;
        sp.ah = sp.al = 0;
    }
    const bool need_dc = sp.ss == 0 && sp.ah == 0, need_ac = sp.se > 0;
    for (int i = 0; i < sp.ncomp; ++i) {
        if (need_dc && !H.dc[sp.td[i]].present) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: scan refers to a missing DC Huffman table");
        if (need_ac && !H.ac[sp.ta[i]].present) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: scan refers to a missing AC Huffman table");
    }
    *data_start = pos + L;
    return ODISE_OK;
}

// Block walk of one scan: interleaved scans visit MCUs (h x v blocks of every component), single-component scans visit the component's
// own ceil(w/8) x ceil(h/8) blocks in raster order (A.2.3 of the standard); `blk(c, by, bx)` = the block inside the MCU-padded grid.
template <class Body>
static void walk_scan(const JpegHeader& H, const ScanSpec& sp, const Segments& S, Body&& body) {
    const size_t nseg = S.begin.size() - 1;
    static const uint8_t zeros[32] = {0};
    size_t seg = 0;
    BitReader br;
    auto open_segment = [&]() {
        if (seg < nseg) br.open(S.bytes.data() + S.begin[seg], S.bytes.data() + S.begin[seg + 1] - 16);
        else br.open(zeros, zeros);  // the stream has fewer restart segments than MCUs need: zero bits
        ++seg;
    };
    open_segment();
    int units_x, units_y;
    if (sp.ncomp > 1) {
        units_x = H.mcus_x;
        units_y = H.mcus_y;
    } else {
        const int c = sp.ci[0];
        const int cw = (H.width * H.h[c] + H.hmax - 1) / H.hmax, ch = (H.height * H.v[c] + H.vmax - 1) / H.vmax;
        units_x = (cw + 7) / 8;
        units_y = (ch + 7) / 8;
    }
    int until_restart = H.restart;
    for (int uy = 0; uy < units_y; ++uy)
        for (int ux = 0; ux < units_x; ++ux) {
            bool restarted = false;
            if (H.restart) {
                if (until_restart == 0) {
                    open_segment();
                    until_restart = H.restart;
                    restarted = true;
                }
                --until_restart;
            }
            body(br, uy, ux, restarted);
        }
}

// Sequential (baseline / extended) scan: DC difference + run-length coded AC coefficients of every block.
static void decode_scan_sequential(const JpegHeader& H, const ScanSpec& sp, const Segments& S, int16_t* coefs) {
    int pred[3] = {0, 0, 0};
    auto block = [&](BitReader& br, int i, int by, int bx) {
        const int c = sp.ci[i];
        const HuffTable& dct = H.dc[sp.td[i]];
        const HuffTable& act = H.ac[sp.ta[i]];
        int16_t* blk = coefs + H.coef_off[c] + ((int64_t)by * H.bx[c] + bx) * 64;
        br.fill();
        int s = br.symbol(dct);
        if (s < 0 || s > 16) s = 0;
        pred[i] += extend(s ? br.take(s) : 0u, s);
        blk[0] = (int16_t)pred[i];
        for (int k = 1; k < 64;) {
            br.fill();
            const unsigned look = br.peek(9);
            const int fl = act.fast_len[look];
            if (fl) {  // code and magnitude bits inside the lookahead
                k += act.fast_run[look];
                if (k > 63) break;
                br.skip(fl);
                blk[kZigzag[k++]] = act.fast_val[look];
                continue;
            }
            const int rs = br.symbol(act);
            if (rs < 0) break;
            const int r = rs >> 4;
            s = rs & 15;
            if (s == 0) {
                if (r != 15) break;  // EOB
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) break;  // corrupt data
            blk[kZigzag[k]] = (int16_t)extend(br.take(s), s);
            ++k;
        }
    };
    walk_scan(H, sp, S, [&](BitReader& br, int uy, int ux, bool restarted) {
        if (restarted) pred[0] = pred[1] = pred[2] = 0;
        if (sp.ncomp > 1) {
            for (int i = 0; i < sp.ncomp; ++i) {
                const int c = sp.ci[i];
                for (int v = 0; v < H.v[c]; ++v)
                    for (int h = 0; h < H.h[c]; ++h) block(br, i, uy * H.v[c] + v, ux * H.h[c] + h);
            }
        } else {
            block(br, 0, uy, ux);
        }
    });
}

// Progressive scans (ITU T.81 annex G, jdphuff.c): DC first / refinement over interleaved MCUs, AC first / refinement over the blocks
// of one component for the band [ss, se], successive approximation bit `al`, end-of-band runs.
static void decode_scan_progressive(const JpegHeader& H, const ScanSpec& sp, const Segments& S, int16_t* coefs) {
    int pred[3] = {0, 0, 0};
    int eobrun = 0;
    const int al = sp.al;
    auto dc_block = [&](BitReader& br, int i, int by, int bx) {
        const int c = sp.ci[i];
        int16_t* blk = coefs + H.coef_off[c] + ((int64_t)by * H.bx[c] + bx) * 64;
        br.fill();
        if (sp.ah == 0) {
            int s = br.symbol(H.dc[sp.td[i]]);
            if (s < 0 || s > 16) s = 0;
            pred[i] += extend(s ? br.take(s) : 0u, s);
            blk[0] = (int16_t)(pred[i] * (1 << al));
        } else if (br.take(1)) {
            blk[0] = (int16_t)(blk[0] | (1 << al));
        }
    };
    auto ac_first = [&](BitReader& br, int16_t* blk) {
        if (eobrun > 0) { --eobrun; return; }
        const HuffTable& act = H.ac[sp.ta[0]];
        for (int k = sp.ss; k <= sp.se;) {
            br.fill();
            const int rs = br.symbol(act);
            if (rs < 0) return;
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {  // EOBr: this block and the next 2^r + extra - 1 blocks have no more coefficients in the band
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += (int)br.take(r);
                    return;
                }
                k += 16;  // ZRL
                continue;
            }
            k += r;
            if (k > 63) return;  // corrupt data
            blk[kZigzag[k]] = (int16_t)(extend(br.take(s), s) * (1 << al));
            ++k;
        }
    };
    auto refine_nonzero = [&](BitReader& br, int16_t& v) {  // one correction bit for a coefficient with history
        br.fill();
        if (br.take(1) && (v & (1 << al)) == 0) v = (int16_t)(v >= 0 ? v + (1 << al) : v - (1 << al));
    };
    auto ac_refine = [&](BitReader& br, int16_t* blk) {
        const HuffTable& act = H.ac[sp.ta[0]];
        int k = sp.ss;
        if (eobrun == 0) {
            while (k <= sp.se) {
                br.fill();
                const int rs = br.symbol(act);
                if (rs < 0) return;
                int r = rs >> 4;
                const int s = rs & 15;
                int value = 0;
                if (s == 0) {
                    if (r < 15) {
                        eobrun = 1 << r;
                        if (r) eobrun += (int)br.take(r);
                        break;  // the rest of this block is handled as part of the run
                    }
                    // ZRL: skip 16 zero-history coefficients
                } else {
                    value = br.take(1) ? (1 << al) : -(1 << al);  // s must be 1
                }
                for (; k <= sp.se; ++k) {
                    int16_t& v = blk[kZigzag[k]];
                    if (v != 0) {
                        refine_nonzero(br, v);
                    } else {
                        if (r == 0) {
                            if (value) v = (int16_t)value;
                            ++k;
                            break;
                        }
                        --r;
                    }
                }
            }
        }
        if (eobrun > 0) {  // inside an end-of-band run: only correction bits for the coefficients that already have history
            for (; k <= sp.se; ++k) {
                int16_t& v = blk[kZigzag[k]];
                if (v != 0) refine_nonzero(br, v);
            }
            --eobrun;
        }
    };
    walk_scan(H, sp, S, [&](BitReader& br, int uy, int ux, bool restarted) {
        if (restarted) { pred[0] = pred[1] = pred[2] = 0; eobrun = 0; }
        if (sp.ss == 0) {
            if (sp.ncomp > 1) {
                for (int i = 0; i < sp.ncomp; ++i) {
                    const int c = sp.ci[i];
                    for (int v = 0; v < H.v[c]; ++v)
                        for (int h = 0; h < H.h[c]; ++h) dc_block(br, i, uy * H.v[c] + v, ux * H.h[c] + h);
                }
            } else {
                dc_block(br, 0, uy, ux);
            }
        } else {
            const int c = sp.ci[0];
            int16_t* blk = coefs + H.coef_off[c] + ((int64_t)uy * H.bx[c] + ux) * 64;
            if (sp.ah == 0) ac_first(br, blk);
            else ac_refine(br, blk);
        }
    });
}

// All scans of the stream into coefs (H.coef_count int16, zero-filled by the caller; natural order inside a block, not dequantised).
// Tables defined between scans (DHT, DRI) take effect for the following scans.
static int entropy_decode(const uint8_t* d, int64_t len, JpegHeader& H, int16_t* coefs) {
    Segments S;
    int64_t pos = H.sos_pos;
    H.scans_started = true;
    for (int scans = 0; pos >= 0; ++scans) {
        if (scans >= 256) JPEG_FAIL(ODISE_ERR_ARG, "jpeg: too many scans");
        ScanSpec sp;
        int64_t data_start = 0;
        int rc = parse_scan_header(d, len, pos, H, sp, &data_start);
        if (rc != ODISE_OK) return scans ? (int)ODISE_OK : rc;  // damage after the first scan: keep what has been decoded (libjpeg warns and goes on)
        const int64_t stop = unstuff(d, len, data_start, S);
        if (H.progressive) decode_scan_progressive(H, sp, S, coefs);
        else decode_scan_sequential(H, sp, S, coefs);
        if (next_sos(d, len, stop, H, &pos) != ODISE_OK) break;
    }
    return ODISE_OK;
}

// ---- device side -----------------------------------------------------------------------------------------------------------------
struct JpegPlanes {
    int ncomp;
    int bx[3], by[3];        // block grids
    int64_t coef_off[3];     // int16 elements
    int64_t plane_off[3];    // bytes
    int64_t first_block[3];  // running block index of the component's first block
    int64_t blocks;
    uint16_t qt[3][64];
};

__device__ __forceinline__ void idct8(const int (&r)[8], int (&o)[8]) {
    // jidctint.c: even part
    int z2 = r[2], z3 = r[6];
    int z1 = (z2 + z3) * 4433;
    const int tmp2 = z1 - z3 * 15137;
    const int tmp3 = z1 + z2 * 6270;
    const int tmp0 = (r[0] + r[4]) << 13;
    const int tmp1 = (r[0] - r[4]) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    // odd part
    int t0 = r[7], t1 = r[5], t2 = r[3], t3 = r[1];
    z1 = t0 + t3;
    z2 = t1 + t2;
    z3 = t0 + t2;
    int z4 = t1 + t3;
    const int z5 = (z3 + z4) * 9633;
    t0 *= 2446;
    t1 *= 16819;
    t2 *= 25172;
    t3 *= 12299;
    z1 *= -7373;
    z2 *= -20995;
    z3 = z3 * -16069 + z5;
    z4 = z4 * -3196 + z5;
    t0 += z1 + z3;
    t1 += z2 + z4;
    t2 += z2 + z3;
    t3 += z1 + z4;
    o[0] = tmp10 + t3;
    o[7] = tmp10 - t3;
    o[1] = tmp11 + t2;
    o[6] = tmp11 - t2;
    o[2] = tmp12 + t1;
    o[5] = tmp12 - t1;
    o[3] = tmp13 + t0;
    o[4] = tmp13 - t0;
}

__global__ void __launch_bounds__(256) jpeg_idct_kernel(const int16_t* __restrict__ coefs, uint8_t* __restrict__ planes, JpegPlanes P) {
    __shared__ int ws[32][8][9];  // [block in the workgroup][row][col], padded
    const int tid = threadIdx.x, lb = tid >> 3, c8 = tid & 7;
    const int64_t blk = (int64_t)blockIdx.x * 32 + lb;
    const bool live = blk < P.blocks;
    int comp = 0;
    if (P.ncomp == 3) comp = blk >= P.first_block[2] ? 2 : (blk >= P.first_block[1] ? 1 : 0);
    const int64_t bi = blk - P.first_block[comp];
    int r[8], o[8];
    if (live) {
        const int16_t* cb = coefs + P.coef_off[comp] + bi * 64;
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = (int)cb[k * 8 + c8] * (int)P.qt[comp][k * 8 + c8];  // pass 1: this lane's column
        idct8(r, o);
#pragma unroll
        for (int k = 0; k < 8; ++k) ws[lb][k][c8] = (o[k] + (1 << 10)) >> 11;
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = ws[lb][c8][k];  // pass 2: this lane's row
        idct8(r, o);
        const int byi = (int)(bi / P.bx[comp]), bxi = (int)(bi - (int64_t)byi * P.bx[comp]);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int val = ((o[k] + (1 << 17)) >> 18) + 128;
            val = val < 0 ? 0 : (val > 255 ? 255 : val);
            if (k < 4) lo |= (uint32_t)val << (8 * k);
            else hi |= (uint32_t)val << (8 * (k - 4));
        }
        uint8_t* dst = planes + P.plane_off[comp] + ((int64_t)(byi * 8 + c8) * P.bx[comp] + bxi) * 8;
        *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
    }
}

struct JpegColor {
    int W, H;            // coded image size
    int OW, OH;          // output size (after the orientation)
    int orientation;
    int ncomp, hx, vx;   // chroma expansion factors (1 | 2)
    int dw, dh;          // real chroma samples
    int pitch[3];
    int64_t plane_off[3];
};

__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ p, int pitch, int dw, int dh, int hx, int vx, int x, int y) {
    if (hx == 1) return p[(int64_t)y * pitch + x];
    const int cx = x >> 1;
    if (dw <= 2) return p[(int64_t)(vx == 2 ? (y >> 1) : y) * pitch + cx];  // jdsample.c: plain replication for narrow components
    const int nb = (x & 1) ? (cx + 1 < dw ? cx + 1 : cx) : (cx > 0 ? cx - 1 : 0);  // the further column (clamped: reproduces the edge rules)
    if (vx == 1) {
        const uint8_t* row = p + (int64_t)y * pitch;
        return (3 * row[cx] + row[nb] + ((x & 1) ? 2 : 1)) >> 2;
    }
    const int cy = y >> 1;
    const int fy = (y & 1) ? (cy + 1 < dh ? cy + 1 : cy) : (cy > 0 ? cy - 1 : 0);
    const uint8_t* rn = p + (int64_t)cy * pitch;
    const uint8_t* rf = p + (int64_t)fy * pitch;
    const int s_this = 3 * rn[cx] + rf[cx], s_nb = 3 * rn[nb] + rf[nb];
    return (3 * s_this + s_nb + ((x & 1) ? 7 : 8)) >> 4;
}

__global__ void __launch_bounds__(256) jpeg_color_kernel(const uint8_t* __restrict__ planes, uint8_t* __restrict__ out, JpegColor G) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)G.OW * G.OH) return;
    const int oy = (int)(idx / G.OW), ox = (int)(idx - (int64_t)oy * G.OW);
    int x = ox, y = oy;
    switch (G.orientation) {  // PIL.ImageOps.exif_transpose as an index map
        case 2: x = G.W - 1 - ox; break;
        case 3: x = G.W - 1 - ox; y = G.H - 1 - oy; break;
        case 4: y = G.H - 1 - oy; break;
        case 5: x = oy; y = ox; break;
        case 6: x = oy; y = G.H - 1 - ox; break;
        case 7: x = G.W - 1 - oy; y = G.H - 1 - ox; break;
        case 8: x = G.W - 1 - oy; y = ox; break;
        default: break;
    }
    const int Y = planes[G.plane_off[0] + (int64_t)y * G.pitch[0] + x];
    int R = Y, Gc = Y, B = Y;
    if (G.ncomp == 3) {
        const int cb = chroma_at(planes + G.plane_off[1], G.pitch[1], G.dw, G.dh, G.hx, G.vx, x, y) - 128;
        const int cr = chroma_at(planes + G.plane_off[2], G.pitch[2], G.dw, G.dh, G.hx, G.vx, x, y) - 128;
        R = Y + ((91881 * cr + 32768) >> 16);                       // jdcolor.c build_ycc_rgb_table
        Gc = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
        B = Y + ((116130 * cb + 32768) >> 16);
        R = R < 0 ? 0 : (R > 255 ? 255 : R);
        Gc = Gc < 0 ? 0 : (Gc > 255 ? 255 : Gc);
        B = B < 0 ? 0 : (B > 255 ? 255 : B);
    }
    uint8_t* o = out + idx * 3;
    o[0] = (uint8_t)R;
    o[1] = (uint8_t)Gc;
    o[2] = (uint8_t)B;
}

static void fill_info(const JpegHeader& H, odise_jpeg_info* info) {
    info->width = H.width;
    info->height = H.height;
    info->components = H.ncomp;
    info->h_samp = H.hmax;
    info->v_samp = H.vmax;
    info->orientation = H.orientation;
    info->restart_interval = H.restart;
    for (int c = 0; c < 3; ++c) {
        info->blocks_x[c] = c < H.ncomp ? H.bx[c] : 0;
        info->blocks_y[c] = c < H.ncomp ? H.by[c] : 0;
    }
    info->coef_count = H.coef_count;
}

void jpeg_release(odise_hip_ctx* ctx) {
    if (ctx->jpeg_host) (void)hipHostFree(ctx->jpeg_host);
    if (ctx->jpeg_dev) (void)hipFree(ctx->jpeg_dev);
    if (ctx->jpeg_ev) (void)hipEventDestroy(ctx->jpeg_ev);
    ctx->jpeg_host = ctx->jpeg_dev = nullptr;
    ctx->jpeg_host_bytes = ctx->jpeg_dev_bytes = 0;
    ctx->jpeg_ev = nullptr;
}

}  // namespace odise

using namespace odise;

extern "C" int odise_hip_jpeg_info(const void* data, int64_t len, odise_jpeg_info* info) {
    ODISE_REQUIRE(data && info && len > 0, "jpeg_info: null argument");
    std::vector<JpegHeader> hv(1);  // the header holds eight Huffman tables: keep it off the stack
    const int rc = parse_header((const uint8_t*)data, len, hv[0]);
    if (rc != ODISE_OK) return rc;
    fill_info(hv[0], info);
    return ODISE_OK;
}

extern "C" int odise_hip_jpeg_entropy_decode(const void* data, int64_t len, int16_t* coefs, int64_t capacity, uint16_t* qtables) {
    ODISE_REQUIRE(data && coefs && len > 0, "jpeg_entropy_decode: null argument");
    std::vector<JpegHeader> hv(1);
    JpegHeader& H = hv[0];
    const int rc = parse_header((const uint8_t*)data, len, H);
    if (rc != ODISE_OK) return rc;
    ODISE_REQUIRE(capacity >= H.coef_count, "jpeg_entropy_decode: coefficient buffer too small (%lld < %lld)", (long long)capacity, (long long)H.coef_count);
    memset(coefs, 0, (size_t)H.coef_count * sizeof(int16_t));
    const int drc = entropy_decode((const uint8_t*)data, len, H, coefs);
    if (drc != ODISE_OK) return drc;
    if (qtables)
        for (int c = 0; c < H.ncomp; ++c) memcpy(qtables + 64 * c, H.qt[H.tq[c]], 64 * sizeof(uint16_t));
    return ODISE_OK;
}

namespace odise {
struct JpegDims {
    int width, height, ncomp, hmax, vmax, orientation;
    int bx[3], by[3];
    int64_t coef_off[3], coef_count;
    uint16_t qt[3][64];
};

// Device half: `fill(int16_t* pinned)` produces the coefficients in the context's pinned staging buffer; upload, IDCT, colour.
template <class Fill>
static int jpeg_device_stage(odise_hip_ctx* ctx, const JpegDims& D, Fill&& fill, void* dst_rgb, int64_t dst_capacity, int apply_orientation, int* out_h,
                             int* out_w) {
    const int orient = apply_orientation ? D.orientation : 1;
    const bool swap = orient >= 5;
    const int OH = swap ? D.width : D.height, OW = swap ? D.height : D.width;
    if (out_h) *out_h = OH;
    if (out_w) *out_w = OW;
    ODISE_REQUIRE(dst_capacity >= (int64_t)OH * OW * 3, "jpeg_decode: output buffer too small for %dx%d RGB", OH, OW);
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));
    // ---- staging: pinned host coefficients (reused once the previous image's upload has finished), device coefficients + planes
    const size_t coef_bytes = (size_t)D.coef_count * sizeof(int16_t);
    JpegPlanes P;
    JpegColor G;
    P.ncomp = G.ncomp = D.ncomp;
    int64_t poff = (int64_t)round_up((int64_t)coef_bytes, 256), nblk = 0;
    for (int c = 0; c < 3; ++c) {
        const bool on = c < D.ncomp;
        P.bx[c] = on ? D.bx[c] : 0;
        P.by[c] = on ? D.by[c] : 0;
        P.coef_off[c] = on ? D.coef_off[c] : 0;
        P.plane_off[c] = G.plane_off[c] = poff;
        P.first_block[c] = nblk;
        G.pitch[c] = P.bx[c] * 8;
        if (on) {
            memcpy(P.qt[c], D.qt[c], sizeof(P.qt[c]));
            poff += (int64_t)round_up((int64_t)P.bx[c] * P.by[c] * 64, 256);
            nblk += (int64_t)P.bx[c] * P.by[c];
        } else {
            memset(P.qt[c], 0, sizeof(P.qt[c]));
        }
    }
    P.blocks = nblk;
    if (!ctx->jpeg_ev) ODISE_CHECK_HIP(hipEventCreateWithFlags(&ctx->jpeg_ev, hipEventDisableTiming));
    else ODISE_CHECK_HIP(hipEventSynchronize(ctx->jpeg_ev));
    if (ctx->jpeg_host_bytes < coef_bytes) {
        if (ctx->jpeg_host) (void)hipHostFree(ctx->jpeg_host);
        ctx->jpeg_host = nullptr;
        ctx->jpeg_host_bytes = 0;
        const size_t want = coef_bytes + coef_bytes / 4;
        if (hipHostMalloc(&ctx->jpeg_host, want, hipHostMallocDefault) != hipSuccess) { set_error("jpeg_decode: cannot pin %zu bytes", want); return ODISE_ERR_NOMEM; }
        ctx->jpeg_host_bytes = want;
    }
    if (ctx->jpeg_dev_bytes < (size_t)poff) {
        ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->jpeg_dev) (void)hipFree(ctx->jpeg_dev);
        ctx->jpeg_dev = nullptr;
        ctx->jpeg_dev_bytes = 0;
        const size_t want = (size_t)poff + (size_t)poff / 4;
        if (hipMalloc(&ctx->jpeg_dev, want) != hipSuccess) { set_error("jpeg_decode: out of device memory (%zu bytes)", want); return ODISE_ERR_NOMEM; }
        ctx->jpeg_dev_bytes = want;
    }
    int16_t* hc = (int16_t*)ctx->jpeg_host;
    const int frc = fill(hc);
    if (frc != ODISE_OK) return frc;
    ODISE_CHECK_HIP(hipMemcpyAsync(ctx->jpeg_dev, hc, coef_bytes, hipMemcpyHostToDevice, ctx->stream));
    ODISE_CHECK_HIP(hipEventRecord(ctx->jpeg_ev, ctx->stream));
    uint8_t* dev = (uint8_t*)ctx->jpeg_dev;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)ceil_div(nblk, 32)), dim3(256), 0, ctx->stream, (const int16_t*)dev, dev, P);
    ODISE_CHECK_HIP(hipGetLastError());
    G.W = D.width;
    G.H = D.height;
    G.OW = OW;
    G.OH = OH;
    G.orientation = orient;
    G.hx = D.hmax;  // chroma is 1x1: its expansion factors are the luma sampling factors
    G.vx = D.vmax;
    G.dw = (D.width + D.hmax - 1) / D.hmax;
    G.dh = (D.height + D.vmax - 1) / D.vmax;
    const int64_t npix = (int64_t)OW * OH;
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)ceil_div(npix, 256)), dim3(256), 0, ctx->stream, (const uint8_t*)dev, (uint8_t*)dst_rgb, G);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
}  // namespace odise

extern "C" int odise_hip_jpeg_decode(odise_hip_ctx* ctx, const void* data, int64_t len, void* dst_rgb, int64_t dst_capacity, int apply_orientation,
                                     int* out_h, int* out_w) {
    ODISE_REQUIRE(ctx && data && dst_rgb && len > 0, "jpeg_decode: null argument");
    std::vector<JpegHeader> hv(1);
    JpegHeader& H = hv[0];
    const int rc = parse_header((const uint8_t*)data, len, H);
    if (rc != ODISE_OK) return rc;
    JpegDims D;
    D.width = H.width;
    D.height = H.height;
    D.ncomp = H.ncomp;
    D.hmax = H.hmax;
    D.vmax = H.vmax;
    D.orientation = H.orientation;
    D.coef_count = H.coef_count;
    for (int c = 0; c < 3; ++c) {
        D.bx[c] = H.bx[c];
        D.by[c] = H.by[c];
        D.coef_off[c] = H.coef_off[c];
        if (c < H.ncomp) memcpy(D.qt[c], H.qt[H.tq[c]], sizeof(D.qt[c]));
    }
    return jpeg_device_stage(ctx, D, [&](int16_t* hc) {
        memset(hc, 0, (size_t)H.coef_count * sizeof(int16_t));
        return entropy_decode((const uint8_t*)data, len, H, hc);
    }, dst_rgb, dst_capacity, apply_orientation, out_h, out_w);
}

extern "C" int odise_hip_jpeg_decode_coefs(odise_hip_ctx* ctx, const odise_jpeg_info* info, const int16_t* coefs, const uint16_t* qtables, void* dst_rgb,
                                           int64_t dst_capacity, int apply_orientation, int* out_h, int* out_w) {
    ODISE_REQUIRE(ctx && info && coefs && qtables && dst_rgb, "jpeg_decode_coefs: null argument");
    JpegDims D;
    D.width = info->width;
    D.height = info->height;
    D.ncomp = info->components;
    D.hmax = info->h_samp;
    D.vmax = info->v_samp;
    D.orientation = (info->orientation >= 1 && info->orientation <= 8) ? info->orientation : 1;
    ODISE_REQUIRE(D.width > 0 && D.height > 0 && D.width <= 65535 && D.height <= 65535 && (D.ncomp == 1 || D.ncomp == 3), "jpeg_decode_coefs: bad image description");
    const bool samp_ok = D.ncomp == 1 ? (D.hmax == 1 && D.vmax == 1) : ((D.hmax == 1 && D.vmax == 1) || (D.hmax == 2 && (D.vmax == 1 || D.vmax == 2)));
    ODISE_REQUIRE(samp_ok, "jpeg_decode_coefs: sampling factors %dx%d are not supported", D.hmax, D.vmax);
    const int mx = (D.width + 8 * D.hmax - 1) / (8 * D.hmax), my = (D.height + 8 * D.vmax - 1) / (8 * D.vmax);
    int64_t off = 0;
    for (int c = 0; c < 3; ++c) {
        D.bx[c] = c < D.ncomp ? mx * (c == 0 ? D.hmax : 1) : 0;
        D.by[c] = c < D.ncomp ? my * (c == 0 ? D.vmax : 1) : 0;
        D.coef_off[c] = off;
        off += (int64_t)D.bx[c] * D.by[c] * 64;
        if (c < D.ncomp) {
            ODISE_REQUIRE(info->blocks_x[c] == D.bx[c] && info->blocks_y[c] == D.by[c], "jpeg_decode_coefs: block grid does not match the image size");
            memcpy(D.qt[c], qtables + 64 * c, sizeof(D.qt[c]));
        }
    }
    D.coef_count = off;
    ODISE_REQUIRE(info->coef_count == off, "jpeg_decode_coefs: coefficient count does not match the image size");
    return jpeg_device_stage(ctx, D, [&](int16_t* hc) { memcpy(hc, coefs, (size_t)off * sizeof(int16_t)); return (int)ODISE_OK; }, dst_rgb, dst_capacity, apply_orientation,
                             out_h, out_w);
}
