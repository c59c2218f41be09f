// Baseline JPEG decoding for the frame ingest in front of hot path A.
//
// What it replaces (paths relative to the reference checkout):
//   src/features_GPU_compute/calcSig_wOF.py:92         cv2.imread(img_NNNNN.jpg, cv2.IMREAD_COLOR)
//   src/features_GPU_compute/calcSig_wOF.py:105-106    cv2.imread(flow_{x,y}_NNNNN.jpg, cv2.IMREAD_GRAYSCALE)
// i.e. libjpeg(-turbo) at its defaults: integer "islow" IDCT, fancy (triangle) chroma upsampling, fixed-point YCbCr -> RGB.
// The arithmetic is the one restated in oracle/jpeg_oracle.py, which is pinned bit for bit against libjpeg-turbo (through
// Pillow): the same bits come out here.
//
// Split of the work: the entropy-coded segment is a serial bit stream -- marker parsing and Huffman decoding (ITU-T T.81
// F.2.2) run on the host, one frame per host thread, into 16-bit coefficient blocks; everything that is data-parallel
// runs on the GPU for the whole batch of frames at once:
//   jpeg_idct_kernel      one thread per 8x8 block: dequantise, two-pass 13-bit fixed-point IDCT in registers, +128, clamp,
//                         eight 8-byte row stores into the component plane
//   jpeg_pixels_kernel    one thread per output pixel: h2v2 / h2v1 triangle-filter upsampling of the chroma planes with
//                         libjpeg's alternating rounding, the 16-bit fixed-point colour transform, BGR (cv2 order) or the
//                         Y plane for a grey read
// Frames of a call share one size (video frames); sampling factors (4:4:4, 4:2:2, 4:2:0, one component) may differ.
// Progressive, arithmetic-coded, 12-bit, multi-scan and CMYK files are refused (VQ_E_UNSUPPORTED) -- nothing in the
// reference's pipeline writes them.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "vq_common.h"

using namespace vq;

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    // lookup of the first 9 bits -> (symbol, code length), 0 length = longer code; canonical tables for the rest (T.81 F.2.2.3)
    uint8_t look_sym[512], look_len[512];
    int maxcode[18], valptr[17], mincode[17];
    uint8_t vals[256];
    bool present = false;
};

bool build_huff(const uint8_t* counts, const uint8_t* symbols, int n_symbols, Huff& h) {
    int code = 0, k = 0;
    memset(h.look_len, 0, sizeof h.look_len);
    for (int ln = 1; ln <= 16; ++ln) {
        h.valptr[ln] = k;
        h.mincode[ln] = code;
        for (int i = 0; i < counts[ln - 1]; ++i) {
            if (k >= n_symbols || k >= 256) return false;
            h.vals[k] = symbols[k];
            if (ln <= 9) {
                const int first = code << (9 - ln), span = 1 << (9 - ln);
                if (first + span > 512) return false;
                for (int q = 0; q < span; ++q) {
                    h.look_sym[first + q] = symbols[k];
                    h.look_len[first + q] = (uint8_t)ln;
                }
            }
            ++code;
            ++k;
        }
        h.maxcode[ln] = counts[ln - 1] ? code - 1 : -1;
        if (code > (1 << ln)) return false;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    h.present = true;
    return true;
}

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;        // blocks per row / column of the decoded plane
};

struct Frame {
    int H = 0, W = 0, nc = 0, hmax = 1, vmax = 1, ri = 0;
    Comp comp[3];
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    size_t scan = 0;
};

struct BitReader {
    const uint8_t* d;
    size_t n, p;
    uint64_t acc = 0;
    int bits = 0;
    bool hit_marker = false;
    void fill() {                       // keep at least 25 bits; behind a marker the stream continues with zeros
        while (bits <= 56) {
            uint32_t b = 0;
            if (!hit_marker && p < n) {
                b = d[p];
                if (b == 0xFF) {
                    const uint8_t nx = p + 1 < n ? d[p + 1] : 0xD9;
                    if (nx == 0) {
                        p += 2;
                    } else {
                        hit_marker = true;
                        b = 0;
                    }
                } else {
                    ++p;
                }
            }
            acc = (acc << 8) | b;
            bits += 8;
        }
    }
    inline uint32_t peek(int k) { return (uint32_t)((acc >> (bits - k)) & ((1u << k) - 1)); }
    inline void skip(int k) { bits -= k; }
    inline uint32_t get(int k) {
        if (k == 0) return 0;
        if (bits < k) fill();
        const uint32_t v = peek(k);
        bits -= k;
        return v;
    }
    bool restart() {                    // discard padding, consume the RSTn marker
        acc = 0;
        bits = 0;
        hit_marker = false;
        while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7)) ++p;
        if (p + 1 >= n) return false;
        p += 2;
        return true;
    }
};

inline int decode_symbol(BitReader& br, const Huff& h) {
    if (br.bits < 16) br.fill();
    const uint32_t look = br.peek(9);
    const int ln = h.look_len[look];
    if (ln) {
        br.skip(ln);
        return h.look_sym[look];
    }
    int code = (int)br.peek(10);
    int l = 10;
    while (l <= 16 && code > h.maxcode[l]) {
        ++l;
        code = (int)br.peek(l);
    }
    if (l > 16) return -1;
    br.skip(l);
    const int idx = h.valptr[l] + code - h.mincode[l];
    return idx >= 0 && idx < 256 ? h.vals[idx] : -1;
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// Marker segments up to the start of the (single) scan.  Returns VQ_OK or an error code with the message set.
int parse_headers(const uint8_t* d, size_t n, Frame& f) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail(VQ_E_INVALID, "not a JPEG file (no SOI marker)");
    size_t p = 2;
    bool have_sof = false;
    for (;;) {
        while (p < n && d[p] != 0xFF) ++p;
        while (p < n && d[p] == 0xFF) ++p;
        if (p >= n) return fail(VQ_E_INVALID, "JPEG: no scan found");
        const int m = d[p++];
        if (m == 0xD9) return fail(VQ_E_INVALID, "JPEG: end of image before any scan");
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;      // markers without a length
        if (p + 2 > n) return fail(VQ_E_INVALID, "JPEG: truncated marker segment");
        const int ln = be16(d + p);
        if (ln < 2 || p + ln > n) return fail(VQ_E_INVALID, "JPEG: marker segment runs past the end of the file");
        const uint8_t* s = d + p + 2;
        const int sl = ln - 2;
        p += ln;
        if (m == 0xDB) {
            for (int q = 0; q < sl;) {
                const int prec = s[q] >> 4, tid = s[q] & 15;
                ++q;
                if (tid > 3 || q + (prec ? 128 : 64) > sl) return fail(VQ_E_INVALID, "JPEG: bad quantisation table");
                for (int k = 0; k < 64; ++k) f.qt[tid][kZigzag[k]] = prec ? (uint16_t)be16(s + q + 2 * k) : s[q + k];
                f.qt_present[tid] = true;
                q += prec ? 128 : 64;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6) return fail(VQ_E_INVALID, "JPEG: bad frame header");
            if (s[0] != 8) return fail(VQ_E_UNSUPPORTED, "JPEG: %d-bit samples (only 8-bit files are decoded)", s[0]);
            f.H = be16(s + 1);
            f.W = be16(s + 3);
            f.nc = s[5];
            if (f.nc != 1 && f.nc != 3) return fail(VQ_E_UNSUPPORTED, "JPEG: %d components (1 or 3 are decoded)", f.nc);
            if (sl < 6 + 3 * f.nc || f.H <= 0 || f.W <= 0) return fail(VQ_E_INVALID, "JPEG: bad frame header");
            for (int i = 0; i < f.nc; ++i) {
                Comp& c = f.comp[i];
                c.id = s[6 + 3 * i];
                c.h = s[7 + 3 * i] >> 4;
                c.v = s[7 + 3 * i] & 15;
                c.tq = s[8 + 3 * i];
                if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return fail(VQ_E_UNSUPPORTED, "JPEG: sampling factors %dx%d", c.h, c.v);
                f.hmax = std::max(f.hmax, c.h);
                f.vmax = std::max(f.vmax, c.v);
            }
            have_sof = true;
        } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xC7) || (m >= 0xC9 && m <= 0xCB) || (m >= 0xCD && m <= 0xCF)) {
            return fail(VQ_E_UNSUPPORTED, "JPEG process with marker FF%02X (progressive / lossless / arithmetic): baseline Huffman files only", m);
        } else if (m == 0xC4) {
            for (int q = 0; q < sl;) {
                if (q + 17 > sl) return fail(VQ_E_INVALID, "JPEG: bad Huffman table");
                const int tc = s[q] >> 4, th = s[q] & 15;
                int cnt = 0;
                for (int k = 0; k < 16; ++k) cnt += s[q + 1 + k];
                if (tc > 1 || th > 3 || cnt > 256 || q + 17 + cnt > sl) return fail(VQ_E_INVALID, "JPEG: bad Huffman table");
                if (!build_huff(s + q + 1, s + q + 17, cnt, tc ? f.ac[th] : f.dc[th])) return fail(VQ_E_INVALID, "JPEG: inconsistent Huffman table");
                q += 17 + cnt;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return fail(VQ_E_INVALID, "JPEG: bad restart interval");
            f.ri = be16(s);
        } else if (m == 0xDA) {
            if (!have_sof) return fail(VQ_E_INVALID, "JPEG: scan before the frame header");
            if (sl < 1 || s[0] != f.nc || sl < 1 + 2 * f.nc + 3) return fail(VQ_E_UNSUPPORTED, "JPEG: multi-scan files are not decoded");
            for (int i = 0; i < f.nc; ++i) {
                Comp* c = nullptr;
                for (int k = 0; k < f.nc; ++k)
                    if (f.comp[k].id == s[1 + 2 * i]) c = &f.comp[k];
                if (!c || c != &f.comp[i]) return fail(VQ_E_UNSUPPORTED, "JPEG: scan components out of frame order");
                c->td = s[2 + 2 * i] >> 4;
                c->ta = s[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || !f.dc[c->td].present || !f.ac[c->ta].present || !f.qt_present[c->tq])
                    return fail(VQ_E_INVALID, "JPEG: scan refers to a table the file does not define");
            }
            f.scan = p;
            if (f.nc == 3) {
                for (int i = 1; i < 3; ++i)
                    if (f.hmax % f.comp[i].h || f.vmax % f.comp[i].v) return fail(VQ_E_UNSUPPORTED, "JPEG: fractional sampling ratios");
                if (f.comp[0].h != f.hmax || f.comp[0].v != f.vmax || f.comp[1].h != f.comp[2].h || f.comp[1].v != f.comp[2].v ||
                    (f.vmax / f.comp[1].v == 2 && f.hmax / f.comp[1].h == 1))
                    return fail(VQ_E_UNSUPPORTED, "JPEG: chroma layout other than 4:4:4, 4:2:2 (h2v1) or 4:2:0 (h2v2)");
            }
            return VQ_OK;
        }
        // APPn, COM and the rest: skipped
    }
}

// Entropy decoding of the scan into natural-order coefficient blocks: [component][block row][block col][64] int16, the
// components back to back at comp_off[] (in blocks).
int decode_scan(const uint8_t* d, size_t n, Frame& f, int16_t* coef, const size_t* comp_off) {
    const bool single = f.nc == 1;
    const int mx = single ? cdiv(f.W, 8) : cdiv(f.W, 8 * f.hmax), my = single ? cdiv(f.H, 8) : cdiv(f.H, 8 * f.vmax);
    BitReader br{d, n, f.scan};
    int pred[3] = {0, 0, 0};
    int count = 0;
    for (int mcu = 0; mcu < mx * my; ++mcu) {
        if (f.ri && count == f.ri) {
            if (!br.restart()) return fail(VQ_E_INVALID, "JPEG: restart marker missing");
            pred[0] = pred[1] = pred[2] = 0;
            count = 0;
        }
        ++count;
        const int my_ = mcu / mx, mx_ = mcu - my_ * mx;
        for (int ci = 0; ci < f.nc; ++ci) {
            const Comp& c = f.comp[ci];
            const int hh = single ? 1 : c.h, vv = single ? 1 : c.v;
            const Huff &hd = f.dc[c.td], &ha = f.ac[c.ta];
            for (int by = 0; by < vv; ++by)
                for (int bx = 0; bx < hh; ++bx) {
                    int16_t* blk = coef + (comp_off[ci] + (size_t)(my_ * vv + by) * c.bw + (size_t)(mx_ * hh + bx)) * 64;
                    int s = decode_symbol(br, hd);
                    if (s < 0 || s > 11) return fail(VQ_E_INVALID, "JPEG: corrupt entropy-coded data (DC)");
                    if (s) pred[ci] += extend((int)br.get(s), s);
                    blk[0] = (int16_t)pred[ci];
                    for (int k = 1; k < 64;) {
                        const int rs = decode_symbol(br, ha);
                        if (rs < 0) return fail(VQ_E_INVALID, "JPEG: corrupt entropy-coded data (AC)");
                        const int r = rs >> 4;
                        s = rs & 15;
                        if (s == 0) {
                            if (r == 15) {
                                k += 16;
                                continue;
                            }
                            break;
                        }
                        k += r;
                        if (k > 63) return fail(VQ_E_INVALID, "JPEG: corrupt entropy-coded data (run past the block)");
                        blk[kZigzag[k]] = (int16_t)extend((int)br.get(s), s);
                        ++k;
                    }
                }
        }
    }
    return VQ_OK;
}

// ---- device side ------------------------------------------------------------------------------------------------------

struct PlaneDesc {             // one component of one frame
    unsigned coef_off;         // first block (in blocks) inside the batch's coefficient buffer
    unsigned plane_off;        // first byte inside the batch's plane buffer
    unsigned first_block;      // index of its first block in the batch-wide block numbering
    int bw, bh;                // blocks per row / column (plane is bh*8 rows of bw*8 bytes)
    int dw, dh;                // real (downsampled) width / height
    int qt;                    // index into the batch's quantisation tables
};

struct FrameDesc {
    int nc;                    // 1 or 3
    int mode;                  // chroma layout: 0 = same size as Y, 1 = h2v1, 2 = h2v2
    PlaneDesc pl[3];
};

constexpr int CONST_BITS = 13, PASS1_BITS = 2;
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jidctint.c jpeg_idct_islow, one 8-point pass: v[0..7] in, out[0..7] = descaled results
__device__ __forceinline__ void idct8(const int* v, int shift, int* out) {
    int z2 = v[2], z3 = v[6];
    int z1 = (z2 + z3) * 4433;
    const int tmp2e = z1 + z3 * (-15137), tmp3e = z1 + z2 * 6270;
    z2 = v[0];
    z3 = v[4];
    const int tmp0e = (z2 + z3) << CONST_BITS, tmp1e = (z2 - z3) << CONST_BITS;
    const int tmp10 = tmp0e + tmp3e, tmp13 = tmp0e - tmp3e, tmp11 = tmp1e + tmp2e, tmp12 = tmp1e - tmp2e;
    int tmp0 = v[7], tmp1 = v[5], tmp2 = v[3], tmp3 = v[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446;
    tmp1 *= 16819;
    tmp2 *= 25172;
    tmp3 *= 12299;
    z1 *= -7373;
    z2 *= -20995;
    z3 = z3 * (-16069) + z5;
    z4 = z4 * (-3196) + z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    out[0] = descale(tmp10 + tmp3, shift);
    out[7] = descale(tmp10 - tmp3, shift);
    out[1] = descale(tmp11 + tmp2, shift);
    out[6] = descale(tmp11 - tmp2, shift);
    out[2] = descale(tmp12 + tmp1, shift);
    out[5] = descale(tmp12 - tmp1, shift);
    out[3] = descale(tmp13 + tmp0, shift);
    out[4] = descale(tmp13 - tmp0, shift);
}

// One thread per 8x8 block of the whole batch.  block_plane[b] = which (frame, component) block b belongs to.
__global__ __launch_bounds__(128) void jpeg_idct_kernel(const int16_t* __restrict__ coef, const uint16_t* __restrict__ qts,
                                                        const FrameDesc* __restrict__ frames, const unsigned* __restrict__ block_plane,
                                                        uint8_t* __restrict__ planes, unsigned n_blocks) {
    const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const unsigned fp = block_plane[b];
    const PlaneDesc& pd = frames[fp >> 2].pl[fp & 3];
    const unsigned local = b - pd.first_block;
    const int16_t* c = coef + (size_t)(pd.coef_off + local) * 64;
    const uint16_t* q = qts + (size_t)pd.qt * 64;
    int ws[64];
#pragma unroll
    for (int col = 0; col < 8; ++col) {           // pass 1: columns
        int v[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = (int)c[r * 8 + col] * (int)q[r * 8 + col];
        idct8(v, CONST_BITS - PASS1_BITS, o);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r * 8 + col] = o[r];
    }
    const int by = local / pd.bw, bx = local - by * pd.bw;
    uint8_t* dst = planes + pd.plane_off + (size_t)(by * 8) * (pd.bw * 8) + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {                 // pass 2: rows, +128, clamp, one 8-byte store
        int o[8];
        idct8(ws + r * 8, CONST_BITS + PASS1_BITS + 3, o);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo |= (unsigned)min(max(o[k] + 128, 0), 255) << (8 * k);
            hi |= (unsigned)min(max(o[4 + k] + 128, 0), 255) << (8 * k);
        }
        *reinterpret_cast<uint2*>(dst + (size_t)r * (pd.bw * 8)) = make_uint2(lo, hi);
    }
}

__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ p, const PlaneDesc& pd, int mode, int x, int y) {
    const int stride = pd.bw * 8;
    if (mode == 0) return p[(size_t)y * stride + x];
    const int i = x >> 1;
    if (pd.dw <= 2) return p[(size_t)(mode == 2 ? y >> 1 : y) * stride + i];   // jinit_upsampler: components up to 2 samples wide are replicated
    if (mode == 1) {                               // h2v1: (3,1)/4, +1 on even, +2 on odd columns; edge columns copied
        const uint8_t* row = p + (size_t)y * stride;
        const int v = row[i];
        if (x & 1) return i == pd.dw - 1 ? v : (3 * v + row[i + 1] + 2) >> 2;
        return i == 0 ? v : (3 * v + row[i - 1] + 1) >> 2;
    }
    // h2v2: 3:1 sums with the nearer / farther row (edge rows replicated), then (3,1) across columns: +8 even, +7 odd, >> 4
    const int j = y >> 1;
    const int jo = (y & 1) ? min(j + 1, pd.dh - 1) : max(j - 1, 0);
    const uint8_t *r0 = p + (size_t)j * stride, *r1 = p + (size_t)jo * stride;
    const int s = 3 * r0[i] + r1[i];
    if (x & 1) return i == pd.dw - 1 ? (s * 4 + 7) >> 4 : (3 * s + (3 * r0[i + 1] + r1[i + 1]) + 7) >> 4;
    return i == 0 ? (s * 4 + 8) >> 4 : (3 * s + (3 * r0[i - 1] + r1[i - 1]) + 8) >> 4;
}

// One thread per output pixel of the batch: out [n][H][W][ch], ch = 3 (B, G, R) or 1 (the Y plane).
__global__ void jpeg_pixels_kernel(const FrameDesc* __restrict__ frames, const uint8_t* __restrict__ planes, uint8_t* __restrict__ out, int n,
                                   int H, int W, int ch) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), f = (int)(i / ((int64_t)H * W));
    const FrameDesc& fd = frames[f];
    const int yv = planes[fd.pl[0].plane_off + (size_t)y * (fd.pl[0].bw * 8) + x];
    if (ch == 1) {
        out[i] = (uint8_t)yv;
        return;
    }
    int r = yv, g = yv, b = yv;
    if (fd.nc == 3) {
        const int cb = chroma_at(planes + fd.pl[1].plane_off, fd.pl[1], fd.mode, x, y) - 128;
        const int cr = chroma_at(planes + fd.pl[2].plane_off, fd.pl[2], fd.mode, x, y) - 128;
        // jdcolor.c: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554, ONE_HALF = 32768
        r = yv + ((91881 * cr + 32768) >> 16);
        b = yv + ((116130 * cb + 32768) >> 16);
        g = yv + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
        r = min(max(r, 0), 255);
        g = min(max(g, 0), 255);
        b = min(max(b, 0), 255);
    }
    uint8_t* o = out + i * 3;
    o[0] = (uint8_t)b;
    o[1] = (uint8_t)g;
    o[2] = (uint8_t)r;
}

}  // namespace

struct vq_jpeg {
    std::mutex mu;
    int device = 0, max_frames = 0, max_h = 0, max_w = 0;
    size_t max_blocks = 0;             // coefficient blocks per frame, worst case (4:4:4 padded to 16 x 16 MCUs)
    int16_t* coef_host = nullptr;      // pinned
    int16_t* coef_dev = nullptr;
    uint8_t* planes_dev = nullptr;
    unsigned* block_plane_host = nullptr;   // pinned
    unsigned* block_plane_dev = nullptr;
    uint16_t* qt_dev = nullptr;
    FrameDesc* desc_dev = nullptr;
    uint8_t* out_dev = nullptr;        // [max_frames][max_h][max_w][3]
};

static void jpeg_free(vq_jpeg* j) {
    if (j->coef_host) (void)hipHostFree(j->coef_host);
    if (j->block_plane_host) (void)hipHostFree(j->block_plane_host);
    if (j->coef_dev) (void)hipFree(j->coef_dev);
    if (j->planes_dev) (void)hipFree(j->planes_dev);
    if (j->block_plane_dev) (void)hipFree(j->block_plane_dev);
    if (j->qt_dev) (void)hipFree(j->qt_dev);
    if (j->desc_dev) (void)hipFree(j->desc_dev);
    if (j->out_dev) (void)hipFree(j->out_dev);
}

extern "C" {

int vq_jpeg_info(const uint8_t* data, int64_t size, int32_t* h, int32_t* w, int32_t* components) {
    VQ_REQUIRE(data && size > 0, "NULL argument");
    Frame f;
    const int rc = parse_headers(data, (size_t)size, f);
    if (rc != VQ_OK) return rc;
    if (h) *h = f.H;
    if (w) *w = f.W;
    if (components) *components = f.nc;
    return VQ_OK;
}

int vq_jpeg_create(int32_t max_frames, int32_t max_h, int32_t max_w, int32_t device, vq_jpeg** out) {
    VQ_REQUIRE(out, "out is NULL");
    *out = nullptr;
    VQ_REQUIRE(max_frames > 0 && max_h > 0 && max_w > 0 && max_h <= 65535 && max_w <= 65535, "bad batch shape");
    int ndev = 0;
    VQ_HIP(hipGetDeviceCount(&ndev));
    VQ_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard g(device);
    auto* j = new vq_jpeg;
    j->device = device;
    j->max_frames = max_frames;
    j->max_h = max_h;
    j->max_w = max_w;
    j->max_blocks = (size_t)3 * cdiv(max_w, 16) * 2 * cdiv(max_h, 16) * 2;
    const size_t blocks = j->max_blocks * max_frames;
    VQ_REQUIRE(blocks < 0x7FFFFFFFull && blocks * 64 < 0xFFFFFFFFull, "batch too large for one JPEG workspace (blocks are numbered with 32 bits)");
    auto bail = [&](const char* what, hipError_t e) {
        jpeg_free(j);
        delete j;
        return fail(e == hipErrorOutOfMemory ? VQ_E_NOMEM : VQ_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipHostMalloc((void**)&j->coef_host, blocks * 64 * sizeof(int16_t))) != hipSuccess) return bail("hipHostMalloc(coefficients)", e);
    if ((e = hipHostMalloc((void**)&j->block_plane_host, blocks * sizeof(unsigned))) != hipSuccess) return bail("hipHostMalloc(block map)", e);
    if ((e = hipMalloc((void**)&j->coef_dev, blocks * 64 * sizeof(int16_t))) != hipSuccess) return bail("hipMalloc(coefficients)", e);
    if ((e = hipMalloc((void**)&j->planes_dev, blocks * 64)) != hipSuccess) return bail("hipMalloc(planes)", e);
    if ((e = hipMalloc((void**)&j->block_plane_dev, blocks * sizeof(unsigned))) != hipSuccess) return bail("hipMalloc(block map)", e);
    if ((e = hipMalloc((void**)&j->qt_dev, (size_t)max_frames * 4 * 64 * sizeof(uint16_t))) != hipSuccess) return bail("hipMalloc(tables)", e);
    if ((e = hipMalloc((void**)&j->desc_dev, (size_t)max_frames * sizeof(FrameDesc))) != hipSuccess) return bail("hipMalloc(descriptors)", e);
    if ((e = hipMalloc((void**)&j->out_dev, (size_t)max_frames * max_h * max_w * 3)) != hipSuccess) return bail("hipMalloc(frames)", e);
    *out = j;
    return VQ_OK;
}

int vq_jpeg_destroy(vq_jpeg* j) {
    if (!j) return VQ_OK;
    {
        DeviceGuard g(j->device);
        (void)hipDeviceSynchronize();
        jpeg_free(j);
    }
    delete j;
    return VQ_OK;
}

int vq_jpeg_decode(vq_jpeg* j, const uint8_t* const* files, const int64_t* sizes, int32_t n, int32_t color, int32_t h, int32_t w,
                   uint8_t* out_host, uint8_t** out_dev, void* hip_stream) {
    VQ_REQUIRE(j && files && sizes, "NULL argument");
    VQ_REQUIRE(n > 0 && n <= j->max_frames, "n %d outside (0,%d]", n, j->max_frames);
    VQ_REQUIRE(h > 0 && w > 0 && h <= j->max_h && w <= j->max_w, "frames of %dx%d do not fit the %dx%d workspace", w, h, j->max_w, j->max_h);
    std::lock_guard<std::mutex> lk(j->mu);
    DeviceGuard g(j->device);
    hipStream_t st = (hipStream_t)hip_stream;
    // ---- headers (serial, cheap): sizes, layouts, where everything goes
    std::vector<Frame> fr((size_t)n);
    std::vector<FrameDesc> desc((size_t)n);
    std::vector<uint16_t> qts((size_t)n * 4 * 64, 0);
    std::vector<size_t> comp_off((size_t)n * 3, 0);
    size_t blocks = 0, plane_bytes = 0;
    for (int i = 0; i < n; ++i) {
        VQ_REQUIRE(files[i] && sizes[i] > 0, "file %d is empty", i);
        Frame& f = fr[i];
        const int rc = parse_headers(files[i], (size_t)sizes[i], f);
        if (rc != VQ_OK) return rc;
        VQ_REQUIRE(f.H == h && f.W == w, "file %d is %dx%d, the call decodes %dx%d frames", i, f.W, f.H, w, h);
        FrameDesc& fd = desc[i];
        memset(&fd, 0, sizeof fd);
        fd.nc = f.nc;
        fd.mode = f.nc == 3 ? (f.hmax / f.comp[1].h == 2 ? (f.vmax / f.comp[1].v == 2 ? 2 : 1) : 0) : 0;
        const bool single = f.nc == 1;
        const int mx = single ? cdiv(w, 8) : cdiv(w, 8 * f.hmax), my = single ? cdiv(h, 8) : cdiv(h, 8 * f.vmax);
        for (int c = 0; c < f.nc; ++c) {
            Comp& cp = f.comp[c];
            cp.bw = single ? mx : mx * cp.h;
            cp.bh = single ? my : my * cp.v;
            PlaneDesc& pd = fd.pl[c];
            pd.coef_off = (unsigned)blocks;
            pd.first_block = (unsigned)blocks;
            pd.plane_off = (unsigned)plane_bytes;
            pd.bw = cp.bw;
            pd.bh = cp.bh;
            pd.dw = cdiv((long long)w * cp.h, f.hmax);
            pd.dh = cdiv((long long)h * cp.v, f.vmax);
            pd.qt = i * 4 + cp.tq;
            comp_off[(size_t)i * 3 + c] = blocks;
            for (size_t b = 0; b < (size_t)cp.bw * cp.bh; ++b) j->block_plane_host[blocks + b] = ((unsigned)i << 2) | (unsigned)c;
            blocks += (size_t)cp.bw * cp.bh;
            plane_bytes += (size_t)cp.bw * cp.bh * 64;
        }
        VQ_REQUIRE(blocks <= j->max_blocks * (size_t)j->max_frames, "JPEG workspace too small");
        for (int t = 0; t < 4; ++t)
            if (f.qt_present[t]) memcpy(&qts[((size_t)i * 4 + t) * 64], f.qt[t], 64 * sizeof(uint16_t));
    }
    // ---- entropy decoding: one frame per host thread
    const int workers = std::max(1, std::min<int>({n, 16, (int)std::thread::hardware_concurrency()}));
    std::vector<int> status((size_t)n, VQ_OK);
    std::vector<std::string> message((size_t)n);
    auto work = [&](int first) {
        for (int i = first; i < n; i += workers) {
            // the frame's blocks start from zero (only non-zero coefficients are written): cleared here, by the frame's own thread
            const size_t b0 = comp_off[(size_t)i * 3], b1 = i + 1 < n ? comp_off[(size_t)(i + 1) * 3] : blocks;
            memset(j->coef_host + b0 * 64, 0, (b1 - b0) * 64 * sizeof(int16_t));
            status[i] = decode_scan(files[i], (size_t)sizes[i], fr[i], j->coef_host, &comp_off[(size_t)i * 3]);
            if (status[i] != VQ_OK) message[i] = last_error_ref();       // thread-local message of this worker
        }
    };
    std::vector<std::thread> pool;
    for (int k = 1; k < workers; ++k) pool.emplace_back(work, k);
    work(0);
    for (std::thread& th : pool) th.join();
    for (int i = 0; i < n; ++i)
        if (status[i] != VQ_OK) return fail(status[i], "file %d: %s", i, message[i].c_str());
    // ---- device: IDCT per block, then pixels
    VQ_HIP(hipMemcpyAsync(j->coef_dev, j->coef_host, blocks * 64 * sizeof(int16_t), hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(j->block_plane_dev, j->block_plane_host, blocks * sizeof(unsigned), hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(j->qt_dev, qts.data(), qts.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(j->desc_dev, desc.data(), desc.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
    jpeg_idct_kernel<<<cdiv((long long)blocks, 128), 128, 0, st>>>(j->coef_dev, j->qt_dev, j->desc_dev, j->block_plane_dev, j->planes_dev, (unsigned)blocks);
    const int ch = color ? 3 : 1;
    const int64_t px = (int64_t)n * h * w;
    jpeg_pixels_kernel<<<cdiv(px, 256), 256, 0, st>>>(j->desc_dev, j->planes_dev, j->out_dev, n, h, w, ch);
    VQ_CHECK_LAUNCH();
    if (out_host) VQ_HIP(hipMemcpyAsync(out_host, j->out_dev, (size_t)px * ch, hipMemcpyDeviceToHost, st));
    if (out_dev) *out_dev = j->out_dev;
    VQ_HIP(hipStreamSynchronize(st));      // qts / desc leave scope; the pinned buffers are reused by the next call
    return VQ_OK;
}

}  // extern "C"
