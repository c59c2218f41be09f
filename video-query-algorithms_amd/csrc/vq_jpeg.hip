// Baseline JPEG decoding for the frame ingest in front of hot path A.
//
// What it replaces (paths relative to the reference checkout):
//   src/features_GPU_compute/calcSig_wOF.py:92         cv2.imread(img_NNNNN.jpg, cv2.IMREAD_COLOR)
//   src/features_GPU_compute/calcSig_wOF.py:105-106    cv2.imread(flow_{x,y}_NNNNN.jpg, cv2.IMREAD_GRAYSCALE)
// i.e. libjpeg(-turbo) at its defaults: integer "islow" IDCT, fancy (triangle) chroma upsampling, fixed-point YCbCr -> RGB.
// The arithmetic is the one restated in oracle/jpeg_oracle.py, which is pinned bit for bit against libjpeg-turbo (through
// Pillow): the same bits come out here.
//
// Split of the work.  An entropy-coded segment is a serial bit stream, but a BATCH of frames is many independent streams (and a
// file written with restart intervals is several): the host only parses the marker segments and strips the byte stuffing
// (FF 00 -> FF, one pass per file on host threads, cut at the RSTn markers), the compressed bytes go to the device as they are
// (a quarter of the size of the coefficients) and Huffman decoding (ITU-T T.81 F.2.2) runs there, one LANE per stream:
//   jpeg_entropy_idct_kernel   a DECODER wave decodes 64 streams side by side (64-bit bit buffer per lane fed from an LDS ring of the
//                         stream's words, an 11-bit look-up + loop-free canonical decoding of the files' Huffman tables in LDS --
//                         streams are grouped by table set: all files of one writer share one --, DC prediction per lane), one block
//                         per lane and round into an LDS image; a WRITER wave of the same workgroup turns the round's 64 blocks into
//                         pixels (dequantisation + IDCT in registers) and keeps the rings filled.  Throughput comes from the number
//                         of streams in flight (a symbol costs a lane ~600 cycles whatever the other lanes do): from LARGE batches.
// The host decoder of rounds 1-2 (one frame per host thread into 16-bit coefficient blocks, then jpeg_idct_kernel) serves the small
// batches (fewer than VQ_JPEG_DEVICE_MIN_STREAMS = 2 048 streams; VQ_JPEG_HOST_HUFFMAN=1 / 0 forces one or the other): same pixels,
// same errors (tested).  Everything that is data-parallel follows on the GPU for the whole batch of frames at once:
//   jpeg_idct_kernel      one thread per 8x8 block: dequantise, two-pass 13-bit fixed-point IDCT in registers, +128, clamp,
//                         eight 8-byte row stores into the component plane
//   jpeg_pixels_kernel    one thread per output pixel: h2v2 / h2v1 triangle-filter upsampling of the chroma planes with
//                         libjpeg's alternating rounding, the 16-bit fixed-point colour transform, BGR (cv2 order) or the
//                         Y plane for a grey read
// Frames of a call share one size (video frames); sampling factors (4:4:4, 4:2:2, 4:2:0, one component) may differ.
// Progressive, arithmetic-coded, 12-bit, multi-scan and CMYK files are refused (VQ_E_UNSUPPORTED) -- nothing in the
// reference's pipeline writes them.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>

#include "vq_common.h"
#include "host/vq_jpeg_host.h"      // the host half: parsing, host entropy decoder, unstuffing, the worker-thread stages

using namespace vq;
using namespace vq::jpeg;

namespace {

// ---- device side ------------------------------------------------------------------------------------------------------

// ---- entropy decoding on the device -----------------------------------------------------------------------------------

struct DevBits {
    const uint32_t* w;
    uint32_t p, end;
    uint64_t acc;
    int bits;
    __device__ __forceinline__ void refill() {            // >= 33 bits afterwards; behind the stream: zeros, as the host reader
        if (bits <= 32) {
            const uint32_t x = p < end ? __builtin_bswap32(w[p]) : 0u;
            ++p;
            acc = (acc << 32) | x;
            bits += 32;
        }
    }
    __device__ __forceinline__ uint32_t peek(int k) const { return (uint32_t)(acc >> (bits - k)) & ((1u << k) - 1u); }
    __device__ __forceinline__ uint32_t get(int k) {
        const uint32_t v = peek(k);
        bits -= k;
        return v;
    }
};

// 64 lanes decode side by side: whatever ONE lane needs, the wave executes.  So the common case is one look-up (11 bits cover all
// but a fraction of a percent of the symbols of a typical table) and the long codes take a loop-free path.  Needs >= 16 bits.
__device__ __forceinline__ int dev_symbol(DevBits& br, const DevHuff& h) {
    const uint32_t look = br.peek(16);
    const uint32_t e = h.fast[look >> (16 - kFastBits)];
    if (e) {
        br.bits -= (int)(e >> 8);
        return (int)(e & 255u);
    }
    int l = kFastBits + 1;
#pragma unroll
    for (int i = 0; i < 16 - kFastBits; ++i) l += look >= h.lim[i] ? 1 : 0;
    if (l > 16) return -1;
    br.bits -= l;
    const int idx = h.valptr[l] + (int)(look >> (16 - l)) - h.mincode[l];
    return idx >= 0 && idx < 256 ? h.vals[idx] : -1;
}

__device__ __forceinline__ int dev_extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }


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

// Entropy decoding + IDCT of the batch: a workgroup = a DECODER wave (one stream per lane) and a WRITER wave.  Per round every
// decoder lane decodes ONE block of its stream into its slot of an LDS image (natural order, 16-bit); behind the round's barrier the
// writer wave takes the 64 blocks over -- dequantisation, the two IDCT passes in registers, eight 8-byte row stores into the
// component plane, the slot cleared for its next use -- while the decoder wave is already in the next block (two images).  The
// split is what makes the decoder fast: a wave's vector-memory operations complete in order, so a lane that stored coefficients
// itself waited for its own scattered stores at every refill of its bit buffer (measured: 1 900 cycles per symbol); now the
// decoder wave only LOADS (the next word of every stream is requested one refill ahead) and the stores are another wave's.
// status[s]: 0 ok, 1 corrupt DC, 2 corrupt AC, 3 run past the block.
struct BlockOut {              // where a decoded block goes (written by the decoder lane, read by the writer lane)
    uint32_t plane_off;        // byte offset of its first pixel in the batch's plane buffer; 0xFFFFFFFF = no block this round
    uint32_t stride;           // bytes per plane row
    uint32_t qt;               // its quantisation table
};
// kUnzig[n] = position in zigzag order of the coefficient with natural index n (the inverse of kZigzag)
constexpr int kUnzig[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
                            10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
constexpr int kBlkPitch = 66;  // 16-bit slots per lane in the LDS image: 33 dwords, so that lanes land in different banks
constexpr int kRing = 128;     // words of every stream kept in LDS ahead of its decoder lane (a block needs 8 on average, 54 at most)
constexpr int kRingPitch = kRing + 1;

__global__ __launch_bounds__(128) void jpeg_entropy_idct_kernel(const SegDesc* __restrict__ segs, const EntFrame* __restrict__ frames,
                                                               const FrameDesc* __restrict__ fdesc, const DevTableSet* __restrict__ sets,
                                                               const uint32_t* __restrict__ words, const uint16_t* __restrict__ qts,
                                                               uint8_t* __restrict__ planes, int* __restrict__ status, long long* stamps) {
    __shared__ DevTableSet ts;
    __shared__ int16_t img[2][64 * kBlkPitch];
    long long t_work = 0, t_wait = 0, t_a = 0, t_b = 0;
    __shared__ BlockOut outd[2][64];
    __shared__ uint32_t ring[64 * kRingPitch];             // ring[lane][word index % kRing]: the stream's next words, filled by the writer wave
    __shared__ uint32_t ring_hi[64], ring_lo[64];          // words [0, ring_hi) are in the ring (written by the writer); the decoder is at ring_lo
    __shared__ int rounds_s;
    const int lane = threadIdx.x & 63;
    const bool decoder = threadIdx.x < 64;
    const int si = blockIdx.x * 64 + lane;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(sets + segs[blockIdx.x * 64].set);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&ts);
        for (int i = threadIdx.x; i < (int)(sizeof(DevTableSet) / 4); i += 128) dst[i] = src[i];
        for (int i = threadIdx.x; i < 2 * 64 * kBlkPitch / 2; i += 128) reinterpret_cast<uint32_t*>(&img[0][0])[i] = 0u;
        if (threadIdx.x == 0) rounds_s = 0;
    }
    const SegDesc sg = segs[si];
    const uint32_t* w = words + sg.word_off;
    // The writer wave keeps every stream's ring topped up: the decoder lanes then read their bit stream from LDS only.  (A wave's
    // vector-memory operations complete in order and are counted per WAVE: with 64 lanes refilling from global memory at their own
    // pace, every symbol of every lane waited for some other lane's load -- 1 250 cycles of waiting per symbol, measured.)
    uint32_t hi = 0;                                        // writer lane: words of its stream already in the ring
    auto top_up = [&](uint32_t lo) {                        // keep [lo, lo + kRing) resident: up to 16 words per call, all loads in flight together
        const uint32_t room = lo + kRing - hi, left = sg.n_words - hi;
        const uint32_t n = min(min(room, left), 16u);
        uint32_t v[16];
#pragma unroll
        for (uint32_t q = 0; q < 16; ++q) v[q] = q < n ? w[hi + q] : 0u;
#pragma unroll
        for (uint32_t q = 0; q < 16; ++q)
            if (q < n) ring[lane * kRingPitch + ((hi + q) & (kRing - 1))] = v[q];
        hi += n;
    };
    if (!decoder && sg.frame >= 0) {
        for (int q = 0; q < kRing / 16; ++q) top_up(0);
        ring_hi[lane] = hi;
    }
    if (decoder) ring_lo[lane] = 0;
    __syncthreads();
    // per-component facts in scalar registers, picked by comparisons: an array indexed by the running component would live in scratch
    // memory, i.e. behind the vector-memory counter of the decoder wave
    int nc = 0, mxw = 1, blocks_per_mcu = 0;
    int ch0 = 1, ch1 = 1, ch2 = 1, cv0 = 1, cv1 = 1, cv2 = 1;
    uint32_t po0 = 0, po1 = 0, po2 = 0, ps0 = 0, ps1 = 0, ps2 = 0, pq0 = 0, pq1 = 0, pq2 = 0;
    if (sg.frame >= 0) {
        const EntFrame fr = frames[sg.frame];
        const FrameDesc fd = fdesc[sg.frame];
        nc = fr.nc;
        mxw = fr.mx;
        ch0 = fr.h[0], cv0 = fr.v[0], po0 = fd.pl[0].plane_off, ps0 = (uint32_t)fd.pl[0].bw * 8u, pq0 = (uint32_t)fd.pl[0].qt;
        if (nc > 1) {
            ch1 = fr.h[1], cv1 = fr.v[1], po1 = fd.pl[1].plane_off, ps1 = (uint32_t)fd.pl[1].bw * 8u, pq1 = (uint32_t)fd.pl[1].qt;
            ch2 = fr.h[2], cv2 = fr.v[2], po2 = fd.pl[2].plane_off, ps2 = (uint32_t)fd.pl[2].bw * 8u, pq2 = (uint32_t)fd.pl[2].qt;
        }
        blocks_per_mcu = ch0 * cv0 + (nc > 1 ? ch1 * cv1 + ch2 * cv2 : 0);
    }
#define VQ_SEL3(c, a0, a1, a2) ((c) == 0 ? (a0) : (c) == 1 ? (a1) : (a2))
    if (decoder) atomicMax(&rounds_s, sg.frame >= 0 ? (sg.mcu1 - sg.mcu0) * blocks_per_mcu : 0);
    __syncthreads();
    const int rounds = rounds_s;
    // decoder state
    DevBits br{w, 0u, sg.n_words, 0ull, 0};
    uint32_t avail = decoder && sg.frame >= 0 ? ring_hi[lane] : 0u;      // words of the stream in the ring, as of the last barrier
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int err = 0, mcu = sg.mcu0, ci = 0, by = 0, bx = 0;
    bool more = decoder && sg.frame >= 0 && mcu < sg.mcu1;
    // Two words of the stream sit in registers: `nextw` (the word that goes into the bit buffer next) and `cand` (the one behind it,
    // read from the ring when its predecessor moved up): a ring read is then consumed one refill after it was issued and its LDS
    // latency is off the symbol-to-symbol chain, on which only the look-up of the symbol's code remains.  The ring is never dry: the
    // writer fills it up at every round (>= 74 words ahead at every barrier) and a block consumes at most 54.
    auto ring_word = [&](uint32_t p) -> uint32_t {          // behind the stream: zeros, as the host reader behind a marker
        const uint32_t v = ring[lane * kRingPitch + (p & (kRing - 1))];
        return p < br.end ? v : 0u;
    };
    uint32_t nextw = 0u, cand = 0u;
    if (decoder && sg.frame >= 0) {
        nextw = ring_word(0);
        cand = ring_word(1);
    }
    auto refill = [&]() {                                   // >= 33 bits afterwards
        if (br.bits <= 32) {
            br.acc = (br.acc << 32) | __builtin_bswap32(nextw);
            br.bits += 32;
            ++br.p;
            if (br.p + 1 >= avail && br.p + 1 < br.end) err = 4;    // cannot happen (see above); never use what is not there
            nextw = cand;
            cand = ring_word(br.p + 1);
        }
    };
    for (int t = 0; t <= rounds; ++t) {
        if (stamps) t_a = (long long)__builtin_readcyclecounter();
        if (decoder) {
            BlockOut bo{0xFFFFFFFFu, 0u, 0u};
            if (more && t < rounds) {
                int16_t* blk = &img[t & 1][lane * kBlkPitch];
                const DevHuff &hd = ts.t[2 * ci], &ha = ts.t[2 * ci + 1];
                refill();
                int s = dev_symbol(br, hd);
                if (s < 0 || s > 11) {
                    err = 1;
                } else {
                    const int diff = s ? dev_extend((int)br.get(s), s) : 0;
                    pred0 += ci == 0 ? diff : 0;
                    pred1 += ci == 1 ? diff : 0;
                    pred2 += ci == 2 ? diff : 0;
                    blk[0] = (int16_t)VQ_SEL3(ci, pred0, pred1, pred2);
                    for (int k = 1; k < 64;) {
                        refill();
                        const int rs = dev_symbol(br, ha);
                        if (rs < 0) {
                            err = 2;
                            break;
                        }
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
                        if (k > 63) {
                            err = 3;
                            break;
                        }
                        blk[k] = (int16_t)dev_extend((int)br.get(s), s);      // in ZIGZAG order: the writer undoes it with constant indices
                        ++k;
                    }
                }
                if (!err) {
                    const int hh = VQ_SEL3(ci, ch0, ch1, ch2), vv = VQ_SEL3(ci, cv0, cv1, cv2);
                    const int my_ = mcu / mxw, mx_ = mcu - my_ * mxw;
                    const int row = my_ * vv + by, col = mx_ * hh + bx;
                    bo.stride = VQ_SEL3(ci, ps0, ps1, ps2);
                    bo.plane_off = VQ_SEL3(ci, po0, po1, po2) + (uint32_t)row * 8u * bo.stride + (uint32_t)col * 8u;
                    bo.qt = VQ_SEL3(ci, pq0, pq1, pq2);
                    if (++bx == hh) {
                        bx = 0;
                        if (++by == vv) {
                            by = 0;
                            if (++ci == nc) {
                                ci = 0;
                                ++mcu;
                            }
                        }
                    }
                    more = mcu < sg.mcu1;
                } else {
                    more = false;                         // (its half-written slot is cleared by the writer like any other)
                }
            }
            outd[t & 1][lane] = bo;
            ring_lo[lane] = min(br.p, br.end);            // words below this are consumed: the writer may overwrite their ring slots
        } else {
            if (t > 0) {
                const int b = (t - 1) & 1;
                const BlockOut bo = outd[b][lane];
                int16_t* c = &img[b][lane * kBlkPitch];
                if (bo.plane_off != 0xFFFFFFFFu) {
                    const uint16_t* q = qts + (size_t)bo.qt * 64;
                    int ws[64];
#pragma unroll
                    for (int col = 0; col < 8; ++col) {       // pass 1: columns
                        int v[8], o[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = (int)c[kUnzig[r * 8 + col]] * (int)q[r * 8 + col];
                        idct8(v, CONST_BITS - PASS1_BITS, o);
#pragma unroll
                        for (int r = 0; r < 8; ++r) ws[r * 8 + col] = o[r];
                    }
                    uint8_t* dst = planes + bo.plane_off;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {             // pass 2: rows, +128, clamp, one 8-byte store
                        int o[8];
                        idct8(ws + r * 8, CONST_BITS + PASS1_BITS + 3, o);
                        unsigned lo = 0, hi8 = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            lo |= (unsigned)min(max(o[k] + 128, 0), 255) << (8 * k);
                            hi8 |= (unsigned)min(max(o[4 + k] + 128, 0), 255) << (8 * k);
                        }
                        *reinterpret_cast<uint2*>(dst + (size_t)r * bo.stride) = make_uint2(lo, hi8);
                    }
                }
                uint32_t* z = reinterpret_cast<uint32_t*>(c);   // the slot starts its next round from zero
#pragma unroll
                for (int i = 0; i < 32; ++i) z[i] = 0u;
            }
            // the decoder's position as of the LAST barrier: everything below it may be overwritten.  (During this round it reads
            // on from there, at most up to ring_hi as of the last barrier -- words the top-up does not touch.)
            if (sg.frame >= 0 && t > 0) {
                const uint32_t lo = ring_lo[lane];
                for (int q = 0; q < kRing / 16 && hi < sg.n_words && hi < lo + kRing; ++q) top_up(lo);
                ring_hi[lane] = hi;
            }
        }
        if (stamps) t_b = (long long)__builtin_readcyclecounter();
        __syncthreads();
        if (stamps) {
            t_work += t_b - t_a;
            t_wait += (long long)__builtin_readcyclecounter() - t_b;
        }
        if (decoder && sg.frame >= 0) avail = ring_hi[lane];
    }
    if (stamps && lane == 0) {                              // [workgroup][decoder | writer][work, wait at the barrier] in cycles
        stamps[(blockIdx.x * 2 + (decoder ? 0 : 1)) * 2 + 0] = t_work;
        stamps[(blockIdx.x * 2 + (decoder ? 0 : 1)) * 2 + 1] = t_wait;
    }
    if (decoder) status[si] = err;
#undef VQ_SEL3
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

// The Y plane alone (ch = 1: the flow frames), four pixels per thread: rows of the planes start at multiples of 8 bytes and W % 4 == 0, so
// both sides move aligned words (a byte per lane made the 8 000-frame flow batch's pass 1.9 ms for 1.4 GB of traffic).
__global__ void jpeg_grey4_kernel(const FrameDesc* __restrict__ frames, const uint8_t* __restrict__ planes, uint8_t* __restrict__ out, int n, int H,
                                  int W4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * H * W4) return;
    const int x4 = (int)(i % W4), y = (int)((i / W4) % H), f = (int)(i / ((int64_t)H * W4));
    const PlaneDesc& pd = frames[f].pl[0];
    const uint32_t v = *reinterpret_cast<const uint32_t*>(planes + pd.plane_off + (size_t)y * (pd.bw * 8) + (size_t)x4 * 4);
    reinterpret_cast<uint32_t*>(out)[i] = v;
}

// The top-left crop x crop pixels of the frames of the LAST decode call straight from its component planes (vq_jpeg_crops): what
// vq_resize_crop computes for frames that already have the resize size -- a copy -- without the whole-frame pixel pass in between
// (a flow batch of a command line: 8 000 grey frames = 0.7 GB written and read again for the 57 % of their pixels that survive the crop).
// Colour: a thread per output pixel, the arithmetic of jpeg_pixels_kernel.
__global__ void jpeg_crop_color_kernel(const FrameDesc* __restrict__ frames, const uint8_t* __restrict__ planes, uint8_t* __restrict__ out, int n,
                                       int crop) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * crop * crop) return;
    const int x = (int)(i % crop), y = (int)((i / crop) % crop), f = (int)(i / ((int64_t)crop * crop));
    const FrameDesc& fd = frames[f];
    const int yv = planes[fd.pl[0].plane_off + (size_t)y * (fd.pl[0].bw * 8) + x];
    int r = yv, g = yv, b = yv;
    if (fd.nc == 3) {
        const int cb = chroma_at(planes + fd.pl[1].plane_off, fd.pl[1], fd.mode, x, y) - 128;
        const int cr = chroma_at(planes + fd.pl[2].plane_off, fd.pl[2], fd.mode, x, y) - 128;
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

// Grey planes handed over plane-major (frame p * n_out + i = plane p of snippet i): out [n_out][crop][crop][C]; a thread owns two adjacent
// pixels and writes their 2 C bytes as whole words.
template <int C>
__global__ void jpeg_crop_planes_kernel(const FrameDesc* __restrict__ frames, const uint8_t* __restrict__ planes, uint8_t* __restrict__ out, int n_out,
                                        int crop) {
    static_assert((2 * C) % 4 == 0, "a thread's two pixels are whole 32-bit words");
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int half = crop / 2;
    if (i >= (int64_t)n_out * crop * half) return;
    const int xp = (int)(i % half), y = (int)((i / half) % crop), s = (int)(i / ((int64_t)half * crop));
    uint32_t words[2 * C / 4];
#pragma unroll
    for (int q = 0; q < 2 * C / 4; ++q) words[q] = 0u;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        const PlaneDesc& pd = frames[ch * n_out + s].pl[0];
        const uint8_t* row = planes + pd.plane_off + (size_t)y * (pd.bw * 8) + 2 * xp;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int pos = e * C + ch;
            words[pos >> 2] |= (uint32_t)row[e] << (8 * (pos & 3));
        }
    }
    uint32_t* o = reinterpret_cast<uint32_t*>(out + (((int64_t)s * crop + y) * crop + 2 * xp) * C);
#pragma unroll
    for (int q = 0; q < 2 * C / 4; ++q) o[q] = words[q];
}

}  // namespace

struct vq_jpeg {
    std::mutex mu;
    int device = 0, max_frames = 0, max_h = 0, max_w = 0;
    int last_n = 0, last_h = 0, last_w = 0, last_color = -1;      // what the component planes hold (vq_jpeg_crops)
    size_t max_blocks = 0;             // coefficient blocks per frame, worst case (4:4:4 padded to 16 x 16 MCUs)
    int16_t* coef_host = nullptr;      // pinned
    int16_t* coef_dev = nullptr;
    uint8_t* planes_dev = nullptr;
    unsigned* block_plane_host = nullptr;   // pinned
    unsigned* block_plane_dev = nullptr;
    uint16_t* qt_dev = nullptr;
    FrameDesc* desc_dev = nullptr;
    uint8_t* out_dev = nullptr;        // [frames of the call][h][w][3]
    size_t cap_coef_host = 0, cap_bmap_host = 0;
    size_t cap_coef = 0, cap_planes = 0, cap_bmap = 0, cap_qt = 0, cap_desc = 0, cap_out = 0;   // bytes; every buffer grows on demand
    // device entropy decoding: the batch's unstuffed streams and their descriptors (grown on demand)
    uint32_t* stream_host = nullptr;   // pinned
    uint32_t* stream_dev = nullptr;
    size_t stream_words = 0;
    void* ent_dev = nullptr;           // SegDesc[] | EntFrame[] | DevTableSet[] | status int[]
    size_t ent_bytes = 0;
    int* status_host = nullptr;        // pinned
    size_t status_cap = 0;
    uint8_t* meta_host = nullptr;      // pinned: the call's descriptors on their way to the device (SegDesc[] | EntFrame[] | DevTableSet[], FrameDesc[], tables)
    size_t meta_cap = 0;
    int host_huffman = -1;             // VQ_JPEG_HOST_HUFFMAN at creation: 1 = always the host decoder of rounds 1-2, 0 = always the device
                                       // decoder, unset = by batch size
    long long dev_min_streams = 2048;  // VQ_JPEG_DEVICE_MIN_STREAMS: batches with at least this many streams decode on the device
    std::mutex files_mu;               // vq_jpeg_decode_files: file_data from the read to the end of the decode
    std::vector<std::vector<uint8_t>> file_data;   // the files of a call by path (kept: see read_files)
    std::vector<Frame> frames;         // the parsed headers of a call (12 KB each: kept, so that a call does not clear 100 MB first)
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
    if (j->stream_host) (void)hipHostFree(j->stream_host);
    if (j->stream_dev) (void)hipFree(j->stream_dev);
    if (j->ent_dev) (void)hipFree(j->ent_dev);
    if (j->status_host) (void)hipHostFree(j->status_host);
    if (j->meta_host) (void)hipHostFree(j->meta_host);
}

namespace {
template <typename T>
int grow_dev(T** p, size_t* cap, size_t need) {
    if (*cap >= need) return VQ_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4;
    VQ_HIP(vq::malloc_trim((void**)p, want));
    *cap = want;
    return VQ_OK;
}
template <typename T>
int grow_host(T** p, size_t* cap, size_t need) {
    if (*cap >= need) return VQ_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4;
    VQ_HIP(hipHostMalloc((void**)p, want));
    *cap = want;
    return VQ_OK;
}
}  // namespace

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
    {
        const char* hh = getenv("VQ_JPEG_HOST_HUFFMAN");
        if (hh && (*hh == '0' || *hh == '1')) j->host_huffman = *hh - '0';
        const char* dm = getenv("VQ_JPEG_DEVICE_MIN_STREAMS");
        if (dm && atoll(dm) > 0) j->dev_min_streams = atoll(dm);
    }
    j->device = device;
    j->max_frames = max_frames;
    j->max_h = max_h;
    j->max_w = max_w;
    j->max_blocks = (size_t)3 * cdiv(max_w, 16) * 2 * cdiv(max_h, 16) * 2;
    // nothing is allocated here: the buffers of the path a call takes (host or device entropy decoding) grow to what the call needs
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
    // VQ_JPEG_HOST_STAMPS=1: wall time of the call's host phases on stderr
    const bool host_stamps = getenv("VQ_JPEG_HOST_STAMPS") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!host_stamps) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "jpeg host phase %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    // ---- headers: parsed by the worker threads (2.6 us per file: 21 ms for the 8 000 flow files of a command-line batch when one thread
    //      did it), then sizes, layouts and where everything goes (serial, cheap)
    if (j->frames.size() < (size_t)n) j->frames.resize((size_t)n);
    Frame* fr = j->frames.data();
    const int workers = batch_workers(n);
    {
        const int rc = parse_batch(files, sizes, n, h, w, fr, workers);
        if (rc != VQ_OK) return rc;
    }
    std::vector<FrameDesc> desc((size_t)n);
    // quantisation tables: the files of one writer share theirs, so the batch uploads each distinct table once (8 000 flow frames: one table
    // instead of 4 MB of copies of it)
    std::vector<uint16_t> qts;
    auto qt_id = [&](const uint16_t* t) {
        const size_t have = qts.size() / 64;
        for (size_t q = have; q-- > 0;)
            if (!memcmp(&qts[q * 64], t, 64 * sizeof(uint16_t))) return (int)q;
        qts.insert(qts.end(), t, t + 64);
        return (int)have;
    };
    std::vector<size_t> comp_off((size_t)n * 3, 0);
    size_t blocks = 0, plane_bytes = 0;
    for (int i = 0; i < n; ++i) {
        Frame& f = fr[i];
        FrameDesc& fd = desc[i];
        memset(&fd, 0, sizeof fd);
        fd.nc = f.nc;
        fd.mode = f.nc == 3 ? (f.hmax / f.comp[1].h == 2 ? (f.vmax / f.comp[1].v == 2 ? 2 : 1) : 0) : 0;
        place_blocks(f, h, w);                        // blocks per row / column of every component plane
        for (int c = 0; c < f.nc; ++c) {
            Comp& cp = f.comp[c];
            PlaneDesc& pd = fd.pl[c];
            pd.coef_off = (unsigned)blocks;
            pd.first_block = (unsigned)blocks;
            pd.plane_off = (unsigned)plane_bytes;
            pd.bw = cp.bw;
            pd.bh = cp.bh;
            pd.dw = cdiv((long long)w * cp.h, f.hmax);
            pd.dh = cdiv((long long)h * cp.v, f.vmax);
            pd.qt = qt_id(f.qt[cp.tq]);
            comp_off[(size_t)i * 3 + c] = blocks;
            blocks += (size_t)cp.bw * cp.bh;
            plane_bytes += (size_t)cp.bw * cp.bh * 64;
        }
    }
    lap("headers");
    const size_t n_desc = desc.size() * sizeof(FrameDesc), b_desc = (n_desc + 15) / 16 * 16, b_qt = qts.size() * sizeof(uint16_t);
    int n_seg_padded = 0;
    // Where the entropy decoding runs: a 66 KB stream costs a device lane ~35 ms whatever the batch (600 cycles per symbol, 64 streams
    // per wave, as many waves as there are streams / 64), a host thread ~0.5 ms: 16 host threads decode 31 k frames/s at any batch
    // size, the device 7 k at 256 frames, 22 k at 1 024, 53 k at 4 096 -- and small files (the grey flow frames, of which the command
    // line hands over 8 000 per batch of 32 clips) cost a lane proportionally less.
    long long total_streams = 0;
    for (int i = 0; i < n; ++i) {
        const Frame& f = fr[i];
        const bool single = f.nc == 1;
        const int nm = (single ? cdiv(w, 8) : cdiv(w, 8 * f.hmax)) * (single ? cdiv(h, 8) : cdiv(h, 8 * f.vmax));
        total_streams += f.ri ? cdiv(nm, f.ri) : 1;
    }
    const bool use_host = j->host_huffman == 1 || (j->host_huffman < 0 && total_streams < j->dev_min_streams);
    VQ_REQUIRE(plane_bytes < 0xFFFFFFFFull, "batch too large for one call (planes are addressed with 32 bits)");
    {
        int rc;
        if ((rc = grow_dev(&j->planes_dev, &j->cap_planes, plane_bytes)) != VQ_OK) return rc;
        if ((rc = grow_dev(&j->qt_dev, &j->cap_qt, qts.size() * sizeof(uint16_t))) != VQ_OK) return rc;
        if ((rc = grow_dev(&j->desc_dev, &j->cap_desc, desc.size() * sizeof(FrameDesc))) != VQ_OK) return rc;
        if ((rc = grow_dev(&j->out_dev, &j->cap_out, (size_t)n * h * w * 3)) != VQ_OK) return rc;
    }
    lap("device buffers");
    if (use_host) {
    // ---- entropy decoding on the host: one frame per host thread
    {
        int rc;
        size_t cap_h = j->cap_coef_host, cap_bh = j->cap_bmap_host;
        if ((rc = grow_host(&j->coef_host, &cap_h, blocks * 64 * sizeof(int16_t))) != VQ_OK) return rc;
        if ((rc = grow_host(&j->block_plane_host, &cap_bh, blocks * sizeof(unsigned))) != VQ_OK) return rc;
        j->cap_coef_host = cap_h;
        j->cap_bmap_host = cap_bh;
        if ((rc = grow_dev(&j->coef_dev, &j->cap_coef, blocks * 64 * sizeof(int16_t))) != VQ_OK) return rc;
        if ((rc = grow_dev(&j->block_plane_dev, &j->cap_bmap, blocks * sizeof(unsigned))) != VQ_OK) return rc;
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < fr[i].nc; ++c) {
                const size_t b0 = comp_off[(size_t)i * 3 + c], nb = (size_t)fr[i].comp[c].bw * fr[i].comp[c].bh;
                for (size_t b = 0; b < nb; ++b) j->block_plane_host[b0 + b] = ((unsigned)i << 2) | (unsigned)c;
            }
    }
    lap("host buffers, block map");
    // The coefficients travel in kCopyGroups pieces (frames [g n / G, (g + 1) n / G)): a piece is queued as soon as its frames are decoded
    // -- the workers walk the frames in index order, so the pieces finish roughly in order -- and the copies (4 ms for the 800 RGB
    // frames of a command-line batch) overlap the decoding of the later pieces instead of following it.
    hipError_t copy_err = hipSuccess;
    const int drc = decode_batch(files, sizes, n, fr, j->coef_host, comp_off.data(), blocks, workers, 4, [&](size_t b0, size_t b1) {
        if (copy_err == hipSuccess)
            copy_err = hipMemcpyAsync(j->coef_dev + b0 * 64, j->coef_host + b0 * 64, (b1 - b0) * 64 * sizeof(int16_t), hipMemcpyHostToDevice, st);
    });
    VQ_HIP(copy_err);
    if (drc != VQ_OK) return drc;
    lap("entropy decoding (threads) + copies queued");
    if (host_stamps) {
        VQ_HIP(hipStreamSynchronize(st));
        lap("coefficients on the device");
    }
    } else {
    // ---- entropy decoding on the device: the host strips the byte stuffing and cuts the scans at the restart markers
    std::vector<int> n_mcu, want_segs;
    std::vector<size_t> region;                                     // byte offset of every frame's region in the stream buffer
    stream_regions(fr, sizes, n, h, w, n_mcu, want_segs, region);
    const size_t need_words = region[n] / 4 + 4;
    if (j->stream_words < need_words) {
        if (j->stream_host) (void)hipHostFree(j->stream_host);
        if (j->stream_dev) (void)hipFree(j->stream_dev);
        j->stream_host = nullptr;
        j->stream_dev = nullptr;
        j->stream_words = 0;
        const size_t cap = need_words + need_words / 4;
        VQ_HIP(hipHostMalloc((void**)&j->stream_host, cap * 4));
        VQ_HIP(vq::malloc_trim((void**)&j->stream_dev, cap * 4));
        j->stream_words = cap;
    }
    // table sets: the (DC, AC) tables of a frame's components; files of one writer share one set -- found by comparing the parsed
    // tables themselves (build_huff defines every byte), so that the device form is built once per set, not once per file
    std::vector<DevTableSet> sets;
    std::vector<int> set_rep;                                        // a frame that uses the set
    std::vector<int> set_of((size_t)n);
    auto same_tables = [&](const Frame& x, const Frame& y) {
        if (x.nc != y.nc) return false;
        for (int c = 0; c < x.nc; ++c)
            if (!same_huff(x.dc[x.comp[c].td], y.dc[y.comp[c].td]) || !same_huff(x.ac[x.comp[c].ta], y.ac[y.comp[c].ta])) return false;
        return true;
    };
    for (int i = 0; i < n; ++i) {
        int found = -1;
        for (size_t q = 0; q < sets.size() && found < 0; ++q)
            if (same_tables(fr[i], fr[set_rep[q]])) found = (int)q;
        if (found < 0) {
            DevTableSet ts;
            memset(&ts, 0, sizeof ts);
            for (int c = 0; c < fr[i].nc; ++c) {
                fill_dev_huff(fr[i].dc[fr[i].comp[c].td], ts.t[2 * c]);
                fill_dev_huff(fr[i].ac[fr[i].comp[c].ta], ts.t[2 * c + 1]);
            }
            for (size_t q = 0; q < sets.size() && found < 0; ++q)    // different source tables, same device form
                if (!memcmp(&sets[q], &ts, sizeof ts)) found = (int)q;
            if (found < 0) {
                sets.push_back(ts);
                set_rep.push_back(i);
                found = (int)sets.size() - 1;
            }
        }
        set_of[i] = found;
    }
    lap("table sets");
    std::vector<std::vector<uint32_t>> seg_off, seg_len;
    {
        // the streams travel in pieces as the workers finish them (100 MB for 8 000 flow files: 2 ms on the link, now under the unstuffing)
        hipError_t copy_err = hipSuccess;
        uint8_t* sh = reinterpret_cast<uint8_t*>(j->stream_host);
        uint8_t* sd = reinterpret_cast<uint8_t*>(j->stream_dev);
        const int rc = unstuff_batch(files, sizes, n, fr, sh, region.data(), want_segs.data(), seg_off, seg_len, workers, 4, [&](size_t b0, size_t b1) {
            if (copy_err == hipSuccess && b1 > b0) copy_err = hipMemcpyAsync(sd + b0, sh + b0, b1 - b0, hipMemcpyHostToDevice, st);
        });
        VQ_HIP(copy_err);
        if (rc != VQ_OK) return rc;
    }
    lap("unstuffing (threads) + copies queued");
    // streams grouped by table set, every group padded to whole waves
    std::vector<SegDesc> segs;
    std::vector<int> seg_frame;
    for (size_t q = 0; q < sets.size(); ++q) {
        for (int i = 0; i < n; ++i) {
            if (set_of[i] != (int)q) continue;
            for (int k = 0; k < want_segs[i]; ++k) {
                SegDesc sd;
                sd.word_off = (uint32_t)((region[i] + seg_off[i][k]) / 4);
                sd.n_words = (seg_len[i][k] + 3) / 4;
                sd.frame = i;
                sd.mcu0 = fr[i].ri ? k * fr[i].ri : 0;
                sd.mcu1 = fr[i].ri ? std::min(n_mcu[i], (k + 1) * fr[i].ri) : n_mcu[i];
                sd.set = (int)q;
                segs.push_back(sd);
            }
        }
        while (segs.size() % 64) segs.push_back(SegDesc{0, 0, -1, 0, 0, (int)q});
    }
    n_seg_padded = (int)segs.size();
    std::vector<EntFrame> ef((size_t)n);
    for (int i = 0; i < n; ++i) {
        const Frame& f = fr[i];
        const bool single = f.nc == 1;
        memset(&ef[i], 0, sizeof(EntFrame));
        ef[i].nc = f.nc;
        ef[i].mx = single ? cdiv(w, 8) : cdiv(w, 8 * f.hmax);
        for (int c = 0; c < f.nc; ++c) {
            ef[i].h[c] = single ? 1 : f.comp[c].h;
            ef[i].v[c] = single ? 1 : f.comp[c].v;
            ef[i].bw[c] = f.comp[c].bw;
            ef[i].coef_off[c] = (uint32_t)comp_off[(size_t)i * 3 + c];
        }
    }
    const size_t b_seg = segs.size() * sizeof(SegDesc), b_fr = (ef.size() * sizeof(EntFrame) + 15) / 16 * 16, b_set = sets.size() * sizeof(DevTableSet),
                 b_stat = segs.size() * sizeof(int);
    const size_t o_fr = (b_seg + 15) / 16 * 16, o_set = o_fr + b_fr, o_stat = o_set + b_set, ent_need = o_stat + b_stat;
    if (j->ent_bytes < ent_need) {
        if (j->ent_dev) (void)hipFree(j->ent_dev);
        j->ent_dev = nullptr;
        j->ent_bytes = 0;
        VQ_HIP(vq::malloc_trim(&j->ent_dev, ent_need * 2));
        j->ent_bytes = ent_need * 2;
    }
    if (j->status_cap < segs.size()) {
        if (j->status_host) (void)hipHostFree(j->status_host);
        j->status_host = nullptr;
        j->status_cap = 0;
        VQ_HIP(hipHostMalloc((void**)&j->status_host, segs.size() * 2 * sizeof(int)));
        j->status_cap = segs.size() * 2;
    }
    uint8_t* eb = static_cast<uint8_t*>(j->ent_dev);
    // the three lists in the device's order in one pinned piece: one copy (pageable vectors went through the runtime's staging, a copy each)
    {
        const int rc = grow_host(&j->meta_host, &j->meta_cap, o_stat + b_desc + b_qt);
        if (rc != VQ_OK) return rc;
    }
    memset(j->meta_host, 0, o_stat);
    memcpy(j->meta_host, segs.data(), b_seg);
    memcpy(j->meta_host + o_fr, ef.data(), ef.size() * sizeof(EntFrame));
    memcpy(j->meta_host + o_set, sets.data(), b_set);
    memcpy(j->meta_host + o_stat, desc.data(), n_desc);
    memcpy(j->meta_host + o_stat + b_desc, qts.data(), b_qt);
    VQ_HIP(hipMemcpyAsync(eb, j->meta_host, o_stat, hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(j->desc_dev, j->meta_host + o_stat, n_desc, hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(j->qt_dev, j->meta_host + o_stat + b_desc, b_qt, hipMemcpyHostToDevice, st));
    lap("stream lists, copies queued");
    long long* stamps_dev = nullptr;                      // VQ_JPEG_STAMPS=1: where the two waves of a workgroup spend their cycles (stderr)
    if (getenv("VQ_JPEG_STAMPS")) VQ_HIP(vq::malloc_trim((void**)&stamps_dev, (size_t)n_seg_padded / 64 * 4 * sizeof(long long)));
    jpeg_entropy_idct_kernel<<<n_seg_padded / 64, 128, 0, st>>>(reinterpret_cast<const SegDesc*>(eb), reinterpret_cast<const EntFrame*>(eb + o_fr),
                                                                j->desc_dev, reinterpret_cast<const DevTableSet*>(eb + o_set), j->stream_dev, j->qt_dev,
                                                                j->planes_dev, reinterpret_cast<int*>(eb + o_stat), stamps_dev);
    VQ_CHECK_LAUNCH();
    if (stamps_dev) {
        std::vector<long long> sh((size_t)n_seg_padded / 64 * 4);
        VQ_HIP(hipMemcpyAsync(sh.data(), stamps_dev, sh.size() * 8, hipMemcpyDeviceToHost, st));
        VQ_HIP(hipStreamSynchronize(st));
        for (size_t g2 = 0; g2 < sh.size() / 4 && g2 < 4; ++g2)
            fprintf(stderr, "jpeg stamps wg %zu: decoder work %lld wait %lld | writer work %lld wait %lld cycles\n", g2, sh[4 * g2], sh[4 * g2 + 1],
                    sh[4 * g2 + 2], sh[4 * g2 + 3]);
        (void)hipFree(stamps_dev);
    }
    VQ_HIP(hipMemcpyAsync(j->status_host, eb + o_stat, b_stat, hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));            // the statuses are wanted before the pixels are handed out
    lap("copies + entropy kernel");
    static const char* const what[8] = {"", "JPEG: corrupt entropy-coded data (DC)", "JPEG: corrupt entropy-coded data (AC)",
                                        "JPEG: corrupt entropy-coded data (run past the block)", "JPEG: internal error (stream ring ran dry)", "", "", ""};
    for (size_t q = 0; q < segs.size(); ++q)
        if (segs[q].frame >= 0 && j->status_host[q] != 0) return fail(VQ_E_INVALID, "file %d: %s", segs[q].frame, what[j->status_host[q] & 7]);
    }
    // ---- device: (host-decoded coefficients: IDCT per block,) then pixels
    if (use_host) {
        VQ_HIP(hipMemcpyAsync(j->block_plane_dev, j->block_plane_host, blocks * sizeof(unsigned), hipMemcpyHostToDevice, st));
        {
            const int rc = grow_host(&j->meta_host, &j->meta_cap, b_desc + b_qt);
            if (rc != VQ_OK) return rc;
        }
        memcpy(j->meta_host, desc.data(), n_desc);
        memcpy(j->meta_host + b_desc, qts.data(), b_qt);
        VQ_HIP(hipMemcpyAsync(j->desc_dev, j->meta_host, n_desc, hipMemcpyHostToDevice, st));
        VQ_HIP(hipMemcpyAsync(j->qt_dev, j->meta_host + b_desc, b_qt, hipMemcpyHostToDevice, st));
        jpeg_idct_kernel<<<cdiv((long long)blocks, 128), 128, 0, st>>>(j->coef_dev, j->qt_dev, j->desc_dev, j->block_plane_dev, j->planes_dev, (unsigned)blocks);
    }
    j->last_n = n;
    j->last_h = h;
    j->last_w = w;
    j->last_color = color & 1;
    if (color & 2) {                       // component planes only (vq_jpeg_crops follows): no pixel pass, nothing handed out
        VQ_REQUIRE(!out_host, "color | 2 leaves the component planes on the device: no host output");
        if (out_dev) *out_dev = nullptr;
        VQ_HIP(hipStreamSynchronize(st));
        lap("planes only");
        return VQ_OK;
    }
    const int ch = (color & 1) ? 3 : 1;
    const int64_t px = (int64_t)n * h * w;
    if (ch == 1 && w % 4 == 0)
        jpeg_grey4_kernel<<<cdiv(px / 4, 256), 256, 0, st>>>(j->desc_dev, j->planes_dev, j->out_dev, n, h, w / 4);
    else
        jpeg_pixels_kernel<<<cdiv(px, 256), 256, 0, st>>>(j->desc_dev, j->planes_dev, j->out_dev, n, h, w, ch);
    VQ_CHECK_LAUNCH();
    if (out_host) VQ_HIP(hipMemcpyAsync(out_host, j->out_dev, (size_t)px * ch, hipMemcpyDeviceToHost, st));
    if (out_dev) *out_dev = j->out_dev;
    VQ_HIP(hipStreamSynchronize(st));      // the pinned buffers are reused by the next call
    lap("pixels kernel");
    return VQ_OK;
}

// The same on file PATHS: the files are read by the library's worker threads (70 000 open / read / close calls per 256 clips at the
// reference's defaults are a second of interpreter time when Python threads make them; here they overlap freely).
int vq_jpeg_decode_files(vq_jpeg* j, const char* const* paths, int32_t n, int32_t color, int32_t h, int32_t w, uint8_t* out_host, uint8_t** out_dev,
                         void* hip_stream) {
    VQ_REQUIRE(j && paths && n > 0, "bad argument");
    std::lock_guard<std::mutex> lk(j->files_mu);
    std::vector<std::vector<uint8_t>>& data = j->file_data;
    {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = read_files(paths, n, data, batch_workers(n));
        if (rc != VQ_OK) return rc;
        if (getenv("VQ_JPEG_HOST_STAMPS"))
            fprintf(stderr, "jpeg host phase %-28s %.2f ms\n", "files read (threads)",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::vector<const uint8_t*> ptrs((size_t)n);
    std::vector<int64_t> sizes((size_t)n);
    for (int i = 0; i < n; ++i) {
        ptrs[i] = data[i].data();
        sizes[i] = (int64_t)data[i].size();
    }
    return vq_jpeg_decode(j, ptrs.data(), sizes.data(), n, color, h, w, out_host, out_dev, hip_stream);
}

// ... on a path LIST: the n paths back to back, each closed by its NUL (one bytes object on the caller's side instead of an array of
// 8 000 pointers built under the interpreter's lock: 3-4 ms per flow batch of the command line).
int vq_jpeg_decode_path_list(vq_jpeg* j, const char* paths, int64_t paths_bytes, int32_t n, int32_t color, int32_t h, int32_t w, uint8_t* out_host,
                             uint8_t** out_dev, void* hip_stream) {
    VQ_REQUIRE(j && paths && n > 0 && paths_bytes > 0, "bad argument");
    VQ_REQUIRE(paths[paths_bytes - 1] == 0, "the path list must end with the NUL of its last path");
    std::vector<const char*> each;
    each.reserve((size_t)n);
    for (int64_t at = 0; at < paths_bytes && (int)each.size() < n;) {
        each.push_back(paths + at);
        at += (int64_t)strlen(paths + at) + 1;
    }
    VQ_REQUIRE((int)each.size() == n, "the path list holds %d paths, the call names %d", (int)each.size(), n);
    return vq_jpeg_decode_files(j, each.data(), n, color, h, w, out_host, out_dev, hip_stream);
}

int vq_jpeg_crops(vq_jpeg* j, int32_t c, int32_t crop, uint8_t* crops_dev, void* hip_stream) {
    VQ_REQUIRE(j && crops_dev, "NULL argument");
    std::lock_guard<std::mutex> lk(j->mu);
    VQ_REQUIRE(j->last_n > 0, "no decoded batch in the handle");
    VQ_REQUIRE(crop > 0 && crop <= j->last_h && crop <= j->last_w, "crop %d does not fit the %dx%d frames", crop, j->last_w, j->last_h);
    DeviceGuard g(j->device);
    hipStream_t st = (hipStream_t)hip_stream;
    if (c == 3) {
        VQ_REQUIRE(j->last_color == 1, "the last call decoded grey planes");
        const int64_t px = (int64_t)j->last_n * crop * crop;
        jpeg_crop_color_kernel<<<cdiv(px, 256), 256, 0, st>>>(j->desc_dev, j->planes_dev, crops_dev, j->last_n, crop);
    } else {
        VQ_REQUIRE(c == 10, "grey planes: the kernel is built for the 10 planes of a flow stack (got %d)", c);
        VQ_REQUIRE(j->last_color == 0, "the last call decoded colour frames");
        VQ_REQUIRE(j->last_n % c == 0 && crop % 2 == 0 && ((uintptr_t)crops_dev & 3u) == 0,
                   "%d frames are not whole stacks of %d planes, or the crop is odd / the output unaligned", j->last_n, c);
        const int n_out = j->last_n / c;
        const int64_t pairs = (int64_t)n_out * crop * (crop / 2);
        jpeg_crop_planes_kernel<10><<<cdiv(pairs, 256), 256, 0, st>>>(j->desc_dev, j->planes_dev, crops_dev, n_out, crop);
    }
    VQ_CHECK_LAUNCH();
    VQ_HIP(hipStreamSynchronize(st));
    return VQ_OK;
}

int vq_jpeg_info_file(const char* path, int32_t* h, int32_t* w, int32_t* components) {
    VQ_REQUIRE(path, "NULL argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(VQ_E_INVALID, "cannot read %s", path);
    std::vector<uint8_t> head(65536);
    const size_t got = fread(head.data(), 1, head.size(), f);
    fclose(f);
    if (got == 0) return fail(VQ_E_INVALID, "%s is empty", path);
    Frame fr;
    // the frame header of a file with large APPn segments may lie beyond the first 64 KB: then the whole file is read
    int rc = parse_headers(head.data(), got, fr);
    if (rc != VQ_OK && fr.H == 0) {
        f = fopen(path, "rb");
        if (!f) return fail(VQ_E_INVALID, "cannot read %s", path);
        std::vector<uint8_t> all;
        uint8_t buf[65536];
        size_t k;
        while ((k = fread(buf, 1, sizeof buf, f)) > 0) all.insert(all.end(), buf, buf + k);
        fclose(f);
        fr = Frame();
        rc = parse_headers(all.data(), all.size(), fr);
        if (rc != VQ_OK && fr.H == 0) return rc;
    }
    if (h) *h = fr.H;
    if (w) *w = fr.W;
    if (components) *components = fr.nc;
    return VQ_OK;
}

}  // extern "C"
