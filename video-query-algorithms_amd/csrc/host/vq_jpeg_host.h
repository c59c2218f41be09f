// The host half of the JPEG decoder (csrc/vq_jpeg.hip): marker / header / table parsing, the host entropy decoder, the unstuffing
// pass in front of the device entropy decoder, the device form of a Huffman table, and the worker-thread stages of a batch.
// Everything here parses UNTRUSTED bytes on several threads and touches no GPU: plain C++ (no HIP include), so that besides the
// product build (build.py) tests/sanitize/Makefile builds it with -fsanitize=address,undefined and -fsanitize=thread and runs the
// mutation corpus and the threaded batch stages against it in the CPU container.
//
// Replaces the cv2.imread calls of src/features_GPU_compute/calcSig_wOF.py:92,105-106 (libjpeg behind them): ITU-T T.81 baseline.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "vq_amd.h"
#include "vq_host.h"

namespace vq {
namespace jpeg {

extern const uint8_t kZigzag[64];

struct Huff {
    // lookup of the first 9 bits -> (code length << 8) | symbol, 0 = longer code; canonical tables for the rest (T.81 F.2.2.3)
    uint16_t fast[512];
    int maxcode[18], valptr[17], mincode[17];
    uint8_t vals[256];
    uint8_t counts[16];        // codes per length and their number: with vals[0, n_vals) these determine everything above
    uint16_t n_vals;
    bool present = false;
};

inline bool same_huff(const Huff& a, const Huff& b) {
    return a.present == b.present && a.n_vals == b.n_vals && !memcmp(a.counts, b.counts, sizeof a.counts) && !memcmp(a.vals, b.vals, a.n_vals);
}

bool build_huff(const uint8_t* counts, const uint8_t* symbols, int n_symbols, Huff& h);

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

// The scan's bits, LEFT-aligned in a 64-bit buffer: the next bit of the stream is bit 63, `bits` of them are valid; what lies below the
// valid ones is either zero or (after the fast refill) a prefix of the bytes that follow -- the very bits the next refill ORs in again.
struct BitReader {
    const uint8_t* d;
    size_t n, p;
    uint64_t acc = 0;
    int bits = 0;
    bool hit_marker = false;
    // At least 57 valid bits afterwards; behind a marker (or the end of the file) the stream continues with zeros.
    void fill() {
        if (!hit_marker && p + 8 <= n) {
            uint64_t wd;
            memcpy(&wd, d + p, 8);                                    // little-endian: the first stream byte is the lowest
            const uint64_t x = ~wd;                                    // a zero byte of x = a 0xFF byte of the stream
            if (!((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull)) {
                // no 0xFF among the next eight bytes (no stuffing, no marker): whole bytes in one go
                acc |= __builtin_bswap64(wd) >> bits;
                p += (size_t)((63 - bits) >> 3);
                bits |= 56;
                return;
            }
        }
        acc = bits ? acc & (~0ull << (64 - bits)) : 0;               // only the valid bits: the byte loop appends behind them
        while (bits <= 56) {
            uint64_t b = 0;
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
            acc |= b << (56 - bits);
            bits += 8;
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(acc >> (64 - k)); }      // 1 <= k <= 32
    inline void skip(int k) {
        acc <<= k;
        bits -= k;
    }
    inline uint32_t get(int k) {
        if (k == 0) return 0;
        if (bits < k) fill();
        const uint32_t v = peek(k);
        skip(k);
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

// The next symbol; the caller has made sure of >= 16 valid bits (fill()).
inline int decode_symbol_filled(BitReader& br, const Huff& h) {
    const uint32_t e = h.fast[br.peek(9)];
    if (e) {
        br.skip((int)(e >> 8));
        return (int)(e & 255);
    }
    int l = 10;
    int code = (int)br.peek(10);
    while (l <= 16 && code > h.maxcode[l]) {
        ++l;
        code = (int)br.peek(l);
    }
    if (l > 16) return -1;
    br.skip(l);
    const int idx = h.valptr[l] + code - h.mincode[l];
    return idx >= 0 && idx < 256 ? h.vals[idx] : -1;
}
inline int decode_symbol(BitReader& br, const Huff& h) {
    if (br.bits < 16) br.fill();
    return decode_symbol_filled(br, h);
}

// T.81 F.2.2.1 EXTEND, without a branch on the value's top bit (a coin toss on real data): 1 <= s <= 16, v < 2^s
inline int extend(int v, int s) { return v + ((((v >> (s - 1)) & 1) - 1) & (1 - (1 << s))); }

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// Marker segments up to the start of the (single) scan.  Returns VQ_OK or an error code with the message set.
int parse_headers(const uint8_t* d, size_t n, Frame& f);
// Entropy decoding of the scan into natural-order coefficient blocks: [component][block row][block col][64] int16, the
// components back to back at comp_off[] (in blocks).
int decode_scan(const uint8_t* d, size_t n, Frame& f, int16_t* coef, const size_t* comp_off);

// ---- what the device entropy decoder reads (plain data: shared with the kernels of vq_jpeg.hip) ----------------------
constexpr int kFastBits = 11;
struct DevHuff {               // one Huffman table as the decoder wants it
    uint16_t fast[1 << kFastBits];   // the next 11 bits -> (code length << 8) | symbol, 0 = the code is longer
    uint32_t lim[8];           // codes of 12..16 bits without a loop: lim[l - 12] = first left-aligned 16-bit pattern that is NOT a code of
                               // <= l bits (non-decreasing); the length is 12 + the number of limits the next 16 bits reach
    int32_t valptr[17];        // canonical decoding (T.81 F.2.2.3): index of the first symbol of every length ...
    int32_t mincode[17];       // ... and its code
    uint8_t vals[256];
    uint8_t pad[8];
};
static_assert(sizeof(DevHuff) == 4096 + 32 + 68 + 68 + 256 + 8 && sizeof(DevHuff) % 16 == 0, "DevHuff layout");
struct DevTableSet {           // the tables a frame's components use: DC, AC of component 0, 1, 2
    DevHuff t[6];
};
struct SegDesc {               // one stream: a frame's scan, or one restart interval of it
    uint32_t word_off, n_words;          // its unstuffed bytes inside the batch's stream buffer (32-bit words, zero padded)
    int32_t frame;                       // -1: padding of a wave
    int32_t mcu0, mcu1;                  // MCUs [mcu0, mcu1) of the frame
    int32_t set;                         // its table set (the same for all 64 streams of a wave)
};
struct EntFrame {              // what the decoder needs to place a frame's blocks
    int32_t nc, mx;                      // components, MCUs per row
    int32_t h[3], v[3], bw[3];           // blocks per MCU in x / y, blocks per plane row
    uint32_t coef_off[3];                // first block of each component in the coefficient buffer
};

int unstuff_scan(const uint8_t* d, size_t n, size_t scan, uint8_t* dst, int max_segs, uint32_t* seg_off_bytes, uint32_t* seg_len_bytes);
void fill_dev_huff(const Huff& h, DevHuff& d);

// blocks per row / column of every component plane of an h x w frame (Comp::bw, Comp::bh); returns the frame's block count
size_t place_blocks(Frame& f, int h, int w);
// the device decoder's stream buffer: per frame its MCU count, the streams it is cut into (restart intervals) and the byte offset
// of its region (region[n] = total; a region holds the scan + 16 bytes of padding per stream + slack, a multiple of 4)
void stream_regions(const Frame* fr, const int64_t* sizes, int n, int h, int w, std::vector<int>& n_mcu, std::vector<int>& want_segs,
                    std::vector<size_t>& region);

// ---- the worker-thread stages of a batch (strided over `workers` threads; thread k takes files k, k + workers, ...) ----
int batch_workers(int n);
// headers of all files; every frame must be h x w.  First failing file: its status, message "file <i>: ...".
int parse_batch(const uint8_t* const* files, const int64_t* sizes, int n, int h, int w, Frame* fr, int workers);
// host entropy decoding of all files into coef_host (every frame's blocks at comp_off[3 i ..], `blocks` in total).  The frames are
// split into `groups` index ranges; group_ready(first block, end block) is called ON THE CALLING THREAD as soon as all frames of a
// group are decoded (in group order) -- the caller queues that piece's copy to the device while later pieces are still decoded.
int decode_batch(const uint8_t* const* files, const int64_t* sizes, int n, Frame* fr, int16_t* coef_host, const size_t* comp_off, size_t blocks,
                 int workers, int groups, const std::function<void(size_t, size_t)>& group_ready);
// the unstuffing pass of all files into stream_host + region[i] (region has n + 1 entries); seg_off / seg_len get want_segs[i] entries per
// file.  With a callback the stream buffer is handed over in `groups` pieces like decode_batch's coefficients: group_ready(first byte, end byte).
int unstuff_batch(const uint8_t* const* files, const int64_t* sizes, int n, const Frame* fr, uint8_t* stream_host, const size_t* region,
                  const int* want_segs, std::vector<std::vector<uint32_t>>& seg_off, std::vector<std::vector<uint32_t>>& seg_len, int workers,
                  int groups = 1, const std::function<void(size_t, size_t)>& group_ready = {});
// whole files into memory: data[i] for i < n (data is grown to n entries, never shrunk -- keep it between calls, its memory is what the
// call costs); VQ_E_INVALID naming the first unreadable one
int read_files(const char* const* paths, int n, std::vector<std::vector<uint8_t>>& data, int workers);

}  // namespace jpeg
}  // namespace vq
