// See vq_jpeg_host.h.  Host-only translation unit: no HIP.
#include "vq_jpeg_host.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <thread>

namespace vq {
namespace jpeg {

#define fail host_fail

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

bool build_huff(const uint8_t* counts, const uint8_t* symbols, int n_symbols, Huff& h) {
    int code = 0, k = 0;
    memset(static_cast<void*>(&h), 0, sizeof h);             // every byte defined: identical tables of different files compare equal
    for (int ln = 1; ln <= 16; ++ln) {
        h.valptr[ln] = k;
        h.mincode[ln] = code;
        for (int i = 0; i < counts[ln - 1]; ++i) {
            if (k >= n_symbols || k >= 256) return false;
            h.vals[k] = symbols[k];
            if (ln <= 9) {
                const int first = code << (9 - ln), span = 1 << (9 - ln);
                if (first + span > 512) return false;
                for (int q = 0; q < span; ++q) h.fast[first + q] = (uint16_t)((ln << 8) | symbols[k]);
            }
            ++code;
            ++k;
        }
        h.maxcode[ln] = counts[ln - 1] ? code - 1 : -1;
        if (code > (1 << ln)) return false;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    memcpy(h.counts, counts, sizeof h.counts);
    h.n_vals = (uint16_t)k;
    h.present = true;
    return true;
}

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
    // The bit buffer lives in LOCALS (registers): the reader object escapes into fill() / restart(), so its members are memory to the
    // compiler.  Every coefficient starts with a refill whose test is the same almost every time (eight bytes ahead without a 0xFF: OR them
    // in, variant 4 of the classic left-aligned readers) instead of one taken when the DATA say so ("fewer than 32 bits left"), which the
    // branch predictor cannot know: on noisy frames that test, the sign test of the value bits and the end-of-block test were half the time.
    uint64_t acc = 0;
    int bits = 0;
    size_t p = br.p;
    const size_t fast_end = n >= 8 ? n - 8 : 0;            // p <= fast_end: eight bytes can be loaded
    bool plain = n >= 8;                                   // no marker met yet
#define VQ_FILL()                                                                          \
    {                                                                                      \
        uint64_t wd_ = 0;                                                                  \
        bool ok_ = plain && p <= fast_end;                                                 \
        if (ok_) {                                                                         \
            memcpy(&wd_, d + p, 8);                                                        \
            const uint64_t x_ = ~wd_;                                                      \
            ok_ = !((x_ - 0x0101010101010101ull) & ~x_ & 0x8080808080808080ull);           \
        }                                                                                  \
        if (__builtin_expect(ok_, 1)) {                                                    \
            acc |= __builtin_bswap64(wd_) >> bits;                                         \
            p += (size_t)((63 - bits) >> 3);                                               \
            bits |= 56;                                                                    \
        } else if (bits < 32) {                                                            \
            br.acc = acc;                                                                  \
            br.bits = bits;                                                                \
            br.p = p;                                                                      \
            br.fill();                                                                     \
            acc = br.acc;                                                                  \
            bits = br.bits;                                                                \
            p = br.p;                                                                      \
            plain = !br.hit_marker;                                                        \
        }                                                                                  \
    }
#define VQ_PEEK(K) ((uint32_t)(acc >> (64 - (K))))
#define VQ_SKIP(K)     \
    {                  \
        acc <<= (K);   \
        bits -= (K);   \
    }
    // the next symbol of table H into SYM (-1: no such code); >= 16 valid bits on entry
#define VQ_SYMBOL(H, SYM)                                                   \
    {                                                                       \
        const uint32_t e_ = (H).fast[VQ_PEEK(9)];                           \
        if (e_) {                                                           \
            VQ_SKIP((int)(e_ >> 8))                                         \
            SYM = (int)(e_ & 255);                                          \
        } else {                                                            \
            int l_ = 10, code_ = (int)VQ_PEEK(10);                          \
            while (l_ <= 16 && code_ > (H).maxcode[l_]) {                   \
                ++l_;                                                       \
                code_ = (int)VQ_PEEK(l_);                                   \
            }                                                               \
            if (l_ > 16) {                                                  \
                SYM = -1;                                                   \
            } else {                                                        \
                VQ_SKIP(l_)                                                 \
                const int idx_ = (H).valptr[l_] + code_ - (H).mincode[l_];  \
                SYM = idx_ >= 0 && idx_ < 256 ? (H).vals[idx_] : -1;        \
            }                                                               \
        }                                                                   \
    }
    int pred[3] = {0, 0, 0};
    int count = 0;
    for (int mcu = 0; mcu < mx * my; ++mcu) {
        if (f.ri && count == f.ri) {
            br.p = p;
            if (!br.restart()) return fail(VQ_E_INVALID, "JPEG: restart marker missing");
            acc = 0;
            bits = 0;
            p = br.p;
            plain = n >= 8;
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
                    // one refill check per coefficient: a code (<= 16 bits) and its value bits (<= 15) fit what fill() leaves (>= 57)
                    VQ_FILL()
                    int s;
                    VQ_SYMBOL(hd, s)
                    if (s < 0 || s > 11) return fail(VQ_E_INVALID, "JPEG: corrupt entropy-coded data (DC)");
                    if (s) {
                        pred[ci] += extend((int)VQ_PEEK(s), s);
                        VQ_SKIP(s)
                    }
                    blk[0] = (int16_t)pred[ci];
                    for (int k = 1; k < 64;) {
                        VQ_FILL()
                        int rs;
                        VQ_SYMBOL(ha, rs)
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
                        blk[kZigzag[k]] = (int16_t)extend((int)VQ_PEEK(s), s);
                        VQ_SKIP(s)
                        ++k;
                    }
                }
        }
    }
#undef VQ_FILL
#undef VQ_PEEK
#undef VQ_SKIP
#undef VQ_SYMBOL
    return VQ_OK;
}

// One pass over a file's scan: byte stuffing removed (FF 00 -> FF), cut at the RSTn markers, every piece zero-padded to whole
// 32-bit words + two words (the decoder reads zeros behind a stream, like the host reader behind a marker).  dst has room for
// n - scan + 16 * (max_segs + 1) bytes.  Returns the number of pieces found (<= max_segs are recorded).
int unstuff_scan(const uint8_t* d, size_t n, size_t scan, uint8_t* dst, int max_segs, uint32_t* seg_off_bytes, uint32_t* seg_len_bytes) {
    size_t p = scan, o = 0;
    int segs = 0;
    size_t start = 0;
    auto close = [&]() {
        if (segs < max_segs) {
            seg_off_bytes[segs] = (uint32_t)start;
            seg_len_bytes[segs] = (uint32_t)(o - start);
        }
        ++segs;
        const size_t padded = ((o + 3) & ~(size_t)3) + 8;
        memset(dst + o, 0, padded - o);
        o = padded;
        start = o;
    };
    while (p < n) {
        const uint8_t* q = (const uint8_t*)memchr(d + p, 0xFF, n - p);
        const size_t run = q ? (size_t)(q - (d + p)) : n - p;
        memcpy(dst + o, d + p, run);
        o += run;
        p += run;
        if (!q) break;
        const uint8_t nx = p + 1 < n ? d[p + 1] : 0xD9;
        if (nx == 0) {
            dst[o++] = 0xFF;
            p += 2;
        } else if (nx >= 0xD0 && nx <= 0xD7) {
            if (segs + 1 >= max_segs) break;  // more restart markers than the frame has intervals: the rest is not decoded
            close();
            p += 2;
        } else if (nx == 0xFF) {              // fill byte before a marker
            ++p;
        } else {
            break;                            // EOI or any other marker: the entropy-coded data ends here
        }
    }
    close();
    return segs;
}

void fill_dev_huff(const Huff& h, DevHuff& d) {
    memset(&d, 0, sizeof d);
    // every code of <= kFastBits bits fills its span of the look-up (codes are left-aligned in the index)
    for (int ln = 1; ln <= kFastBits; ++ln) {
        if (h.maxcode[ln] < 0) continue;
        for (int code = h.mincode[ln]; code <= h.maxcode[ln]; ++code) {
            const int sym = h.vals[h.valptr[ln] + code - h.mincode[ln]];
            const int first = code << (kFastBits - ln), span = 1 << (kFastBits - ln);
            for (int q = 0; q < span && first + q < (1 << kFastBits); ++q) d.fast[first + q] = (uint16_t)((ln << 8) | sym);
        }
    }
    uint32_t run = 0;
    for (int ln = 1; ln <= 16; ++ln) {
        if (h.maxcode[ln] >= 0) run = std::max(run, (uint32_t)(h.maxcode[ln] + 1) << (16 - ln));
        if (ln > kFastBits) d.lim[ln - kFastBits - 1] = run;
    }
    for (int i = 0; i < 17; ++i) {
        d.valptr[i] = h.valptr[i];
        d.mincode[i] = h.mincode[i];
    }
    memcpy(d.vals, h.vals, 256);
}

#undef fail

size_t place_blocks(Frame& f, int h, int w) {
    const bool single = f.nc == 1;
    const int mx = single ? cdiv(w, 8) : cdiv(w, 8 * f.hmax), my = single ? cdiv(h, 8) : cdiv(h, 8 * f.vmax);
    size_t blocks = 0;
    for (int c = 0; c < f.nc; ++c) {
        Comp& cp = f.comp[c];
        cp.bw = single ? mx : mx * cp.h;
        cp.bh = single ? my : my * cp.v;
        blocks += (size_t)cp.bw * cp.bh;
    }
    return blocks;
}

void stream_regions(const Frame* fr, const int64_t* sizes, int n, int h, int w, std::vector<int>& n_mcu, std::vector<int>& want_segs,
                    std::vector<size_t>& region) {
    n_mcu.assign((size_t)n, 0);
    want_segs.assign((size_t)n, 0);
    region.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) {
        const Frame& f = fr[i];
        const bool single = f.nc == 1;
        n_mcu[i] = (single ? cdiv(w, 8) : cdiv(w, 8 * f.hmax)) * (single ? cdiv(h, 8) : cdiv(h, 8 * f.vmax));
        want_segs[i] = f.ri ? cdiv(n_mcu[i], f.ri) : 1;
        region[i + 1] = region[i] + (((size_t)sizes[i] - f.scan + 16 * ((size_t)want_segs[i] + 2)) + 3) / 4 * 4;
    }
}

int batch_workers(int n) { return std::max(1, std::min<int>({n, 16, (int)std::thread::hardware_concurrency()})); }

namespace {
template <typename F>
void strided(int workers, F&& body) {           // body(first): the calling thread is worker 0
    std::vector<std::thread> pool;
    for (int k = 1; k < workers; ++k) pool.emplace_back(body, k);
    body(0);
    for (std::thread& th : pool) th.join();
}

int first_failure(const std::vector<int>& status, const std::vector<std::string>& message) {
    for (size_t i = 0; i < status.size(); ++i)
        if (status[i] != VQ_OK) return host_fail(status[i], "file %d: %s", (int)i, message[i].c_str());
    return VQ_OK;
}
}  // namespace

int parse_batch(const uint8_t* const* files, const int64_t* sizes, int n, int h, int w, Frame* fr, int workers) {
    std::vector<int> status((size_t)n, VQ_OK);
    std::vector<std::string> message((size_t)n);
    strided(workers, [&](int first) {
        for (int i = first; i < n; i += workers) {
            fr[i] = Frame();
            if (!files[i] || sizes[i] <= 0) {
                status[i] = VQ_E_INVALID;
                message[i] = "file is empty";
                continue;
            }
            status[i] = parse_headers(files[i], (size_t)sizes[i], fr[i]);
            if (status[i] != VQ_OK) {
                message[i] = last_error_ref();                        // thread-local message of this worker
            } else if (fr[i].H != h || fr[i].W != w) {
                status[i] = VQ_E_INVALID;
                char buf[96];
                snprintf(buf, sizeof buf, "is %dx%d, the call decodes %dx%d frames", fr[i].W, fr[i].H, w, h);
                message[i] = buf;
            }
        }
    });
    return first_failure(status, message);
}

namespace {
// Items [0, n) on `workers` threads (item i on thread i % workers, so the items finish roughly in index order), handed to the caller in
// `groups` runs of consecutive items as each run completes: ready(i0, i1) on the calling thread, in order.
template <typename Item, typename Ready>
void in_groups(int n, int workers, int groups, Item&& item, Ready&& ready) {
    groups = std::max(1, std::min(groups, n));
    std::vector<std::atomic<int>> group_done((size_t)groups);
    for (auto& g : group_done) g.store(0);
    auto group_of = [&](int i) { return (int)((long long)i * groups / n); };
    std::vector<std::thread> pool;
    for (int k = 0; k < workers; ++k)
        pool.emplace_back([&, k] {
            for (int i = k; i < n; i += workers) {
                item(i);
                group_done[(size_t)group_of(i)].fetch_add(1, std::memory_order_release);
            }
        });
    for (int g = 0, i0 = 0; g < groups; ++g) {
        int i1 = i0;
        while (i1 < n && group_of(i1) == g) ++i1;
        if (i1 == i0) continue;
        while (group_done[(size_t)g].load(std::memory_order_acquire) < i1 - i0) std::this_thread::yield();
        ready(i0, i1);
        i0 = i1;
    }
    for (std::thread& th : pool) th.join();
}
}  // namespace

int decode_batch(const uint8_t* const* files, const int64_t* sizes, int n, Frame* fr, int16_t* coef_host, const size_t* comp_off, size_t blocks,
                 int workers, int groups, const std::function<void(size_t, size_t)>& group_ready) {
    std::vector<int> status((size_t)n, VQ_OK);
    std::vector<std::string> message((size_t)n);
    in_groups(
        n, workers, groups,
        [&](int i) {
            // the frame's blocks start from zero (only non-zero coefficients are written): cleared here, by the frame's own thread
            const size_t b0 = comp_off[(size_t)i * 3], b1 = i + 1 < n ? comp_off[(size_t)(i + 1) * 3] : blocks;
            memset(coef_host + b0 * 64, 0, (b1 - b0) * 64 * sizeof(int16_t));
            status[i] = decode_scan(files[i], (size_t)sizes[i], fr[i], coef_host, &comp_off[(size_t)i * 3]);
            if (status[i] != VQ_OK) message[i] = last_error_ref();       // thread-local message of this worker
        },
        [&](int i0, int i1) { group_ready(comp_off[(size_t)i0 * 3], i1 < n ? comp_off[(size_t)i1 * 3] : blocks); });
    return first_failure(status, message);
}

int unstuff_batch(const uint8_t* const* files, const int64_t* sizes, int n, const Frame* fr, uint8_t* stream_host, const size_t* region,
                  const int* want_segs, std::vector<std::vector<uint32_t>>& seg_off, std::vector<std::vector<uint32_t>>& seg_len, int workers,
                  int groups, const std::function<void(size_t, size_t)>& group_ready) {
    std::vector<int> status((size_t)n, VQ_OK);
    std::vector<std::string> message((size_t)n);
    seg_off.assign((size_t)n, {});
    seg_len.assign((size_t)n, {});
    in_groups(
        n, workers, groups,
        [&](int i) {
            seg_off[i].assign((size_t)want_segs[i], 0);
            seg_len[i].assign((size_t)want_segs[i], 0);
            const int got = unstuff_scan(files[i], (size_t)sizes[i], fr[i].scan, stream_host + region[i], want_segs[i], seg_off[i].data(), seg_len[i].data());
            if (got < want_segs[i]) {
                status[i] = VQ_E_INVALID;
                message[i] = "JPEG: restart marker missing";
            }
        },
        [&](int i0, int i1) {
            if (group_ready) group_ready(region[i0], region[i1]);
        });
    return first_failure(status, message);
}

int read_files(const char* const* paths, int n, std::vector<std::vector<uint8_t>>& data, int workers) {
    // The caller's vectors are REUSED (only grown): fresh heap memory for 8 000 files is 100 MB of first-touch page faults per call --
    // 28 ms on 16 threads, twice the decoding -- while a second call into the same vectors reads the page cache at memcpy speed.
    if (data.size() < (size_t)n) data.resize((size_t)n);
    std::vector<int> bad((size_t)n, 0);
    // open / fstat / read / close: four system calls per file (stdio made nine, and read every file through a buffer of its own)
    strided(workers, [&](int first) {
        for (int i = first; i < n; i += workers) {
            const int fd = paths[i] ? open(paths[i], O_RDONLY | O_CLOEXEC) : -1;
            if (fd < 0) {
                bad[i] = 1;
                continue;
            }
            struct stat st;
            if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
                data[i].resize((size_t)st.st_size);
                size_t got = 0;
                while (got < data[i].size()) {
                    const ssize_t r = read(fd, data[i].data() + got, data[i].size() - got);
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) break;
                    got += (size_t)r;
                }
                if (got != data[i].size()) bad[i] = 1;
            } else {
                bad[i] = 1;
            }
            close(fd);
        }
    });
    for (int i = 0; i < n; ++i)
        if (bad[i]) return host_fail(VQ_E_INVALID, "cannot read file %d: %s", i, paths[i] ? paths[i] : "(null)");
    return VQ_OK;
}

}  // namespace jpeg
}  // namespace vq
