// Hot path B on gfx950: resident feature DB, similarity scan, scoring, selection, top-k.
//
// Reference semantics restated here (paths relative to the reference checkout):
//   src/models/ticket.py:120-163   compute_similarities  -> scan_kernel
//   src/models/ticket.py:165-180   compute_scores        -> score_from_avg (scan epilogue, rescore_kernel)
//   src/models/ticket.py:311-356   select_clips_to_review-> select_* kernels (stable partition + near argmax)
//   src/models/ticket.py:266       final report ordering -> topk (radix select + stable compaction)
//   src/models/target_clip.py:311-313 _scale_feature     -> scale_query_kernel
//   src/models/hyperparameter.py:57-58 40 rescorings     -> grid_kernel
//
// The scan is an HBM-bound stream (2 FLOP per 4 bytes): every clip's S*E vectors are read exactly
// once with 16-byte coalesced loads, multiplied against the fp64 query held in LDS, accumulated in
// fp64 per lane and reduced across the 64-lane wavefront.  This file is compiled with
// -ffp-contract=off: the score arithmetic must round exactly like the reference's numpy scalars;
// fused multiply-adds are written explicitly where they are wanted (the dot products).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "vq_common.h"

namespace vq {
std::string& last_error_ref() {
    static thread_local std::string s;
    return s;
}
}  // namespace vq

using namespace vq;

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ticket.py:172-180 -- same IEEE operations in the same order (contraction is off for this file).
__device__ __forceinline__ double score_from_avg(const double* a, const double* w, int S) {
    double ssum = 0.0, denom = 0.0;
    for (int s = 0; s < S; ++s) {
        const double term = w[s] * (1.0 - a[s]);
        ssum = ssum + term * term;
        denom = denom + w[s] * w[s];
    }
    return 1.0 - sqrt(ssum / denom);
}

// LDS image of the query: element k of vector v lives at lds_index(v, k).  Inside each 256-element
// chunk the two 16-byte halves of a lane's 4 doubles are stored in separate lane-linear planes, so
// both ds_read_b128 of a wave are 64 x 16 contiguous bytes (bank-conflict free).
__device__ __forceinline__ int t_lds_index(int D, int v, int k) {
    const int j = k >> 8, r = k & 255, lane = r >> 2, q = r & 3;
    return v * D + (j << 8) + ((q >> 1) << 7) + (lane << 1) + (q & 1);
}

// The one-query scan reads the database once, whole 1 KB per wave instruction, and it is far larger than every cache: its loads
// carry the non-temporal hint (6.15 ms per pass at cfg 4 instead of 6.8-7.0 on the same box: 6.66 TB/s).
typedef float vq_f4 __attribute__((ext_vector_type(4)));
typedef double vq_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 stream_load4(const float* p) {
    const vq_f4 r = __builtin_nontemporal_load(reinterpret_cast<const vq_f4*>(p));
    return make_float4(r.x, r.y, r.z, r.w);
}

template <typename T>
struct Quad;
template <>
struct Quad<float> {
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = stream_load4(p); }
    __device__ __forceinline__ double x0() const { return (double)v.x; }
    __device__ __forceinline__ double x1() const { return (double)v.y; }
    __device__ __forceinline__ double x2() const { return (double)v.z; }
    __device__ __forceinline__ double x3() const { return (double)v.w; }
};
template <>
struct Quad<double> {
    double2 a, b;
    __device__ __forceinline__ void load(const double* p) {
        const vq_d2 lo = __builtin_nontemporal_load(reinterpret_cast<const vq_d2*>(p));
        const vq_d2 hi = __builtin_nontemporal_load(reinterpret_cast<const vq_d2*>(p + 2));
        a = make_double2(lo.x, lo.y);
        b = make_double2(hi.x, hi.y);
    }
    __device__ __forceinline__ double x0() const { return a.x; }
    __device__ __forceinline__ double x1() const { return a.y; }
    __device__ __forceinline__ double x2() const { return b.x; }
    __device__ __forceinline__ double x3() const { return b.y; }
};

struct ScanArgs {
    const void* feats;
    const double* t;          // [NV][D] natural order, global
    const uint8_t* present;   // [N][S][E] or null
    const double* w;          // [S] or null
    double* sims;             // [N][S][E] or null
    double* avg;              // [N][S]
    int32_t* ne;              // [N][S]
    double* scores;           // [N] (written only if w)
    int64_t n;
    int32_t S, E, D;
};

// Per-clip bookkeeping, carried by every lane (wave-uniform values): ensemble mean
// (ticket.py:155-160: sequential sum over the splits present, divided by their count) and the
// weighted score (ticket.py:172-180).
struct ClipAcc {
    double acc = 0.0;
    int cnt = 0;
    double av[8];
};

__device__ __forceinline__ void clip_add(const ScanArgs& a, ClipAcc& st, int64_t c, int s, int e, double sim, int lane) {
    const int64_t idx = (c * a.S + s) * a.E + e;
    const bool p = a.present ? a.present[idx] != 0 : true;
    if (p) {
        st.acc = st.acc + sim;
        ++st.cnt;
    }
    if (a.sims && lane == 0) a.sims[idx] = sim;
}

__device__ __forceinline__ double clip_close_stream(const ScanArgs& a, ClipAcc& st, int64_t c, int s, int lane) {
    const double m = st.acc / (double)st.cnt;
    if (lane == 0) {
        a.avg[c * a.S + s] = m;
        a.ne[c * a.S + s] = st.cnt;
    }
    st.acc = 0.0;
    st.cnt = 0;
    return m;
}

// One wavefront per clip, grid-strided.  S streams x E splits per clip and CH = D/256 chunks per
// vector are compile-time, so every index is static and everything stays in registers.  The next
// vector (possibly of the wave's next clip) is in flight while the current one is multiplied, so
// every wave keeps 2 x CH x 1 KiB of HBM reads outstanding.
// LEAN (round 6): left alone, the compiler keeps the WHOLE query slice of a lane in registers across clips (the LDS reads do not depend on
// the clip: 192-320 VGPRs, one wave per SIMD) -- right for a database of a million clips, where every wave streams for milliseconds, and wrong
// for one of ten thousand (configs[0]: three clips per wave), whose scan then runs at 4.3 TB/s with 4 waves per compute unit in flight.  The
// LEAN instantiation re-reads the query from LDS for every clip and fits three waves per SIMD (what the 48 KB LDS image allows per unit).
template <typename T, int S, int E, int CH, bool LEAN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LEAN ? 3 : 1, LEAN ? 3 : 8))) void scan_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) double t_lds[];
    constexpr int NV = S * E;
    constexpr int D = CH * 256;
    for (int i = threadIdx.x; i < NV * D; i += blockDim.x) t_lds[t_lds_index(D, i / D, i % D)] = a.t[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const T* feats = static_cast<const T*>(a.feats);
    constexpr int64_t clip_elems = (int64_t)NV * D;
    double w[S];
    if (a.w) {
#pragma unroll
        for (int s = 0; s < S; ++s) w[s] = a.w[s];
    }

    Quad<T> cur[CH], nxt[CH];
    int64_t c = wave;
    if (c < a.n) {
#pragma unroll
        for (int j = 0; j < CH; ++j) cur[j].load(feats + c * clip_elems + j * 256 + lane * 4);
    }
    while (c < a.n) {
        if constexpr (LEAN) asm volatile("" ::: "memory");             // the query fragments are read again, not carried over
        const int64_t cn = c + nwaves;
        ClipAcc st;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const T* nb = (v + 1 < NV) ? feats + c * clip_elems + (int64_t)(v + 1) * D
                                       : (cn < a.n ? feats + cn * clip_elems : nullptr);
            if (nb) {
#pragma unroll
                for (int j = 0; j < CH; ++j) nxt[j].load(nb + j * 256 + lane * 4);
            }
            double p = 0.0;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const double2 ta = *reinterpret_cast<const double2*>(&t_lds[v * D + j * 256 + lane * 2]);
                const double2 tb = *reinterpret_cast<const double2*>(&t_lds[v * D + j * 256 + 128 + lane * 2]);
                p = fma(cur[j].x0(), ta.x, p);
                p = fma(cur[j].x1(), ta.y, p);
                p = fma(cur[j].x2(), tb.x, p);
                p = fma(cur[j].x3(), tb.y, p);
            }
            clip_add(a, st, c, v / E, v % E, wave_sum(p), lane);
            if (v % E == E - 1) st.av[v / E] = clip_close_stream(a, st, c, v / E, lane);
#pragma unroll
            for (int j = 0; j < CH; ++j) cur[j] = nxt[j];
        }
        if (a.w && lane == 0) a.scores[c] = score_from_avg(st.av, w, S);
        c = cn;
    }
}

// Any shape with D % 4 == 0: runtime loops, query read from global memory (L2-resident).
template <typename T>
__global__ __launch_bounds__(256) void scan_generic_kernel(ScanArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const T* feats = static_cast<const T*>(a.feats);
    const int NV = a.S * a.E, D = a.D;
    double w[8];
    if (a.w)
        for (int s = 0; s < a.S; ++s) w[s] = a.w[s];
    for (int64_t c = wave; c < a.n; c += nwaves) {
        ClipAcc st;
        for (int s = 0; s < a.S; ++s) {
            for (int e = 0; e < a.E; ++e) {
                const int v = s * a.E + e;
                const T* x = feats + (c * NV + v) * (int64_t)D;
                const double* t = a.t + (int64_t)v * D;
                double p = 0.0;
                for (int k = lane * 4; k < D; k += 256) {
                    Quad<T> q;
                    q.load(x + k);
                    p = fma(q.x0(), t[k + 0], p);
                    p = fma(q.x1(), t[k + 1], p);
                    p = fma(q.x2(), t[k + 2], p);
                    p = fma(q.x3(), t[k + 3], p);
                }
                clip_add(a, st, c, s, e, wave_sum(p), lane);
            }
            const double m = clip_close_stream(a, st, c, s, lane);
            // runtime-indexed store into a small per-lane array; the generic path is not the fast path
            st.av[s] = m;
        }
        if (a.w && lane == 0) a.scores[c] = score_from_avg(st.av, w, a.S);
    }
}

// The one-query scan over a TILED database (vq_db_set_layout): [tile of 16 clips][slice][k / 4][clip][4], fp32.  A wave owns a tile:
// lane l = (clip l % 16, k quarter l / 16), and because piece g = 4 i + l / 16 of the slice sits at g * 64 + (l % 16) * 4 floats, load i
// of a lane is at i * 256 + l * 4 -- a wave instruction still takes 1 KB of contiguous memory and a whole tile (all its slices) is ONE
// contiguous run of NV * 64 KB.  A lane multiplies 256 of its clip's 1024 elements per slice (k = 16 i + 4 (l / 16) + 0..3; the
// matching query values are four 32-byte groups per instruction in LDS: broadcast, conflict-free) and the clip's dot is the sum
// over its four lanes: two butterfly steps per 16 clips and slice where the row-major kernel pays six per clip.  G loads of 1 KB
// per wave are in flight behind the G being multiplied.  Same bookkeeping per clip as scan_kernel (ensemble mean over the splits
// present, weighted score); the k order of a dot differs, so the two layouts agree to rounding (<= 1e-12, tested), not bit for bit.
template <int S, int E, int CH, int G>
__global__ __launch_bounds__(256) void scan_tiled_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) double t_lds[];    // [NV][D], natural order
    constexpr int NV = S * E;
    constexpr int D = CH * 256;
    constexpr int LOADS = D / 16;                                     // 1 KB loads of a wave per slice
    constexpr int GROUPS = LOADS / G;
    static_assert(LOADS % G == 0, "group size must divide the slice");
    for (int i = threadIdx.x; i < NV * D; i += blockDim.x) t_lds[i] = a.t[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, col = lane & 15, kk = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t ntiles = (a.n + 15) / 16;
    constexpr int64_t tile_elems = (int64_t)NV * 16 * D;
    const float* feats = static_cast<const float*>(a.feats) + lane * 4;
    double w[S];
    if (a.w) {
#pragma unroll
        for (int s = 0; s < S; ++s) w[s] = a.w[s];
    }
    float4 cur[G], nxt[G];
    int64_t tile = wave;
    if (tile < ntiles) {
#pragma unroll
        for (int j = 0; j < G; ++j) cur[j] = stream_load4(feats + tile * tile_elems + j * 256);
    }
    while (tile < ntiles) {
        const int64_t tn = tile + nwaves;
        const float* x = feats + tile * tile_elems;
        // after the wave's last tile the ring is refilled from the tile just read: the refill stays unconditional
        const float* xn = tn < ntiles ? feats + tn * tile_elems : x;
        const int64_t c = tile * 16 + col;
        const bool mine = kk == 0 && c < a.n;
        const int64_t cc = min(c, a.n - 1);                            // the clip whose presence bits this lane follows
        double acc = 0.0, av[S];
        int cnt = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const double* tq = t_lds + v * D + 4 * kk;
            double p = 0.0;
#pragma unroll 1
            for (int g = 0; g < GROUPS; ++g) {
                const float* src = (v + 1 < NV || g + 1 < GROUPS) ? x + ((int64_t)v * GROUPS + g + 1) * (G * 256) : xn;
#pragma unroll
                for (int j = 0; j < G; ++j) nxt[j] = stream_load4(src + j * 256);
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const double2 ta = *reinterpret_cast<const double2*>(tq + 16 * (g * G + j));
                    const double2 tb = *reinterpret_cast<const double2*>(tq + 16 * (g * G + j) + 2);
                    p = fma((double)cur[j].x, ta.x, p);
                    p = fma((double)cur[j].y, ta.y, p);
                    p = fma((double)cur[j].z, tb.x, p);
                    p = fma((double)cur[j].w, tb.y, p);
                }
#pragma unroll
                for (int j = 0; j < G; ++j) cur[j] = nxt[j];
            }
            p += __shfl_xor(p, 16, 64);
            p += __shfl_xor(p, 32, 64);
            const int s = v / E, e = v % E;
            const int64_t idx = (cc * S + s) * E + e;
            const bool here = a.present ? a.present[idx] != 0 : true;
            if (here) {
                acc = acc + p;
                ++cnt;
            }
            if (a.sims && mine) a.sims[idx] = p;
            if (e == E - 1) {
                const double m = acc / (double)cnt;
                if (mine) {
                    a.avg[c * S + s] = m;
                    a.ne[c * S + s] = cnt;
                }
                av[s] = m;
                acc = 0.0;
                cnt = 0;
            }
        }
        if (a.w && mine) a.scores[c] = score_from_avg(av, w, S);
        tile = tn;
    }
}

// ------------------------------------------------------------------------------------------------
// batched scan: Q queries per pass over the database, on the fp64 matrix cores
// ------------------------------------------------------------------------------------------------
// The single-query scan keeps the whole fp64 query (S*E*D*8 = 80 KB at cfg 4) in LDS; Q of them do not fit.  The
// batched scan therefore walks the database one (stream, split) slice at a time: the slice's query vectors (16 slots x
// D x 8 bytes = 128 KB, unused slots zero) sit in LDS and a wave multiplies a tile of 16 clips against all 16 slots with
// v_mfma_f64_16x16x4_f64: D[clip][query] += A[clip][k] B[k][query], 256 MFMAs per tile and slice.  Operand layouts
// (tools/ubench/mfma_f64_layout.hip): A lane l = (clip l % 16, k l / 16), B lane l = (k l / 16, query l % 16), D lane l
// register r = (clip l / 16 + 4 r, query l % 16).  Per 64-element chunk a lane loads four 16-byte pieces such that one load
// instruction covers 64 contiguous bytes of each of the 16 clips, converts to fp64 and feeds 16 MFMAs; the matching query
// values come from LDS with one ds_read_b64 per MFMA (rows swizzled: conflict-free).
// The fp64 matrix rate (64 cycles per MFMA and SIMD = 78.6 TFLOP/s) puts 16 queries x 41 GB at 4.2 ms of matrix time,
// under the 6.6 ms the HBM stream takes: the pass is HBM-bound, i.e. 16 queries cost what one does.
// The k order of a dot differs from scan_kernel's (matrix-core accumulation), so batched and single-query scores agree to rounding
// (<= 1e-12, tested), not bit for bit.  (Round 2 ran this as ten slice launches into a [slice][query][clip] matrix plus a finalising
// launch -- 2.6 GB written and read back per pass at cfg 4; batch_fused_kernel below replaced it in round 3 and was checked against
// it bit for bit until the two-kernel form was removed in round 5.)
constexpr int kBatchSlots = 16;
#ifndef VQ_TILED_GROUP
#define VQ_TILED_GROUP 4
#endif
constexpr int kTiledGroup = VQ_TILED_GROUP;     // 1 KB loads a wave of scan_tiled_kernel keeps in flight behind the ones it multiplies

typedef double doublex4 __attribute__((ext_vector_type(4)));

// One 64-element chunk of a clip's vector as a lane sees it: piece m (0..3) holds elements 16 m + 4 kk .. + 3 of the chunk
// (kk = the lane's k quarter), so that ONE load instruction covers 64 contiguous bytes per clip across the four k-lanes.
template <typename T>
struct Chunk;
template <>
struct Chunk<float> {
    float4 v[4];
    __device__ __forceinline__ void load(const float* p) {      // p = chunk start + 4 kk
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = *reinterpret_cast<const float4*>(p + 16 * m);
    }
    __device__ __forceinline__ double at(int m, int e) const {
        return (double)(e == 0 ? v[m].x : e == 1 ? v[m].y : e == 2 ? v[m].z : v[m].w);
    }
};
template <>
struct Chunk<double> {
    double2 v[4][2];
    __device__ __forceinline__ void load(const double* p) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            v[m][0] = *reinterpret_cast<const double2*>(p + 16 * m);
            v[m][1] = *reinterpret_cast<const double2*>(p + 16 * m + 2);
        }
    }
    __device__ __forceinline__ double at(int m, int e) const { return (e & 1) ? v[m][e >> 1].y : v[m][e >> 1].x; }
};

// The same with P pieces of 16 elements per chunk (the fused kernel: a refill is then 64 P contiguous bytes of each clip).
template <typename T, int P>
struct ChunkP;
template <int P>
struct ChunkP<float, P> {
    float4 v[P];
    __device__ __forceinline__ void load(const float* p) {      // plain loads: two pieces share a 128-byte line (with the non-temporal
#pragma unroll                                                  // hint of the one-query scan the 16-query pass took 8.3 ms instead of 7.5)
        for (int m = 0; m < P; ++m) v[m] = *reinterpret_cast<const float4*>(p + 16 * m);
    }
    __device__ __forceinline__ void load_tiled(const float* p) {   // the tiled mirror: a piece is 1 KB of contiguous memory for the wave,
#pragma unroll                                                     // pieces 256 floats apart; whole lines, read once: non-temporal
        for (int m = 0; m < P; ++m) v[m] = stream_load4(p + 256 * m);
    }
    __device__ __forceinline__ double at(int m, int e) const {
        return (double)(e == 0 ? v[m].x : e == 1 ? v[m].y : e == 2 ? v[m].z : v[m].w);
    }
};
template <int P>
struct ChunkP<double, P> {
    double2 v[P][2];
    __device__ __forceinline__ void load(const double* p) {
#pragma unroll
        for (int m = 0; m < P; ++m) {
            v[m][0] = *reinterpret_cast<const double2*>(p + 16 * m);
            v[m][1] = *reinterpret_cast<const double2*>(p + 16 * m + 2);
        }
    }
    __device__ __forceinline__ void load_tiled(const double* p) { load(p); }   // never used: the mirror is fp32 only
    __device__ __forceinline__ double at(int m, int e) const { return (e & 1) ? v[m][e >> 1].y : v[m][e >> 1].x; }
};

// Where query slot q starts in LDS (in doubles): rows of D doubles plus a swizzle that makes the 32 lanes of a
// ds_read_b64 group (16 queries x 2 k quarters, 4 doubles apart) hit 32 distinct 8-byte bank slots: D is a multiple
// of 32, so slot = (q % 4) + 8 (q / 4) + 4 kk + const is a bijection onto 0..31.
__device__ __forceinline__ int query_row(int q, int D) { return q * D + (q & 3) + 8 * (q >> 2); }

// The 16-query pass's view of an fp32 database: [tile of 16 clips][slice][k / 4][clip][4] -- every load instruction of a wave then
// takes 1 KB of contiguous memory (whole 128-byte lines: the non-temporal hint pays), where the row-major layout gives it 64 bytes of
// each of 16 clips 40 KB apart.  One workgroup per (tile, slice): the 16 rows come in whole (a wave reads 1 KB of one clip per
// instruction), turn in LDS and go out in the tiled order.  Rows past n repeat the last clip (the pass never stores them).
__global__ __launch_bounds__(256) void mirror_build_kernel(const float* feats, float* mirror, int64_t n, int NV, int D) {
    extern __shared__ __attribute__((aligned(16))) float tile_lds[];          // [16][D + 4]
    const int64_t tile = blockIdx.x / NV;
    const int v = (int)(blockIdx.x - tile * NV);
    const int d4 = D / 4, row = D + 4;
    for (int i = threadIdx.x; i < 16 * d4; i += 256) {
        const int c = i / d4, g = i - c * d4;
        const int64_t clip = min(tile * 16 + c, n - 1);
        *reinterpret_cast<float4*>(tile_lds + c * row + 4 * g) = stream_load4(feats + (clip * NV + v) * (int64_t)D + 4 * g);
    }
    __syncthreads();
    float* out = mirror + (tile * NV + v) * 16 * (int64_t)D;
    for (int j = threadIdx.x; j < 16 * d4; j += 256) {
        const int g = j >> 4, c = j & 15;
        *reinterpret_cast<float4*>(out + 4 * (int64_t)j) = *reinterpret_cast<const float4*>(tile_lds + c * row + 4 * g);
    }
}

// the inverse of mirror_build_kernel: one workgroup per (tile, slice) of a tiled block -> the 16 row segments (rows past n: none)
__global__ __launch_bounds__(256) void untile_kernel(const float* tiled, float* rows, int64_t n, int NV, int D) {
    extern __shared__ __attribute__((aligned(16))) float tile_lds[];          // [16][D + 4]
    const int64_t tile = blockIdx.x / NV;
    const int v = (int)(blockIdx.x - tile * NV);
    const int d4 = D / 4, row = D + 4;
    const float* in = tiled + (tile * NV + v) * 16 * (int64_t)D;
    for (int j = threadIdx.x; j < 16 * d4; j += 256) {
        const int g = j >> 4, c = j & 15;
        *reinterpret_cast<float4*>(tile_lds + c * row + 4 * g) = *reinterpret_cast<const float4*>(in + 4 * (int64_t)j);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * d4; i += 256) {
        const int c = i / d4, g = i - c * d4;
        const int64_t clip = tile * 16 + c;
        if (clip < n) *reinterpret_cast<float4*>(rows + (clip * NV + v) * (int64_t)D + 4 * g) = *reinterpret_cast<const float4*>(tile_lds + c * row + 4 * g);
    }
}

// vq_db_upload into a tiled database: staged row-major rows -> their places in the tiles (16 bytes per thread)
__global__ void scatter_rows_tiled_kernel(const float4* staged, float4* tiled, int64_t row0, int64_t nrows, int NV, int d4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_row = (int64_t)NV * d4;
    if (i >= nrows * per_row) return;
    const int64_t r = i / per_row, rem = i - r * per_row;
    const int v = (int)(rem / d4), g = (int)(rem - (int64_t)v * d4);
    const int64_t clip = row0 + r, tile = clip >> 4;
    tiled[((tile * NV + v) * d4 + g) * 16 + (clip & 15)] = staged[i];
}

// ---- the pass as ONE launch: no dot matrix in memory ----------------------------------------------------------------
// A workgroup (16 waves, one per CU) owns TW tiles of 16 clips per wave and "round" and walks the (stream, split) slices
// itself: barrier, the slice's sixteen query rows into LDS (128 KB; from L2 after the first workgroup), barrier, then every
// wave multiplies its tiles against them.  What a tile carries from slice to slice lives in registers: the running sum
// over the splits of the stream being walked (ticket.py:155-160) and the running sum of the weighted squared misses over the
// streams already closed (ticket.py:172-180) -- 16 VGPRs per tile -- so no [slice][query][clip] matrix (2.6 GB per pass at cfg 4)
// and no per-slice launch ramps exist.  Per (clip, query): the dots of the splits present, their mean per stream in split order, the
// weighted squared misses in stream order (clip_add / clip_close_stream / score_from_avg, shared with scan_kernel's epilogue).
// The feature stream never stops at a slice change: the ring of chunk buffers is refilled from the NEXT (slice, tile) of
// the wave while the last chunks of the current one are multiplied, so the loads are in flight across the two barriers (a
// plain __syncthreads() does not wait for vector memory) and HBM stays busy while the matrix pipe waits for the new rows.
// Tiles are dealt so that a short last round spreads over ALL workgroups (tile = base + slot * gridDim + block).
struct BatchFusedArgs {
    const void* feats;
    const double* t;          // [Q][NV][D]
    const double* w;          // [Q][S]
    const uint8_t* present;   // [n][S][E] or null
    double* scores;           // [Q][n]
    int64_t n;
    int32_t Q, S, E, D;
};

template <typename T, int CH, int TW, bool PRES, int P, bool TILED = false>
__global__ __launch_bounds__(1024) void batch_fused_kernel(BatchFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) double tq[];      // 16 swizzled rows of D doubles (+ 32 of slack)
    constexpr int D = CH * 256, CE = 16 * P, NCHUNK = D / CE;       // a chunk = P pieces of 16 elements
    constexpr int NBUF = (sizeof(T) == 4 ? 16 : 8) / P;              // the ring holds 64 VGPRs of features per lane
    static_assert(NCHUNK % NBUF == 0, "chunk ring must divide the vector");
    const int lane = threadIdx.x & 63, col = lane & 15, kk = lane >> 4;
    // everything that indexes tiles is wave-uniform: kept on the scalar unit (the register budget of 4 waves per SIMD is
    // the ring of chunk buffers + the per-tile sums)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NV = a.S * a.E;
    const int64_t ntiles = (a.n + 15) / 16;
    const int64_t per_round = (int64_t)gridDim.x * 16 * TW;
    const int rounds = (int)((ntiles + per_round - 1) / per_round);
    const int64_t clip_elems = (int64_t)NV * D;
    const T* feats = static_cast<const T*>(a.feats);
    const double* tb = tq + query_row(col, D) + 4 * kk;               // this lane's query (B operand column), its k quarter
    for (int i = threadIdx.x; i < kBatchSlots * D; i += blockDim.x) {  // slots beyond Q stay zero for the whole pass
        const int q = i / D, k = i - q * D;
        if (q >= a.Q) tq[query_row(q, D) + k] = 0.0;
    }
    // tile j of this wave in round r (scalar), and how many of its TW tiles exist (a prefix: the tiles grow with j)
    auto tile_of = [&](int r, int j) -> int64_t { return (int64_t)r * per_round + (int64_t)(wv * TW + j) * gridDim.x + blockIdx.x; };
    auto valid_in = [&](int r) -> int {
        int nv = 0;
        for (int j = 0; j < TW; ++j) nv += (r < rounds && tile_of(r, j) < ntiles) ? 1 : 0;
        return nv;
    };
    // a tile's slice = a scalar base + this lane's 32-bit element offset (its clip row, its k quarter); rows past n re-read the
    // last clip and are never stored
    // TILED: a.feats is the tile-interleaved mirror [tile][slice][k / 4][16 clips][4] (mirror_build_kernel): element (clip, k) of a
    // (tile, slice) block sits at (k / 4) * 64 + clip * 4 + k % 4, so the lane that owns (clip col, k quarter kk) reads piece m of chunk c
    // at ((c P + m) * 4 + kk) * 64 + col * 4 -- the same k = 16 (c P + m) + 4 kk + e as in the row-major form: same operands, same bits.
    constexpr int CSTEP = TILED ? 256 * P : CE;
    auto base_of = [&](int64_t tile, int v) -> const T* {
        return TILED ? feats + (tile * NV + v) * 16 * D : feats + tile * 16 * clip_elems + (int64_t)v * D;
    };
    auto lane_off = [&](int64_t tile) -> uint32_t {
        if (TILED) return (uint32_t)(kk * 64 + col * 4);
        const int last = (int)min((int64_t)15, a.n - 1 - tile * 16);
        return (uint32_t)(min(col, last) * (int)clip_elems + 4 * kk);
    };
    ChunkP<T, P> buf[NBUF];
    if (valid_in(0) > 0) {
        const T* x0 = base_of(tile_of(0, 0), 0) + lane_off(tile_of(0, 0));
#pragma unroll
        for (int bq = 0; bq < NBUF; ++bq) {
            if (TILED) buf[bq].load_tiled(x0 + CSTEP * bq);
            else buf[bq].load(x0 + CSTEP * bq);
        }
    }
    for (int r = 0; r < rounds; ++r) {
        const int nval = valid_in(r), nval_next = valid_in(r + 1);
        doublex4 miss[TW];                                            // sum over closed streams of (w_s (1 - avg_s))^2
#pragma unroll
        for (int j = 0; j < TW; ++j) miss[j] = {0.0, 0.0, 0.0, 0.0};
        double denom = 0.0;
        for (int s = 0; s < a.S; ++s) {
            doublex4 ssum[TW];
            int cnt[TW][4];
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                ssum[j] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) cnt[j][rr] = 0;
            }
            for (int e = 0; e < a.E; ++e) {
                const int v = s * a.E + e;
                __syncthreads();                                      // every wave is done with the previous slice's rows
                for (int i = threadIdx.x; i < a.Q * D; i += blockDim.x) {
                    const int q = i / D, k = i - q * D;
                    tq[query_row(q, D) + k] = a.t[((size_t)q * NV + v) * D + k];
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    if (j >= nval) continue;                          // wave-uniform
                    const int64_t tile = tile_of(r, j);
                    const T* x = base_of(tile, v) + lane_off(tile);
                    // the wave's next (tile, slice): next tile of this slice, first tile of the next slice, first of the next round
                    // (after the wave's very last one the ring is refilled from the lines just read: the refill stays
                    // unconditional -- a branch around it makes the compiler drain the whole ring at every loop head)
                    int64_t tn = tile;
                    int vn = v;
                    if (j + 1 < nval) tn = tile_of(r, j + 1);
                    else if (v + 1 < NV) { tn = tile_of(r, 0); vn = v + 1; }
                    else if (nval_next > 0) { tn = tile_of(r + 1, 0); vn = 0; }
                    const T* xn = base_of(tn, vn) + lane_off(tn);
                    doublex4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
                    for (int ch0 = 0; ch0 < NCHUNK; ch0 += NBUF) {
                        const T* src = ch0 + NBUF < NCHUNK ? x + CSTEP * (ch0 + NBUF) : xn;
#pragma unroll
                        for (int bq = 0; bq < NBUF; ++bq) {
                            int off = CE * (ch0 + bq);
                            // the query rows of a slice do not change while a wave walks its tiles: the compiler would hoist all the LDS
                            // reads of a tile out of the tile loop (512 registers: spilled); an opaque offset keeps every read next to its MFMA
                            asm volatile("" : "+v"(off));
                            const double* tbc = tb + off;
#pragma unroll
                            for (int m = 0; m < P; ++m)
#pragma unroll
                                for (int ee = 0; ee < 4; ++ee)
                                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(buf[bq].at(m, ee), tbc[16 * m + ee], acc, 0, 0, 0);
                            if (TILED) buf[bq].load_tiled(src + CSTEP * bq);
                            else buf[bq].load(src + CSTEP * bq);
                        }
                    }
                    // ticket.py:155-160: the mean runs over the splits PRESENT for the clip
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        bool here = true;
                        if (PRES) {
                            const int64_t cc = min(tile * 16 + kk + 4 * rr, a.n - 1);
                            here = a.present[(cc * a.S + s) * a.E + e] != 0;
                        }
                        if (here) {
                            ssum[j][rr] = ssum[j][rr] + acc[rr];
                            if (PRES) ++cnt[j][rr];
                        }
                    }
                }
            }
            // the stream closes: avg = sum / count, then the terms of ticket.py:172-180 in stream order
            const double ws = col < a.Q ? a.w[col * a.S + s] : 0.0;
            denom = denom + ws * ws;
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const double av = ssum[j][rr] / (double)(PRES ? cnt[j][rr] : a.E);
                    const double term = ws * (1.0 - av);
                    miss[j][rr] = miss[j][rr] + term * term;
                }
        }
        if (col < a.Q) {
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                if (j >= nval) continue;
                const int64_t tile = tile_of(r, j);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int64_t cc = tile * 16 + kk + 4 * rr;
                    if (cc < a.n) a.scores[(int64_t)col * a.n + cc] = 1.0 - sqrt(miss[j][rr] / denom);
                }
            }
        }
    }
}

__global__ void rescore_kernel(const double* avg, const double* w, double* scores, int64_t n, int S) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    double av[8];
    for (int s = 0; s < S; ++s) av[s] = avg[c * S + s];
    scores[c] = score_from_avg(av, w, S);
}

// out[g][l] = score(rows[l]; w_grid[g])  (hyperparameter.py:57-58 restricted to labelled clips)
__global__ void grid_kernel(const double* avg, const double* w_grid, const int64_t* rows, double* out, int G, int L,
                            int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * L) return;
    const int g = i / L, l = i % L;
    double av[8], w[8];
    for (int s = 0; s < S; ++s) {
        av[s] = avg[rows[l] * S + s];
        w[s] = w_grid[g * S + s];
    }
    out[i] = score_from_avg(av, w, S);
}

// hyperparameter.py:57-64 in one launch: the 40 rescorings of the labelled rows AND the loss surface over the threshold grid.  Block g =
// grid weight g: its L scores (score_from_avg, as grid_kernel) go to LDS, then thread t walks the labels IN ORDER for threshold t:
//     surface[g][t] = 0.5 th[t] + sum_l (heaviside(score_l - th[t], 1) - y_l) * (score_l - th[t]) * (1 + y_l * ballast)
// -- the elementwise operations of the reference's numpy expression in its order (this file is compiled without contraction), one clip
// at a time in the dict's order: every cell sees the reference's sequence of fp64 operations (the division by L stays on the host).
__global__ void loss_surface_kernel(const double* avg, const double* w_grid, const int64_t* rows, const double* labels, const double* th_grid,
                                    double* out, int L, int T, int S, double ballast) {
    extern __shared__ double ls_scores[];                      // [L]
    const int g = blockIdx.x;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        double av[8], w[8];
        for (int s = 0; s < S; ++s) {
            av[s] = avg[rows[l] * S + s];
            w[s] = w_grid[g * S + s];
        }
        ls_scores[l] = score_from_avg(av, w, S);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const double th = th_grid[t];
        double surf = 0.5 * th;
        for (int l = 0; l < L; ++l) {
            const double y = labels[l];
            const double margin = ls_scores[l] - th;
            const double h = margin < 0.0 ? 0.0 : (margin >= 0.0 ? 1.0 : margin);     // np.heaviside(margin, 1); NaN stays NaN
            const double c = 1.0 + y * ballast;
            surf = surf + ((h - y) * margin) * c;
        }
        out[g * T + t] = surf;
    }
}

// target_clip.py:311-313: t = r / (r . r), one block per (s,e) vector of row `row`, fp64 throughout.
// element k of vector (row, v): row-major, or inside the tiled layout [tile][v][k / 4][row % 16][4]
__device__ __forceinline__ int64_t feat_index(bool tiled, int64_t row, int v, int k, int NV, int D) {
    return tiled ? (((row >> 4) * NV + v) * (int64_t)(D / 4) + (k >> 2)) * 64 + (row & 15) * 4 + (k & 3) : (row * NV + v) * (int64_t)D + k;
}

template <typename T>
__global__ __launch_bounds__(256) void scale_query_kernel(const T* feats, int64_t row, int NV, int D, double* t, bool tiled) {
    __shared__ double part[4];
    const int v = blockIdx.x;
    double p = 0.0;
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        const double x = (double)feats[feat_index(tiled, row, v, k, NV, D)];
        p = fma(x, x, p);
    }
    p = wave_sum(p);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = p;
    __syncthreads();
    const double rr = ((part[0] + part[1]) + part[2]) + part[3];
    for (int k = threadIdx.x; k < D; k += blockDim.x) t[(int64_t)v * D + k] = (double)feats[feat_index(tiled, row, v, k, NV, D)] / rr;
}

// counter-based synthetic rows (shared with oracle/sim_oracle.py::synth_features)
__device__ __forceinline__ uint32_t synth_u24(uint64_t seed_mul, uint64_t idx) {
    uint64_t z = idx + seed_mul;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 40);
}

template <typename T>
__global__ void generate_kernel(T* feats, int64_t total, uint64_t seed_mul, uint64_t idx0, int per_stream /*E*D*/,
                                int S, const float* scales) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t k = i + q;
        if (k >= total) break;
        const int s = (int)((k / per_stream) % S);
        const float u = (float)synth_u24(seed_mul, idx0 + (uint64_t)k) * 5.9604644775390625e-08f;  // 2^-24
        feats[k] = (T)(u * scales[s]);
    }
}

// ------------------------------------------------------------------------------------------------
// selection: stable partition of the score array (ticket.py:325-340)
// ------------------------------------------------------------------------------------------------
constexpr int SEL_ITEMS = 8;                       // consecutive rows per thread (keeps order)
constexpr int SEL_BLOCK = 256;
constexpr int SEL_CHUNK = SEL_ITEMS * SEL_BLOCK;   // rows per block

// predicate mode 0: bit0 = v >= th, bit1 = lower <= v < th   (select)
// predicate mode 1: bit0 = key > pivot,  bit1 = key == pivot  (top-k; keys are order-mapped scores)
__device__ __forceinline__ uint64_t order_key(double v) {
    if (v != v) return 0ull;                                  // NaN sorts last
    if (v == 0.0) v = 0.0;                                     // -0 == +0
    uint64_t u = (uint64_t)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);       // ascending in v; > 0 for every non-NaN
}

struct SelPred {
    int mode;
    double th, lower;
    uint64_t pivot;
    __device__ __forceinline__ int operator()(double v) const {
        if (mode == 0) return (v >= th ? 1 : 0) | ((lower <= v && v < th) ? 2 : 0);
        const uint64_t k = order_key(v);
        return (k > pivot ? 1 : 0) | ((k == pivot && k != 0ull) ? 2 : 0);
    }
};

__device__ __forceinline__ int2 block_scan_excl2(int2 v, int2* total) {
    // exclusive scan of two counters over the 256 threads of a block (4 waves)
    __shared__ int2 wsum[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int2 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int ax = __shfl_up(inc.x, off, 64), ay = __shfl_up(inc.y, off, 64);
        if (lane >= off) {
            inc.x += ax;
            inc.y += ay;
        }
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int2 base = make_int2(0, 0), tot = make_int2(0, 0);
    for (int i = 0; i < 4; ++i) {
        if (i < wid) {
            base.x += wsum[i].x;
            base.y += wsum[i].y;
        }
        tot.x += wsum[i].x;
        tot.y += wsum[i].y;
    }
    __syncthreads();
    *total = tot;
    return make_int2(base.x + inc.x - v.x, base.y + inc.y - v.y);
}

// pass 1: per-block counts of both classes + block-local first-argmax of class 1 (near band)
__global__ __launch_bounds__(SEL_BLOCK) void select_count_kernel(const double* scores, int64_t n, SelPred pred,
                                                                  int2* blk_cnt, double* blk_max, int64_t* blk_arg) {
    const int64_t base = (int64_t)blockIdx.x * SEL_CHUNK + (int64_t)threadIdx.x * SEL_ITEMS;
    int2 cnt = make_int2(0, 0);
    double bmax = -INFINITY;
    int64_t barg = -1;
#pragma unroll
    for (int i = 0; i < SEL_ITEMS; ++i) {
        const int64_t r = base + i;
        if (r < n) {
            const double v = scores[r];
            const int f = pred(v);
            cnt.x += f & 1;
            cnt.y += (f >> 1) & 1;
            if ((f & 2) && (barg < 0 || v > bmax)) {
                bmax = v;
                barg = r;
            }
        }
    }
    int2 tot;
    block_scan_excl2(cnt, &tot);
    // first-argmax across the block: larger value wins, ties go to the smaller row
    __shared__ double smax[SEL_BLOCK];
    __shared__ int64_t sarg[SEL_BLOCK];
    smax[threadIdx.x] = bmax;
    sarg[threadIdx.x] = barg;
    __syncthreads();
    for (int off = SEL_BLOCK / 2; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
            const double ov = smax[threadIdx.x + off];
            const int64_t oa = sarg[threadIdx.x + off];
            const int64_t ma = sarg[threadIdx.x];
            if (oa >= 0 && (ma < 0 || ov > smax[threadIdx.x] || (ov == smax[threadIdx.x] && oa < ma))) {
                smax[threadIdx.x] = ov;
                sarg[threadIdx.x] = oa;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        blk_cnt[blockIdx.x] = tot;
        blk_max[blockIdx.x] = smax[0];
        blk_arg[blockIdx.x] = sarg[0];
    }
}

// pass 2 (one block): exclusive scan of the block counts, totals and global first-argmax
// result[0] = n_class0, result[1] = n_class1, result[2] = argmax row (-1 if none)
__global__ __launch_bounds__(SEL_BLOCK) void select_scan_kernel(int2* blk_cnt, const double* blk_max,
                                                                 const int64_t* blk_arg, int nblk, int64_t* result,
                                                                 int64_t limit1 /* cap on class-1 picks, <0 = none */) {
    __shared__ int2 carry;
    __shared__ double gmax;
    __shared__ int64_t garg;
    if (threadIdx.x == 0) {
        carry = make_int2(0, 0);
        gmax = -INFINITY;
        garg = -1;
    }
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += SEL_BLOCK) {
        const int b = b0 + threadIdx.x;
        int2 v = b < nblk ? blk_cnt[b] : make_int2(0, 0);
        int2 tot;
        const int2 ex = block_scan_excl2(v, &tot);
        const int2 c0 = carry;
        if (b < nblk) blk_cnt[b] = make_int2(c0.x + ex.x, c0.y + ex.y);
        __syncthreads();
        if (threadIdx.x == 0) {
            carry = make_int2(c0.x + tot.x, c0.y + tot.y);
            for (int i = b0; i < min(nblk, b0 + SEL_BLOCK); ++i) {   // serial, in order: first max wins
                if (blk_arg[i] >= 0 && (garg < 0 || blk_max[i] > gmax)) {
                    gmax = blk_max[i];
                    garg = blk_arg[i];
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        result[0] = carry.x;
        result[1] = (limit1 >= 0 && carry.y > limit1) ? limit1 : carry.y;
        result[2] = garg;
    }
}

// pass 3: scatter rows in original order.  class-1 rows beyond `limit1` (in order) are dropped.
__global__ __launch_bounds__(SEL_BLOCK) void select_scatter_kernel(const double* scores, int64_t n, SelPred pred,
                                                                    const int2* blk_off, int64_t* out0, int64_t* out1,
                                                                    int64_t limit1, int64_t* pre0, int64_t* pre1, int64_t prefix) {
    const int64_t base = (int64_t)blockIdx.x * SEL_CHUNK + (int64_t)threadIdx.x * SEL_ITEMS;
    int flags[SEL_ITEMS];
    int2 cnt = make_int2(0, 0);
#pragma unroll
    for (int i = 0; i < SEL_ITEMS; ++i) {
        const int64_t r = base + i;
        flags[i] = r < n ? pred(scores[r]) : 0;
        cnt.x += flags[i] & 1;
        cnt.y += (flags[i] >> 1) & 1;
    }
    int2 tot;
    const int2 ex = block_scan_excl2(cnt, &tot);
    int64_t o0 = (int64_t)blk_off[blockIdx.x].x + ex.x;
    int64_t o1 = (int64_t)blk_off[blockIdx.x].y + ex.y;
#pragma unroll
    for (int i = 0; i < SEL_ITEMS; ++i) {
        if (flags[i] & 1) {
            if (o0 < prefix) pre0[o0] = base + i;            // the head of each list a second time, next to the counts (one copy back)
            out0[o0++] = base + i;
        }
        if (flags[i] & 2) {
            if (limit1 < 0 || o1 < limit1) {
                out1[o1] = base + i;
                if (o1 < prefix) pre1[o1] = base + i;
            }
            ++o1;
        }
    }
}

// The three passes above as ONE launch of one workgroup for a small database (n <= kSelSmallN): a thread owns a run of consecutive
// rows, counts / first-arg-max / offsets meet in LDS.  Same lists, same counts, same arg-max as the three-kernel form (tested).
constexpr int kSelSmallN = 16384, kSelSmallThreads = 1024;
__global__ __launch_bounds__(kSelSmallThreads) void select_small_kernel(const double* scores, int n, SelPred pred, int64_t* result, int64_t* out0,
                                                                        int64_t* out1, int64_t* pre0, int64_t* pre1, int prefix) {
    __shared__ int2 s_wave[kSelSmallThreads / 64];
    __shared__ double s_max[kSelSmallThreads / 64];
    __shared__ int s_arg[kSelSmallThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int per = (n + kSelSmallThreads - 1) / kSelSmallThreads;          // <= 16
    const int r0 = tid * per, r1 = min(n, r0 + per);
    int2 cnt = make_int2(0, 0);
    unsigned flags = 0;                                                     // 2 bits per row of the run
    double bmax = -INFINITY;
    int barg = -1;
    for (int r = r0; r < r1; ++r) {
        const double v = scores[r];
        const int f = pred(v);
        flags |= (unsigned)f << (2 * (r - r0));
        cnt.x += f & 1;
        cnt.y += (f >> 1) & 1;
        if ((f & 2) && (barg < 0 || v > bmax)) {                             // first maximum of the run
            bmax = v;
            barg = r;
        }
    }
    int2 inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int ax = __shfl_up(inc.x, off, 64), ay = __shfl_up(inc.y, off, 64);
        if (lane >= off) {
            inc.x += ax;
            inc.y += ay;
        }
    }
    // first arg-max across the wave: larger value wins, ties go to the smaller row
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_down(bmax, off, 64);
        const int oa = __shfl_down(barg, off, 64);
        if (lane + off < 64 && oa >= 0 && (barg < 0 || ov > bmax || (ov == bmax && oa < barg))) {
            bmax = ov;
            barg = oa;
        }
    }
    if (lane == 63) s_wave[wid] = inc;
    if (lane == 0) {
        s_max[wid] = bmax;
        s_arg[wid] = barg;
    }
    __syncthreads();
    int2 base = make_int2(0, 0), tot = make_int2(0, 0);
    for (int i = 0; i < kSelSmallThreads / 64; ++i) {
        if (i < wid) {
            base.x += s_wave[i].x;
            base.y += s_wave[i].y;
        }
        tot.x += s_wave[i].x;
        tot.y += s_wave[i].y;
    }
    int o0 = base.x + inc.x - cnt.x, o1 = base.y + inc.y - cnt.y;
    for (int r = r0; r < r1; ++r) {
        const unsigned f = (flags >> (2 * (r - r0))) & 3u;
        if (f & 1u) {
            if (o0 < prefix) pre0[o0] = r;
            out0[o0++] = r;
        }
        if (f & 2u) {
            if (o1 < prefix) pre1[o1] = r;
            out1[o1++] = r;
        }
    }
    if (tid == 0) {
        double gmax = -INFINITY;
        int garg = -1;
        for (int i = 0; i < kSelSmallThreads / 64; ++i)                      // in row order: the first maximum wins
            if (s_arg[i] >= 0 && (garg < 0 || s_max[i] > gmax)) {
                gmax = s_max[i];
                garg = s_arg[i];
            }
        result[0] = tot.x;
        result[1] = tot.y;
        result[2] = garg;
    }
}

// ------------------------------------------------------------------------------------------------
// top-k: MSB-first radix select on the order-mapped scores, then the stable compaction above
// ------------------------------------------------------------------------------------------------
// state[0] = prefix (high bits fixed so far), state[1] = k still to find inside the prefix bucket
__global__ __launch_bounds__(256) void topk_hist_kernel(const double* scores, int64_t n, int pass,
                                                         const uint64_t* state, unsigned int* hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const uint64_t prefix = state[0];
    const uint64_t mask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = order_key(scores[r]);
        if (k != 0ull && (k & mask) == prefix) atomicAdd(&h[(k >> shift) & 255], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void topk_pick_kernel(int pass, uint64_t* state, unsigned int* hist) {
    if (threadIdx.x != 0) return;
    const int shift = 56 - 8 * pass;
    uint64_t need = state[1];
    int b = 255;
    for (; b > 0; --b) {
        if (hist[b] >= need) break;
        need -= hist[b];
    }
    state[0] |= (uint64_t)b << shift;
    state[1] = need;
    for (int i = 0; i < 256; ++i) hist[i] = 0;
}

// The same selection for a SMALL database (n <= kTopkSmallN, k <= kTopkSmallK) as ONE launch of one workgroup: the order keys live in
// LDS, the eight histogram passes and the stable compaction are separated by barriers instead of 19 launches and three
// synchronisations (a 10k-clip top-20: ~0.15 ms of launch sequence -> one launch and one copy).  Same survivors in the same (row)
// order as the general path: keys above the pivot first, then the first `need` rows that tie with it.
constexpr int kTopkSmallN = 16384, kTopkSmallK = 1024, kTopkSmallThreads = 1024;
__global__ __launch_bounds__(kTopkSmallThreads) void topk_small_kernel(const double* scores, int n, int k, int64_t* out_rows, double* out_vals,
                                                                       int64_t* out_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char topk_lds[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(topk_lds);          // [n]
    __shared__ unsigned int hist[256];
    __shared__ uint64_t s_prefix;
    __shared__ unsigned int s_need;
    __shared__ int2 s_wave[kTopkSmallThreads / 64];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += kTopkSmallThreads) keys[i] = order_key(scores[i]);
    if (tid == 0) {
        s_prefix = 0ull;
        s_need = (unsigned)k;
    }
    for (int pass = 0; pass < 8; ++pass) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const int shift = 56 - 8 * pass;
        const uint64_t prefix = s_prefix;
        const uint64_t mask = pass == 0 ? 0ull : (~0ull << (shift + 8));
        // the keys of a pass crowd into few bins (every score of a ranking shares its sign and most of its exponent): the lanes that
        // agree with the wave's first live lane are counted by ONE atomic, the others add themselves
        for (int i0 = 0; i0 < n; i0 += kTopkSmallThreads) {
            const int i = i0 + tid;
            const uint64_t key = i < n ? keys[i] : 0ull;
            const bool live = key != 0ull && (key & mask) == prefix;
            const int digit = live ? (int)((key >> shift) & 255) : -1;
            const unsigned long long act = __ballot(live);
            if (act) {
                const int lead = __ffsll((long long)act) - 1;
                const int first = __shfl(digit, lead, 64);
                const unsigned long long same = __ballot(digit == first);
                if ((tid & 63) == lead)
                    atomicAdd(&hist[first], (unsigned)__popcll(same));
                else if (live && digit != first)
                    atomicAdd(&hist[digit], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {                                              // topk_pick_kernel, by one wave: lane l owns bins 4l .. 4l+3
            const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            unsigned above = h0 + h1 + h2 + h3;                      // inclusive suffix sum over the lanes: bins >= 4l
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_down(above, off, 64);
                if (tid + off < 64) above += o;
            }
            const unsigned need = s_need;
            const unsigned gt3 = above - (h0 + h1 + h2 + h3);        // elements in bins > 4l+3
            // the bin b the serial walk from 255 downwards stops in: the highest b with count(bins >= b) >= need (b = 0 if none)
            int found = -1;
            unsigned rest = 0;
            if (gt3 + h3 >= need) {
                found = 4 * tid + 3;
                rest = need - gt3;
            } else if (gt3 + h3 + h2 >= need) {
                found = 4 * tid + 2;
                rest = need - gt3 - h3;
            } else if (gt3 + h3 + h2 + h1 >= need) {
                found = 4 * tid + 1;
                rest = need - gt3 - h3 - h2;
            } else if (gt3 + h3 + h2 + h1 + h0 >= need) {
                found = 4 * tid;
                rest = need - gt3 - h3 - h2 - h1;
            }
            const unsigned long long have = __ballot(found >= 0 && gt3 < need);   // lanes whose own bins contain the stop
            // the stop lies in the HIGHEST lane for which bins above it hold fewer than `need` and its own close the gap
            const int top = have ? 63 - __builtin_clzll(have) : -1;
            if (top < 0) {
                if (tid == 0) {                                      // fewer than `need` keys under this prefix: bin 0, need minus all above it
                    s_prefix = prefix;
                    s_need = need - (above - h0);
                }
            } else if (tid == top) {
                if (found == 0) rest = need - (above - h0);          // the serial walk never tests bin 0: it takes what is left
                s_prefix = prefix | ((uint64_t)found << shift);
                s_need = rest;
            }
        }
        __syncthreads();
    }
    const uint64_t pivot = s_prefix;
    const int need = (int)s_need;
    // stable compaction: a thread owns a run of consecutive rows
    const int per = (n + kTopkSmallThreads - 1) / kTopkSmallThreads;
    const int r0 = tid * per, r1 = min(n, r0 + per);
    int2 cnt = make_int2(0, 0);
    for (int r = r0; r < r1; ++r) {
        const uint64_t key = keys[r];
        cnt.x += key > pivot ? 1 : 0;
        cnt.y += (key == pivot && key != 0ull) ? 1 : 0;
    }
    const int lane = tid & 63, wid = tid >> 6;
    int2 inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int ax = __shfl_up(inc.x, off, 64), ay = __shfl_up(inc.y, off, 64);
        if (lane >= off) {
            inc.x += ax;
            inc.y += ay;
        }
    }
    if (lane == 63) s_wave[wid] = inc;
    __syncthreads();
    int2 base = make_int2(0, 0), tot = make_int2(0, 0);
    for (int i = 0; i < kTopkSmallThreads / 64; ++i) {
        if (i < wid) {
            base.x += s_wave[i].x;
            base.y += s_wave[i].y;
        }
        tot.x += s_wave[i].x;
        tot.y += s_wave[i].y;
    }
    int o0 = base.x + inc.x - cnt.x, o1 = base.y + inc.y - cnt.y;
    const int n_gt = tot.x, n_eq = min(tot.y, need);
    for (int r = r0; r < r1; ++r) {
        const uint64_t key = keys[r];
        if (key > pivot) {
            out_rows[o0] = r;
            out_vals[o0] = scores[r];
            ++o0;
        } else if (key == pivot && key != 0ull) {
            if (o1 < need) {
                out_rows[n_gt + o1] = r;
                out_vals[n_gt + o1] = scores[r];
            }
            ++o1;
        }
    }
    if (tid == 0) out_count[0] = n_gt + n_eq;
}

// rows of the database -> a dense [cnt][row_elems] block (vq_db_read_rows: the few validated clips a sharded round sends to the
// rank that solves the bootstrapping problems).  16 bytes per thread.
__global__ void gather_rows_kernel(const uint4* feats, const int64_t* rows, int64_t cnt, int64_t row_vec16, uint4* out, int tiled_nv, int tiled_d4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt * row_vec16) return;
    const int64_t r = i / row_vec16, k = i - r * row_vec16;
    if (tiled_nv) {                                   // fp32, tiled: piece g of (clip, v) is one 16-byte unit of the tile
        const int v = (int)(k / tiled_d4), g = (int)(k - (int64_t)v * tiled_d4);
        const int64_t clip = rows[r];
        out[i] = feats[(((clip >> 4) * tiled_nv + v) * tiled_d4 + g) * 16 + (clip & 15)];
    } else {
        out[i] = feats[rows[r] * row_vec16 + k];
    }
}

__global__ void gather_scores_kernel(const double* scores, const int64_t* rows, int64_t cnt, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) out[i] = scores[rows[i]];
}

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
// The results of a query round, device block and (behind the query and the weights) host block alike: avg [N][S] f64 | ne [N][S] i32 |
// scores [N] f64 | result [4] i64 (n_match, n_near, near_argmax, 0) | first kRoundPrefix match rows | first kRoundPrefix near rows;
// every piece starts on a 64-byte boundary.  off[0..5] = the pieces, off[6] = bytes.
constexpr int64_t kRoundPrefix = 1024;
static void round_layout(int64_t n, int S, int64_t off[8]) {
    auto up = [](int64_t v) { return (v + 63) / 64 * 64; };
    off[0] = 0;
    off[1] = off[0] + up(n * S * 8);
    off[2] = off[1] + up(n * S * 4);
    off[3] = off[2] + up(n * 8);
    off[4] = off[3] + 64;
    off[5] = off[4] + kRoundPrefix * 8;
    off[6] = off[5] + kRoundPrefix * 8;
    off[7] = 0;
}

struct vq_db {
    std::mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t n = 0;
    int S = 0, E = 0, D = 0, dtype = VQ_F32;
    int cus = 256;
    void* feats = nullptr;
    bool owns_feats = true;
    // VQ_LAYOUT_ROWS: [N][S][E][D].  VQ_LAYOUT_TILED (fp32, own memory): [tile of 16 clips][S*E][D/4][clip][4] in the SAME block (a
    // tile's 16 rows and its tiled form cover the same bytes; the block is allocated for whole tiles) -- vq_db_set_layout
    int layout = VQ_LAYOUT_ROWS;
    bool feats_exposed = false;      // adopted memory, or the raw pointer was handed out: the library no longer controls the layout
    uint8_t* present = nullptr;
    double* t = nullptr;        // [S*E*D]
    bool have_query = false, have_avg = false, have_scores = false, have_sims = false;
    double* w = nullptr;        // [8]
    double* sims = nullptr;     // lazily [N][S][E]
    double* avg = nullptr;      // [N][S]
    int32_t* ne = nullptr;      // [N][S]
    double* scores = nullptr;   // [N]
    // avg | ne | scores | sel_result | match prefix | near prefix are ONE device allocation (round_dev), in the order of the host
    // block of vq_db_query_round: a round's results go back in one copy
    char* round_dev = nullptr;
    int64_t round_off[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // byte offsets of avg, ne, scores, result, matchp, nearp; [6] = total
    int64_t* matchp = nullptr;       // first kRoundPrefix rows of the last selection's match list
    int64_t* nearp = nullptr;
    // selection scratch
    int nblk = 0;
    int2* blk_cnt = nullptr;
    double* blk_max = nullptr;
    int64_t* blk_arg = nullptr;
    int64_t* sel_result = nullptr;   // [3] device
    int64_t* rows0 = nullptr;        // [N]
    int64_t* rows1 = nullptr;        // [N]
    int64_t last_n0 = 0, last_n1 = 0;
    uint64_t* tk_state = nullptr;    // [2]
    unsigned int* tk_hist = nullptr; // [256]
    double* batch_buf = nullptr;     // batched scan: queries [Q][NV][D] | weights [Q][S] | (two-kernel form: sims [NV][Q][N]) | scores [Q][N]
    double* batch_scores = nullptr;  // where the scores of the last batched scan start inside batch_buf
    int64_t batch_cap = 0;           // doubles
    int batch_q = 0;                 // queries of the last batched scan
    double* grid_buf = nullptr;      // scratch for grid / gathers
    int64_t grid_cap = 0;
    size_t elem() const { return dtype == VQ_F64 ? 8 : 4; }
};

static int db_free(vq_db* db) {
    if (db->owns_feats && db->feats) (void)hipFree(db->feats);
    void* ptrs[] = {db->present, db->t,       db->sims,  db->round_dev, db->blk_cnt,
                    db->blk_max, db->blk_arg, db->rows0, db->rows1, db->tk_state, db->tk_hist, db->grid_buf, db->batch_buf};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    return VQ_OK;
}

static int ensure_grid_buf(vq_db* db, int64_t bytes);

// rows (by number) -> a dense row-major [L][S][E][D] block in the handle's scratch, whatever the layout; caller holds the lock
static int gather_rows_device(vq_db* db, const int64_t* rows_host, int L, const void** dense) {
    const int64_t row_bytes = (int64_t)db->S * db->E * db->D * (int64_t)db->elem();      // D % 4 == 0: whole 16-byte pieces
    const int64_t rb = ((int64_t)L * 8 + 15) / 16 * 16;
    const int rc = ensure_grid_buf(db, rb + (int64_t)L * row_bytes);
    if (rc != VQ_OK) return rc;
    char* base = (char*)db->grid_buf;
    VQ_HIP(hipMemcpyAsync(base, rows_host, (size_t)L * 8, hipMemcpyHostToDevice, db->stream));
    const int64_t vec = row_bytes / 16;
    const bool tiled = db->layout == VQ_LAYOUT_TILED;
    gather_rows_kernel<<<cdiv((int64_t)L * vec, 256), 256, 0, db->stream>>>((const uint4*)db->feats, (const int64_t*)base, L, vec, (uint4*)(base + rb),
                                                                            tiled ? db->S * db->E : 0, db->D / 4);
    VQ_CHECK_LAUNCH();
    *dense = base + rb;
    return VQ_OK;
}

// In place: a tile's 16 rows and its tiled form occupy the same bytes, so the block is converted a bounded run of tiles at a time
// through a scratch block (rows -> scratch in the new order -> back).  One sweep of reads and writes each way; caller holds the lock.
static int convert_layout(vq_db* db, int to) {
    const int NV = db->S * db->E;
    const int64_t ntiles = (db->n + 15) / 16, tile_bytes = (int64_t)16 * NV * db->D * 4;
    const int64_t per = std::max<int64_t>(1, std::min<int64_t>(ntiles, (int64_t)(256u << 20) / tile_bytes));
    float* scratch = nullptr;
    hipError_t e = hipMalloc((void**)&scratch, (size_t)(per * tile_bytes));
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        device_pool_trim();
        e = hipMalloc((void**)&scratch, (size_t)(per * tile_bytes));
    }
    if (e != hipSuccess) return fail(VQ_E_NOMEM, "no room for the %lld MB conversion block: %s", (long long)(per * tile_bytes >> 20), hipGetErrorString(e));
    const size_t lds = 16 * (size_t)(db->D + 4) * sizeof(float);         // 65 792 bytes at D = 1024: above the 64 KB default limit
    {   // raised per (kernel, device) like every other large-LDS launch (vq_common.h); the caller holds a DeviceGuard
        const auto raise = [&]() -> int {
            VQ_DYN_LDS(mirror_build_kernel, lds);
            VQ_DYN_LDS(untile_kernel, lds);
            return VQ_OK;
        };
        const int rc0 = raise();
        if (rc0 != VQ_OK) {
            (void)hipFree(scratch);
            return rc0;
        }
    }
    int rc = VQ_OK;
    for (int64_t t0 = 0; t0 < ntiles && rc == VQ_OK; t0 += per) {
        const int64_t k = std::min(per, ntiles - t0);
        float* block = (float*)db->feats + t0 * 16 * NV * db->D;
        const int64_t rows_here = std::min<int64_t>(db->n - t0 * 16, k * 16);
        if (to == VQ_LAYOUT_TILED)
            mirror_build_kernel<<<(unsigned)(k * NV), 256, lds, db->stream>>>(block, scratch, rows_here, NV, db->D);
        else
            untile_kernel<<<(unsigned)(k * NV), 256, lds, db->stream>>>(block, scratch, rows_here, NV, db->D);
        e = hipGetLastError();
        // tiled -> rows leaves the rows past n of the last tile unwritten in the scratch block: copy whole rows only
        const size_t bytes = to == VQ_LAYOUT_TILED ? (size_t)(k * tile_bytes) : (size_t)rows_here * NV * db->D * 4;
        if (e == hipSuccess) e = hipMemcpyAsync(block, scratch, bytes, hipMemcpyDeviceToDevice, db->stream);
        if (e != hipSuccess) rc = fail(VQ_E_HIP, "layout conversion failed: %s", hipGetErrorString(e));
    }
    e = hipStreamSynchronize(db->stream);
    if (rc == VQ_OK && e != hipSuccess) rc = fail(VQ_E_HIP, "layout conversion failed: %s", hipGetErrorString(e));
    (void)hipFree(scratch);
    if (rc == VQ_OK) db->layout = to;
    return rc;
}

namespace vq {
int db_copy_scores_ordered(vq_db* db, double* dst_dev, int64_t* n_out, hipStream_t st) {
    VQ_REQUIRE(db && dst_dev, "NULL argument");
    std::lock_guard<std::mutex> lk(db->mu);
    if (!db->have_scores) return fail(VQ_E_STATE, "no scores: scan (or rescore) first");
    DeviceGuard g(db->device);
    if (n_out) *n_out = db->n;
    if (!db->n) return VQ_OK;
    if (st != db->stream) {                     // the scan was only ENQUEUED on db->stream: nothing else orders st behind it
        hipEvent_t ev;
        VQ_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, db->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, ev, 0);
        (void)hipEventDestroy(ev);              // destruction is deferred until the event has completed
        if (e != hipSuccess) return fail(VQ_E_HIP, "ordering the score copy behind the scan failed: %s", hipGetErrorString(e));
    }
    VQ_HIP(hipMemcpyAsync(dst_dev, db->scores, (size_t)db->n * 8, hipMemcpyDeviceToDevice, st));
    return VQ_OK;
}
}  // namespace vq

extern "C" {

const char* vq_last_error(void) { return last_error_ref().c_str(); }
int vq_abi_version(void) { return VQ_ABI_VERSION; }

int vq_device_count(int* count) {
    VQ_REQUIRE(count, "count is NULL");
    VQ_HIP(hipGetDeviceCount(count));
    return VQ_OK;
}

struct vq_timer_t {
    hipEvent_t a, b;
};
int vq_timer_create(void** timer) {
    VQ_REQUIRE(timer, "timer is NULL");
    auto* t = new vq_timer_t;
    VQ_HIP(hipEventCreate(&t->a));
    VQ_HIP(hipEventCreate(&t->b));
    *timer = t;
    return VQ_OK;
}
int vq_timer_start(void* timer, void* stream) {
    VQ_HIP(hipEventRecord(static_cast<vq_timer_t*>(timer)->a, (hipStream_t)stream));
    return VQ_OK;
}
int vq_timer_stop(void* timer, void* stream) {
    VQ_HIP(hipEventRecord(static_cast<vq_timer_t*>(timer)->b, (hipStream_t)stream));
    return VQ_OK;
}
int vq_timer_elapsed_ms(void* timer, float* ms) {
    auto* t = static_cast<vq_timer_t*>(timer);
    VQ_HIP(hipEventSynchronize(t->b));
    VQ_HIP(hipEventElapsedTime(ms, t->a, t->b));
    return VQ_OK;
}
int vq_timer_destroy(void* timer) {
    auto* t = static_cast<vq_timer_t*>(timer);
    if (!t) return VQ_OK;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
    return VQ_OK;
}

int vq_db_create(int64_t n, int32_t S, int32_t E, int32_t D, int32_t dtype, int32_t device, vq_db** out) {
    VQ_REQUIRE(out, "out is NULL");
    *out = nullptr;
    VQ_REQUIRE(n > 0 && S > 0 && E > 0 && D > 0, "n, S, E, D must be positive (got %lld, %d, %d, %d)", (long long)n, S, E, D);
    VQ_REQUIRE(S <= 8, "at most 8 streams are supported (got %d)", S);
    VQ_REQUIRE(S * E <= 64, "S*E must be <= 64 (got %d)", S * E);
    VQ_REQUIRE(D % 4 == 0, "D must be a multiple of 4 (got %d)", D);
    VQ_REQUIRE(dtype == VQ_F32 || dtype == VQ_F64, "dtype must be VQ_F32 or VQ_F64");
    VQ_REQUIRE(n < (1ll << 31) * (long long)SEL_CHUNK, "n too large");
    int ndev = 0;
    VQ_HIP(hipGetDeviceCount(&ndev));
    VQ_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard g(device);
    auto* db = new vq_db;
    db->device = device;
    db->n = n;
    db->S = S;
    db->E = E;
    db->D = D;
    db->dtype = dtype;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) db->cus = prop.multiProcessorCount;
    db->nblk = cdiv(n, SEL_CHUNK);
    const size_t fbytes = (size_t)((n + 15) / 16 * 16) * S * E * D * db->elem();      // whole tiles of 16 clips (vq_db_set_layout)
#define A_(ptr, bytes)                                    \
    do {                                                  \
        hipError_t e_ = hipMalloc((void**)&(ptr), bytes); \
        if (e_ == hipErrorOutOfMemory) {                  \
            (void)hipGetLastError();                      \
            device_pool_trim();    /* blocks closed extractors left for reuse (vq_tsn.hip) */ \
            e_ = hipMalloc((void**)&(ptr), bytes);        \
        }                                                 \
        if (e_ != hipSuccess) {                           \
            db_free(db);                                  \
            delete db;                                    \
            return fail(VQ_E_NOMEM, "vq::malloc_trim(%zu bytes) failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
        }                                                 \
    } while (0)
    A_(db->feats, fbytes);
    if (n % 16) (void)hipMemset((char*)db->feats + (size_t)n * S * E * D * db->elem(), 0, fbytes - (size_t)n * S * E * D * db->elem());
    A_(db->t, ((size_t)S * E * D * 8 + 63) / 64 * 64 + 64);      // query | weights [8]: adjacent, like the head of a round's host block
    db->w = (double*)((char*)db->t + ((size_t)S * E * D * 8 + 63) / 64 * 64);
    round_layout(n, S, db->round_off);
    A_(db->round_dev, (size_t)db->round_off[6]);
    db->avg = (double*)(db->round_dev + db->round_off[0]);
    db->ne = (int32_t*)(db->round_dev + db->round_off[1]);
    db->scores = (double*)(db->round_dev + db->round_off[2]);
    db->sel_result = (int64_t*)(db->round_dev + db->round_off[3]);
    db->matchp = (int64_t*)(db->round_dev + db->round_off[4]);
    db->nearp = (int64_t*)(db->round_dev + db->round_off[5]);
    A_(db->blk_cnt, (size_t)db->nblk * sizeof(int2));
    A_(db->blk_max, (size_t)db->nblk * 8);
    A_(db->blk_arg, (size_t)db->nblk * 8);
    A_(db->rows0, (size_t)n * 8);
    A_(db->rows1, (size_t)n * 8);
    A_(db->tk_state, 2 * 8);
    A_(db->tk_hist, 256 * 4);
#undef A_
    *out = db;
    return VQ_OK;
}

int vq_db_destroy(vq_db* db) {
    if (!db) return VQ_OK;
    {
        DeviceGuard g(db->device);
        (void)hipStreamSynchronize(db->stream);
        db_free(db);
    }
    delete db;
    return VQ_OK;
}

int vq_db_set_stream(vq_db* db, void* s) {
    VQ_REQUIRE(db, "db is NULL");
    std::lock_guard<std::mutex> lk(db->mu);
    db->stream = (hipStream_t)s;
    return VQ_OK;
}

int vq_db_shape(vq_db* db, int64_t* n, int32_t* S, int32_t* E, int32_t* D, int32_t* dtype) {
    VQ_REQUIRE(db, "db is NULL");
    if (n) *n = db->n;
    if (S) *S = db->S;
    if (E) *E = db->E;
    if (D) *D = db->D;
    if (dtype) *dtype = db->dtype;
    return VQ_OK;
}

int vq_db_upload(vq_db* db, int64_t row0, int64_t nrows, const void* feats_host) {
    VQ_REQUIRE(db && feats_host, "NULL argument");
    VQ_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= db->n, "rows [%lld,%lld) outside [0,%lld)", (long long)row0,
               (long long)(row0 + nrows), (long long)db->n);
    std::lock_guard<std::mutex> lk(db->mu);
    DeviceGuard g(db->device);
    const size_t row_bytes = (size_t)db->S * db->E * db->D * db->elem();
    if (db->layout == VQ_LAYOUT_TILED) {
        // rows land in a staging block and are dealt into their tiles by a kernel, a bounded chunk at a time
        const int64_t chunk = std::max<int64_t>(1, (int64_t)(64u << 20) / (int64_t)row_bytes);
        void* stage = nullptr;
        VQ_HIP(vq::malloc_trim(&stage, (size_t)std::min<int64_t>(chunk, std::max<int64_t>(nrows, 1)) * row_bytes));
        const int NV = db->S * db->E, d4 = db->D / 4;
        for (int64_t r = 0; r < nrows; r += chunk) {
            const int64_t k = std::min(chunk, nrows - r);
            hipError_t e = hipMemcpyAsync(stage, (const char*)feats_host + r * row_bytes, k * row_bytes, hipMemcpyHostToDevice, db->stream);
            if (e == hipSuccess) {
                scatter_rows_tiled_kernel<<<cdiv(k * NV * d4, 256), 256, 0, db->stream>>>((const float4*)stage, (float4*)db->feats, row0 + r, k, NV, d4);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(db->stream);
            if (e != hipSuccess) {
                (void)hipFree(stage);
                return fail(VQ_E_HIP, "upload into the tiled layout failed: %s", hipGetErrorString(e));
            }
        }
        VQ_HIP(hipFree(stage));
    } else {
        VQ_HIP(hipMemcpyAsync((char*)db->feats + row0 * row_bytes, feats_host, nrows * row_bytes, hipMemcpyHostToDevice,
                              db->stream));
        VQ_HIP(hipStreamSynchronize(db->stream));
    }
    db->have_avg = db->have_scores = db->have_sims = false;
    return VQ_OK;
}

int vq_db_adopt_device(vq_db* db, void* feats_dev) {
    VQ_REQUIRE(db && feats_dev, "NULL argument");
    VQ_REQUIRE(((uintptr_t)feats_dev & 15) == 0, "device feature block must be 16-byte aligned");
    std::lock_guard<std::mutex> lk(db->mu);
    DeviceGuard g(db->device);
    if (db->owns_feats && db->feats) VQ_HIP(hipFree(db->feats));
    db->feats = feats_dev;
    db->owns_feats = false;
    db->have_avg = db->have_scores = db->have_sims = false;
    db->feats_exposed = true;                 // the caller owns the memory (row-major, possibly not whole tiles): never re-tiled
    db->layout = VQ_LAYOUT_ROWS;
    return VQ_OK;
}

int vq_db_set_present(vq_db* db, const uint8_t* present_host) {
    VQ_REQUIRE(db, "db is NULL");
    std::lock_guard<std::mutex> lk(db->mu);
    DeviceGuard g(db->device);
    const size_t bytes = (size_t)db->n * db->S * db->E;
    if (!present_host) {
        if (db->present) VQ_HIP(hipFree(db->present));
        db->present = nullptr;
    } else {
        if (!db->present) VQ_HIP(vq::malloc_trim((void**)&db->present, bytes));
        VQ_HIP(hipMemcpyAsync(db->present, present_host, bytes, hipMemcpyHostToDevice, db->stream));
        VQ_HIP(hipStreamSynchronize(db->stream));
    }
    db->have_avg = db->have_scores = db->have_sims = false;
    return VQ_OK;
}

int vq_db_generate(vq_db* db, uint64_t seed, int64_t global_row0, const float* scales_host) {
    VQ_REQUIRE(db && scales_host, "NULL argument");
    std::lock_guard<std::mutex> lk(db->mu);
    DeviceGuard g(db->device);
    float* scales_dev = reinterpret_cast<float*>(db->w);   // 64 bytes of scratch; w is rewritten by every scan
    VQ_HIP(hipMemcpyAsync(scales_dev, scales_host, db->S * sizeof(float), hipMemcpyHostToDevice, db->stream));
    const int64_t per_row = (int64_t)db->S * db->E * db->D;
    const int64_t total = db->n * per_row;
    const uint64_t seed_mul = seed * 0x9E3779B97F4A7C15ull;
    const int64_t threads = (total + 3) / 4;
    const int64_t blocks = (threads + 255) / 256;
    VQ_REQUIRE(blocks < (1ll << 31), "DB too large for one generate launch");
    if (db->dtype == VQ_F32)
        generate_kernel<float><<<(unsigned)blocks, 256, 0, db->stream>>>((float*)db->feats, total, seed_mul,
                                                                          (uint64_t)(global_row0 * per_row),
                                                                          db->E * db->D, db->S, scales_dev);
    else
        generate_kernel<double><<<(unsigned)blocks, 256, 0, db->stream>>>((double*)db->feats, total, seed_mul,
                                                                           (uint64_t)(global_row0 * per_row),
                                                                           db->E * db->D, db->S, scales_dev);
    VQ_CHECK_LAUNCH();
    VQ_HIP(hipStreamSynchronize(db->stream));
    db->have_avg = db->have_scores = db->have_sims = false;
    if (db->layout == VQ_LAYOUT_TILED) {          // the generator writes rows; a tiled database gets its layout back
        db->layout = VQ_LAYOUT_ROWS;
        return convert_layout(db, VQ_LAYOUT_TILED);
    }
    return VQ_OK;
}

int vq_db_feats_devptr(vq_db* db, void** p) {
    VQ_REQUIRE(db && p, "NULL argument");
    std::lock_guard<std::mutex> lk(db->mu);
    if (db->layout != VQ_LAYOUT_ROWS) return fail(VQ_E_STATE, "the database is tiled: its block is not [N][S][E][D] (vq_db_set_layout(VQ_LAYOUT_ROWS) first)");
    *p = db->feats;
    db->feats_exposed = true;                 // whoever holds the raw pointer assumes row-major memory: the layout is frozen
    return VQ_OK;
}

int vq_db_set_query(vq_db* db, const double* t_host) {
    VQ_REQUIRE(db && t_host, "NULL argument");
    std::lock_guard<std::mutex> lk(db->mu);
    DeviceGuard g(db->device);
    VQ_HIP(hipMemcpyAsync(db->t, t_host, (size_t)db->S * db->E * db->D * 8, hipMemcpyHostToDevice, db->stream));
    VQ_HIP(hipStreamSynchronize(db->stream));
    db->have_query = true;
    db->have_avg = db->have_scores = db->have_sims = false;
    return VQ_OK;
}

int vq_db_bootstrap_target(vq_db* db, const int64_t* valid_rows, int32_t n_valid, const int64_t* invalid_rows, int32_t n_invalid,
                           double mu, double* targets_host, int32_t set_query) {
    VQ_REQUIRE(db && valid_rows, "NULL argument");
    VQ_REQUIRE(n_valid >= 1 && n_invalid >= 0 && (n_invalid == 0 || invalid_rows), "need >= 1 validated match (and invalid_rows if n_invalid > 0)");
    for (int k = 0; k < n_valid; ++k)
        VQ_REQUIRE(valid_rows[k] >= 0 && valid_rows[k] < db->n, "valid row %lld outside [0,%lld)", (long long)valid_rows[k], (long long)db->n);
    for (int k = 0; k < n_invalid; ++k)
        VQ_REQUIRE(invalid_rows[k] >= 0 && invalid_rows[k] < db->n, "invalid row %lld outside [0,%lld)", (long long)invalid_rows[k],
                   (long long)db->n);
    std::lock_guard<std::mutex> lk(db->mu);
    DeviceGuard g(db->device);
    const int P = db->S * db->E, stride = n_valid + n_invalid;
    // the validated rows are gathered into a dense [stride][P][D] block first (whatever the layout; tens of rows of 40 KB)
    std::vector<int64_t> rows((size_t)stride);
    for (int k = 0; k < stride; ++k) rows[k] = k < n_valid ? valid_rows[k] : invalid_rows[k - n_valid];
    const void* dense = nullptr;
    int rc = gather_rows_device(db, rows.data(), stride, &dense);
    if (rc != VQ_OK) return rc;
    std::vector<int64_t> off((size_t)P * stride);
    std::vector<int32_t> nv(P, n_valid), ni(P, n_invalid);
    for (int p = 0; p < P; ++p)
        for (int k = 0; k < stride; ++k) off[(size_t)p * stride + k] = ((int64_t)k * P + p) * db->D;      // p = s * E + e
    rc = bootstrap_from_device_rows(dense, db->dtype, off, stride, nv.data(), ni.data(), P, db->D, mu, db->stream, targets_host,
                                    set_query ? db->t : nullptr);
    if (rc != VQ_OK) return rc;
    if (set_query) {
        db->have_query = true;
        db->have_avg = db->have_scores = db->have_sims = false;
    }
    return VQ_OK;
}

int vq_db_set_query_from_row(vq_db* db, int64_t row, double* t_out_host) {
    VQ_REQUIRE(db, "db is NULL");
    VQ_REQUIRE(row >= 0 && row < db->n, "row %lld outside [0,%lld)", (long long)row, (long long)db->n);
    std::lock_guard<std::mutex> lk(db->mu);
    DeviceGuard g(db->device);
    const int NV = db->S * db->E;
    if (db->dtype == VQ_F32)
        scale_query_kernel<float><<<NV, 256, 0, db->stream>>>((const float*)db->feats, row, NV, db->D, db->t, db->layout == VQ_LAYOUT_TILED);
    else
        scale_query_kernel<double><<<NV, 256, 0, db->stream>>>((const double*)db->feats, row, NV, db->D, db->t, false);
    VQ_CHECK_LAUNCH();
    if (t_out_host) {
        VQ_HIP(hipMemcpyAsync(t_out_host, db->t, (size_t)NV * db->D * 8, hipMemcpyDeviceToHost, db->stream));
        VQ_HIP(hipStreamSynchronize(db->stream));
    }
    db->have_query = true;
    db->have_avg = db->have_scores = db->have_sims = false;
    return VQ_OK;
}

}  // extern "C"

template <typename T, int S, int E, int CH>
static int launch_scan_t(vq_db* db, const ScanArgs& a) {
    const size_t lds = (size_t)S * E * CH * 256 * 8;
    int per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds);
    if (per_cu < 1) per_cu = 1;
    if (db->layout == VQ_LAYOUT_TILED) {
        if constexpr (sizeof(T) == 4) {
            auto kern = scan_tiled_kernel<S, E, CH, kTiledGroup>;
            VQ_DYN_LDS(kern, lds);
            const int64_t want = ((a.n + 15) / 16 + 3) / 4;   // 4 waves (tiles in flight) per block
            const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)db->cus * per_cu));
            kern<<<grid, 256, lds, db->stream>>>(a);
            VQ_CHECK_LAUNCH();
            return VQ_OK;
        }
        return fail(VQ_E_STATE, "internal: a tiled database is fp32");
    }
    // a database whose clips are a few per wave: the lean instantiation, three waves per SIMD (VQ_SCAN_LEAN=0|1 forces one or the other)
    static const int force_lean = getenv("VQ_SCAN_LEAN") ? atoi(getenv("VQ_SCAN_LEAN")) : -1;
    // (measured, MI355X, fp32 rows, lean against not: 10 000 clips x 2 x 3: 49 against 59 us; 50 000: 0.77 against 0.72 of the HBM peak; 1 M x 2 x 3:
    // 0.81 against 0.76; 1 M x 2 x 5: 0.81 against 0.82; fp64 rows 200 000 x 2 x 3: 0.75 against 0.76)
    const bool lean = force_lean >= 0 ? force_lean != 0 : (a.n < (int64_t)db->cus * per_cu * 4 * 16 || (sizeof(T) == 4 && S * E <= 6));
    if (lean) {
        auto kern = scan_kernel<T, S, E, CH, true>;
        VQ_DYN_LDS(kern, lds);
        const int64_t want = (a.n + 3) / 4;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)db->cus * per_cu));
        kern<<<grid, 256, lds, db->stream>>>(a);
        VQ_CHECK_LAUNCH();
        return VQ_OK;
    }
    auto kern = scan_kernel<T, S, E, CH>;
    VQ_DYN_LDS(kern, lds);
    // (as many waves as fit the chip.  Fewer waves that each walk the same number of clips -- 2 500 waves of exactly four clips for a
    // 10 000-clip database instead of 3 072 of three or four -- were measured in round 6: 62.9 against 56.8 us per scan; the streams in
    // flight matter more than the ragged last round.)
    const int64_t want = (a.n + 3) / 4;   // 4 waves (clips in flight) per block
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)db->cus * per_cu));
    kern<<<grid, 256, lds, db->stream>>>(a);
    VQ_CHECK_LAUNCH();
    return VQ_OK;
}

// shapes the fast one-query kernels (and with them the tiled layout) exist for
static bool fast_scan_shape(const vq_db* db) { return db->D == 1024 && db->S >= 1 && db->S <= 2 && db->E >= 1 && db->E <= 5; }

template <typename T>
static int launch_scan(vq_db* db, const ScanArgs& a) {
    if (db->D == 1024) {
#define C_(s_, e_) \
    if (db->S == s_ && db->E == e_) return launch_scan_t<T, s_, e_, 4>(db, a);
        C_(1, 1) C_(1, 2) C_(1, 3) C_(1, 4) C_(1, 5)
        C_(2, 1) C_(2, 2) C_(2, 3) C_(2, 4) C_(2, 5)
#undef C_
    }
    if (db->layout != VQ_LAYOUT_ROWS) return fail(VQ_E_STATE, "internal: tiled layout on a shape without a tiled scan kernel");
    const int64_t want = (a.n + 3) / 4;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)db->cus * 8));
    scan_generic_kernel<T><<<grid, 256, 0, db->stream>>>(a);
    VQ_CHECK_LAUNCH();
    return VQ_OK;
}

static int run_select(vq_db* db, const SelPred& pred, int64_t limit1, int64_t host_result[3], bool with_prefix = false) {
    if (db->n <= kSelSmallN && limit1 < 0) {             // a small database: one launch
        select_small_kernel<<<1, kSelSmallThreads, 0, db->stream>>>(db->scores, (int)db->n, pred, db->sel_result, db->rows0, db->rows1, db->matchp,
                                                                    db->nearp, with_prefix ? (int)kRoundPrefix : 0);
        VQ_CHECK_LAUNCH();
        if (!host_result) return VQ_OK;
        VQ_HIP(hipMemcpyAsync(host_result, db->sel_result, 3 * 8, hipMemcpyDeviceToHost, db->stream));
        VQ_HIP(hipStreamSynchronize(db->stream));
        return VQ_OK;
    }
    select_count_kernel<<<db->nblk, SEL_BLOCK, 0, db->stream>>>(db->scores, db->n, pred, db->blk_cnt, db->blk_max,
                                                                 db->blk_arg);
    VQ_CHECK_LAUNCH();
    select_scan_kernel<<<1, SEL_BLOCK, 0, db->stream>>>(db->blk_cnt, db->blk_max, db->blk_arg, db->nblk, db->sel_result,
                                                         limit1);
    VQ_CHECK_LAUNCH();
    select_scatter_kernel<<<db->nblk, SEL_BLOCK, 0, db->stream>>>(db->scores, db->n, pred, db->blk_cnt, db->rows0,
                                                                   db->rows1, limit1, db->matchp, db->nearp, with_prefix ? kRoundPrefix : 0);
    VQ_CHECK_LAUNCH();
    if (!host_result) return VQ_OK;                  // vq_db_query_round: the counts travel with the round's block
    VQ_HIP(hipMemcpyAsync(host_result, db->sel_result, 3 * 8, hipMemcpyDeviceToHost, db->stream));
    VQ_HIP(hipStreamSynchronize(db->stream));
    return VQ_OK;
}

extern "C" {

int vq_db_scan(vq_db* db, const double* w_host, int32_t keep_sims) {
    VQ_REQUIRE(db, "db is NULL");
    std::lock_guard<std::mutex> lk(db->mu);
    if (!db->have_query) return fail(VQ_E_STATE, "vq_db_scan: no query set (call vq_db_set_query first)");
    DeviceGuard g(db->device);
    if (keep_sims && !db->sims) VQ_HIP(vq::malloc_trim((void**)&db->sims, (size_t)db->n * db->S * db->E * 8));
    if (w_host) VQ_HIP(hipMemcpyAsync(db->w, w_host, db->S * 8, hipMemcpyHostToDevice, db->stream));
    ScanArgs a;
    a.feats = db->feats;
    a.t = db->t;
    a.present = db->present;
    a.w = w_host ? db->w : nullptr;
    a.sims = keep_sims ? db->sims : nullptr;
    a.avg = db->avg;
    a.ne = db->ne;
    a.scores = db->scores;
    a.n = db->n;
    a.S = db->S;
    a.E = db->E;
    a.D = db->D;
    const int rc = db->dtype == VQ_F32 ? launch_scan<float>(db, a) : launch_scan<double>(db, a);
    if (rc != VQ_OK) return rc;
    db->have_avg = true;
    db->have_sims = keep_sims != 0;
    db->have_scores = w_host != nullptr;
    return VQ_OK;
}

int vq_db_scan_batch(vq_db* db, int32_t n_queries, const double* t_host, const double* w_host, double* scores_host) {
    VQ_REQUIRE(db && t_host && w_host, "NULL argument");
    VQ_REQUIRE(n_queries >= 1 && n_queries <= kBatchSlots, "n_queries must be in [1,%d] (got %d)", kBatchSlots, n_queries);
    VQ_REQUIRE(db->D % 256 == 0 && db->D <= 1024, "batched scan needs D in {256,512,768,1024} (got %d)", db->D);
    VQ_REQUIRE(db->S <= 8, "at most 8 streams");
    std::lock_guard<std::mutex> lk(db->mu);
    DeviceGuard g(db->device);
    const int Q = n_queries, NV = db->S * db->E;
    const int64_t n_t = (int64_t)Q * NV * db->D, n_w = (int64_t)Q * db->S, n_sc = (int64_t)Q * db->n;
    const int64_t need = n_t + n_w + n_sc;
    if (db->batch_cap < need) {
        if (db->batch_buf) VQ_HIP(hipFree(db->batch_buf));
        db->batch_buf = nullptr;
        db->batch_cap = 0;
        VQ_HIP(vq::malloc_trim((void**)&db->batch_buf, (size_t)need * 8));
        db->batch_cap = need;
    }
    double* d_t = db->batch_buf;
    double* d_w = d_t + n_t;
    double* d_scores = d_w + n_w;
    db->batch_scores = d_scores;
    VQ_HIP(hipMemcpyAsync(d_t, t_host, (size_t)n_t * 8, hipMemcpyHostToDevice, db->stream));
    VQ_HIP(hipMemcpyAsync(d_w, w_host, (size_t)n_w * 8, hipMemcpyHostToDevice, db->stream));
    const size_t lds = ((size_t)kBatchSlots * db->D + 32) * 8;   // sixteen swizzled query rows
    const int ch = db->D / 256;
    {
        const bool tiled = db->layout == VQ_LAYOUT_TILED;      // whole 128-byte lines per load instruction, non-temporal: 0.75 of the HBM peak against 0.71
        BatchFusedArgs f;
        f.feats = db->feats;
        f.t = d_t;
        f.w = d_w;
        f.present = db->present;
        f.scores = d_scores;
        f.n = db->n;
        f.Q = Q;
        f.S = db->S;
        f.E = db->E;
        f.D = db->D;
        constexpr int TW = 2;                                     // tiles of 16 clips per wave and round
        // chunks of 8 pieces = 128 elements: a refill reads 512 contiguous bytes of each of the 16 clips (two ring buffers).  With 64-
        // element chunks and four buffers the pass took 7.60 ms instead of 7.45 at cfg 4 (65 k slow streams of 256-byte reads)
        const int64_t ntiles = (db->n + 15) / 16;
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((ntiles + 16 * TW - 1) / (16 * TW), (int64_t)db->cus));
#define VQ_FUSED_LAUNCH(T, CH)                                                                                             \
    {                                                                                                                      \
        auto kern = db->present ? batch_fused_kernel<T, CH, TW, true, 8> : batch_fused_kernel<T, CH, TW, false, 8>;       \
        if (tiled) kern = db->present ? batch_fused_kernel<T, CH, TW, true, 8, true> : batch_fused_kernel<T, CH, TW, false, 8, true>; \
        VQ_DYN_LDS(kern, (kBatchSlots * CH * 256 + 32) * 8);                                                                \
        kern<<<blocks, 1024, lds, db->stream>>>(f);                                                                        \
    }
        if (db->dtype == VQ_F32) {
            if (ch == 4) VQ_FUSED_LAUNCH(float, 4) else if (ch == 3) VQ_FUSED_LAUNCH(float, 3) else if (ch == 2) VQ_FUSED_LAUNCH(float, 2) else VQ_FUSED_LAUNCH(float, 1)
        } else {
            if (ch == 4) VQ_FUSED_LAUNCH(double, 4) else if (ch == 3) VQ_FUSED_LAUNCH(double, 3) else if (ch == 2) VQ_FUSED_LAUNCH(double, 2) else VQ_FUSED_LAUNCH(double, 1)
        }
#undef VQ_FUSED_LAUNCH
        VQ_CHECK_LAUNCH();

    }
    db->batch_q = Q;
    if (scores_host) {
        VQ_HIP(hipMemcpyAsync(scores_host, d_scores, (size_t)n_sc * 8, hipMemcpyDeviceToHost, db->stream));
        VQ_HIP(hipStreamSynchronize(db->stream));
    }
    return VQ_OK;
}

int vq_db_batch_scores_devptr(vq_db* db, void** dev_ptr, int32_t* n_queries) {
    VQ_REQUIRE(db && dev_ptr, "NULL argument");
    std::lock_guard<std::mutex> lk(db->mu);
    if (db->batch_q == 0) return fail(VQ_E_STATE, "no batched scan has run");
    const int NV = db->S * db->E;
    (void)NV;
    *dev_ptr = db->batch_scores;
    if (n_queries) *n_queries = db->batch_q;
    return VQ_OK;
}

int vq_db_rescore(vq_db* db, const double* w_host) {
    VQ_REQUIRE(db && w_host, "NULL argument");
    std::lock_guard<std::mutex> lk(db->mu);
    if (!db->have_avg) return fail(VQ_E_STATE, "vq_db_rescore: no similarities cached (call vq_db_scan first)");
    DeviceGuard g(db->device);
    VQ_HIP(hipMemcpyAsync(db->w, w_host, db->S * 8, hipMemcpyHostToDevice, db->stream));
    rescore_kernel<<<cdiv(db->n, 256), 256, 0, db->stream>>>(db->avg, db->w, db->scores, db->n, db->S);
    VQ_CHECK_LAUNCH();
    db->have_scores = true;
    return VQ_OK;
}

int vq_db_read_similarities(vq_db* db, double* avg_host, int32_t* ne_host, double* sims_host) {
    VQ_REQUIRE(db, "db is NULL");
    std::lock_guard<std::mutex> lk(db->mu);
    if (!db->have_avg) return fail(VQ_E_STATE, "no similarities cached (call vq_db_scan first)");
    if (sims_host && !db->have_sims) return fail(VQ_E_STATE, "per-split similarities were not kept (scan with keep_sims=1)");
    DeviceGuard g(db->device);
    if (avg_host) VQ_HIP(hipMemcpyAsync(avg_host, db->avg, (size_t)db->n * db->S * 8, hipMemcpyDeviceToHost, db->stream));
    if (ne_host) VQ_HIP(hipMemcpyAsync(ne_host, db->ne, (size_t)db->n * db->S * 4, hipMemcpyDeviceToHost, db->stream));
    if (sims_host)
        VQ_HIP(hipMemcpyAsync(sims_host, db->sims, (size_t)db->n * db->S * db->E * 8, hipMemcpyDeviceToHost, db->stream));
    VQ_HIP(hipStreamSynchronize(db->stream));
    return VQ_OK;
}

int vq_db_read_scores(vq_db* db, double* scores_host) {
    VQ_REQUIRE(db && scores_host, "NULL argument");
    std::lock_guard<std::mutex> lk(db->mu);
    if (!db->have_scores) return fail(VQ_E_STATE, "no scores computed (scan with weights, or rescore)");
    DeviceGuard g(db->device);
    VQ_HIP(hipMemcpyAsync(scores_host, db->scores, (size_t)db->n * 8, hipMemcpyDeviceToHost, db->stream));
    VQ_HIP(hipStreamSynchronize(db->stream));
    return VQ_OK;
}

int vq_db_scores_devptr(vq_db* db, void** p) {
    VQ_REQUIRE(db && p, "NULL argument");
    *p = db->scores;
    return VQ_OK;
}
int vq_db_avg_devptr(vq_db* db, void** p) {
    VQ_REQUIRE(db && p, "NULL argument");
    *p = db->avg;
    return VQ_OK;
}
int vq_db_ne_devptr(vq_db* db, void** p) {
    VQ_REQUIRE(db && p, "NULL argument");
    *p = db->ne;
    return VQ_OK;
}

int vq_db_read_scores_at(vq_db* db, const int64_t* rows_host, int32_t L, double* out_host) {
    VQ_REQUIRE(db && out_host, "NULL argument");
    VQ_REQUIRE(L >= 0 && (L == 0 || rows_host), "bad rows");
    for (int l = 0; l < L; ++l)
        VQ_REQUIRE(rows_host[l] >= 0 && rows_host[l] < db->n, "rows[%d] = %lld outside [0,%lld)", l, (long long)rows_host[l], (long long)db->n);
    if (L == 0) return VQ_OK;
    std::lock_guard<std::mutex> lk(db->mu);
    if (!db->have_scores) return fail(VQ_E_STATE, "no scores computed (scan with weights, or rescore)");
    DeviceGuard g(db->device);
    int rc = ensure_grid_buf(db, (int64_t)L * 16);
    if (rc != VQ_OK) return rc;
    int64_t* rdev = (int64_t*)db->grid_buf;
    double* vdev = (double*)((char*)db->grid_buf + (int64_t)L * 8);
    VQ_HIP(hipMemcpyAsync(rdev, rows_host, (size_t)L * 8, hipMemcpyHostToDevice, db->stream));
    gather_scores_kernel<<<cdiv(L, 256), 256, 0, db->stream>>>(db->scores, rdev, L, vdev);
    VQ_CHECK_LAUNCH();
    VQ_HIP(hipMemcpyAsync(out_host, vdev, (size_t)L * 8, hipMemcpyDeviceToHost, db->stream));
    VQ_HIP(hipStreamSynchronize(db->stream));
    return VQ_OK;
}

int vq_db_read_rows(vq_db* db, const int64_t* rows_host, int32_t L, void* out_host) {
    VQ_REQUIRE(db && out_host, "NULL argument");
    VQ_REQUIRE(L >= 0 && (L == 0 || rows_host), "bad rows");
    for (int l = 0; l < L; ++l)
        VQ_REQUIRE(rows_host[l] >= 0 && rows_host[l] < db->n, "rows[%d] = %lld outside [0,%lld)", l, (long long)rows_host[l], (long long)db->n);
    if (L == 0) return VQ_OK;
    std::lock_guard<std::mutex> lk(db->mu);
    DeviceGuard g(db->device);
    const void* dense = nullptr;
    const int rc = gather_rows_device(db, rows_host, L, &dense);
    if (rc != VQ_OK) return rc;
    VQ_HIP(hipMemcpyAsync(out_host, dense, (size_t)L * db->S * db->E * db->D * db->elem(), hipMemcpyDeviceToHost, db->stream));
    VQ_HIP(hipStreamSynchronize(db->stream));
    return VQ_OK;
}

int vq_db_layout(vq_db* db, int32_t* layout) {
    VQ_REQUIRE(db && layout, "NULL argument");
    std::lock_guard<std::mutex> lk(db->mu);
    *layout = db->layout;
    return VQ_OK;
}

int vq_db_set_layout(vq_db* db, int32_t layout) {
    VQ_REQUIRE(db, "db is NULL");
    VQ_REQUIRE(layout == VQ_LAYOUT_ROWS || layout == VQ_LAYOUT_TILED, "layout must be VQ_LAYOUT_ROWS or VQ_LAYOUT_TILED");
    std::lock_guard<std::mutex> lk(db->mu);
    if (layout == db->layout) return VQ_OK;
    if (layout == VQ_LAYOUT_TILED) {
        if (db->dtype != VQ_F32 || !fast_scan_shape(db))
            return fail(VQ_E_UNSUPPORTED, "the tiled layout exists for fp32 databases with D = 1024, S <= 2, E <= 5 (got dtype %d, D %d, S %d, E %d)", db->dtype,
                        db->D, db->S, db->E);
        if (!db->owns_feats || db->feats_exposed)
            return fail(VQ_E_STATE, "the feature block is not the library's alone (adopted, or its address was handed out): it stays row-major");
    }
    DeviceGuard g(db->device);
    return convert_layout(db, layout);
}

int vq_db_write_avg(vq_db* db, const double* avg_host, const int32_t* ne_host) {
    VQ_REQUIRE(db && avg_host, "NULL argument");
    std::lock_guard<std::mutex> lk(db->mu);
    DeviceGuard g(db->device);
    VQ_HIP(hipMemcpyAsync(db->avg, avg_host, (size_t)db->n * db->S * 8, hipMemcpyHostToDevice, db->stream));
    if (ne_host) VQ_HIP(hipMemcpyAsync(db->ne, ne_host, (size_t)db->n * db->S * 4, hipMemcpyHostToDevice, db->stream));
    VQ_HIP(hipStreamSynchronize(db->stream));
    db->have_avg = true;
    db->have_scores = false;
    return VQ_OK;
}

}  // extern "C"
static int ensure_grid_buf(vq_db* db, int64_t bytes) {
    if (db->grid_cap >= bytes) return VQ_OK;
    if (db->grid_buf) VQ_HIP(hipFree(db->grid_buf));
    db->grid_buf = nullptr;
    db->grid_cap = 0;
    VQ_HIP(vq::malloc_trim((void**)&db->grid_buf, bytes));
    db->grid_cap = bytes;
    return VQ_OK;
}
extern "C" {

int vq_db_scores_grid(vq_db* db, const double* w_grid_host, int32_t G, const int64_t* rows_host, int32_t L,
                      double* out_host) {
    VQ_REQUIRE(db && w_grid_host && rows_host && out_host, "NULL argument");
    VQ_REQUIRE(G > 0 && L > 0, "G and L must be positive");
    for (int l = 0; l < L; ++l)
        VQ_REQUIRE(rows_host[l] >= 0 && rows_host[l] < db->n, "rows[%d] = %lld outside [0,%lld)", l,
                   (long long)rows_host[l], (long long)db->n);
    std::lock_guard<std::mutex> lk(db->mu);
    if (!db->have_avg) return fail(VQ_E_STATE, "no similarities cached (call vq_db_scan first)");
    DeviceGuard g(db->device);
    const int64_t wb = (int64_t)G * db->S * 8, rb = (int64_t)L * 8, ob = (int64_t)G * L * 8;
    int rc = ensure_grid_buf(db, wb + rb + ob);
    if (rc != VQ_OK) return rc;
    char* base = (char*)db->grid_buf;
    VQ_HIP(hipMemcpyAsync(base, w_grid_host, wb, hipMemcpyHostToDevice, db->stream));
    VQ_HIP(hipMemcpyAsync(base + wb, rows_host, rb, hipMemcpyHostToDevice, db->stream));
    grid_kernel<<<cdiv((int64_t)G * L, 256), 256, 0, db->stream>>>(db->avg, (const double*)base, (const int64_t*)(base + wb),
                                                                    (double*)(base + wb + rb), G, L, db->S);
    VQ_CHECK_LAUNCH();
    VQ_HIP(hipMemcpyAsync(out_host, base + wb + rb, ob, hipMemcpyDeviceToHost, db->stream));
    VQ_HIP(hipStreamSynchronize(db->stream));
    return VQ_OK;
}

int vq_db_loss_surface(vq_db* db, const double* w_grid_host, int32_t G, const int64_t* rows_host, const double* labels_host, int32_t L,
                       const double* th_grid_host, int32_t T, double ballast, double* out_host) {
    VQ_REQUIRE(db && w_grid_host && rows_host && labels_host && th_grid_host && out_host, "NULL argument");
    VQ_REQUIRE(G > 0 && L > 0 && T > 0 && L <= 4096, "G, L, T must be positive (L <= 4096 labelled clips)");
    for (int l = 0; l < L; ++l)
        VQ_REQUIRE(rows_host[l] >= 0 && rows_host[l] < db->n, "rows[%d] = %lld outside [0,%lld)", l, (long long)rows_host[l], (long long)db->n);
    std::lock_guard<std::mutex> lk(db->mu);
    if (!db->have_avg) return fail(VQ_E_STATE, "no similarities cached (call vq_db_scan first)");
    DeviceGuard g(db->device);
    // one staging block: weights | rows | labels | thresholds in, the surface out
    const int64_t wb = (int64_t)G * db->S * 8, rb = (int64_t)L * 8, tb = (int64_t)T * 8, ob = (int64_t)G * T * 8;
    int rc = ensure_grid_buf(db, wb + 2 * rb + tb + ob);
    if (rc != VQ_OK) return rc;
    std::vector<char> stage((size_t)(wb + 2 * rb + tb));
    memcpy(stage.data(), w_grid_host, (size_t)wb);
    memcpy(stage.data() + wb, rows_host, (size_t)rb);
    memcpy(stage.data() + wb + rb, labels_host, (size_t)rb);
    memcpy(stage.data() + wb + 2 * rb, th_grid_host, (size_t)tb);
    char* base = (char*)db->grid_buf;
    VQ_HIP(hipMemcpyAsync(base, stage.data(), stage.size(), hipMemcpyHostToDevice, db->stream));      // pageable: staged before the call returns
    loss_surface_kernel<<<G, 64, (size_t)L * 8, db->stream>>>(db->avg, (const double*)base, (const int64_t*)(base + wb), (const double*)(base + wb + rb),
                                                               (const double*)(base + wb + 2 * rb), (double*)(base + wb + 2 * rb + tb), L, T, db->S, ballast);
    VQ_CHECK_LAUNCH();
    VQ_HIP(hipMemcpyAsync(out_host, base + wb + 2 * rb + tb, ob, hipMemcpyDeviceToHost, db->stream));
    VQ_HIP(hipStreamSynchronize(db->stream));
    return VQ_OK;
}

int vq_db_select(vq_db* db, double threshold, double lower, int64_t* n_match, int64_t* n_near, int64_t* near_argmax) {
    VQ_REQUIRE(db, "db is NULL");
    std::lock_guard<std::mutex> lk(db->mu);
    if (!db->have_scores) return fail(VQ_E_STATE, "no scores computed (scan with weights, or rescore)");
    DeviceGuard g(db->device);
    SelPred pred;
    pred.mode = 0;
    pred.th = threshold;
    pred.lower = lower;
    pred.pivot = 0;
    int64_t res[3];
    const int rc = run_select(db, pred, -1, res);
    if (rc != VQ_OK) return rc;
    db->last_n0 = res[0];
    db->last_n1 = res[1];
    if (n_match) *n_match = res[0];
    if (n_near) *n_near = res[1];
    if (near_argmax) *near_argmax = res[2];
    return VQ_OK;
}

static int fetch_selected(vq_db* db, int64_t* match_rows_host, int64_t cap_match, int64_t* near_rows_host, int64_t cap_near);

int vq_db_select_fetch(vq_db* db, int64_t* match_rows_host, int64_t cap_match, int64_t* near_rows_host, int64_t cap_near) {
    VQ_REQUIRE(db, "db is NULL");
    std::lock_guard<std::mutex> lk(db->mu);
    return fetch_selected(db, match_rows_host, cap_match, near_rows_host, cap_near);
}

int vq_db_select_rows(vq_db* db, double threshold, double lower, int64_t* match_rows_host, int64_t cap_match,
                      int64_t* near_rows_host, int64_t cap_near, int64_t* n_match, int64_t* n_near, int64_t* near_argmax) {
    VQ_REQUIRE(db && match_rows_host && near_rows_host && n_match && n_near, "NULL argument");
    std::lock_guard<std::mutex> lk(db->mu);          // partition and copy-out under ONE lock: no other thread's
    if (!db->have_scores) return fail(VQ_E_STATE, "no scores computed (scan with weights, or rescore)");
    DeviceGuard g(db->device);                       // selection or top-k can slip in between
    SelPred pred;
    pred.mode = 0;
    pred.th = threshold;
    pred.lower = lower;
    pred.pivot = 0;
    int64_t res[3];
    const int rc = run_select(db, pred, -1, res);
    if (rc != VQ_OK) return rc;
    db->last_n0 = res[0];
    db->last_n1 = res[1];
    *n_match = res[0];
    *n_near = res[1];
    if (near_argmax) *near_argmax = res[2];
    return fetch_selected(db, match_rows_host, cap_match, near_rows_host, cap_near);
}

int vq_host_alloc(void** ptr, int64_t bytes) {
    VQ_REQUIRE(ptr && bytes > 0, "vq_host_alloc: bad argument");
    *ptr = nullptr;
    VQ_HIP(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
    return VQ_OK;
}

int vq_host_free(void* ptr) {
    if (ptr) VQ_HIP(hipHostFree(ptr));
    return VQ_OK;
}

int vq_db_round_layout(vq_db* db, int64_t off[10]) {
    VQ_REQUIRE(db && off, "NULL argument");
    auto up = [](int64_t v) { return (v + 63) / 64 * 64; };
    const int64_t head = up((int64_t)db->S * db->E * db->D * 8) + 64;         // query | weights [8]
    off[0] = 0;
    off[1] = head - 64;
    for (int i = 0; i < 6; ++i) off[2 + i] = head + db->round_off[i];
    off[8] = head + db->round_off[6];
    off[9] = kRoundPrefix;
    return VQ_OK;
}

int vq_db_query_round(vq_db* db, void* block, int64_t block_bytes, int32_t flags, double threshold, double lower) {
    VQ_REQUIRE(db && block, "NULL argument");
    VQ_REQUIRE((flags & ~7) == 0 && flags != 0, "flags: 1 = scan, 2 = scores under the block's weights, 4 = select");
    const bool do_scan = flags & 1, do_scores = flags & 2, do_select = flags & 4;
    VQ_REQUIRE(!do_select || do_scores, "a selection needs the scores of the same call");
    int64_t off[10];
    vq_db_round_layout(db, off);
    VQ_REQUIRE(block_bytes >= off[8], "the round block needs %lld bytes (vq_db_round_layout), got %lld", (long long)off[8], (long long)block_bytes);
    std::lock_guard<std::mutex> lk(db->mu);
    if (!do_scan && !db->have_avg) return fail(VQ_E_STATE, "vq_db_query_round: no similarities cached (scan first, or set flag 1)");
    DeviceGuard g(db->device);
    char* host = (char*)block;
    if (do_scan) {                                   // query and weights in ONE copy (adjacent on both sides)
        VQ_HIP(hipMemcpyAsync(db->t, host + off[0], (size_t)(off[1] - off[0]) + (do_scores ? db->S * 8 : 0), hipMemcpyHostToDevice, db->stream));
        db->have_query = true;
        db->have_avg = db->have_scores = db->have_sims = false;
    } else if (do_scores) {
        VQ_HIP(hipMemcpyAsync(db->w, host + off[1], db->S * 8, hipMemcpyHostToDevice, db->stream));
    }
    if (do_scan) {
        ScanArgs a;
        a.feats = db->feats;
        a.t = db->t;
        a.present = db->present;
        a.w = do_scores ? db->w : nullptr;
        a.sims = nullptr;
        a.avg = db->avg;
        a.ne = db->ne;
        a.scores = db->scores;
        a.n = db->n;
        a.S = db->S;
        a.E = db->E;
        a.D = db->D;
        const int rc = db->dtype == VQ_F32 ? launch_scan<float>(db, a) : launch_scan<double>(db, a);
        if (rc != VQ_OK) return rc;
        db->have_avg = true;
        db->have_scores = do_scores;
    } else if (do_scores) {
        rescore_kernel<<<cdiv(db->n, 256), 256, 0, db->stream>>>(db->avg, db->w, db->scores, db->n, db->S);
        VQ_CHECK_LAUNCH();
        db->have_scores = true;
    }
    if (do_select) {
        SelPred pred;
        pred.mode = 0;
        pred.th = threshold;
        pred.lower = lower;
        pred.pivot = 0;
        const int rc = run_select(db, pred, -1, nullptr, true);
        if (rc != VQ_OK) return rc;
    }
    // one copy back: from the similarities (a scan) or from the scores (a re-weighting) to the end of what was computed
    const int first = do_scan ? 0 : 2;
    const int64_t end = do_select ? db->round_off[6] : (do_scores ? db->round_off[3] : db->round_off[2]);
    VQ_HIP(hipMemcpyAsync(host + off[2] + db->round_off[first], db->round_dev + db->round_off[first], (size_t)(end - db->round_off[first]),
                          hipMemcpyDeviceToHost, db->stream));
    VQ_HIP(hipStreamSynchronize(db->stream));
    if (do_select) {
        const int64_t* res = (const int64_t*)(host + off[5]);
        db->last_n0 = res[0];
        db->last_n1 = res[1];
    }
    return VQ_OK;
}

static int fetch_selected(vq_db* db, int64_t* match_rows_host, int64_t cap_match, int64_t* near_rows_host, int64_t cap_near) {
    if (db->last_n0 < 0 || db->last_n1 < 0)
        return fail(VQ_E_STATE, "the selection lists were overwritten by a later top-k on this handle (select again)");
    DeviceGuard g(db->device);
    if (match_rows_host) {
        VQ_REQUIRE(cap_match >= db->last_n0, "match buffer too small (%lld < %lld)", (long long)cap_match, (long long)db->last_n0);
        if (db->last_n0) VQ_HIP(hipMemcpyAsync(match_rows_host, db->rows0, db->last_n0 * 8, hipMemcpyDeviceToHost, db->stream));
    }
    if (near_rows_host) {
        VQ_REQUIRE(cap_near >= db->last_n1, "near buffer too small (%lld < %lld)", (long long)cap_near, (long long)db->last_n1);
        if (db->last_n1) VQ_HIP(hipMemcpyAsync(near_rows_host, db->rows1, db->last_n1 * 8, hipMemcpyDeviceToHost, db->stream));
    }
    VQ_HIP(hipStreamSynchronize(db->stream));
    return VQ_OK;
}

int vq_db_topk(vq_db* db, int64_t k, int64_t* rows_host, double* vals_host, int64_t* k_out) {
    VQ_REQUIRE(db && rows_host && vals_host && k_out, "NULL argument");
    VQ_REQUIRE(k > 0, "k must be positive");
    std::lock_guard<std::mutex> lk(db->mu);
    if (!db->have_scores) return fail(VQ_E_STATE, "no scores computed (scan with weights, or rescore)");
    DeviceGuard g(db->device);
    if (k > db->n) k = db->n;
    if (db->n <= kTopkSmallN && k <= kTopkSmallK) {                  // a small resident database: one launch, one copy back
        const int64_t bytes = 64 + k * 16;
        int rc = ensure_grid_buf(db, bytes);
        if (rc != VQ_OK) return rc;
        char* base = (char*)db->grid_buf;
        auto kern = topk_small_kernel;
        const size_t lds = (size_t)db->n * 8;
        VQ_DYN_LDS(kern, (size_t)kTopkSmallN * 8);
        kern<<<1, kTopkSmallThreads, lds, db->stream>>>(db->scores, (int)db->n, (int)k, (int64_t*)(base + 64), (double*)(base + 64 + k * 8), (int64_t*)base);
        VQ_CHECK_LAUNCH();
        std::vector<char> host((size_t)bytes);
        VQ_HIP(hipMemcpyAsync(host.data(), base, (size_t)bytes, hipMemcpyDeviceToHost, db->stream));
        VQ_HIP(hipStreamSynchronize(db->stream));
        const int64_t tot = *(const int64_t*)host.data();
        VQ_REQUIRE(tot >= 0 && tot <= k, "internal: top-k produced %lld > k=%lld rows", (long long)tot, (long long)k);
        const int64_t* rows = (const int64_t*)(host.data() + 64);
        const double* vals = (const double*)(host.data() + 64 + k * 8);
        std::vector<int64_t> order((size_t)tot);
        for (int64_t i = 0; i < tot; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) {
            if (vals[x] != vals[y]) return vals[x] > vals[y];
            return rows[x] < rows[y];
        });
        for (int64_t i = 0; i < tot; ++i) {
            rows_host[i] = rows[order[i]];
            vals_host[i] = vals[order[i]];
        }
        *k_out = tot;
        return VQ_OK;
    }
    // radix select: find the key of the k-th largest score
    uint64_t st[2] = {0ull, (uint64_t)k};
    VQ_HIP(hipMemcpyAsync(db->tk_state, st, 16, hipMemcpyHostToDevice, db->stream));
    VQ_HIP(hipMemsetAsync(db->tk_hist, 0, 256 * 4, db->stream));
    const unsigned hgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((db->n + 255) / 256, (int64_t)db->cus * 8));
    for (int pass = 0; pass < 8; ++pass) {
        topk_hist_kernel<<<hgrid, 256, 0, db->stream>>>(db->scores, db->n, pass, db->tk_state, db->tk_hist);
        VQ_CHECK_LAUNCH();
        topk_pick_kernel<<<1, 64, 0, db->stream>>>(pass, db->tk_state, db->tk_hist);
        VQ_CHECK_LAUNCH();
    }
    VQ_HIP(hipMemcpyAsync(st, db->tk_state, 16, hipMemcpyDeviceToHost, db->stream));
    VQ_HIP(hipStreamSynchronize(db->stream));
    // st[0] = key of the k-th largest (0 if fewer than k non-NaN scores), st[1] = how many equal to it are needed
    SelPred pred;
    pred.mode = 1;
    pred.th = pred.lower = 0.0;
    pred.pivot = st[0];
    int64_t res[3];
    db->last_n0 = db->last_n1 = -1;      // the shared row lists no longer hold a vq_db_select partition
    int rc = run_select(db, pred, (int64_t)st[1], res);
    if (rc != VQ_OK) return rc;
    const int64_t n_gt = res[0], n_eq = res[1];
    const int64_t tot = n_gt + n_eq;
    VQ_REQUIRE(tot <= k, "internal: top-k produced %lld > k=%lld rows", (long long)tot, (long long)k);
    rc = ensure_grid_buf(db, std::max<int64_t>(tot, 1) * 8);
    if (rc != VQ_OK) return rc;
    std::vector<int64_t> rows((size_t)tot);
    std::vector<double> vals((size_t)tot);
    if (n_gt) VQ_HIP(hipMemcpyAsync(rows.data(), db->rows0, n_gt * 8, hipMemcpyDeviceToHost, db->stream));
    if (n_eq) VQ_HIP(hipMemcpyAsync(rows.data() + n_gt, db->rows1, n_eq * 8, hipMemcpyDeviceToHost, db->stream));
    if (n_gt) {
        gather_scores_kernel<<<cdiv(n_gt, 256), 256, 0, db->stream>>>(db->scores, db->rows0, n_gt, (double*)db->grid_buf);
        VQ_CHECK_LAUNCH();
        VQ_HIP(hipMemcpyAsync(vals.data(), db->grid_buf, n_gt * 8, hipMemcpyDeviceToHost, db->stream));
    }
    VQ_HIP(hipStreamSynchronize(db->stream));
    if (n_eq) {
        gather_scores_kernel<<<cdiv(n_eq, 256), 256, 0, db->stream>>>(db->scores, db->rows1, n_eq, (double*)db->grid_buf);
        VQ_CHECK_LAUNCH();
        VQ_HIP(hipMemcpyAsync(vals.data() + n_gt, db->grid_buf, n_eq * 8, hipMemcpyDeviceToHost, db->stream));
        VQ_HIP(hipStreamSynchronize(db->stream));
    }
    // final ordering of the k survivors: descending score, ties by ascending row (ticket.py:266)
    std::vector<int64_t> order((size_t)tot);
    for (int64_t i = 0; i < tot; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) {
        if (vals[x] != vals[y]) return vals[x] > vals[y];
        return rows[x] < rows[y];
    });
    for (int64_t i = 0; i < tot; ++i) {
        rows_host[i] = rows[order[i]];
        vals_host[i] = vals[order[i]];
    }
    *k_out = tot;
    return VQ_OK;
}

int vq_db_min_score(vq_db* db, const int64_t* rows_host, int32_t L, double* min_out) {
    VQ_REQUIRE(db && min_out, "NULL argument");
    VQ_REQUIRE(L >= 0 && (L == 0 || rows_host), "bad rows");
    for (int l = 0; l < L; ++l)
        VQ_REQUIRE(rows_host[l] >= 0 && rows_host[l] < db->n, "rows[%d] outside the DB", l);
    std::lock_guard<std::mutex> lk(db->mu);
    if (!db->have_scores) return fail(VQ_E_STATE, "no scores computed");
    DeviceGuard g(db->device);
    double m = 1.0;   // ticket.py:302 min_score = 1
    if (L > 0) {
        int rc = ensure_grid_buf(db, (int64_t)L * 16);
        if (rc != VQ_OK) return rc;
        int64_t* rdev = (int64_t*)db->grid_buf;
        double* vdev = (double*)((char*)db->grid_buf + (int64_t)L * 8);
        VQ_HIP(hipMemcpyAsync(rdev, rows_host, (size_t)L * 8, hipMemcpyHostToDevice, db->stream));
        gather_scores_kernel<<<cdiv(L, 256), 256, 0, db->stream>>>(db->scores, rdev, L, vdev);
        VQ_CHECK_LAUNCH();
        std::vector<double> v((size_t)L);
        VQ_HIP(hipMemcpyAsync(v.data(), vdev, (size_t)L * 8, hipMemcpyDeviceToHost, db->stream));
        VQ_HIP(hipStreamSynchronize(db->stream));
        for (int l = 0; l < L; ++l) m = std::min(m, v[l]);   // Python min(a, b): b if b < a else a
    }
    *min_out = m;
    return VQ_OK;
}

}  // extern "C"
