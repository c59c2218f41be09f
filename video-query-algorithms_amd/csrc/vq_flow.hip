// TV-L1 optical flow on gfx950: the arithmetic behind the reference's flow frames.
//
// What it replaces (paths relative to the reference checkout):
//   src/features_GPU_compute/build_wof_clips.py:55-76   run_warp_optical_flow(): os.system("<TSN_ROOT>/lib/dense_flow/build/
//       extract_warp_gpu -f <video> -x flow_x -y flow_y -b 20 -t 1 -d <gpu> -s 1 -o dir") -- a third-party binary (OpenCV CUDA
//       TV-L1; not in the reference tree, no pinned version).  "-t 1" = TV-L1, "-b 20" = clamp to +-20 px and quantise to 8 bits.
// PARITY UNPINNED: the reference holds neither frames nor flow images nor the binary.  The kernels follow oracle/
// tvl1_oracle.py -- the PUBLISHED algorithm (Zach, Pock & Bischof 2007 as formulated in IPOL 2013, Algorithm 1) with OpenCV's
// default parameters and interpolation choices -- operation for operation in fp32 (contraction off, correctly rounded
// division and square root), so device and oracle agree to rounding.  dense_flow's camera-motion "warp": vq_flow_tvl1 applies a
// given homography to the second frame; vq_flow_good_features + vq_flow_ransac_homography (end of this file, oracle/
// warp_oracle.py) estimate it from corners moved by the first-pass flow.  The SURF matches the binary adds are not built.
//
// Shape of the work: a BATCH of independent frame pairs (one 340 x 256 pair is only 87 k pixels).  Per pyramid level and warp:
// one warp kernel, then the inner iterations in blocks of kBlkIters per launch on tiles whose fields stay in registers / LDS
// (tvl1_tile_kernel below; rounds 1-2 streamed every plane through the caches twice per iteration with a primal and a dual launch,
// ~90 bytes per pixel and iteration, round 3 ran square 64 x 64 tiles on 1 024 threads: both were checked bit for bit against the
// present kernel until they were removed in round 5 -- git history).
// A pair that has converged (mean squared update <= epsilon^2, or the iteration cap) is switched off on the device and its
// workgroups exit at once; the host looks at the number of live pairs every few launches only.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "vq_common.h"
#include "host/vq_corners.h"

using namespace vq;

namespace {

constexpr float kGradIsZero = 1e-10f;

struct Level {
    int h, w;
    size_t off;          // float offset of this level inside a per-plane pyramid buffer (per pair: see plane())
};

// bilinear sample positions of cv::resize INTER_LINEAR (fp64 coordinate, fp32 weights -- oracle.resize_bilinear)
__global__ void resize_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int ih, int iw, int oh, int ow, float gain) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * oh * ow) return;
    const int x = (int)(i % ow), y = (int)((i / ow) % oh);
    const int64_t p = i / ((int64_t)oh * ow);
    double ys = ((double)y + 0.5) * (double)ih / (double)oh - 0.5, xs = ((double)x + 0.5) * (double)iw / (double)ow - 0.5;
    ys = fmin(fmax(ys, 0.0), (double)(ih - 1));
    xs = fmin(fmax(xs, 0.0), (double)(iw - 1));
    const int y0 = (int)floor(ys), x0 = (int)floor(xs);
    const int y1 = min(y0 + 1, ih - 1), x1 = min(x0 + 1, iw - 1);
    const float wy = (float)(ys - (double)y0), wx = (float)(xs - (double)x0);
    const float* a = src + p * (int64_t)ih * iw;
    const float top = a[y0 * iw + x0] * (1.0f - wx) + a[y0 * iw + x1] * wx;
    const float bot = a[y1 * iw + x0] * (1.0f - wx) + a[y1 * iw + x1] * wx;
    dst[i] = (top * (1.0f - wy) + bot * wy) * gain;
}

__global__ void u8_to_float_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) dst[i] = (float)src[i];
}

// centered gradient of I1: 0.5 (I[x+1] - I[x-1]), the missing neighbour at the border replaced by the pixel itself
__global__ void gradient_kernel(const float* __restrict__ img, float* __restrict__ gx, float* __restrict__ gy, int n, int h, int w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * h * w) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const float* a = img + (i - (int64_t)y * w - x);
    gx[i] = 0.5f * (a[y * w + min(x + 1, w - 1)] - a[y * w + max(x - 1, 0)]);
    gy[i] = 0.5f * (a[min(y + 1, h - 1) * w + x] - a[max(y - 1, 0) * w + x]);
}

__device__ __forceinline__ float sample_bilinear(const float* __restrict__ a, int h, int w, float xs, float ys) {
    xs = fminf(fmaxf(xs, 0.0f), (float)(w - 1));
    ys = fminf(fmaxf(ys, 0.0f), (float)(h - 1));
    const int x0 = (int)floorf(xs), y0 = (int)floorf(ys);
    const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    const float wx = xs - (float)x0, wy = ys - (float)y0;
    const float top = a[y0 * w + x0] * (1.0f - wx) + a[y0 * w + x1] * wx;
    const float bot = a[y1 * w + x0] * (1.0f - wx) + a[y1 * w + x1] * wx;
    return top * (1.0f - wy) + bot * wy;
}

#ifndef VQ_FLOW_BLOCK_ITERS
#define VQ_FLOW_BLOCK_ITERS 4
#endif
constexpr int kBlkIters = VQ_FLOW_BLOCK_ITERS;      // inner iterations per launch of the blocked form (= the halo of a tile)
#ifndef VQ_FLOW_TILE_THREADS
#define VQ_FLOW_TILE_THREADS 512
#endif
#ifndef VQ_FLOW_TILE_NC
#define VQ_FLOW_TILE_NC 4
#endif
#ifndef VQ_FLOW_TILE_WPE
#define VQ_FLOW_TILE_WPE 4
#endif
constexpr int kTileNC = VQ_FLOW_TILE_NC;          // cells per thread
constexpr int kTileThreads = VQ_FLOW_TILE_THREADS;  // tvl1_tile_kernel: threads of a workgroup; two workgroups share a compute unit (128 VGPRs, 48 KB of LDS each)
constexpr int kTileCells = kTileNC * kTileThreads;   // cells of a tile = floats of one of its six LDS planes
// The cut of a w x h level into nx x ny tiles of ceil(w / nx) x ceil(h / ny) own pixels that costs `pairs` pairs the least on `slots`
// workgroup slots (two per compute unit): a workgroup's time goes with its cells (halo included) in whole waves, a launch's with its rounds.
struct TileCut {
    int nx, ny, tw, th, ew, eh;
};
inline TileCut fit_tiles(int w, int h, int pairs, int slots) {
    TileCut best{0, 0, 0, 0, 0, 0};
    long long best_cost = -1;
    for (int nx = 1; nx <= (w + 7) / 8; ++nx) {
        const int tw = (w + nx - 1) / nx, ew = tw + 2 * kBlkIters;
        for (int ny = 1; ny <= (h + 7) / 8; ++ny) {
            const int th = (h + ny - 1) / ny, eh = th + 2 * kBlkIters;
            if ((long long)ew * eh > kTileCells) continue;
            const long long waves = ((long long)ew * eh + 63) / 64;
            const long long rounds = ((long long)nx * ny * pairs + slots - 1) / slots;
            const long long cost = rounds * waves;
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best = TileCut{nx, ny, tw, th, ew, eh};
            }
        }
    }
    return best;
}
struct BlkSched {                 // what a pair does in one launch of the blocked form
    int mode;                     // kBlkRun: a block of n iterations; kBlkReplay: the exact n iterations of a block that ran past the stop; kBlkDone
    int src;                      // the set of planes the launch reads (it writes the other one)
    int base;                     // inner iterations completed before this launch
    int n;                        // iterations this launch runs
};
constexpr int kBlkRun = 0, kBlkReplay = 1, kBlkDone = 2;
struct alignas(128) PairState {   // own cache lines per pair: the error sums of different pairs never contend for a line
    double err[3][kBlkIters];     // sums of squared primal updates; slot L % 3 belongs to launch L (the two-launch form uses err[0][0])
    BlkSched blk[3];              // slot L % 3: the schedule of launch L (written by its first thread, read by launch L + 1)
    int stop_iter;                // two-launch form: iterations >= stop_iter of the current warp do not run
    int iters;                    // inner iterations run in the current warp
    int final_set;                // the set of planes that holds the pair's fields when the warp's loop is over
};
constexpr int kNoStop = 0x7FFFFFFF;

// Start of a warp: I1 and its gradient sampled at x + u, |grad|^2, the constant part of rho; the pair becomes active.
__global__ void tvl1_warp_kernel(const float* __restrict__ i0, const float* __restrict__ i1, const float* __restrict__ i1x,
                                 const float* __restrict__ i1y, const float* __restrict__ u1, const float* __restrict__ u2,
                                 float* __restrict__ i1wx, float* __restrict__ i1wy, float* __restrict__ grad, float* __restrict__ rho_c,
                                 PairState* __restrict__ st, int* __restrict__ n_active, int* __restrict__ live_flag, int n, int h, int w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * h * w) return;
    if (i == 0) {                    // every pair starts the warp's inner loop live
        *n_active = n;
        __hip_atomic_store(live_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const int64_t p = i / ((int64_t)h * w), base = p * (int64_t)h * w;
    const float a = u1[i], b = u2[i];
    const float xs = (float)x + a, ys = (float)y + b;
    const float w0 = sample_bilinear(i1 + base, h, w, xs, ys);
    const float wx = sample_bilinear(i1x + base, h, w, xs, ys);
    const float wy = sample_bilinear(i1y + base, h, w, xs, ys);
    i1wx[i] = wx;
    i1wy[i] = wy;
    grad[i] = wx * wx + wy * wy;
    rho_c[i] = w0 - wx * a - wy * b - i0[i];
    if (x == 0 && y == 0) {
        for (int q = 0; q < 3; ++q) {
            for (int m = 0; m < kBlkIters; ++m) st[p].err[q][m] = 0.0;
            st[p].blk[q] = BlkSched{kBlkRun, 0, 0, 0};
        }
        st[p].final_set = 0;
        st[p].stop_iter = kNoStop;
        st[p].iters = 0;
    }
}

// End of a warp: the inner iterations each pair ran go to the log (read by the host once, after the last level).
__global__ void log_iters_kernel(const PairState* __restrict__ st, int* __restrict__ log, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) log[p] = st[p].iters;
}

// The divisions and square roots of an inner iteration (one quotient in the primal step, four quotients over two denominators and two
// roots in the dual step) are ~70 of its 273 vector instructions when IEEE-rounded, and the kernel is bound by vector-ALU issue
// (0.42 of the rate).  FAST (VQ_FLOW_FAST=1 at creation; OFF by default) = the hardware's own v_rcp_f32 / v_sqrt_f32 (1 ulp) and one
// reciprocal per denominator: 23.1 -> 17.9 ms of inner loops per batch of 64 pairs, +26 % pairs/s plain, +21 % warped (tools/flow_exact_ab.py).
// It is an option and not the default because TV-L1 does not forgive an ulp: with ANY of the three substitutions alone the 8-bit flow
// images equal the IEEE form's on 99.6 % of the pixels and differ by up to 15 px on the rest (threshold decisions flip where the flow is
// not determined: occluded margins, flat regions), which is outside the tolerances this file is tested to against oracle/tvl1_oracle.py
// (1e-4 px at a fixed iteration count).  The default keeps the IEEE operations, operation for operation with the oracle.
template <bool FAST>
__device__ __forceinline__ float tv_rcp(float x) {
    return FAST ? __builtin_amdgcn_rcpf(x) : 1.0f / x;
}
template <bool FAST>
__device__ __forceinline__ float tv_sqrt(float x) {
    return FAST ? __builtin_amdgcn_sqrtf(x) : sqrtf(x);
}
// p <- (p + taut * grad u) / (1 + taut * |grad u|) for the two components of one flow field
template <bool FAST>
__device__ __forceinline__ void dual_pair(float& pa, float& pb, float ux, float uy, float taut) {
    const float ng = 1.0f + taut * tv_sqrt<FAST>(ux * ux + uy * uy);
    if (FAST) {
        const float inv = __builtin_amdgcn_rcpf(ng);
        pa = (pa + taut * ux) * inv;
        pb = (pb + taut * uy) * inv;
    } else {
        pa = (pa + taut * ux) / ng;
        pb = (pb + taut * uy) / ng;
    }
}

// ---- two cells per instruction ---------------------------------------------------------------------------------------------------
// gfx950 executes v_pk_{add,mul,fma}_f32 -- two fp32 operations per lane -- at the rate of the one-operation forms, and the iteration is
// bound by the vector ALU's issue rate.  A thread's cells therefore go through the iteration in PAIRS: every addition / multiplication of the
// primal and the dual step on a float2, no branch around a dead cell or around the three cases of the thresholding step (selects instead),
// and the correctly rounded division written out -- v_div_scale, v_rcp, the six-operation refinement, v_div_fmas, v_div_fixup: exactly the
// sequence the compiler emits for `a / b` (AMDGPUTargetLowering::LowerFDIV32) -- so that its refinement runs packed as well: 16 instructions
// per two quotients instead of 22.  Per cell the operations, their operands and their order are unchanged: same bits as the cell-by-cell
// form (tools/flow_bits_ab.py compares the two libraries).  The square root is the library's expansion without its scaling and class steps where they cannot act
// (vq_flow_math.h).
#include "vq_flow_math.h"

// primal_pixel on two cells (same operations per cell)
__device__ __forceinline__ void primal_pair(f2 ux, f2 uy, f2 gx, f2 gy, f2 gr, f2 rc, f2 div1, f2 div2, float l_t, float theta, f2& n1, f2& n2,
                                            f2& err) {
    const f2 rho = rc + (gx * ux + gy * uy);
    const f2 lg = l_t * gr;                                   // -l_t * gr = -(l_t * gr), bit for bit
    const f2 fi = div_ieee(-rho, gr);                         // used where |rho| <= l_t * gr and gr > kGradIsZero only (0 / 0 elsewhere: dropped)
    f2 s;
    bool any[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const bool below = rho[e] < -lg[e], above = rho[e] > lg[e];
        s[e] = below ? l_t : (above ? -l_t : fi[e]);
        any[e] = below || above || gr[e] > kGradIsZero;
    }
    f2 d1 = s * gx, d2 = s * gy;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        d1[e] = any[e] ? d1[e] : 0.0f;
        d2[e] = any[e] ? d2[e] : 0.0f;
    }
    n1 = (ux + d1) + theta * div1;
    n2 = (uy + d2) + theta * div2;
    const f2 a1 = n1 - ux, a2 = n2 - uy;
    err = a1 * a1 + a2 * a2;
}

// dual_pair (exact form) on two cells
__device__ __forceinline__ void dual_pair2(f2& pa, f2& pb, f2 ux, f2 uy, float taut) {
    const f2 sq = ux * ux + uy * uy;
    const f2 rt = sqrt_ieee(sq);
    const f2 ng = 1.0f + taut * rt;
    pa = div_ieee(pa + taut * ux, ng);
    pb = div_ieee(pb + taut * uy, ng);
}

// ---- the blocked form: kBlkIters inner iterations per launch, the fields of a tile resident in LDS ----------------------------
// The two-launch form streams every plane through the caches twice per inner iteration (22 floats per pixel) and needs two
// dependent launches for it; on the coarse levels a launch is a few microseconds of work.  Here a workgroup loads a tile of
// E x E pixels -- T x T of its own plus a halo of K = kBlkIters on every side -- ONCE: u1, u2, p11..p22 into LDS, the four constant
// planes into registers (a thread owns fixed cells), runs n <= K iterations on it (primal step in place, barrier, dual step in
// place, barrier: the primal step reads the dual variables of the left / upper neighbour, the dual step the new primal values of
// the right / lower one, so the region whose values are exact shrinks by one pixel per iteration and side and is the tile itself
// after K) and writes its T x T pixels to the OTHER set of planes.  Per pixel the operations and their order are those of the
// kernels above: same bits.  Traffic per pixel and iteration: (10 x (E/T)^2 + 6) / K floats (4.8 at E = 64) instead of 22.  What the
// kernel is bound by after that is arithmetic: five correctly rounded divisions and two square roots per pixel and iteration.
// Shapes measured on 64 pairs of 340 x 256 (two-launch form: 39.2 ms per batch): E = 56 with 256 threads and every field in LDS
// 41.6 ms; 512 threads 41.5; own cells in registers, LDS for the neighbours' values only 36.5; the same without branches around
// dead cells and with the thresholding as selects 39.1 (the division then runs for every cell); forced to 128 VGPRs for two
// workgroups per CU 49.0 (spills); E = 64 with 1024 threads (four cells per thread, no idle slots, 16 waves per CU) 29.9.
//
// The stopping rule stays EXACT (iteration k + 1 runs iff the mean squared update of iteration k exceeds epsilon^2): a block runs
// its iterations speculatively and records every iteration's sum; the next launch reads them.  If the rule stopped inside the
// block -- at its iteration j, not the last -- the block's output is too far: the set it READ is still intact (it wrote the
// other one), so the pair runs exactly j + 1 iterations from it again ("replay": once per pair and warp) and is done.  Every
// workgroup of a pair derives the schedule from the same numbers of the previous launch (slots L % 3: launch L writes its own,
// reads those of L - 1 and clears those of L + 1, which nobody touches meanwhile); one thread records it.
struct BlockArgs {
    const float *i1wx, *i1wy, *grad, *rho_c;
    float* set[2][6];                 // u1, u2, p11, p12, p21, p22 of set 0 and set 1
    PairState* st;
    int* n_active;
    int* live_flag;                   // in pinned host memory: 1 while a pair of the warp is live, 0 once the last one has stopped (the host reads
                                      // it behind an event instead of copying n_active back: 141 copy kernels per batch of 64 pairs)
    int h, w, L, max_iters;
    float l_t, theta, taut;
    double eps2;
    int ew, eh, tw, th;               // tvl1_tile_kernel: a tile of ew x eh cells = tw x th own pixels + the halo; ew * eh <= 4 x its threads
    float inv_ew;
};

template <bool FAST>
__device__ __forceinline__ void primal_pixel(float ux, float uy, float gx, float gy, float gr, float rc, float div1, float div2, float l_t,
                                             float theta, float& n1, float& n2, float& err) {
    const float rho = rc + (gx * ux + gy * uy);
    float d1, d2;
    if (rho < -l_t * gr) {
        d1 = l_t * gx;
        d2 = l_t * gy;
    } else if (rho > l_t * gr) {
        d1 = -l_t * gx;
        d2 = -l_t * gy;
    } else if (gr > kGradIsZero) {
        const float fi = FAST ? -rho * __builtin_amdgcn_rcpf(gr) : -rho / gr;
        d1 = fi * gx;
        d2 = fi * gy;
    } else {
        d1 = d2 = 0.0f;
    }
    n1 = (ux + d1) + theta * div1;
    n2 = (uy + d2) + theta * div2;
    err = (n1 - ux) * (n1 - ux) + (n2 - uy) * (n2 - uy);
}

__device__ __forceinline__ BlkSched next_schedule(const PairState& st, int L, int hw, double eps2, int max_iters) {
    if (L == 0) return BlkSched{kBlkRun, 0, 0, min(kBlkIters, max_iters)};
    const BlkSched prev = st.blk[(L - 1) % 3];
    if (prev.mode == kBlkDone) return prev;
    if (prev.mode == kBlkReplay) return BlkSched{kBlkDone, 1 - prev.src, prev.base + prev.n, 0};
    int j = -1;
    for (int m = prev.n - 1; m >= 0; --m)
        if (!(st.err[(L - 1) % 3][m] / (double)hw > eps2)) j = m;                 // the FIRST iteration whose update was small enough
    if (j < 0) {
        const int base = prev.base + prev.n;
        if (base >= max_iters) return BlkSched{kBlkDone, 1 - prev.src, base, 0};
        return BlkSched{kBlkRun, 1 - prev.src, base, min(kBlkIters, max_iters - base)};
    }
    if (j == prev.n - 1) return BlkSched{kBlkDone, 1 - prev.src, prev.base + prev.n, 0};
    return BlkSched{kBlkReplay, prev.src, prev.base, j + 1};
}

// The product form: tiles FITTED to the level, two workgroups per compute unit.  What round 4 measured about the kernel above: a
// workgroup takes 11-12 us whether 41 or 64 of its waves' cells are live (a level cut into fewer, emptier or fuller 4 096-cell tiles
// costs the same per round of workgroups) -- with ONE workgroup of 16 waves per unit (100 KB of LDS, 118 VGPRs) every barrier, every
// LDS round trip and every division chain of its 4 iterations is exposed; the vector ALUs issue 45 % of the time.  So: 512 threads and
// at most 2 048 cells per workgroup (48 KB of LDS, the same 4 cells and <= 128 VGPRs per thread), TWO workgroups per unit whose phases
// overlap, and the level cut into nx x ny tiles of ceil(w / nx) x ceil(h / ny) own pixels chosen on the host for the fewest rounds x
// waves (fit_tiles): inner loops 22.1 -> 20.0 ms per batch of 64 pairs (256 threads / 1 024 cells: 22.3, the halo eats it; 1 024 threads
// with fitted tiles: 22.1).  The tile's cells are dealt to the threads in row-major order (cell tid + NT k of the ew x eh tile, which is
// also its place in the LDS planes: no padding needed, neighbours in a row are neighbours in a wave), so the cells a tile does NOT have are
// whole waves of its last quarter.  Per pixel the arithmetic and its order are those of the kernel above: same bits (tested).
template <int NT, bool FAST>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(VQ_FLOW_TILE_WPE, VQ_FLOW_TILE_WPE))) void tvl1_tile_kernel(BlockArgs a) {
    constexpr int K = kBlkIters, NC = kTileNC;                 // a tile has at most NC x NT cells
    extern __shared__ float lds[];                       // u1, u2, p11, p12, p21, p22: [eh][ew] each, a cell at its index tid + NT k
    __shared__ double part[NT / 64];
    const int ew = a.ew, plane = a.ew * a.eh;
    float* __restrict__ U1 = lds;
    float* __restrict__ U2 = lds + plane;
    float* __restrict__ P11 = lds + 2 * plane;
    float* __restrict__ P12 = lds + 3 * plane;
    float* __restrict__ P21 = lds + 4 * plane;
    float* __restrict__ P22 = lds + 5 * plane;
    const int p = blockIdx.z;
    PairState& st = a.st[p];
    const int hw = a.h * a.w;
    const BlkSched cur = next_schedule(st, a.L, hw, a.eps2, a.max_iters);
    const bool scribe = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    if (scribe) {
        st.blk[a.L % 3] = cur;
        for (int m = 0; m < K; ++m) st.err[(a.L + 1) % 3][m] = 0.0;
        if (cur.mode == kBlkDone && (a.L == 0 || st.blk[(a.L - 1) % 3].mode != kBlkDone)) {      // the pair has just finished
            st.final_set = cur.src;
            st.iters = cur.base;
            if (atomicSub(a.n_active, 1) == 1) __hip_atomic_store(a.live_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (cur.mode == kBlkDone) return;
    const float* const* src = a.set[cur.src];
    float* const* dst = a.set[1 - cur.src];
    const int xo = (int)blockIdx.x * a.tw - K, yo = (int)blockIdx.y * a.th - K;
    const int64_t base = (int64_t)p * hw;
    const int tid = (int)threadIdx.x;
    float cgx[NC], cgy[NC], cgr[NC], crc[NC], ru1[NC], ru2[NC], r11[NC], r12[NC], r21[NC], r22[NC];
    // what the iterations ask about a cell, decided once (lane masks): it exists; its left / upper neighbour is in the tile AND in the image
    // (a tile-edge cell is outside the exact region, and subtracting the 0 the square kernel reads there gives the same bits as not
    // subtracting); its right / lower one; it is one of the tile's own pixels
    bool live[NC], hl[NC], hu[NC], rgt[NC], blw[NC], own[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int idx = tid + NT * k;
        const int cy = (int)(((float)idx + 0.5f) * a.inv_ew), cx = idx - cy * ew, x = xo + cx, y = yo + cy;
        const bool in_tile = idx < plane;
        cgx[k] = cgy[k] = cgr[k] = crc[k] = 0.f;
        ru1[k] = ru2[k] = r11[k] = r12[k] = r21[k] = r22[k] = 0.f;
        live[k] = in_tile && x >= 0 && x < a.w && y >= 0 && y < a.h;
        hl[k] = cx > 0 && x > 0;
        hu[k] = cy > 0 && y > 0;
        rgt[k] = x + 1 < a.w && cx + 1 < ew;
        blw[k] = y + 1 < a.h && cy + 1 < a.eh;
        own[k] = cx >= K && cx < ew - K && cy >= K && cy < a.eh - K;
        if (live[k]) {
            const int64_t g = base + (int64_t)y * a.w + x;
            ru1[k] = src[0][g];
            ru2[k] = src[1][g];
            r11[k] = src[2][g];
            r12[k] = src[3][g];
            r21[k] = src[4][g];
            r22[k] = src[5][g];
            cgx[k] = a.i1wx[g];
            cgy[k] = a.i1wy[g];
            cgr[k] = a.grad[g];
            crc[k] = a.rho_c[g];
        }
        if (in_tile) {
            P11[idx] = r11[k];
            P12[idx] = r12[k];
            P21[idx] = r21[k];
            P22[idx] = r22[k];
        }
    }
    __syncthreads();
    // the neighbour masks of a cell that does not exist are off: the pair form below has no branch around a dead cell (its fields are
    // zeros and stay zeros: rho = 0, no case of the thresholding applies, |grad u| = 0; nothing of it is stored)
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        hl[k] = hl[k] && live[k];
        hu[k] = hu[k] && live[k];
        rgt[k] = rgt[k] && live[k];
        blw[k] = blw[k] && live[k];
    }
    for (int m = 0; m < cur.n; ++m) {
        double local = 0.0;
        if constexpr (!FAST) {
            static_assert(NC % 2 == 0, "cells go through the iteration in pairs");
            // primal step (a cell reads its own old u and p, the p11 / p21 of its left and the p12 / p22 of its upper neighbour)
#pragma unroll
            for (int j = 0; j < NC; j += 2) {
                if (live[j] || live[j + 1]) {
                    const int ca = tid + NT * j, cb = ca + NT;
                    const f2 q11 = {r11[j], r11[j + 1]}, q12 = {r12[j], r12[j + 1]}, q21 = {r21[j], r21[j + 1]}, q22 = {r22[j], r22[j + 1]};
                    // (subtracting the 0 a masked neighbour stands for gives the bits of not subtracting)
                    const f2 l11 = {hl[j] ? P11[ca - 1] : 0.0f, hl[j + 1] ? P11[cb - 1] : 0.0f};
                    const f2 l21 = {hl[j] ? P21[ca - 1] : 0.0f, hl[j + 1] ? P21[cb - 1] : 0.0f};
                    const f2 t12 = {hu[j] ? P12[ca - ew] : 0.0f, hu[j + 1] ? P12[cb - ew] : 0.0f};
                    const f2 t22 = {hu[j] ? P22[ca - ew] : 0.0f, hu[j + 1] ? P22[cb - ew] : 0.0f};
                    const f2 div1 = (q11 - l11) + (q12 - t12), div2 = (q21 - l21) + (q22 - t22);
                    f2 n1, n2, err;
                    primal_pair(f2{ru1[j], ru1[j + 1]}, f2{ru2[j], ru2[j + 1]}, f2{cgx[j], cgx[j + 1]}, f2{cgy[j], cgy[j + 1]}, f2{cgr[j], cgr[j + 1]},
                                f2{crc[j], crc[j + 1]}, div1, div2, a.l_t, a.theta, n1, n2, err);
                    ru1[j] = n1.x;
                    ru1[j + 1] = n1.y;
                    ru2[j] = n2.x;
                    ru2[j + 1] = n2.y;
                    if (live[j]) {
                        U1[ca] = n1.x;
                        U2[ca] = n2.x;
                        if (own[j]) local += (double)err.x;
                    }
                    if (live[j + 1]) {
                        U1[cb] = n1.y;
                        U2[cb] = n2.y;
                        if (own[j + 1]) local += (double)err.y;
                    }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (live[k]) {
                    const int c = tid + NT * k;
                    const float q11 = r11[k], q12 = r12[k], q21 = r21[k], q22 = r22[k];
                    const float div1 = (hl[k] ? q11 - P11[c - 1] : q11) + (hu[k] ? q12 - P12[c - ew] : q12);
                    const float div2 = (hl[k] ? q21 - P21[c - 1] : q21) + (hu[k] ? q22 - P22[c - ew] : q22);
                    float n1, n2, err;
                    primal_pixel<FAST>(ru1[k], ru2[k], cgx[k], cgy[k], cgr[k], crc[k], div1, div2, a.l_t, a.theta, n1, n2, err);
                    ru1[k] = n1;
                    ru2[k] = n2;
                    U1[c] = n1;
                    U2[c] = n2;
                    if (own[k]) local += (double)err;
                }
            }
        }
        // the tile's squared update of this iteration: one atomic per workgroup
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
        __syncthreads();                                 // also: every new primal value is in LDS, every old p has been read
        if (threadIdx.x == 0) {
            double sum = 0.0;
#pragma unroll
            for (int q = 0; q < NT / 64; ++q) sum += part[q];
            if (sum != 0.0) atomicAdd(&st.err[a.L % 3][m], sum);
        }
        // dual step (a cell reads its own p and new u, the new u of its right and lower neighbour)
        if constexpr (!FAST) {
#pragma unroll
            for (int j = 0; j < NC; j += 2) {
                if (live[j] || live[j + 1]) {
                    const int ca = tid + NT * j, cb = ca + NT;
                    const f2 c1 = {ru1[j], ru1[j + 1]}, c2 = {ru2[j], ru2[j + 1]};
                    // (a masked neighbour: the cell's own value, whose difference is the 0 of the cell-by-cell form)
                    const f2 e1 = {rgt[j] ? U1[ca + 1] : c1.x, rgt[j + 1] ? U1[cb + 1] : c1.y};
                    const f2 s1 = {blw[j] ? U1[ca + ew] : c1.x, blw[j + 1] ? U1[cb + ew] : c1.y};
                    const f2 e2 = {rgt[j] ? U2[ca + 1] : c2.x, rgt[j + 1] ? U2[cb + 1] : c2.y};
                    const f2 s2 = {blw[j] ? U2[ca + ew] : c2.x, blw[j + 1] ? U2[cb + ew] : c2.y};
                    f2 pa1 = {r11[j], r11[j + 1]}, pb1 = {r12[j], r12[j + 1]}, pa2 = {r21[j], r21[j + 1]}, pb2 = {r22[j], r22[j + 1]};
                    dual_pair2(pa1, pb1, e1 - c1, s1 - c1, a.taut);
                    dual_pair2(pa2, pb2, e2 - c2, s2 - c2, a.taut);
                    r11[j] = pa1.x;
                    r11[j + 1] = pa1.y;
                    r12[j] = pb1.x;
                    r12[j + 1] = pb1.y;
                    r21[j] = pa2.x;
                    r21[j + 1] = pa2.y;
                    r22[j] = pb2.x;
                    r22[j + 1] = pb2.y;
                    if (live[j]) {
                        P11[ca] = pa1.x;
                        P12[ca] = pb1.x;
                        P21[ca] = pa2.x;
                        P22[ca] = pb2.x;
                    }
                    if (live[j + 1]) {
                        P11[cb] = pa1.y;
                        P12[cb] = pb1.y;
                        P21[cb] = pa2.y;
                        P22[cb] = pb2.y;
                    }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (live[k]) {
                    const int c = tid + NT * k;
                    const float c1 = ru1[k], c2 = ru2[k];
                    const float u1x = rgt[k] ? U1[c + 1] - c1 : 0.0f, u1y = blw[k] ? U1[c + ew] - c1 : 0.0f;
                    const float u2x = rgt[k] ? U2[c + 1] - c2 : 0.0f, u2y = blw[k] ? U2[c + ew] - c2 : 0.0f;
                    dual_pair<FAST>(r11[k], r12[k], u1x, u1y, a.taut);
                    dual_pair<FAST>(r21[k], r22[k], u2x, u2y, a.taut);
                    P11[c] = r11[k];
                    P12[c] = r12[k];
                    P21[c] = r21[k];
                    P22[c] = r22[k];
                }
            }
        }
        __syncthreads();                                 // every new p is in LDS, every new u has been read
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        if (live[k] && own[k]) {
            const int idx = tid + NT * k;
            const int cy = (int)(((float)idx + 0.5f) * a.inv_ew), cx = idx - cy * ew;
            const int64_t g = base + (int64_t)(yo + cy) * a.w + (xo + cx);
            dst[0][g] = ru1[k];
            dst[1][g] = ru2[k];
            dst[2][g] = r11[k];
            dst[3][g] = r12[k];
            dst[4][g] = r21[k];
            dst[5][g] = r22[k];
        }
    }
}

// After the inner loop of a warp: a pair whose fields ended in set 1 gets them copied back to set 0.
struct SettleArgs {
    float* set0[6];
    const float* set1[6];
};
__global__ void tvl1_settle_kernel(const PairState* __restrict__ st, SettleArgs a, int hw) {
    const int p = blockIdx.y;
    if (st[p].final_set == 0) return;
    for (int q = 0; q < 6; ++q) {
        const float* src = a.set1[q] + (int64_t)p * hw;
        float* dst = a.set0[q] + (int64_t)p * hw;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) dst[i] = src[i];
    }
}

// Where the first-pass flow moves the corners: moved = (xi + u1[yi][xi], yi + u2[yi][xi]) with (xi, yi) the corner rounded to a pixel.
__global__ void move_corners_kernel(const float* __restrict__ corners, const int* __restrict__ counts, const float* __restrict__ u1,
                                    const float* __restrict__ u2, float* __restrict__ moved, int n, int max_corners, int h, int w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * max_corners) return;
    const int p = i / max_corners, k = i - p * max_corners;
    float mx = 0.f, my = 0.f;
    if (k < counts[p]) {
        const int xi = min(max((int)rintf(corners[2 * i]), 0), w - 1), yi = min(max((int)rintf(corners[2 * i + 1]), 0), h - 1);
        const int64_t g = ((int64_t)p * h + yi) * w + xi;
        mx = (float)xi + u1[g];
        my = (float)yi + u2[g];
    }
    moved[2 * i] = mx;
    moved[2 * i + 1] = my;
}

__global__ void flow_to_image_kernel(const float* __restrict__ flow, uint8_t* __restrict__ img, int64_t total, float bound) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double v = ((double)flow[i] + (double)bound) * (255.0 / (2.0 * (double)bound));
    img[i] = (uint8_t)fmin(fmax(floor(v + 0.5), 0.0), 255.0);
}

__global__ void homography_warp_kernel(const float* __restrict__ src, float* __restrict__ dst, const double* __restrict__ hinv, int n, int h,
                                       int w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * h * w) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const int64_t p = i / ((int64_t)h * w);
    const double* m = hinv + p * 9;
    const double den = m[6] * x + m[7] * y + m[8];
    const float sx = (float)((m[0] * x + m[1] * y + m[2]) / den), sy = (float)((m[3] * x + m[4] * y + m[5]) / den);
    // oracle.warp_homography: bilinear sample at (x + (sx - x), y + (sy - y)) -- the same fp32 additions
    dst[i] = sample_bilinear(src + p * (int64_t)h * w, h, w, (float)x + (sx - (float)x), (float)y + (sy - (float)y));
}

// ------------------------------------------------------------------------------------------------
// Camera-motion estimation ("warp" half of extract_warp_gpu, the flow-match branch): Shi-Tomasi corners on the first frame,
// matched through the first-pass flow, a RANSAC homography over the matches (oracle/warp_oracle.py).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// cv::cornerMinEigenVal(blockSize 3, Sobel 3, BORDER_REFLECT_101) on an 8-bit frame: derivatives scaled by 1 / (4 * 3 * 255),
// their products summed over the 3 x 3 block (rows first), smaller eigenvalue of the 2 x 2 matrix.
__global__ void corner_strength_kernel(const uint8_t* __restrict__ frames, float* __restrict__ strength, unsigned* __restrict__ frame_max, int n,
                                       int h, int w) {
    // grid (pixel blocks, frame): a block lies inside one frame, so the frame maximum costs one atomic per block
    __shared__ float wave_top[4];
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.y;
    const bool live = q < h * w;
    const int x = live ? q % w : 0, y = live ? q / w : 0;
    const int64_t i = p * (int64_t)h * w + q;
    const uint8_t* im = frames + p * (int64_t)h * w;
    const float scale = (float)(1.0 / (4.0 * 3.0 * 255.0));
    float a = 0.f, b = 0.f, c = 0.f;
    for (int dy = -1; dy <= 1; ++dy) {
        float ra = 0.f, rb = 0.f, rc = 0.f;
        const int yy = reflect101(y + dy, h);
        const int y0 = reflect101(yy - 1, h), y2 = reflect101(yy + 1, h);
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = reflect101(x + dx, w);
            const int x0 = reflect101(xx - 1, w), x2 = reflect101(xx + 1, w);
            const int gx = ((int)im[y0 * w + x2] + 2 * (int)im[yy * w + x2] + (int)im[y2 * w + x2]) -
                           ((int)im[y0 * w + x0] + 2 * (int)im[yy * w + x0] + (int)im[y2 * w + x0]);
            const int gy = ((int)im[y2 * w + x0] + 2 * (int)im[y2 * w + xx] + (int)im[y2 * w + x2]) -
                           ((int)im[y0 * w + x0] + 2 * (int)im[y0 * w + xx] + (int)im[y0 * w + x2]);
            const float fx = (float)gx * scale, fy = (float)gy * scale;
            ra += fx * fx;
            rb += fx * fy;
            rc += fy * fy;
        }
        a += ra;
        b += rb;
        c += rc;
    }
    a *= 0.5f;
    c *= 0.5f;
    const float d = a - c;
    const float v = (a + c) - __fsqrt_rn(d * d + b * b);
    if (live) strength[i] = v;
    float top = live ? fmaxf(v, 0.f) : 0.f;
    for (int off = 32; off > 0; off >>= 1) top = fmaxf(top, __shfl_xor(top, off));
    if ((threadIdx.x & 63) == 0) wave_top[threadIdx.x >> 6] = top;
    __syncthreads();
    if (threadIdx.x == 0) {
        top = fmaxf(fmaxf(wave_top[0], wave_top[1]), fmaxf(wave_top[2], wave_top[3]));
        if (top > 0.f) atomicMax(frame_max + p, __float_as_uint(top));   // positive floats order like their bit patterns
    }
}

// A candidate corner is an interior pixel that equals the maximum of its 3 x 3 neighbourhood (cv::dilate + compare); others read 0.
__global__ void corner_peaks_kernel(const float* __restrict__ strength, float* __restrict__ peaks, int n, int h, int w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * h * w) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const float* s = strength + (i - (int64_t)y * w - x);
    float out = 0.f;
    if (x >= 1 && y >= 1 && x < w - 1 && y < h - 1) {
        const float v = s[y * w + x];
        float m = v;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) m = fmaxf(m, s[(y + dy) * w + x + dx]);
        out = (v == m) ? v : 0.f;
    }
    peaks[i] = out;
}

__host__ __device__ inline uint32_t mix32(uint32_t x) {   // "lowbias32" integer hash: the hypothesis generator of the RANSAC
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

__device__ inline double orient(const float* a, const float* b, const float* c) {
    return ((double)b[0] - (double)a[0]) * ((double)c[1] - (double)a[1]) - ((double)b[1] - (double)a[1]) * ((double)c[0] - (double)a[0]);
}

// The homography through 4 correspondences (h33 = 1): 8 x 8 system, Gaussian elimination with partial pivoting, fp64.
__device__ bool homography_4pt(const float* const* s, const float* const* d, double* H) {
    double A[8][9];
    for (int k = 0; k < 4; ++k) {
        const double x = s[k][0], y = s[k][1], u = d[k][0], v = d[k][1];
        double* r0 = A[2 * k];
        double* r1 = A[2 * k + 1];
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * x; r0[7] = -u * y; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -v * x; r1[7] = -v * y; r1[8] = v;
    }
    for (int c = 0; c < 8; ++c) {
        int piv = c;
        double best = fabs(A[c][c]);
        for (int r = c + 1; r < 8; ++r)
            if (fabs(A[r][c]) > best) {
                best = fabs(A[r][c]);
                piv = r;
            }
        if (best < 1e-9) return false;
        if (piv != c)
            for (int q = c; q < 9; ++q) {
                const double t = A[c][q];
                A[c][q] = A[piv][q];
                A[piv][q] = t;
            }
        for (int r = c + 1; r < 8; ++r) {
            const double f = A[r][c] / A[c][c];
            for (int q = c; q < 9; ++q) A[r][q] -= f * A[c][q];
        }
    }
    for (int c = 7; c >= 0; --c) {
        double v = A[c][8];
        for (int q = c + 1; q < 8; ++q) v -= A[c][q] * H[q];
        H[c] = v / A[c][c];
    }
    H[8] = 1.0;
    return true;
}

// One workgroup per pair, one hypothesis per thread and round: 4 distinct matches drawn with mix32, rejected when the two
// quadrilaterals are not consistently oriented (cv's checkSubset), scored by the number of matches whose forward
// reprojection error is <= thr2.  The winner is the hypothesis with the most inliers, the lowest index among equals.
__global__ void ransac_homography_kernel(const float* __restrict__ src, const float* __restrict__ dst, const int* __restrict__ counts, int max_points,
                                         int hypotheses, uint32_t seed, double thr2, double* __restrict__ best_h, int* __restrict__ best_count,
                                         int* __restrict__ best_index, uint8_t* __restrict__ mask) {
    extern __shared__ float pts[];               // [max_points][4]: sx, sy, dx, dy
    __shared__ int win_count[256], win_index[256];
    __shared__ double win_h[9];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int n = counts[p];
    for (int i = tid; i < n; i += blockDim.x) {
        pts[4 * i] = src[((size_t)p * max_points + i) * 2];
        pts[4 * i + 1] = src[((size_t)p * max_points + i) * 2 + 1];
        pts[4 * i + 2] = dst[((size_t)p * max_points + i) * 2];
        pts[4 * i + 3] = dst[((size_t)p * max_points + i) * 2 + 1];
    }
    __syncthreads();
    int my_count = -1, my_index = 0x7fffffff;
    double my_h[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (n >= 4) {
        for (int j = tid; j < hypotheses; j += blockDim.x) {
            int idx[4];
            bool ok = true;
            const uint32_t base = mix32(mix32(seed + (uint32_t)p) + (uint32_t)j);
            for (int k = 0; k < 4 && ok; ++k) {
                ok = false;
                for (uint32_t a = 0; a < 64 && !ok; ++a) {
                    const int cand = (int)(mix32(base + 16u * a + (uint32_t)k) % (uint32_t)n);
                    bool fresh = true;
                    for (int q = 0; q < k; ++q) fresh = fresh && idx[q] != cand;
                    if (fresh) {
                        idx[k] = cand;
                        ok = true;
                    }
                }
            }
            if (!ok) continue;
            const float* s4[4];
            const float* d4[4];
            for (int k = 0; k < 4; ++k) {
                s4[k] = pts + 4 * idx[k];
                d4[k] = pts + 4 * idx[k] + 2;
            }
            // every triple keeps its orientation (and is not collinear) under a homography that maps the quadrilateral properly
            for (int a = 0; a < 4 && ok; ++a) {
                const int i0 = a, i1 = (a + 1) & 3, i2 = (a + 2) & 3;
                const double os = orient(s4[i0], s4[i1], s4[i2]), od = orient(d4[i0], d4[i1], d4[i2]);
                ok = os * od > 0.0;
            }
            if (!ok) continue;
            double H[9];
            if (!homography_4pt(s4, d4, H)) continue;
            int cnt = 0;
            for (int i = 0; i < n; ++i) {
                const double x = pts[4 * i], y = pts[4 * i + 1];
                const double wq = H[6] * x + H[7] * y + 1.0;
                const double ex = (H[0] * x + H[1] * y + H[2]) / wq - (double)pts[4 * i + 2];
                const double ey = (H[3] * x + H[4] * y + H[5]) / wq - (double)pts[4 * i + 3];
                cnt += (ex * ex + ey * ey <= thr2) ? 1 : 0;
            }
            if (cnt > my_count) {     // j ascends per thread: the first best stays
                my_count = cnt;
                my_index = j;
                for (int q = 0; q < 9; ++q) my_h[q] = H[q];
            }
        }
    }
    win_count[tid] = my_count;
    win_index[tid] = my_index;
    __syncthreads();
    for (int step = blockDim.x / 2; step > 0; step >>= 1) {
        if (tid < step) {
            const int oc = win_count[tid + step], oi = win_index[tid + step];
            if (oc > win_count[tid] || (oc == win_count[tid] && oi < win_index[tid])) {
                win_count[tid] = oc;
                win_index[tid] = oi;
            }
        }
        __syncthreads();
    }
    const int w_count = win_count[0], w_index = win_index[0];
    if (my_index == w_index && w_count >= 0)
        for (int q = 0; q < 9; ++q) win_h[q] = my_h[q];
    if (w_count < 0 && tid == 0)
        for (int q = 0; q < 9; ++q) win_h[q] = (q % 4 == 0) ? 1.0 : 0.0;
    __syncthreads();
    if (tid < 9) best_h[(size_t)p * 9 + tid] = win_h[tid];
    if (tid == 0) {
        best_count[p] = w_count < 0 ? 0 : w_count;
        best_index[p] = w_count < 0 ? -1 : w_index;
    }
    for (int i = tid; i < max_points; i += blockDim.x) {
        uint8_t in = 0;
        if (i < n && w_count >= 0) {
            const double x = pts[4 * i], y = pts[4 * i + 1];
            const double wq = win_h[6] * x + win_h[7] * y + 1.0;
            const double ex = (win_h[0] * x + win_h[1] * y + win_h[2]) / wq - (double)pts[4 * i + 2];
            const double ey = (win_h[3] * x + win_h[4] * y + win_h[5]) / wq - (double)pts[4 * i + 3];
            in = (ex * ex + ey * ey <= thr2) ? 1 : 0;
        }
        mask[(size_t)p * max_points + i] = in;
    }
}

}  // namespace

struct vq_flow {
    std::recursive_mutex mu;               // recursive: vq_flow_warped holds it across the calls it is composed of
    int device = 0, max_pairs = 0, h = 0, w = 0;
    vq_tvl1_params prm;
    std::vector<Level> levels;
    size_t pyr_floats = 0;                 // floats of one pair's pyramid
    float *pyr0 = nullptr, *pyr1 = nullptr;      // [level][pair][h_l][w_l]
    float* plane[12] = {nullptr};          // i1x, i1y, i1wx, i1wy, grad, rho_c, u1, u2 / p11, p12, p21, p22 at the current level ...
    float* alt[6] = {nullptr};             // second set of u1, u2, p11, p12, p21, p22 (the one-launch iteration ping-pongs between the sets)
    float* tmp[2] = {nullptr, nullptr};    // flow of the coarser level while it is resized
    uint8_t* frames_dev[2] = {nullptr, nullptr};
    uint8_t* img_dev[2] = {nullptr, nullptr};
    PairState* st = nullptr;
    int* n_active = nullptr;
    int* iters_log = nullptr;              // [levels][warps][pairs]
    int* live_host = nullptr;              // pinned: [0], [1] the two most recent polls of n_active (two-launch form); [2] the blocked form's live flag
    int* live_flag_dev = nullptr;          // live_host + 2 as the device sees it
    hipEvent_t poll_ev[2] = {nullptr, nullptr};
    double* hinv_dev = nullptr;
    unsigned* frame_max = nullptr;         // [max_pairs] bit pattern of the largest corner strength of a frame
    void* match_dev = nullptr;             // RANSAC scratch (matches, winners, masks), grown on demand
    size_t match_bytes = 0;
    float* peaks_host = nullptr;           // pinned: the corner-peak maps of a batch on their way to the host's selection
    size_t peaks_host_floats = 0;
    void* warp_dev = nullptr;              // vq_flow_warped: corners, moved corners, counts
    size_t warp_bytes = 0;
    float* corner_plane[2] = {nullptr, nullptr};   // corner strength / peak maps: their own memory, so that the corner search of a
    hipStream_t side_stream = nullptr;             // batch can run (on this stream) beside its first flow pass
    hipEvent_t side_ev = nullptr;
    std::vector<hipEvent_t> loop_ev;       // a start / stop pair around the inner loop of every (level, warp) of a call
    double last_inner_ms = 0.0;            // device time of those loops in the last vq_flow_tvl1 call (sum of the pairs)
    int last_iter_launches = 0;            // iteration-kernel launches of the last call
    bool exact_math = true;                // VQ_FLOW_FAST=1 at creation switches to hardware reciprocals / roots in the inner iterations (see tv_rcp)
    int n_cus = 256;                       // compute units of the device (tile fitting)
};

static void flow_free(vq_flow* f) {
    for (float* p : f->plane)
        if (p) (void)hipFree(p);
    for (float* p : f->tmp)
        if (p) (void)hipFree(p);
    for (float* p : f->alt)
        if (p) (void)hipFree(p);
    if (f->pyr0) (void)hipFree(f->pyr0);
    if (f->pyr1) (void)hipFree(f->pyr1);
    for (int k = 0; k < 2; ++k) {
        if (f->frames_dev[k]) (void)hipFree(f->frames_dev[k]);
        if (f->img_dev[k]) (void)hipFree(f->img_dev[k]);
    }
    if (f->st) (void)hipFree(f->st);
    if (f->n_active) (void)hipFree(f->n_active);
    if (f->iters_log) (void)hipFree(f->iters_log);
    if (f->live_host) (void)hipHostFree(f->live_host);
    for (hipEvent_t e : f->poll_ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : f->loop_ev)
        if (e) (void)hipEventDestroy(e);
    if (f->hinv_dev) (void)hipFree(f->hinv_dev);
    if (f->frame_max) (void)hipFree(f->frame_max);
    if (f->match_dev) (void)hipFree(f->match_dev);
    if (f->warp_dev) (void)hipFree(f->warp_dev);
    if (f->peaks_host) (void)hipHostFree(f->peaks_host);
    for (float* p : f->corner_plane)
        if (p) (void)hipFree(p);
    if (f->side_stream) (void)hipStreamDestroy(f->side_stream);
    if (f->side_ev) (void)hipEventDestroy(f->side_ev);
}

extern "C" {

int vq_tvl1_default_params(vq_tvl1_params* p) {
    VQ_REQUIRE(p, "NULL argument");
    p->tau = 0.25f;
    p->lambda = 0.15f;
    p->theta = 0.3f;
    p->epsilon = 0.01f;
    p->scale_step = 0.8f;
    p->nscales = 5;
    p->warps = 5;
    p->iterations = 300;
    p->bound = 20.0f;
    return VQ_OK;
}

int vq_flow_create(int32_t max_pairs, int32_t h, int32_t w, const vq_tvl1_params* params, int32_t device, vq_flow** out) {
    VQ_REQUIRE(out, "out is NULL");
    *out = nullptr;
    VQ_REQUIRE(max_pairs > 0 && h >= 16 && w >= 16 && (int64_t)max_pairs * h * w < (1ll << 31), "bad batch shape %d x %d x %d", max_pairs, h, w);
    vq_tvl1_params prm;
    vq_tvl1_default_params(&prm);
    if (params) prm = *params;
    VQ_REQUIRE(prm.nscales >= 1 && prm.nscales <= 16 && prm.warps >= 1 && prm.warps <= 64 && prm.iterations >= 1 && prm.theta > 0 &&
                   prm.scale_step > 0 && prm.scale_step < 1 && prm.bound > 0 && prm.epsilon >= 0,
               "TV-L1 parameters out of range");
    int ndev = 0;
    VQ_HIP(hipGetDeviceCount(&ndev));
    VQ_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard g(device);
    auto* f = new vq_flow;
    f->device = device;
    f->max_pairs = max_pairs;
    f->h = h;
    f->w = w;
    f->prm = prm;
    {
        const char* e3 = getenv("VQ_FLOW_FAST");
        f->exact_math = !(e3 && *e3 == '1');
        hipDeviceProp_t prop;
        VQ_HIP(hipGetDeviceProperties(&prop, device));
        f->n_cus = std::max(1, prop.multiProcessorCount);
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tvl1_tile_kernel<kTileThreads, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   6 * kTileCells * (int)sizeof(float)));
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tvl1_tile_kernel<kTileThreads, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   6 * kTileCells * (int)sizeof(float)));
    }
    // level sizes, finest first: round(previous * scale_step), stop before 16 pixels (oracle.pyramid_sizes)
    size_t off = 0;
    int lh = h, lw = w;
    for (int s = 0; s < prm.nscales; ++s) {
        if (s > 0) {
            const int nh = (int)std::nearbyint((double)lh * (double)prm.scale_step), nw = (int)std::nearbyint((double)lw * (double)prm.scale_step);
            if (nh < 16 || nw < 16) break;
            lh = nh;
            lw = nw;
        }
        f->levels.push_back(Level{lh, lw, off});
        off += (size_t)max_pairs * lh * lw;
    }
    f->pyr_floats = off;
    auto bail = [&](const char* what, hipError_t e) {
        flow_free(f);
        delete f;
        return fail(e == hipErrorOutOfMemory ? VQ_E_NOMEM : VQ_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    const size_t full = (size_t)max_pairs * h * w;
    hipError_t e;
    if ((e = vq::malloc_trim((void**)&f->pyr0, off * sizeof(float))) != hipSuccess) return bail("vq::malloc_trim(pyramid)", e);
    if ((e = vq::malloc_trim((void**)&f->pyr1, off * sizeof(float))) != hipSuccess) return bail("vq::malloc_trim(pyramid)", e);
    for (float*& p : f->plane)
        if ((e = vq::malloc_trim((void**)&p, full * sizeof(float))) != hipSuccess) return bail("vq::malloc_trim(plane)", e);
    for (float*& p : f->tmp)
        if ((e = vq::malloc_trim((void**)&p, full * sizeof(float))) != hipSuccess) return bail("vq::malloc_trim(plane)", e);
    for (float*& p : f->alt)
        if ((e = vq::malloc_trim((void**)&p, full * sizeof(float))) != hipSuccess) return bail("vq::malloc_trim(plane)", e);
    for (int k = 0; k < 2; ++k) {
        if ((e = vq::malloc_trim((void**)&f->frames_dev[k], full)) != hipSuccess) return bail("vq::malloc_trim(frames)", e);
        if ((e = vq::malloc_trim((void**)&f->img_dev[k], full)) != hipSuccess) return bail("vq::malloc_trim(images)", e);
    }
    if ((e = vq::malloc_trim((void**)&f->st, (size_t)max_pairs * sizeof(PairState))) != hipSuccess) return bail("vq::malloc_trim(state)", e);
    if ((e = vq::malloc_trim((void**)&f->n_active, sizeof(int))) != hipSuccess) return bail("vq::malloc_trim(state)", e);
    if ((e = vq::malloc_trim((void**)&f->iters_log, (size_t)f->levels.size() * prm.warps * max_pairs * sizeof(int))) != hipSuccess)
        return bail("vq::malloc_trim(log)", e);
    if ((e = hipHostMalloc((void**)&f->live_host, 4 * sizeof(int))) != hipSuccess) return bail("hipHostMalloc(poll)", e);
    if ((e = hipHostGetDevicePointer((void**)&f->live_flag_dev, f->live_host + 2, 0)) != hipSuccess) return bail("hipHostGetDevicePointer(poll)", e);
    for (hipEvent_t& ev : f->poll_ev)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    f->loop_ev.assign((size_t)2 * f->levels.size() * prm.warps, nullptr);
    for (hipEvent_t& ev : f->loop_ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = vq::malloc_trim((void**)&f->hinv_dev, (size_t)max_pairs * 9 * sizeof(double))) != hipSuccess) return bail("vq::malloc_trim(homographies)", e);
    if ((e = vq::malloc_trim((void**)&f->frame_max, (size_t)max_pairs * sizeof(unsigned))) != hipSuccess) return bail("vq::malloc_trim(state)", e);
    *out = f;
    return VQ_OK;
}

int vq_flow_destroy(vq_flow* f) {
    if (!f) return VQ_OK;
    {
        DeviceGuard g(f->device);
        (void)hipDeviceSynchronize();
        flow_free(f);
    }
    delete f;
    return VQ_OK;
}

int vq_flow_levels(vq_flow* f, int32_t* n_levels, int32_t* sizes_hw, int32_t cap) {
    VQ_REQUIRE(f && n_levels, "NULL argument");
    *n_levels = (int)f->levels.size();
    for (int s = 0; s < (int)f->levels.size() && s < cap && sizes_hw; ++s) {
        sizes_hw[2 * s] = f->levels[s].h;
        sizes_hw[2 * s + 1] = f->levels[s].w;
    }
    return VQ_OK;
}

int vq_flow_tvl1(vq_flow* f, const uint8_t* frames0, const uint8_t* frames1, int32_t frames_on_device, int32_t n_pairs,
                 const double* homographies_host, float* u1_host, float* u2_host, uint8_t* flow_x_host, uint8_t* flow_y_host,
                 int32_t* iters_host, void* hip_stream) {
    VQ_REQUIRE(f && frames0 && frames1, "NULL argument");
    VQ_REQUIRE(n_pairs > 0 && n_pairs <= f->max_pairs, "n_pairs %d outside (0,%d]", n_pairs, f->max_pairs);
    std::lock_guard<std::recursive_mutex> lk(f->mu);
    DeviceGuard g(f->device);
    hipStream_t st = (hipStream_t)hip_stream;
    const vq_tvl1_params& P = f->prm;
    const int h = f->h, w = f->w;
    const int64_t full = (int64_t)n_pairs * h * w;
    const uint8_t *d0 = frames0, *d1 = frames1;
    if (!frames_on_device) {
        VQ_HIP(hipMemcpyAsync(f->frames_dev[0], frames0, (size_t)full, hipMemcpyHostToDevice, st));
        VQ_HIP(hipMemcpyAsync(f->frames_dev[1], frames1, (size_t)full, hipMemcpyHostToDevice, st));
        d0 = f->frames_dev[0];
        d1 = f->frames_dev[1];
    }
    float *i1x = f->plane[0], *i1y = f->plane[1], *i1wx = f->plane[2], *i1wy = f->plane[3], *grad = f->plane[4], *rho_c = f->plane[5];
    float *u1 = f->plane[6], *u2 = f->plane[7], *p11 = f->plane[8], *p12 = f->plane[9], *p21 = f->plane[10], *p22 = f->plane[11];
    const int nl = (int)f->levels.size();
    // level 0 of the pyramids: the frames as floats (0..255); the second frame optionally through a homography first
    u8_to_float_kernel<<<cdiv(full, 256), 256, 0, st>>>(d0, f->pyr0, full);
    if (homographies_host) {
        std::vector<double> inv((size_t)n_pairs * 9);
        for (int p = 0; p < n_pairs; ++p) {          // 3x3 inverse by cofactors (fp64), as numpy.linalg.inv does to rounding
            const double* m = homographies_host + (size_t)p * 9;
            const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
            VQ_REQUIRE(std::fabs(det) > 1e-300, "homography %d is singular", p);
            double* o = inv.data() + (size_t)p * 9;
            o[0] = (m[4] * m[8] - m[5] * m[7]) / det;
            o[1] = (m[2] * m[7] - m[1] * m[8]) / det;
            o[2] = (m[1] * m[5] - m[2] * m[4]) / det;
            o[3] = (m[5] * m[6] - m[3] * m[8]) / det;
            o[4] = (m[0] * m[8] - m[2] * m[6]) / det;
            o[5] = (m[2] * m[3] - m[0] * m[5]) / det;
            o[6] = (m[3] * m[7] - m[4] * m[6]) / det;
            o[7] = (m[1] * m[6] - m[0] * m[7]) / det;
            o[8] = (m[0] * m[4] - m[1] * m[3]) / det;
        }
        VQ_HIP(hipMemcpyAsync(f->hinv_dev, inv.data(), inv.size() * sizeof(double), hipMemcpyHostToDevice, st));
        VQ_HIP(hipStreamSynchronize(st));           // `inv` leaves scope
        u8_to_float_kernel<<<cdiv(full, 256), 256, 0, st>>>(d1, f->tmp[0], full);
        homography_warp_kernel<<<cdiv(full, 256), 256, 0, st>>>(f->tmp[0], f->pyr1, f->hinv_dev, n_pairs, h, w);
    } else {
        u8_to_float_kernel<<<cdiv(full, 256), 256, 0, st>>>(d1, f->pyr1, full);
    }
    VQ_CHECK_LAUNCH();
    for (int s = 1; s < nl; ++s) {
        const Level &a = f->levels[s - 1], &b = f->levels[s];
        const int64_t tot = (int64_t)n_pairs * b.h * b.w;
        resize_kernel<<<cdiv(tot, 256), 256, 0, st>>>(f->pyr0 + a.off, f->pyr0 + b.off, n_pairs, a.h, a.w, b.h, b.w, 1.0f);
        resize_kernel<<<cdiv(tot, 256), 256, 0, st>>>(f->pyr1 + a.off, f->pyr1 + b.off, n_pairs, a.h, a.w, b.h, b.w, 1.0f);
    }
    VQ_CHECK_LAUNCH();
    {
        const Level& c = f->levels[nl - 1];
        VQ_HIP(hipMemsetAsync(u1, 0, (size_t)n_pairs * c.h * c.w * sizeof(float), st));
        VQ_HIP(hipMemsetAsync(u2, 0, (size_t)n_pairs * c.h * c.w * sizeof(float), st));
    }
    const double eps2 = (double)P.epsilon * (double)P.epsilon;      // oracle: float(float32(epsilon)) ** 2
    int iter_launches = 0;
    for (int s = nl - 1; s >= 0; --s) {
        const Level& L = f->levels[s];
        const int64_t tot = (int64_t)n_pairs * L.h * L.w;
        const float *i0 = f->pyr0 + L.off, *i1 = f->pyr1 + L.off;
        gradient_kernel<<<cdiv(tot, 256), 256, 0, st>>>(i1, i1x, i1y, n_pairs, L.h, L.w);
        for (float* p : {p11, p12, p21, p22}) VQ_HIP(hipMemsetAsync(p, 0, (size_t)tot * sizeof(float), st));
        const float l_t = (float)((double)P.lambda * (double)P.theta);      // oracle: float32(lam * theta) on the float32 parameters
        const float taut = (float)((double)P.tau / (double)P.theta);
        const TileCut cut = fit_tiles(L.w, L.h, n_pairs, std::max(1, VQ_FLOW_TILE_WPE * 256 / kTileThreads) * f->n_cus);
        const dim3 tgrid((unsigned)cut.nx, (unsigned)cut.ny, (unsigned)n_pairs);
        const size_t tlds = (size_t)6 * cut.eh * cut.ew * sizeof(float);
        BlockArgs ba;
        ba.ew = cut.ew;
        ba.eh = cut.eh;
        ba.tw = cut.tw;
        ba.th = cut.th;
        ba.inv_ew = 1.0f / (float)cut.ew;
        ba.i1wx = i1wx;
        ba.i1wy = i1wy;
        ba.grad = grad;
        ba.rho_c = rho_c;
        ba.st = f->st;
        ba.n_active = f->n_active;
        ba.live_flag = f->live_flag_dev;
        ba.h = L.h;
        ba.w = L.w;
        ba.max_iters = P.iterations;
        ba.l_t = l_t;
        ba.theta = P.theta;
        ba.taut = taut;
        ba.eps2 = eps2;
        float* set[2][6] = {{u1, u2, p11, p12, p21, p22}, {f->alt[0], f->alt[1], f->alt[2], f->alt[3], f->alt[4], f->alt[5]}};
        SettleArgs sa;
        for (int q = 0; q < 6; ++q) {
            sa.set0[q] = ba.set[0][q] = set[0][q];
            sa.set1[q] = ba.set[1][q] = set[1][q];
        }
        for (int wp = 0; wp < P.warps; ++wp) {
            tvl1_warp_kernel<<<cdiv(tot, 256), 256, 0, st>>>(i0, i1, i1x, i1y, u1, u2, i1wx, i1wy, grad, rho_c, f->st, f->n_active, f->live_flag_dev, n_pairs, L.h, L.w);
            // Converged pairs switch themselves off on the device (their workgroups exit at once).  The host polls the number
            // of live pairs once per chunk of iterations, one chunk BEHIND what it has queued: the stream never runs dry while
            // the host waits, at the price of at most one chunk of empty launches after the last pair has stopped.
            int chunk_no = 0;
            const size_t ev_i = 2 * ((size_t)(nl - 1 - s) * P.warps + wp);
            VQ_HIP(hipEventRecord(f->loop_ev[ev_i], st));
            {
                // blocks of kBlkIters iterations; a pair needs at most ceil(iterations / K) blocks, one replay and one closing launch
                const int max_launches = cdiv(P.iterations, kBlkIters) + 2;
                for (int l0 = 0; l0 < max_launches; ++chunk_no) {
                    const int chunk = std::min(max_launches - l0, l0 < 4 ? 2 : 4);
                    for (int k = 0; k < chunk; ++k) {
                        ba.L = l0 + k;
                        if (f->exact_math) tvl1_tile_kernel<kTileThreads, false><<<tgrid, kTileThreads, tlds, st>>>(ba);
                        else tvl1_tile_kernel<kTileThreads, true><<<tgrid, kTileThreads, tlds, st>>>(ba);
                        ++iter_launches;
                    }
                    VQ_CHECK_LAUNCH();
                    l0 += chunk;
                    VQ_HIP(hipEventRecord(f->poll_ev[chunk_no & 1], st));
                    if (chunk_no > 0) {
                        VQ_HIP(hipEventSynchronize(f->poll_ev[(chunk_no - 1) & 1]));
                        // the kernels clear the flag in host memory when the last pair stops: no copy in the stream
                        if (__atomic_load_n(const_cast<volatile int*>(f->live_host + 2), __ATOMIC_ACQUIRE) == 0) break;
                    }
                }
            }
            VQ_HIP(hipEventRecord(f->loop_ev[ev_i + 1], st));
            tvl1_settle_kernel<<<dim3((unsigned)std::min(cdiv((int64_t)L.h * L.w, 256), 32), (unsigned)n_pairs), 256, 0, st>>>(f->st, sa, L.h * L.w);
            VQ_CHECK_LAUNCH();
            if (iters_host)
                log_iters_kernel<<<cdiv(n_pairs, 256), 256, 0, st>>>(f->st, f->iters_log + ((size_t)(nl - 1 - s) * P.warps + wp) * n_pairs, n_pairs);
        }
        if (s > 0) {          // to the next finer level: bilinear resize, flow values divided by the scale step
            const Level& F = f->levels[s - 1];
            const int64_t ftot = (int64_t)n_pairs * F.h * F.w;
            const float inv = (float)(1.0 / (double)P.scale_step);
            VQ_HIP(hipMemcpyAsync(f->tmp[0], u1, (size_t)tot * sizeof(float), hipMemcpyDeviceToDevice, st));
            VQ_HIP(hipMemcpyAsync(f->tmp[1], u2, (size_t)tot * sizeof(float), hipMemcpyDeviceToDevice, st));
            resize_kernel<<<cdiv(ftot, 256), 256, 0, st>>>(f->tmp[0], u1, n_pairs, L.h, L.w, F.h, F.w, inv);
            resize_kernel<<<cdiv(ftot, 256), 256, 0, st>>>(f->tmp[1], u2, n_pairs, L.h, L.w, F.h, F.w, inv);
            VQ_CHECK_LAUNCH();
        }
    }
    if (iters_host)
        VQ_HIP(hipMemcpyAsync(iters_host, f->iters_log, (size_t)nl * P.warps * n_pairs * sizeof(int), hipMemcpyDeviceToHost, st));
    if (u1_host) VQ_HIP(hipMemcpyAsync(u1_host, u1, (size_t)full * sizeof(float), hipMemcpyDeviceToHost, st));
    if (u2_host) VQ_HIP(hipMemcpyAsync(u2_host, u2, (size_t)full * sizeof(float), hipMemcpyDeviceToHost, st));
    if (flow_x_host || flow_y_host) {
        flow_to_image_kernel<<<cdiv(full, 256), 256, 0, st>>>(u1, f->img_dev[0], full, P.bound);
        flow_to_image_kernel<<<cdiv(full, 256), 256, 0, st>>>(u2, f->img_dev[1], full, P.bound);
        VQ_CHECK_LAUNCH();
        if (flow_x_host) VQ_HIP(hipMemcpyAsync(flow_x_host, f->img_dev[0], (size_t)full, hipMemcpyDeviceToHost, st));
        if (flow_y_host) VQ_HIP(hipMemcpyAsync(flow_y_host, f->img_dev[1], (size_t)full, hipMemcpyDeviceToHost, st));
    }
    VQ_HIP(hipStreamSynchronize(st));
    f->last_iter_launches = iter_launches;
    f->last_inner_ms = 0.0;
    for (size_t q = 0; q + 1 < f->loop_ev.size(); q += 2) {
        float ms = 0.f;
        VQ_HIP(hipEventElapsedTime(&ms, f->loop_ev[q], f->loop_ev[q + 1]));
        f->last_inner_ms += ms;
    }
    return VQ_OK;
}

int vq_flow_last_timing(vq_flow* f, double* inner_loops_ms, int32_t* iteration_launches) {
    VQ_REQUIRE(f, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(f->mu);
    if (inner_loops_ms) *inner_loops_ms = f->last_inner_ms;
    if (iteration_launches) *iteration_launches = f->last_iter_launches;
    return VQ_OK;
}

}  // extern "C"

// ---- camera-motion estimation -------------------------------------------------------------------------------------------

namespace {

bool solve_dense(std::vector<double>& A, std::vector<double>& b, int n) {   // Gaussian elimination, partial pivoting; b <- solution
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)piv * n + c])) piv = r;
        if (std::fabs(A[(size_t)piv * n + c]) < 1e-300) return false;
        if (piv != c) {
            for (int q = 0; q < n; ++q) std::swap(A[(size_t)c * n + q], A[(size_t)piv * n + q]);
            std::swap(b[c], b[piv]);
        }
        for (int r = c + 1; r < n; ++r) {
            const double f = A[(size_t)r * n + c] / A[(size_t)c * n + c];
            for (int q = c; q < n; ++q) A[(size_t)r * n + q] -= f * A[(size_t)c * n + q];
            b[r] -= f * b[c];
        }
    }
    for (int c = n - 1; c >= 0; --c) {
        double v = b[c];
        for (int q = c + 1; q < n; ++q) v -= A[(size_t)c * n + q] * b[q];
        b[c] = v / A[(size_t)c * n + c];
    }
    return true;
}

// Least-squares homography (h33 = 1 in normalised coordinates) over the inliers: Hartley normalisation of both point sets,
// normal equations of the 2k x 8 system in fp64, de-normalised and scaled to H[8] = 1.
bool refit_homography(const float* src, const float* dst, const uint8_t* mask, int n, double* H) {
    int k = 0;
    double cs[2] = {0, 0}, cd[2] = {0, 0};
    for (int i = 0; i < n; ++i)
        if (mask[i]) {
            cs[0] += src[2 * i];
            cs[1] += src[2 * i + 1];
            cd[0] += dst[2 * i];
            cd[1] += dst[2 * i + 1];
            ++k;
        }
    if (k < 4) return false;
    for (int q = 0; q < 2; ++q) {
        cs[q] /= k;
        cd[q] /= k;
    }
    double ms = 0, md = 0;
    for (int i = 0; i < n; ++i)
        if (mask[i]) {
            ms += std::sqrt((src[2 * i] - cs[0]) * (src[2 * i] - cs[0]) + (src[2 * i + 1] - cs[1]) * (src[2 * i + 1] - cs[1]));
            md += std::sqrt((dst[2 * i] - cd[0]) * (dst[2 * i] - cd[0]) + (dst[2 * i + 1] - cd[1]) * (dst[2 * i + 1] - cd[1]));
        }
    if (ms <= 0 || md <= 0) return false;
    const double ss = std::sqrt(2.0) * k / ms, sd = std::sqrt(2.0) * k / md;
    std::vector<double> N(64, 0.0), r(8, 0.0);
    for (int i = 0; i < n; ++i) {
        if (!mask[i]) continue;
        const double x = (src[2 * i] - cs[0]) * ss, y = (src[2 * i + 1] - cs[1]) * ss;
        const double u = (dst[2 * i] - cd[0]) * sd, v = (dst[2 * i + 1] - cd[1]) * sd;
        const double r0[8] = {x, y, 1, 0, 0, 0, -u * x, -u * y}, r1[8] = {0, 0, 0, x, y, 1, -v * x, -v * y};
        for (int a = 0; a < 8; ++a) {
            for (int b = 0; b < 8; ++b) N[a * 8 + b] += r0[a] * r0[b] + r1[a] * r1[b];
            r[a] += r0[a] * u + r1[a] * v;
        }
    }
    if (!solve_dense(N, r, 8)) return false;
    const double Hn[9] = {r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], 1.0};
    // H = Td^-1 Hn Ts with Ts = [ss 0 -ss cs0; 0 ss -ss cs1; 0 0 1], Td^-1 = [1/sd 0 cd0; 0 1/sd cd1; 0 0 1]
    double M[9];
    for (int a = 0; a < 3; ++a) {
        M[a * 3] = Hn[a * 3] * ss;
        M[a * 3 + 1] = Hn[a * 3 + 1] * ss;
        M[a * 3 + 2] = -Hn[a * 3] * ss * cs[0] - Hn[a * 3 + 1] * ss * cs[1] + Hn[a * 3 + 2];
    }
    double G[9];
    for (int q = 0; q < 3; ++q) {
        G[q] = M[q] / sd + cd[0] * M[6 + q];
        G[3 + q] = M[3 + q] / sd + cd[1] * M[6 + q];
        G[6 + q] = M[6 + q];
    }
    if (std::fabs(G[8]) < 1e-300) return false;
    for (int q = 0; q < 9; ++q) H[q] = G[q] / G[8];
    return true;
}

}  // namespace

// The corner search of n frames already in device memory, on `st`; the caller holds the handle's lock (or is the helper thread
// vq_flow_warped starts while it holds it).  Touches only the corner planes, frame_max and the pinned peak buffer.
static int good_features_core(vq_flow* f, const uint8_t* d, int n, int max_corners, float quality, float min_distance, float* corners_host,
                              int32_t* counts_host, hipStream_t st) {
    const int h = f->h, w = f->w;
    const int64_t full = (int64_t)n * h * w;
    for (float*& p : f->corner_plane)
        if (!p) VQ_HIP(vq::malloc_trim((void**)&p, (size_t)f->max_pairs * h * w * sizeof(float)));
    float *strength = f->corner_plane[0], *peaks = f->corner_plane[1];
    VQ_HIP(hipMemsetAsync(f->frame_max, 0, (size_t)n * sizeof(unsigned), st));
    corner_strength_kernel<<<dim3((unsigned)cdiv((int64_t)h * w, 256), (unsigned)n), 256, 0, st>>>(d, strength, f->frame_max, n, h, w);
    corner_peaks_kernel<<<cdiv(full, 256), 256, 0, st>>>(strength, peaks, n, h, w);
    VQ_CHECK_LAUNCH();
    if (f->peaks_host_floats < (size_t)f->max_pairs * h * w) {          // pinned, once: 22 MB per batch of 64 frames come back through it
        if (f->peaks_host) (void)hipHostFree(f->peaks_host);
        f->peaks_host = nullptr;
        f->peaks_host_floats = 0;
        VQ_HIP(hipHostMalloc((void**)&f->peaks_host, (size_t)f->max_pairs * h * w * sizeof(float)));
        f->peaks_host_floats = (size_t)f->max_pairs * h * w;
    }
    float* host_peaks = f->peaks_host;
    std::vector<unsigned> top((size_t)n);
    VQ_HIP(hipMemcpyAsync(host_peaks, peaks, (size_t)full * sizeof(float), hipMemcpyDeviceToHost, st));
    VQ_HIP(hipMemcpyAsync(top.data(), f->frame_max, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));
    // the selection is per frame and sequential inside a frame: frames are spread over host threads (csrc/host/vq_corners.cc)
    vq::select_corners_batch(host_peaks, top.data(), n, h, w, max_corners, quality, min_distance, corners_host, counts_host);
    return VQ_OK;
}

extern "C" {

int vq_flow_good_features(vq_flow* f, const uint8_t* frames, int32_t frames_on_device, int32_t n, int32_t max_corners, float quality,
                          float min_distance, float* corners_host, int32_t* counts_host, void* hip_stream) {
    VQ_REQUIRE(f && frames && corners_host && counts_host, "NULL argument");
    VQ_REQUIRE(n > 0 && n <= f->max_pairs, "n %d outside (0,%d]", n, f->max_pairs);
    VQ_REQUIRE(max_corners > 0 && quality > 0.f && quality < 1.f && min_distance >= 0.f, "corner parameters out of range");
    std::lock_guard<std::recursive_mutex> lk(f->mu);
    DeviceGuard g(f->device);
    hipStream_t st = (hipStream_t)hip_stream;
    const uint8_t* d = frames;
    if (!frames_on_device) {
        VQ_HIP(hipMemcpyAsync(f->frames_dev[0], frames, (size_t)n * f->h * f->w, hipMemcpyHostToDevice, st));
        d = f->frames_dev[0];
    }
    return good_features_core(f, d, n, max_corners, quality, min_distance, corners_host, counts_host, st);
}

int vq_flow_ransac_homography(vq_flow* f, const float* src_host, const float* dst_host, const int32_t* counts_host, int32_t n,
                              int32_t max_points, float threshold, int32_t hypotheses, uint32_t seed, int32_t refit, double* h_host,
                              int32_t* inliers_host, int32_t* winner_host, uint8_t* mask_host, void* hip_stream) {
    VQ_REQUIRE(f && src_host && dst_host && counts_host && h_host && inliers_host, "NULL argument");
    VQ_REQUIRE(n > 0 && max_points >= 4 && max_points <= 8192 && hypotheses > 0 && hypotheses <= (1 << 20) && threshold > 0.f,
               "RANSAC parameters out of range (at most 8192 matches per pair)");
    for (int p = 0; p < n; ++p) VQ_REQUIRE(counts_host[p] >= 0 && counts_host[p] <= max_points, "pair %d: %d matches of at most %d", p, counts_host[p], max_points);
    std::lock_guard<std::recursive_mutex> lk(f->mu);
    DeviceGuard g(f->device);
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t pts_b = (size_t)n * max_points * 2 * sizeof(float);
    const size_t need = 2 * pts_b + (size_t)n * (3 * sizeof(int) + 9 * sizeof(double)) + (size_t)n * max_points + 64;
    if (need > f->match_bytes) {
        if (f->match_dev) (void)hipFree(f->match_dev);
        f->match_dev = nullptr;
        f->match_bytes = 0;
        VQ_HIP(vq::malloc_trim(&f->match_dev, need));
        f->match_bytes = need;
    }
    char* base = (char*)f->match_dev;
    double* h_dev = (double*)base;                                   // 8-byte aligned things first
    float* src_dev = (float*)(base + (size_t)n * 9 * sizeof(double));
    float* dst_dev = (float*)((char*)src_dev + pts_b);
    int* cnt_dev = (int*)((char*)dst_dev + pts_b);
    int* best_dev = cnt_dev + n;
    int* win_dev = best_dev + n;
    uint8_t* mask_dev = (uint8_t*)(win_dev + n);
    VQ_HIP(hipMemcpyAsync(src_dev, src_host, pts_b, hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(dst_dev, dst_host, pts_b, hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(cnt_dev, counts_host, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    const size_t lds = (size_t)max_points * 4 * sizeof(float);
    VQ_DYN_LDS(ransac_homography_kernel, 8192 * 16);
    ransac_homography_kernel<<<n, 256, lds, st>>>(src_dev, dst_dev, cnt_dev, max_points, hypotheses, seed, (double)threshold * (double)threshold,
                                                  h_dev, best_dev, win_dev, mask_dev);
    VQ_CHECK_LAUNCH();
    std::vector<uint8_t> mask((size_t)n * max_points);
    std::vector<int> win((size_t)n);
    VQ_HIP(hipMemcpyAsync(h_host, h_dev, (size_t)n * 9 * sizeof(double), hipMemcpyDeviceToHost, st));
    VQ_HIP(hipMemcpyAsync(inliers_host, best_dev, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st));
    VQ_HIP(hipMemcpyAsync(win.data(), win_dev, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st));
    VQ_HIP(hipMemcpyAsync(mask.data(), mask_dev, mask.size(), hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));
    if (refit)
        for (int p = 0; p < n; ++p)
            if (inliers_host[p] >= 4) {
                double H[9];
                if (refit_homography(src_host + (size_t)p * max_points * 2, dst_host + (size_t)p * max_points * 2, mask.data() + (size_t)p * max_points,
                                     counts_host[p], H))
                    memcpy(h_host + (size_t)p * 9, H, sizeof H);
            }
    if (winner_host) memcpy(winner_host, win.data(), (size_t)n * sizeof(int));
    if (mask_host) memcpy(mask_host, mask.data(), mask.size());
    return VQ_OK;
}

// The warped flow of extract_warp_gpu (flow-match branch) in one call, the frames uploaded once and the first-pass fields never leaving the
// device: TV-L1 -> Shi-Tomasi corners of the first frame (selection on host threads) -> the corners moved by the flow (device) -> RANSAC
// homography (device kernel + host refit) with dense_flow's guards (> 50 matches, > 25 inliers, else identity) -> the second frame warped
// back by it -> TV-L1 again.  tsn/flow.py:Tvl1Flow.warped_steps is the same sequence call by call (tests compare the two).
int vq_flow_warped(vq_flow* f, const uint8_t* frames0, const uint8_t* frames1, int32_t n_pairs, uint32_t seed, int32_t hypotheses, float* u1_host,
                   float* u2_host, uint8_t* flow_x_host, uint8_t* flow_y_host, double* h_host, int32_t* matches_host, int32_t* inliers_host,
                   void* hip_stream) {
    VQ_REQUIRE(f && frames0 && frames1, "NULL argument");
    VQ_REQUIRE(n_pairs > 0 && n_pairs <= f->max_pairs, "n_pairs %d outside (0,%d]", n_pairs, f->max_pairs);
    std::lock_guard<std::recursive_mutex> lk(f->mu);
    DeviceGuard g(f->device);
    hipStream_t st = (hipStream_t)hip_stream;
    constexpr int kMaxCorners = 1000, kMinMatches = 50, kMinInliers = 25;
    const int n = n_pairs, h = f->h, w = f->w;
    std::vector<float> corners((size_t)n * kMaxCorners * 2), moved((size_t)n * kMaxCorners * 2);
    std::vector<int32_t> counts((size_t)n), inl((size_t)n);
    // The corners depend on the first frames alone: their search (two small kernels, 22 MB of peak maps to the host, the selection on
    // host threads) runs on a stream and a thread of its own beside the first flow pass instead of between the two passes.
    if (!f->side_stream) VQ_HIP(hipStreamCreateWithFlags(&f->side_stream, hipStreamNonBlocking));
    if (!f->side_ev) VQ_HIP(hipEventCreateWithFlags(&f->side_ev, hipEventDisableTiming));
    const size_t full = (size_t)n * h * w;
    VQ_HIP(hipMemcpyAsync(f->frames_dev[0], frames0, full, hipMemcpyHostToDevice, st));
    VQ_HIP(hipEventRecord(f->side_ev, st));
    VQ_HIP(hipMemcpyAsync(f->frames_dev[1], frames1, full, hipMemcpyHostToDevice, st));
    VQ_HIP(hipStreamWaitEvent(f->side_stream, f->side_ev, 0));
    int rc_corners = VQ_OK;
    std::string err_corners;
    std::thread side([&] {
        DeviceGuard gs(f->device);
        rc_corners = good_features_core(f, f->frames_dev[0], n, kMaxCorners, 0.001f, 3.0f, corners.data(), counts.data(), f->side_stream);
        if (rc_corners != VQ_OK) err_corners = last_error_ref();        // the message lives in the helper thread's slot
    });
    int rc = vq_flow_tvl1(f, f->frames_dev[0], f->frames_dev[1], 1, n_pairs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, hip_stream);
    side.join();
    if (rc != VQ_OK) return rc;
    if (rc_corners != VQ_OK) return fail(rc_corners, "%s", err_corners.c_str());
    const size_t cb = corners.size() * sizeof(float);
    const size_t need = 2 * cb + (size_t)n * sizeof(int);
    if (need > f->warp_bytes) {
        if (f->warp_dev) (void)hipFree(f->warp_dev);
        f->warp_dev = nullptr;
        f->warp_bytes = 0;
        VQ_HIP(vq::malloc_trim(&f->warp_dev, need));
        f->warp_bytes = need;
    }
    float* c_dev = (float*)f->warp_dev;
    float* m_dev = (float*)((char*)f->warp_dev + cb);
    int* n_dev = (int*)((char*)f->warp_dev + 2 * cb);
    VQ_HIP(hipMemcpyAsync(c_dev, corners.data(), cb, hipMemcpyHostToDevice, st));
    VQ_HIP(hipMemcpyAsync(n_dev, counts.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    move_corners_kernel<<<cdiv((int64_t)n * kMaxCorners, 256), 256, 0, st>>>(c_dev, n_dev, f->plane[6], f->plane[7], m_dev, n, kMaxCorners, h, w);
    VQ_CHECK_LAUNCH();
    VQ_HIP(hipMemcpyAsync(moved.data(), m_dev, cb, hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));
    std::vector<double> H((size_t)n * 9), Hinv((size_t)n * 9);
    rc = vq_flow_ransac_homography(f, corners.data(), moved.data(), counts.data(), n, kMaxCorners, 1.0f, hypotheses, seed, 1, H.data(), inl.data(), nullptr,
                                   nullptr, hip_stream);
    if (rc != VQ_OK) return rc;
    for (int p = 0; p < n; ++p) {
        double* m = H.data() + (size_t)p * 9;
        double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        if (counts[p] <= kMinMatches || inl[p] <= kMinInliers || !(std::fabs(det) > 1e-300)) {
            const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            memcpy(m, eye, sizeof eye);
            det = 1.0;
        }
        // vq_flow_tvl1(homographies = G) shows the second frame as out(x) = frame1(G^-1 x); the compensated frame is frame1(H x): G = H^-1
        double* o = Hinv.data() + (size_t)p * 9;
        o[0] = (m[4] * m[8] - m[5] * m[7]) / det;
        o[1] = (m[2] * m[7] - m[1] * m[8]) / det;
        o[2] = (m[1] * m[5] - m[2] * m[4]) / det;
        o[3] = (m[5] * m[6] - m[3] * m[8]) / det;
        o[4] = (m[0] * m[8] - m[2] * m[6]) / det;
        o[5] = (m[2] * m[3] - m[0] * m[5]) / det;
        o[6] = (m[3] * m[7] - m[4] * m[6]) / det;
        o[7] = (m[1] * m[6] - m[0] * m[7]) / det;
        o[8] = (m[0] * m[4] - m[1] * m[3]) / det;
    }
    if (h_host) memcpy(h_host, H.data(), H.size() * sizeof(double));
    if (matches_host) memcpy(matches_host, counts.data(), (size_t)n * sizeof(int32_t));
    if (inliers_host) memcpy(inliers_host, inl.data(), (size_t)n * sizeof(int32_t));
    return vq_flow_tvl1(f, f->frames_dev[0], f->frames_dev[1], 1, n, Hinv.data(), u1_host, u2_host, flow_x_host, flow_y_host, nullptr, hip_stream);
}

}  // extern "C"
