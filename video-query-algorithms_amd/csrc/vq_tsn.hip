// Hot path A on gfx950: the frozen TSN BN-Inception forward pass + segment consensus.
//
// What it replaces (paths relative to the reference checkout):
//   src/features_GPU_compute/calcSig_wOF.py:88-113  per-snippet CaffeNet forward (rgb / flow)
//   src/features_GPU_compute/calcSig_wOF.py:82      np.array(frame_features).mean(axis=0)  (consensus)
//   src/features_GPU_compute/models/ucf101/tsn_bn_inception_{rgb,flow}_deploy.prototxt  (the layers)
//
// Layout: activations NHWC fp32 in HBM, one tensor slot per blob; a Concat output is one slot and
// its producers write at their channel offset (zero-copy concat).  Convolution + frozen BN + ReLU is a
// single kernel launch over all B*T crops of a batch:
//   * direct form (this file): implicit GEMM, M = crops*Ho*Wo pixels, N = Cout, K = k*k*Cin, on the fp32
//     matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, 256 FLOP/clk/CU) from LDS-staged,
//     register-prefetched tiles -- the 1x1, stride-2 and stem layers;
//   * Winograd F(2x2,3x3) form (vq_wino.hip) -- the 3x3 / stride-1 layers.
// The executor below validates the plan once, autotunes the tiling per layer and batch size, and runs the
// layer list on one stream (profiling) or as sub-batches on several (production).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <tuple>
#include <type_traits>
#include <vector>

#include "vq_common.h"
#include "vq_tsn_kernels.h"
#include "host/vq_block_pool.h"

using namespace vq;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((__vector_size__(4 * sizeof(unsigned))));

// ------------------------------------------------------------------------------------------------
// preprocess: uint8 NHWC crops -> fp32 NHWC (channels padded to a multiple of 4), minus mean
// ------------------------------------------------------------------------------------------------
// One thread per 16-byte chunk of the output (4 consecutive channels of one pixel): coalesced float4 stores.
__global__ void preprocess_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t nchunk, int C, int Cpad,
                                  const float* __restrict__ mean) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nchunk) return;
    const int cpp = Cpad >> 2;                      // chunks per pixel
    const int64_t p = i / cpp;
    const int c0 = (int)(i - p * cpp) * 4;
    floatx4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = c0 + e < C ? (float)src[p * C + c0 + e] - mean[c0 + e] : 0.0f;
    *reinterpret_cast<floatx4*>(dst + p * Cpad + c0) = v;
}

// Space-to-depth(2) variant (vq_input_desc.s2d_pad >= 0): slot0[n][Y][X][(p*2+q)*C + c] = crop[2Y+p-pad][2X+q-pad][c] -
// mean[c], zeros outside the crop (the first convolution's own zero padding, applied after the mean like Caffe does).
// One thread per 16-byte chunk of the 4*C output channels of a cell.
__global__ void preprocess_s2d_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t nchunk, int H, int W, int C,
                                      int Hs, int Ws, int pad, const float* __restrict__ mean, int xmajor) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nchunk) return;
    const int64_t cell = i / C;                     // 4*C channels = C chunks of 4 per cell
    const int j0 = (int)(i - cell * C) * 4;
    const int X = (int)(cell % Ws), Y = (int)((cell / Ws) % Hs);
    const int64_t n = cell / ((int64_t)Ws * Hs);
    floatx4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = j0 + e, pq = j / C, c = j - pq * C;
        const int p = xmajor ? pq & 1 : pq >> 1, q = xmajor ? pq >> 1 : pq & 1;       // s2d_order 1: cell channels (q*2+p)*C + c
        const int y = 2 * Y + p - pad, x = 2 * X + q - pad;
        const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        v[e] = in ? (float)src[((n * H + y) * W + x) * C + c] - mean[c] : 0.0f;
    }
    *reinterpret_cast<floatx4*>(dst + cell * (4 * C) + j0) = v;
}

// The RGB stem's form (C = 3): one thread per CELL -- its 2 x 2 pixels are two runs of 6 bytes, its 12 floats three 16-byte
// stores; two divisions per thread instead of a dozen per chunk (32 us -> 17 us per 96-crop step).  Same values.
__global__ __launch_bounds__(256) void preprocess_s2d3_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t ncell, int H, int W,
                                                              int Hs, int Ws, int pad, const float* __restrict__ mean, int xmajor) {
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= ncell) return;
    const int X = (int)(cell % Ws);
    const int64_t ny = cell / Ws;
    const int Y = (int)(ny % Hs);
    const int64_t n = ny / Hs;
    const float m0 = mean[0], m1 = mean[1], m2 = mean[2];
    float v[12];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int y = 2 * Y + p - pad;
        const bool row_in = (unsigned)y < (unsigned)H;
        const uint8_t* r = src + ((n * H + (row_in ? y : 0)) * W) * 3;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int x = 2 * X + q - pad;
            const bool in = row_in && (unsigned)x < (unsigned)W;
            const uint8_t* px = r + (in ? x : 0) * 3;
            const float a = (float)px[0] - m0, b = (float)px[1] - m1, c = (float)px[2] - m2;
            const int slot = xmajor ? q * 2 + p : p * 2 + q;        // s2d_order 1: (q, p, c) -- the x-taps of a kernel row side by side
            v[slot * 3 + 0] = in ? a : 0.0f;
            v[slot * 3 + 1] = in ? b : 0.0f;
            v[slot * 3 + 2] = in ? c : 0.0f;
        }
    }
    floatx4* o = reinterpret_cast<floatx4*>(dst + cell * 12);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        floatx4 t;
        t[0] = v[4 * k], t[1] = v[4 * k + 1], t[2] = v[4 * k + 2], t[3] = v[4 * k + 3];
        o[k] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// convolution (+ folded BN + ReLU) as implicit GEMM on v_mfma_f32_32x32x2_f32
// ------------------------------------------------------------------------------------------------
// Where a group of 32 output columns of a convolution goes.  One convolution may feed several tensors (sibling 1x1
// convolutions of an inception block merged into one GEMM): the table has one entry per 32-column group.
struct ConvSeg {
    float* out_base;     // destination slot + its channel offset - first column of the segment: index with the GEMM column
    int Cs;              // channel stride of the destination slot
    int relu;
};

struct ConvArgs {
    const float* in;
    const ConvSeg* segs; // [ceil(Cout / 32)]
    float* out;
    const float* w;      // [Cout][Kp], k index = (kh*k + kw)*Cin + c, zero padded to Kp
    const float* bias;   // [Cout]
    const float* zeros;  // >= 16 bytes of zeros: where masked-off lanes load from (no branches around loads)
    int H, W, Cs_in, coff_in, Cin;
    int Ho, Wo, Cs_out, coff_out, Cout;
    int k, stride, pad;  // k kernel rows ...
    int kw;              // ... of kw taps each (= k, or 1 when a kernel row is folded into the channel axis, see launch_conv_layer)
    int col_off;         // aligned mode: floats between the 16-byte chunk columns of a staged row (4 = consecutive channels), and
    int c_step;          // floats the channel offset advances per K-step (BK).  The x-major space-to-depth stem stages ROW r of its
                         // four kernel rows in chunk column r instead: col_off = the slot's row pitch, c_step = 4 (launch_conv_layer)
    int M, Kp, relu;
    int out_row0;        // first output row (pixel) of this launch inside the destination slots (sub-batch launches)
    int pool_k, pool_s;  // POOL kernels: the 1x1 convolution reads max over a pool_k x pool_k window (stride pool_s, no padding,
                         // clipped at the image edge) of the H x W source instead of a pixel; Ho x Wo is the pooled size
    int tiles_m, tiles_n;
    unsigned in_bytes, w_bytes;   // extents for the buffer descriptors (out-of-range offsets read as zero)
    int ksplit;          // > 1: the grid is ksplit x tiles; slice ks sums K floats [ks * k_per, (ks + 1) * k_per) only and stores
    int k_per;           // its raw partial sums (zero bias, no ReLU) through segs[ks * seg_stride ..] into a scratch plane;
    int seg_stride;      // splitk_combine_kernel adds the planes in slice order, then bias and ReLU.  Aligned Cin only.
    // conv_igemm_pipe_kernel: the pixel decoding of a workgroup without a division (fill_pixel_walk)
    unsigned m_tiles_n;            // magic_div constant for tiles_n
    FullDiv d_howo, d_wo;          // first pixel of the tile -> (image, row, column) on the scalar unit
    unsigned s_wo, s_ho;           // recip22 constants: the per-lane walk of at most BM pixels from there
};

constexpr int KPAD = 32;       // weights are packed [Cout][Kp] with Kp a multiple of 32
// ... except the x-major space-to-depth stem (7x7 / stride 2 over c channels): four kernel rows of 7 x-taps x 2 rows x c floats each,
// every row padded to whole 16-byte chunks, walked side by side (launch_conv_layer)
static inline int stem_rows_kp(int c) { return 4 * ((14 * c + 3) / 4 * 4); }

template <int BM, int BN, int BK>
struct ConvSmem {
    // padded rows (BK + 4 floats): the 16 rows one ds_read_b128 lane group touches land on 16 distinct
    // 16-byte bank slots for BK = 16 and BK = 32 alike
    float a[2][BM][BK + 4];
    float b[2][BN][BK + 4];
};

// Epilogue shared by both kernels: + bias (folded BN), ReLU, store at the channel offset of the destination slot.
// In the MFMA C/D layout a lane holds ONE channel of 16 pixels (column = lane & 31, row = (r & 3) + 8 (r >> 2) +
// 4 (lane >> 5)), which would mean 16 four-byte stores per accumulator tile; the store tail of such a wave is
// issue-bound and cost ~20 % of a K = 576 tile.  Each wave therefore transposes its 32x32 tile through a private
// 4.5 KB LDS patch and writes 16 bytes per lane: 4 store instructions per tile, each 8 pixels x 128 bytes.
#define VQ_EPILOGUE()                                                                                             \
    {                                                                                                             \
        __syncthreads(); /* every wave is done with the K-loop image of the LDS */                               \
        float* patch = reinterpret_cast<float*>(smem_raw) + wave * (32 * 36);                                     \
        const int prow = lane >> 3, pc4 = lane & 7;                                                               \
        const bool full_m = m0 + BM <= a.M;                                                                       \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                          \
            const int nb = n0 + wn * (BN / WN) + 32 * j;                                                          \
            if (nb < a.Cout) {                                                                                    \
                const ConvSeg sg = seg_v[j];                                                                      \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                  \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                \
                        patch[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + l31] = acc[i][j][r];                     \
                    const int mb = m0 + wm * (BM / WM) + 32 * i;                                                  \
                    _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                            \
                        const int row = prow + 8 * ps;                                                            \
                        floatx4 v = *reinterpret_cast<const floatx4*>(&patch[row * 36 + pc4 * 4]);                \
                        v += bias_v[j];                                                                           \
                        if (sg.relu) {                                                                            \
                            v[0] = fmaxf(v[0], 0.f);                                                              \
                            v[1] = fmaxf(v[1], 0.f);                                                              \
                            v[2] = fmaxf(v[2], 0.f);                                                              \
                            v[3] = fmaxf(v[3], 0.f);                                                              \
                        }                                                                                         \
                        if ((full_m || mb + row < a.M) && nb + pc4 * 4 < a.Cout)                                  \
                            *reinterpret_cast<floatx4*>(sg.out_base + (size_t)(a.out_row0 + mb + row) * sg.Cs + nb + pc4 * 4) = v; \
                    }                                                                                             \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
    }

// The same epilogue with 32-bit offsets into a buffer descriptor per destination segment (conv_igemm_pipe_kernel, whose wave index is a
// scalar): the row / segment part of an address is uniform and rides in the store's scalar offset, the lane part (row in the 8-row group,
// 16-byte column) is formed once per segment, a masked-off lane stores at 0xFFFFFFFF (dropped by the range check).  No 64-bit vector
// arithmetic, no multiply per store.  A launch never spans 2^31 bytes of a destination slot (LaunchItem::max_crops).
#define VQ_EPILOGUE_BUF()                                                                                         \
    {                                                                                                             \
        __syncthreads(); /* every wave is done with the K-loop image of the LDS */                               \
        float* patch = reinterpret_cast<float*>(smem_raw) + wave * (32 * 36);                                     \
        const int prow = lane >> 3, pc4 = lane & 7;                                                               \
        const bool full_m = m0 + BM <= a.M;                                                                       \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                          \
            const int nb = n0 + wn * (BN / WN) + 32 * j;                                                          \
            if (nb < a.Cout) {                                                                                    \
                const ConvSeg sg = seg_v[j];                                                                      \
                const __amdgpu_buffer_rsrc_t o_rsrc =                                                             \
                    __builtin_amdgcn_make_buffer_rsrc(sg.out_base + (size_t)a.out_row0 * sg.Cs, 0, 0x7FFFFFF0u, 0x00020000); \
                const unsigned lane_off = nb + pc4 * 4 < a.Cout ? (unsigned)((imul24(prow, sg.Cs) + pc4 * 4) * 4) : 0xFFFFFFFFu; \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                  \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                \
                        patch[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + l31] = acc[i][j][r];                     \
                    const int mb = m0 + wm * (BM / WM) + 32 * i;                                                  \
                    _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                            \
                        const int row = prow + 8 * ps;                                                            \
                        floatx4 v = *reinterpret_cast<const floatx4*>(&patch[row * 36 + pc4 * 4]);                \
                        v += bias_v[j];                                                                           \
                        if (sg.relu) {                                                                            \
                            v[0] = fmaxf(v[0], 0.f);                                                              \
                            v[1] = fmaxf(v[1], 0.f);                                                              \
                            v[2] = fmaxf(v[2], 0.f);                                                              \
                            v[3] = fmaxf(v[3], 0.f);                                                              \
                        }                                                                                         \
                        const unsigned off = (full_m || mb + row < a.M) ? lane_off : 0xFFFFFFFFu;                 \
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), o_rsrc, off,        \
                                                               ((mb + 8 * ps) * sg.Cs + nb) * 4, 0);              \
                    }                                                                                             \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
    }

// One workgroup = 4 waves = one BM x BN output tile; K is walked in steps of BK through a double-buffered LDS
// image that is filled through registers (global -> VGPR under the MFMAs of the current step -> LDS).
template <int BM, int BN, int WM, int WN, int BK, bool SMALL_CIN, bool POOL = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BK == 16 || BK == 32, "BK is 16 or 32");
    static_assert(!(POOL && SMALL_CIN), "the pooled loader walks whole channel chunks");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;   // 32x32 accumulator tiles per wave
    constexpr int CPR = BK / 4;                            // 16-byte chunks per staged row
    constexpr int NA = (BM * CPR + 255) / 256, NB = (BN * CPR + 255) / 256;   // chunks staged per thread
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ConvSmem<BM, BN, BK>& sm = *reinterpret_cast<ConvSmem<BM, BN, BK>*>(smem_raw);

    const int tid = threadIdx.x;
    const int n_tiles = a.tiles_m * a.tiles_n;
    const int ks = a.ksplit > 1 ? (int)blockIdx.x / n_tiles : 0;            // K slice of this workgroup (uniform)
    const int kbeg = ks * a.k_per;
    const int tile = xcd_remap(blockIdx.x - ks * n_tiles, n_tiles);
    const int m0 = (tile / a.tiles_n) * BM;
    const int n0 = (tile % a.tiles_n) * BN;

    // ---- staging roles: chunk id = tid + 256 i -> (row, 16-byte column).
    // Rows past M get an input row far outside the image, so the ordinary bounds test routes them to the
    // zero page; every pointer below stays derived from a kernel argument (global address space -- a null
    // alternative would demote the loads to flat_load, which also counts against lgkmcnt).
    const float* a_img[NA];
    int a_ih0[NA], a_iw0[NA], a_row[NA], a_col[NA];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = tid + 256 * i;
        a_row[i] = c / CPR;
        a_col[i] = c % CPR;
        const int m = m0 + a_row[i];
        const bool ok = m < a.M && a_row[i] < BM;
        const int mm = ok ? m : 0;
        const int n_img = mm / HoWo, rem = mm - n_img * HoWo;
        const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
        a_img[i] = a.in + (size_t)n_img * a.H * a.W * a.Cs_in + a.coff_in;
        a_ih0[i] = ok ? oh * a.stride - a.pad : -(1 << 20);
        a_iw0[i] = ow * a.stride - a.pad;
        if (POOL) {   // first row / column of the pooling window (always inside the image: Caffe's ceil rule)
            a_ih0[i] = ok ? oh * a.pool_s : -(1 << 20);
            a_iw0[i] = ow * a.pool_s;
        }
    }
    const float* b_ptr[NB];
    bool b_ok[NB];
    int b_row[NB], b_col[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int c = tid + 256 * i;
        b_row[i] = c / CPR;
        b_col[i] = c % CPR;
        const int n = n0 + b_row[i];
        b_ok[i] = n < a.Cout && b_row[i] < BN;
        b_ptr[i] = a.w + (size_t)(b_ok[i] ? n : 0) * a.Kp + b_col[i] * 4 + kbeg;
    }

    floatx4 ra[NA], rb[NB];   // ext-vector values (a HIP float4 struct copy becomes a memcpy the optimiser leaves in scratch)
    constexpr int PT = POOL ? 8 : 1;
    floatx4 rp[NA][PT];       // POOL: the other window taps, in flight under the MFMAs; folded into ra when the tile is stored
    int kh = 0, kw = 0, c0 = 0;   // aligned mode: walk (tap, channel chunk) without divisions
    if (!SMALL_CIN && kbeg > 0) {  // a later K slice starts its walk in the middle
        const int tap0 = kbeg / a.Cin;
        c0 = kbeg - tap0 * a.Cin;
        kh = tap0 / a.kw;
        kw = tap0 - kh * a.kw;
    }
    // Every lane always issues its loads: lanes that fall into the zero padding (or past M / Cout) read the
    // zero page instead, so there is no branch and no exec-mask juggling around the global loads.
    // (Macros, not lambdas: by-reference captures of the staging arrays ended up in scratch memory.)
#define VQ_LOAD_TILES(KC)                                                                                          \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                           \
            const float* src = a.zeros;                                                                            \
            if (SMALL_CIN) {                                                                                       \
                const int kidx = ((KC) * CPR + a_col[i]) * 4;                                                      \
                const int tap = kidx / a.Cin, c = kidx - tap * a.Cin;                                              \
                const int th = tap / a.kw, tw = tap - th * a.kw;                                                   \
                const int ih = a_ih0[i] + th, iw = a_iw0[i] + tw;                                                  \
                if (tap < a.k * a.kw && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)              \
                    src = a_img[i] + ((size_t)ih * a.W + iw) * a.Cs_in + c;                                        \
            } else if (POOL) {                                                                                     \
                /* 3x3 window: taps outside the image re-read the window's first pixel (max is unchanged) */      \
                const bool ok = a_ih0[i] >= 0;                                                                     \
                if (ok) src = a_img[i] + ((size_t)a_ih0[i] * a.W + a_iw0[i]) * a.Cs_in + c0 + a_col[i] * a.col_off; \
                _Pragma("unroll") for (int t = 1; t < 9; ++t) {                                                    \
                    const int ih = a_ih0[i] + t / 3, iw = a_iw0[i] + t % 3;                                        \
                    const float* q = src;                                                                          \
                    if (ok && ih < a.H && iw < a.W) q = a_img[i] + ((size_t)ih * a.W + iw) * a.Cs_in + c0 + a_col[i] * a.col_off; \
                    rp[i][t - 1] = *reinterpret_cast<const floatx4*>(q);                                           \
                }                                                                                                  \
            } else {                                                                                               \
                const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;                                                  \
                if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)                                  \
                    src = a_img[i] + ((size_t)ih * a.W + iw) * a.Cs_in + c0 + a_col[i] * a.col_off;                \
            }                                                                                                      \
            ra[i] = *reinterpret_cast<const floatx4*>(src);                                                        \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                             \
            rb[i] = *reinterpret_cast<const floatx4*>(b_ok[i] ? b_ptr[i] + (size_t)(KC) * BK : a.zeros);           \
        if (!SMALL_CIN) {                                                                                          \
            c0 += a.c_step ? a.c_step : BK;                                                                        \
            if (c0 >= a.Cin) {                                                                                     \
                c0 = 0;                                                                                            \
                if (++kw == a.kw) {                                                                                \
                    kw = 0;                                                                                        \
                    ++kh;                                                                                          \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
    }
#define VQ_STORE_TILES(BUF)                                                                                        \
    {                                                                                                              \
        if (POOL) {                                                                                                \
            _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                         \
                _Pragma("unroll") for (int t = 0; t < 8; ++t)                                                      \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) ra[i][e] = fmaxf(ra[i][e], rp[i][t][e]);         \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                             \
            if ((BM * CPR) % 256 == 0 || a_row[i] < BM)                                                            \
                *reinterpret_cast<floatx4*>(&sm.a[BUF][a_row[i]][a_col[i] * 4]) = ra[i];                           \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                             \
            if ((BN * CPR) % 256 == 0 || b_row[i] < BN)                                                            \
                *reinterpret_cast<floatx4*>(&sm.b[BUF][b_row[i]][b_col[i] * 4]) = rb[i];                           \
    }

    // ---- compute roles
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int arow0 = wm * (BM / WM) + l31, brow0 = wn * (BN / WN) + l31;
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // folded-BN bias of this lane's output columns: fetched now so that its latency (and its vmcnt wait)
    // is long gone when the epilogue starts -- a load inside the store loop would serialise every store
    floatx4 bias_v[TN];   // this lane's 4 output channels in the transposed store (see VQ_EPILOGUE)
    ConvSeg seg_v[TN];    // where each 32-column group of this wave goes (wave-uniform)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb_ = n0 + wn * (BN / WN) + 32 * j;
        const int n = nb_ + (lane & 7) * 4;
        bias_v[j] = *reinterpret_cast<const floatx4*>(a.bias + (n < a.Cout ? n : 0));
        seg_v[j] = a.segs[ks * a.seg_stride + (nb_ < a.Cout ? nb_ >> 5 : 0)];
    }

    const int nk = (a.ksplit > 1 ? a.k_per : a.Kp) / BK;
    VQ_LOAD_TILES(0)
    VQ_STORE_TILES(0)
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) VQ_LOAD_TILES(kc + 1)   // global loads in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            // lane (row r, half h) holds k = 8 kk + 4 h + j for MFMA step j: the two halves of a step cover a
            // k pair {8kk + j, 8kk + 4 + j}; A and B use the same assignment, so every k is summed once.
            floatx4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const floatx4*>(&sm.a[buf][arow0 + 32 * i][kk * 8 + half * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const floatx4*>(&sm.b[buf][brow0 + 32 * j][kk * 8 + half * 4]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (kc + 1 < nk) VQ_STORE_TILES(buf ^ 1)
        __syncthreads();
    }

    VQ_EPILOGUE()
#undef VQ_LOAD_TILES
#undef VQ_STORE_TILES
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined variant (aligned Cin only).  Same tiles, same k order (bit-identical results), but the
// instruction stream of a K-step is laid out by hand: every MFMA is followed by one small slice of the side
// work -- a fragment prefetch for the next 8-wide k group, one global load (with its address arithmetic) of the
// next tile, or one LDS store of it -- and a scheduling fence, so a wave keeps its SIMD's matrix pipe fed
// while it stages (an in-order wave cannot overlap a BLOCK of vector work with more than one MFMA).
// ------------------------------------------------------------------------------------------------
// (round 6: the 128 x 64 x 16 and 64 x 128 x 16 tilings need 134 registers left alone; held to 128 -- no spill -- they fit FOUR workgroups per
// compute unit instead of three: conv1 0.245 -> 0.239 ms, the cfg-2 step in product mode +2 % in alternating builds on one box.  The BK = 32
// tilings of the same area spill 140-232 bytes per lane under that limit and keep their 154-160 registers.)
#ifndef VQ_PIPE_WPE
#define VQ_PIPE_WPE 4
#endif
template <int BM, int BN, int WM, int WN, int BK, bool SMALL_CIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BM * BN == 128 * 64 && BK == 16 ? VQ_PIPE_WPE : 1, 8)))
void conv_igemm_pipe_kernel(ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int NT = 256;                    // threads per workgroup
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int CPR = BK / 4;
    constexpr int NA = (BM * CPR + NT - 1) / NT, NB = (BN * CPR + NT - 1) / NT;
    constexpr int NKK = BK / 8;              // 8-wide k groups per K-step
    constexpr int MFK = 4 * TM * TN;         // MFMAs per k group
    constexpr int NMF = NKK * MFK;           // MFMAs per K-step
    constexpr int NFR = TM + TN;             // fragment reads per k group
    constexpr int NG = NA + NB;              // global loads (= LDS stores) per K-step
    constexpr int NFREE = NMF - (NKK - 1) * NFR;
    static_assert(NFREE >= 2 * NG, "not enough MFMA gaps for the staging slices");
    // EARLY: the LDS stores of tile kc+1 are all issued before the last k group of step kc, so a barrier right behind
    // them lets that last group's gaps prefetch the first fragments of step kc+1: no LDS latency is exposed at the
    // start of a step.
    constexpr bool EARLY = (NKK - 1) * (MFK - NFR) >= NG && NFREE - NFR >= 2 * NG;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ConvSmem<BM, BN, BK>& sm = *reinterpret_cast<ConvSmem<BM, BN, BK>*>(smem_raw);

    const int tid = threadIdx.x;
    const int n_tiles = a.tiles_m * a.tiles_n;
    const int ks = a.ksplit > 1 ? (int)blockIdx.x / n_tiles : 0;            // K slice of this workgroup (uniform)
    const int kbeg = ks * a.k_per;
    const int tile = xcd_remap(blockIdx.x - ks * n_tiles, n_tiles);
    // Which pixels, which channels: on the scalar unit, with reciprocals from the host (launch_conv_pipe_t).  An integer division costs
    // ~30 vector instructions, a third of them at a quarter of the rate, and the vector unit's time is the matrix pipe's: the prologue
    // of a K = 256 tile used to be 11 % of its issue slots.
    const int tm = (int)magic_div((unsigned)tile, a.m_tiles_n);
    const int m0 = tm * BM;
    const int n0 = (tile - tm * a.tiles_n) * BN;
    const unsigned img0 = full_div((unsigned)m0, a.d_howo), rem0 = (unsigned)m0 - img0 * (unsigned)(a.Ho * a.Wo);
    const unsigned oh0 = full_div(rem0, a.d_wo), ow0 = rem0 - oh0 * (unsigned)a.Wo;

    // Staging addresses are 32-bit byte offsets into two buffer descriptors (activation slot, weights).  A lane
    // whose tap falls into the zero padding (or whose row is past M / Cout) uses offset 0xFFFFFFFF: the
    // hardware range check of buffer_load returns zeros, so a load costs one add and one select -- no 64-bit
    // address arithmetic, no branches, no zero page.  Which of the k*k taps are inside the image is a per-pixel
    // bit mask computed once per tile.
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
    int a_off[NA];            // byte offset of (pixel, tap (0,0), channel chunk) -- may be "negative" in the padding
    unsigned long long a_mask[NA];   // bit kh*k+kw set: that tap reads inside the image (k*k <= 64)
    int a_row[NA], a_col[NA];
    const unsigned HW = (unsigned)(a.H * a.W);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = tid + NT * i;
        a_row[i] = c / CPR;
        a_col[i] = c % CPR;
        const bool ok = m0 + a_row[i] < a.M && a_row[i] < BM;
        // the a_row-th pixel behind the tile's first one: a walk of < BM pixels (every factor below 2^24, launch_conv_pipe_t)
        const unsigned lin = ow0 + (unsigned)a_row[i];
        const unsigned q1 = umul24(lin, a.s_wo) >> kRecipShift;
        const unsigned ly = oh0 + q1;
        const unsigned q2 = umul24(ly, a.s_ho) >> kRecipShift;
        const int ow = (int)(lin - umul24(q1, (unsigned)a.Wo)), oh = (int)(ly - umul24(q2, (unsigned)a.Ho));
        const int ih0 = imul24(oh, a.stride) - a.pad, iw0 = imul24(ow, a.stride) - a.pad;
        // aligned mode: the chunk column is a channel offset inside the tap; small-Cin mode: it selects the tap itself
        a_off[i] = (imul24((int)umul24(img0 + q2, HW) + imul24(ih0, a.W) + iw0, a.Cs_in) + a.coff_in + (SMALL_CIN ? 0 : imul24(a_col[i], a.col_off))) * 4;
        unsigned long long mask = 0;
        for (int th = 0; th < a.k; ++th)
            for (int tw = 0; tw < a.kw; ++tw)
                if (ok && (unsigned)(ih0 + th) < (unsigned)a.H && (unsigned)(iw0 + tw) < (unsigned)a.W) mask |= 1ull << (th * a.kw + tw);
        a_mask[i] = mask;
    }
    unsigned b_off[NB];
    int b_row[NB], b_col[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int c = tid + NT * i;
        b_row[i] = c / CPR;
        b_col[i] = c % CPR;
        const int n = n0 + b_row[i];
        b_off[i] = (n < a.Cout && b_row[i] < BN) ? (unsigned)((imul24(n, a.Kp) + b_col[i] * 4 + kbeg) * 4) : 0xFFFFFFFFu;
    }

    floatx4 ra[2][NA], rb[2][NB];   // two staging sets: tile kc+1 waits in one while tile kc+2 is fetched into the other
    int kh = 0, kw = 0, c0 = 0, tap = 0;
    // Aligned mode: the address of a staged chunk is (pixel + tap) [per lane, changes only when the tap changes] +
    // c0 [uniform, changes every step].  The first part lives in a_voff (0xFFFFFFFF when the tap falls into the
    // padding) and is recomputed on tap changes only; the second rides in the scalar offset of the buffer load, which
    // the hardware range check accounts for.  The VALU and the matrix pipe share issue bandwidth
    // (tools/ubench/mfma_coissue.hip), so the steady state of the K loop carries no address arithmetic at all.
    unsigned a_voff[NA];
    if (!SMALL_CIN && kbeg > 0) {   // a later K slice starts its walk in the middle
        tap = kbeg / a.Cin;
        c0 = kbeg - tap * a.Cin;
        kh = tap / a.kw;
        kw = tap - kh * a.kw;
    }
    {
        const int tap_pix0 = ((kh * a.W + kw) * a.Cs_in) * 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) a_voff[i] = ((a_mask[i] >> tap) & 1ull) ? (unsigned)(a_off[i] + tap_pix0) : 0xFFFFFFFFu;
    }
    const int c_step = a.c_step ? a.c_step : BK;
    const unsigned cpt = (unsigned)a.Cin >> 2, inv_cpt = 65536u / cpt + 1u, inv_k = 65536u / (unsigned)a.kw + 1u;   // small-Cin decode

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int arow0 = wm * (BM / WM) + l31, brow0 = wn * (BN / WN) + l31;
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    floatx4 bias_v[TN];   // this lane's 4 output channels in the transposed store (see VQ_EPILOGUE_BUF)
    ConvSeg seg_v[TN];    // where each 32-column group of this wave goes (wave-uniform: scalar loads)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb_ = n0 + wn * (BN / WN) + 32 * j;
        const int n = nb_ + (lane & 7) * 4;
        bias_v[j] = *reinterpret_cast<const floatx4*>(a.bias + (n < a.Cout ? n : 0));
        seg_v[j] = a.segs[ks * a.seg_stride + (nb_ < a.Cout ? nb_ >> 5 : 0)];
    }

// Small-Cin mode (the 7x7 stem: Cin padded to 4 or 12): a 32-wide K-step spans several taps, so every staged
// 16-byte chunk decodes its own tap from its K index -- g = KC * CPR + column, tap = g / (Cin/4), (th, tw) = tap / k,
// tap % k -- with multiply-shift divisions (exact for the < 2^10 values that occur).
#define VQ_G_LOAD(IDX, KC, SET)                                                                                \
    {                                                                                                          \
        if ((IDX) < NA) {                                                                                      \
            const int ii = (IDX) < NA ? (IDX) : 0;                                                             \
            unsigned off;                                                                                      \
            if (SMALL_CIN) {                                                                                   \
                const unsigned g_ = (unsigned)(KC) * CPR + a_col[ii];                                          \
                const unsigned tap_ = (g_ * inv_cpt) >> 16, c4_ = g_ - tap_ * cpt;                             \
                const unsigned th_ = (tap_ * inv_k) >> 16, tw_ = tap_ - th_ * a.kw;                            \
                const unsigned ok_ = (unsigned)(a_mask[ii] >> (tap_ & 63u)) & (tap_ < 64u ? 1u : 0u);          \
                off = ok_ ? (unsigned)(a_off[ii] + (int)(((th_ * a.W + tw_) * a.Cs_in + c4_ * 4) * 4)) : 0xFFFFFFFFu; \
            } else {                                                                                           \
                off = a_voff[ii];                                                                              \
            }                                                                                                  \
            ra[SET][ii] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, SMALL_CIN ? 0 : c0 * 4, 0)); \
        } else {                                                                                               \
            const int ii = (IDX) >= NA ? (IDX) - NA : 0;                                                       \
            rb[SET][ii] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_off[ii], (KC) * (BK * 4), 0)); \
        }                                                                                                      \
    }
#define VQ_G_STORE(IDX, BUF, SET)                                                                              \
    {                                                                                                          \
        if ((IDX) < NA) {                                                                                      \
            const int ii = (IDX) < NA ? (IDX) : 0;                                                             \
            if ((BM * CPR) % NT == 0 || a_row[ii] < BM)                                                        \
                *reinterpret_cast<floatx4*>(&sm.a[BUF][a_row[ii]][a_col[ii] * 4]) = ra[SET][ii];               \
        } else {                                                                                               \
            const int ii = (IDX) >= NA ? (IDX) - NA : 0;                                                       \
            if ((BN * CPR) % NT == 0 || b_row[ii] < BN)                                                        \
                *reinterpret_cast<floatx4*>(&sm.b[BUF][b_row[ii]][b_col[ii] * 4]) = rb[SET][ii];               \
        }                                                                                                      \
    }
#define VQ_ADVANCE_TAP()                                                                     \
    if (!SMALL_CIN) {                                                                        \
        c0 += c_step;                                                                        \
        if (c0 >= a.Cin) {                                                                   \
            c0 = 0;                                                                          \
            ++tap;                                                                           \
            if (++kw == a.kw) {                                                              \
                kw = 0;                                                                      \
                ++kh;                                                                        \
            }                                                                                \
            asm volatile("" ::: "memory"); /* keep this a (rarely taken, uniform) branch: if-converted, the    \
                                              recomputation below would run on every step */                 \
            const int tap_pix = ((kh * a.W + kw) * a.Cs_in) * 4;                             \
            _Pragma("unroll") for (int i = 0; i < NA; ++i)                                   \
                a_voff[i] = ((a_mask[i] >> tap) & 1ull) ? (unsigned)(a_off[i] + tap_pix) : 0xFFFFFFFFu; \
        }                                                                                    \
    }
// One K-step on LDS buffer BUF (compute on tile KC); its first fragments are already in fa[0] / fb[0].
// DO_STORE: tile KC+1, fetched during the previous step into staging set SS, goes to LDS buffer BUF^1 behind the first
// MFMAs (a whole K-step after its loads were issued, so the vmcnt wait in front of the LDS store does not stall); with
// EARLY a barrier follows the last of these stores and the last k group prefetches the first fragments of step KC+1.
// DO_LOAD: tile KC+2 is fetched into staging set SL.
#define VQ_PIPE_STEP(BUF, KC, DO_STORE, DO_LOAD, SS, SL)                                                       \
    {                                                                                                          \
        if (!EARLY) {                                                                                          \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                     \
                fa[0][i] = *reinterpret_cast<const floatx4*>(&sm.a[BUF][arow0 + 32 * i][half * 4]);            \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                     \
                fb[0][j] = *reinterpret_cast<const floatx4*>(&sm.b[BUF][brow0 + 32 * j][half * 4]);            \
        }                                                                                                      \
        _Pragma("unroll") for (int q = 0; q < NMF; ++q) {                                                      \
            const int kk = q / MFK, r = q % MFK, s_ = r / (TM * TN), i_ = (r / TN) % TM, j_ = r % TN;          \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i_][s_], fb[kk & 1][j_][s_], acc[i_][j_], 0, 0, 0); \
            if (kk + 1 < NKK && r < NFR) {                                                                     \
                if (r < TM)                                                                                    \
                    fa[(kk + 1) & 1][r < TM ? r : 0] = *reinterpret_cast<const floatx4*>(                      \
                        &sm.a[BUF][arow0 + 32 * (r < TM ? r : 0)][(kk + 1) * 8 + half * 4]);                   \
                else                                                                                           \
                    fb[(kk + 1) & 1][r >= TM ? r - TM : 0] = *reinterpret_cast<const floatx4*>(                \
                        &sm.b[BUF][brow0 + 32 * (r >= TM ? r - TM : 0)][(kk + 1) * 8 + half * 4]);             \
            } else if (EARLY && (DO_STORE) && kk + 1 == NKK && r < NFR) {                                      \
                if (r < TM)                                                                                    \
                    fa[0][r < TM ? r : 0] = *reinterpret_cast<const floatx4*>(                                 \
                        &sm.a[(BUF) ^ 1][arow0 + 32 * (r < TM ? r : 0)][half * 4]);                            \
                else                                                                                           \
                    fb[0][r >= TM ? r - TM : 0] = *reinterpret_cast<const floatx4*>(                           \
                        &sm.b[(BUF) ^ 1][brow0 + 32 * (r >= TM ? r - TM : 0)][half * 4]);                      \
            } else {                                                                                           \
                const int fidx = q - (kk + 1 < NKK ? (kk + 1) * NFR : (NKK - 1) * NFR + (EARLY && (DO_STORE) ? NFR : 0)); \
                if ((DO_STORE) && fidx < NG) {                                                                 \
                    VQ_G_STORE(fidx < NG ? fidx : 0, (BUF) ^ 1, SS)                                            \
                    if (EARLY && fidx == NG - 1) __syncthreads(); /* tile KC+1 is complete in LDS */           \
                } else if ((DO_LOAD) && fidx >= NG && fidx < 2 * NG)                                           \
                    VQ_G_LOAD(fidx - NG < NG ? fidx - NG : 0, (KC) + 2, SL)                                    \
            }                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }                                                                                                      \
    }

    const int nk = (a.ksplit > 1 ? a.k_per : a.Kp) / BK;
    _Pragma("unroll") for (int i = 0; i < NG; ++i) VQ_G_LOAD(i, 0, 0)
    _Pragma("unroll") for (int i = 0; i < NG; ++i) VQ_G_STORE(i, 0, 0)
    if (nk > 1) {
        VQ_ADVANCE_TAP()              // (tap, c0) now address tile 1
        _Pragma("unroll") for (int i = 0; i < NG; ++i) VQ_G_LOAD(i, 1, 1)
    }
    __syncthreads();
    floatx4 fa[2][TM], fb[2][TN];     // MFMA operand fragments, double-buffered over the 8-wide k groups
    if (EARLY) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const floatx4*>(&sm.a[0][arow0 + 32 * i][half * 4]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const floatx4*>(&sm.b[0][brow0 + 32 * j][half * 4]);
    }
    // Steady state (steps 0 .. nk-3: store tile kc+1, fetch tile kc+2), unrolled by two so that the staging sets have
    // static names, as a straight-line pair with no exit in the middle (a mid-loop exit made the register allocator
    // shuttle the accumulators between VGPRs and AGPRs on every trip).
    // End-of-step barrier: every wave is done READING buffer b before anybody stores tile kc+2 into it.  With EARLY
    // the only LDS operations still in flight are the NFR fragment prefetches of the next step (from the other
    // buffer), so the wait is counted and the barrier raw -- a full __syncthreads() would wait for those prefetches
    // and put the LDS latency right back in front of the next step's first MFMA.
#define VQ_END_BARRIER()                                                     \
    {                                                                        \
        if (EARLY) {                                                         \
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NFR) : "memory");     \
            __builtin_amdgcn_s_barrier();                                    \
            asm volatile("" ::: "memory");                                   \
        } else {                                                             \
            __syncthreads();                                                 \
        }                                                                    \
    }
    const int n_steady = nk > 2 ? nk - 2 : 0;
    int kc = 0;
    for (; kc + 1 < n_steady; kc += 2) {
        VQ_ADVANCE_TAP()              // tile kc + 2
        VQ_PIPE_STEP(0, kc, true, true, 1, 0)
        VQ_END_BARRIER()
        VQ_ADVANCE_TAP()              // tile kc + 3
        VQ_PIPE_STEP(1, kc + 1, true, true, 0, 1)
        VQ_END_BARRIER()
    }
    if (kc < n_steady) {              // odd number of steady steps: one more even-indexed step
        VQ_ADVANCE_TAP()
        VQ_PIPE_STEP(0, kc, true, true, 1, 0)
        VQ_END_BARRIER()
        ++kc;
    }
    // tail: the step that still has a successor to store (nothing left to fetch), then the last step
    if (kc + 1 < nk) {
        if (kc & 1) {
            VQ_PIPE_STEP(1, kc, true, false, 0, 0)
        } else {
            VQ_PIPE_STEP(0, kc, true, false, 1, 1)
        }
        VQ_END_BARRIER()
        ++kc;
    }
    if (kc & 1) {
        VQ_PIPE_STEP(1, kc, false, false, 0, 0)
    } else {
        VQ_PIPE_STEP(0, kc, false, false, 0, 0)
    }
    VQ_EPILOGUE_BUF()
#undef VQ_G_LOAD
#undef VQ_G_STORE
#undef VQ_ADVANCE_TAP
#undef VQ_PIPE_STEP
#undef VQ_END_BARRIER
}

// ------------------------------------------------------------------------------------------------
// max-pool + 1x1 convolution with the pooled activations of a workgroup resident in LDS (the stem pools: pool1 ->
// conv2/3x3_reduce, pool2 -> the sibling 1x1 group of inception_3a).  The POOL instantiation of conv_igemm_kernel above stages
// BOTH operands through a double-buffered LDS ring, K-step by K-step.  Here
//   * the pooled image of the workgroup's BM pixels is written ONCE, slab by slab (32 channels of BM = 64 pixels), into a
//     [BM][K] LDS image that is never overwritten -- so there is no ring and only one barrier per slab;
//   * the 18 tap loads of slab s + 1 are in flight under the MFMAs of slab s (64 per wave on the 64 x 256 tile), their maxima
//     (v_max3) and two ds_write_b128 follow those MFMAs;
//   * the weight fragments go straight from L2 into registers one k-group ahead: every wave owns its own BN / WN output
//     columns, nothing about the weights is shared inside a workgroup, so no LDS stage and no barrier exists for them.
// Measured on pool2 -> inception_3a (75 264 x 224 x 192 at 96 crops): the pooling stream alone takes 0.061 ms, the GEMM alone
// 0.069 ms; un-overlapped (all maxima first, then the GEMM; or the K-step ring of conv_igemm_kernel) 0.111 ms.
// Same k order per output element as every other direct kernel (k-groups of 8 ascending, pairs {s, 4 + s} inside): same bits.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void pool_gemm_kernel(ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BM == 64 || BM == 128, "two staged chunks per thread and slab");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int SLAB = 2048 / BM;                              // channels per slab: 512 16-byte chunks = 2 per thread
    constexpr int CPS = SLAB / 4;                                // chunks per row and slab
    constexpr int GS = SLAB / 8;                                 // k-groups per slab (2 or 4)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* As = reinterpret_cast<float*>(smem_raw);               // [BM][K + 4]: rows 16 bytes longer than K -> conflict-free b128 reads
    const int K = a.Cin, pitch = K + 4;
    const int tid = threadIdx.x;
    const int tile = xcd_remap(blockIdx.x, a.tiles_m * a.tiles_n);
    const int m0 = (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * BN;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);

    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    floatx4 bias_v[TN];
    ConvSeg seg_v[TN];
    unsigned b_voff[TN];                                          // this lane's weight row, k = 4 half: + 32 bytes per k-group
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb_ = n0 + wn * (BN / WN) + 32 * j;
        const int n = nb_ + (lane & 7) * 4;
        bias_v[j] = *reinterpret_cast<const floatx4*>(a.bias + (n < a.Cout ? n : 0));
        seg_v[j] = a.segs[nb_ < a.Cout ? nb_ >> 5 : 0];
        b_voff[j] = nb_ + l31 < a.Cout ? (unsigned)(((nb_ + l31) * a.Kp + half * 4) * 4) : 0xFFFFFFFFu;
    }

    // ---- staging role: chunk u of a slab = row (tid + 256 u) / CPS, 16-byte column (tid + 256 u) % CPS -- the same two rows for
    // every slab, so the nine tap offsets of each are computed once (a tap outside the image re-reads the window's first pixel:
    // the maximum is unchanged; a row past M reads zeros through the range check) and a slab adds its scalar channel offset
    unsigned tap_base[2], tap_ok[2];                               // first window pixel (0xFFFFFFFF: row past M); bit t: tap t is inside the image
    int a_dst[2];
    const int px = a.Cs_in * 4, rowb = a.W * px;                   // bytes to the next pixel / next image row (uniform)
    {
        const int HoWo = a.Ho * a.Wo;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + 256 * u, row = idx / CPS, col = idx % CPS;
            const int m = m0 + row;
            const bool ok = m < a.M;
            const int mm = ok ? m : 0;
            const int n_img = mm / HoWo, rem = mm - n_img * HoWo;
            const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
            const int ih0 = oh * a.pool_s, iw0 = ow * a.pool_s;
            tap_base[u] = ok ? (unsigned)((((n_img * a.H + ih0) * a.W + iw0) * a.Cs_in + a.coff_in + col * 4) * 4) : 0xFFFFFFFFu;
            const int nvy = min(3, a.H - ih0), nvx = min(3, a.W - iw0);
            unsigned mask = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if (ok && t / 3 < nvy && t % 3 < nvx) mask |= 1u << t;
            tap_ok[u] = mask;
            a_dst[u] = row * pitch + col * 4;
        }
    }
    floatx4 v[2][9];
#define VQ_PG_TAPS(S)                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                 \
        unsigned inside = tap_ok[u];                                                                                \
        asm volatile("" : "+v"(inside)); /* opaque: the 18 offsets must not be hoisted out of the slab loop into registers */ \
        _Pragma("unroll") for (int t = 0; t < 9; ++t)                                                               \
            v[u][t] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(                            \
                in_rsrc, tap_base[u] + (((inside >> t) & 1u) ? (unsigned)((t / 3) * rowb + (t % 3) * px) : 0u), (S) * (SLAB * 4), 0)); \
    }
#define VQ_PG_POOL(S)                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                 \
        floatx4 r;                                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                             \
            const float x = __builtin_fmaxf(__builtin_fmaxf(v[u][0][e], v[u][1][e]), v[u][2][e]);                  \
            const float y = __builtin_fmaxf(__builtin_fmaxf(v[u][3][e], v[u][4][e]), v[u][5][e]);                  \
            const float z = __builtin_fmaxf(__builtin_fmaxf(v[u][6][e], v[u][7][e]), v[u][8][e]);                  \
            r[e] = __builtin_fmaxf(__builtin_fmaxf(x, y), z);                                                       \
        }                                                                                                           \
        *reinterpret_cast<floatx4*>(As + a_dst[u] + (S) * SLAB) = r;                                                \
    }

    // ---- compute role.  Lane (row r, half h) holds k = 8 g + 4 h + s for MFMA step s of k-group g, as in conv_igemm_kernel.
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* a_frag = As + (wm * (BM / WM) + l31) * pitch + half * 4;      // + 32 i rows, + 8 g floats
    // A wave's vector-memory loads return IN ORDER as far as s_waitcnt vmcnt can tell: a weight fragment fetched behind the 18 tap
    // loads of the next slab could not be consumed before every one of those (HBM) loads had landed -- the first form of this loop did
    // that, and the memory stream and the matrix pipe took turns (0.108 ms for 0.061 + 0.069).  So ALL weight fragments of slab s + 1 are
    // requested FIRST (8 loads, a second register set), the taps of slab s + 1 behind them; the MFMAs of slab s then touch only
    // registers that arrived before the previous slab's closing wait and LDS (lgkmcnt): no vmcnt wait inside a slab.
    floatx4 fa[2][TM], fbs[2][GS][TN];
#define VQ_PG_FETCH_B(SET, S)                                                                                       \
    _Pragma("unroll") for (int q = 0; q < GS; ++q)                                                                  \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                              \
            fbs[SET][q][j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_voff[j], ((S) * GS + q) * 32, 0));
#define VQ_PG_FETCH_A(SET, G)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                  \
        fa[SET][i] = *reinterpret_cast<const floatx4*>(a_frag + 32 * i * pitch + (G) * 8);
#define VQ_PG_MFMA(SET, BSET, Q)                                                                                    \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                                \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                              \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                          \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][i][s_], fbs[BSET][Q][j][s_], acc[i][j], 0, 0, 0);
// one slab: BSET = the register set its weight fragments are in (slab parity)
#define VQ_PG_SLAB(S, BSET)                                                                                         \
    {                                                                                                               \
        const bool more = (S) + 1 < ns;                                                                             \
        if (more) {                                                                                                 \
            VQ_PG_FETCH_B((BSET) ^ 1, (S) + 1)                                                                      \
            VQ_PG_TAPS((S) + 1)                              /* in flight under this slab's MFMAs */                \
        }                                                                                                           \
        VQ_PG_FETCH_A(0, (S) * GS)                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        _Pragma("unroll") for (int q = 0; q < GS; ++q) {                                                            \
            if (q + 1 < GS) VQ_PG_FETCH_A((q + 1) & 1, (S) * GS + q + 1)                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
            VQ_PG_MFMA(q & 1, BSET, q)                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
        }                                                                                                           \
        if (more) {                                                                                                 \
            VQ_PG_POOL((S) + 1)                                                                                     \
            __syncthreads();                                 /* slab S + 1 is complete in LDS (nothing is ever overwritten) */ \
        }                                                                                                           \
    }

    const int ns = K / SLAB;
    VQ_PG_FETCH_B(0, 0)
    VQ_PG_TAPS(0)
    VQ_PG_POOL(0)
    __syncthreads();
    int s = 0;
    for (; s + 1 < ns; s += 2) {                                   // by two: the weight register sets have static names
        VQ_PG_SLAB(s, 0)
        VQ_PG_SLAB(s + 1, 1)
    }
    if (s < ns) VQ_PG_SLAB(s, 0)
#undef VQ_PG_SLAB
#undef VQ_PG_TAPS
#undef VQ_PG_POOL
#undef VQ_PG_FETCH_A
#undef VQ_PG_FETCH_B
#undef VQ_PG_MFMA
    VQ_EPILOGUE()
}

// ------------------------------------------------------------------------------------------------
// pooling (Caffe semantics: ceil-mode output size; MAX ignores padding; AVE divides by the window
// clipped to the padded extent and accumulates h-major in fp32)
// ------------------------------------------------------------------------------------------------
// PoolArgs / pool_one: vq_tsn_kernels.h (shared with the grouped Winograd launch, which also carries a level's pooling)
template <bool IS_MAX>
__global__ __launch_bounds__(256) void pool_kernel(PoolArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.total) pool_one<IS_MAX>(a, i);
}

// global average pool (Caffe AVE, kernel = whole map): sequential fp32 sum over h, w then / (H*W)
__global__ void gavgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int HW, int Cs_in, int coff_in,
                                int C, int Cs_out, int coff_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C) return;
    const int c = i % C, img = i / C;
    const float* p = in + (size_t)img * HW * Cs_in + coff_in + c;
    float acc = 0.f;
    for (int q = 0; q < HW; ++q) acc += p[(size_t)q * Cs_in];
    out[(size_t)img * Cs_out + coff_out + c] = acc / (float)HW;
}

// Second half of a convolution that was split over K (the 7x7-map layers with long K chains: at M = 49 x crops there
// are fewer output tiles than compute units, and one wave per SIMD cannot hide its own load latency): the slices' partial
// sums are added in slice order, then bias and ReLU -- a fixed order for every batch size and tiling, so the rule "which
// layers are split" (layer geometry only) keeps the features independent of how crops are batched.
__global__ void splitk_combine_kernel(const float* __restrict__ part, int ksplit, size_t slice_stride, int M, int Cout,
                                      const float* __restrict__ bias, int relu, float* __restrict__ out, int Cs_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = Cout / 4;
    if (i >= (int64_t)M * c4) return;
    const int m = (int)(i / c4), n = (int)(i - (int64_t)m * c4) * 4;
    floatx4 v = *reinterpret_cast<const floatx4*>(part + (size_t)m * Cout + n);
    for (int q = 1; q < ksplit; ++q) v += *reinterpret_cast<const floatx4*>(part + q * slice_stride + (size_t)m * Cout + n);
    v += *reinterpret_cast<const floatx4*>(bias + n);
    if (relu) {
        v[0] = fmaxf(v[0], 0.f);
        v[1] = fmaxf(v[1], 0.f);
        v[2] = fmaxf(v[2], 0.f);
        v[3] = fmaxf(v[3], 0.f);
    }
    *reinterpret_cast<floatx4*>(out + (size_t)m * Cs_out + n) = v;
}

// global average pool + segment consensus in one pass: a snippet's global_pool value is formed exactly
// like gavgpool_kernel does (sequential fp32 sum over h, w, then / (H*W)) and stored as the per-snippet blob
// (calcSig_wOF.py:95,112); the T values of a clip meet in LDS and are summed in fp64 in numpy's axis-0 order
// (calcSig_wOF.py:82) -- the same bits as the two separate kernels.  Clips of up to kMaxFusedT snippets.
constexpr int kMaxFusedT = 16;
__global__ void gavgpool_consensus_kernel(const float* __restrict__ in, float* __restrict__ out, double* __restrict__ feat, int T,
                                          int HW, int Cs_in, int coff_in, int C, int Cs_out, int coff_out) {
    // block = 64 channels x T snippets of clip blockIdx.y; thread (c, t) pools snippet t, thread (c, 0) forms the consensus
    __shared__ float pooled[kMaxFusedT][64];
    const int c = blockIdx.x * 64 + threadIdx.x, t = threadIdx.y, b = blockIdx.y;
    float v = 0.f;
    if (c < C) {
        const int img = b * T + t;
        const float* p = in + (size_t)img * HW * Cs_in + coff_in + c;
        // the sum keeps Caffe's order (h, then w: one sequential fp32 chain); the LOADS of seven pixels are issued together
        float acc = 0.f;
        int q = 0;
        for (; q + 7 <= HW; q += 7) {
            float x[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) x[u] = p[(size_t)(q + u) * Cs_in];
#pragma unroll
            for (int u = 0; u < 7; ++u) acc += x[u];
        }
        for (; q < HW; ++q) acc += p[(size_t)q * Cs_in];
        v = acc / (float)HW;
        out[(size_t)img * Cs_out + coff_out + c] = v;
    }
    pooled[t][threadIdx.x] = v;
    __syncthreads();
    if (t == 0 && c < C) {
        double cons = (double)pooled[0][threadIdx.x];
        for (int s = 1; s < T; ++s) cons = cons + (double)pooled[s][threadIdx.x];
        feat[(size_t)b * C + c] = cons / (double)T;
    }
}

// segment consensus (calcSig_wOF.py:82): fp64 mean over the T snippets of a clip, sequential like numpy's
// axis-0 reduction of the (T, 1, D) float64 array
__global__ void consensus_kernel(const float* __restrict__ per_snippet, double* __restrict__ feat, int B, int T, int D, int Cs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int d = i % D, b = i / D;
    double acc = (double)per_snippet[(size_t)(b * T) * Cs + d];
    for (int t = 1; t < T; ++t) acc = acc + (double)per_snippet[(size_t)(b * T + t) * Cs + d];
    feat[i] = acc / (double)T;
}

// ------------------------------------------------------------------------------------------------
// handle + executor
// ------------------------------------------------------------------------------------------------
// k*k*Cin of a convolution in ALGORITHMIC terms: a layer that reads slot 0 counts the un-padded input channels and,
// in space-to-depth form, the kernel size of the original convolution.
static inline double first_layer_k2c(const vq_input_desc& in, const vq_layer_desc& L) {
    if (L.src != 0) return (double)L.k * L.k * L.cin;
    const int k = (in.s2d_pad >= 0 && in.s2d_kernel > 0) ? in.s2d_kernel : L.k;
    return (double)k * k * in.c;
}

static inline bool is_wino(int op) { return op == VQ_OP_CONV_WINOGRAD || op == VQ_OP_CONV_WINOGRAD16; }
static inline bool is_conv(int op) { return op == VQ_OP_CONV || is_wino(op); }

struct ConvTile {
    int bm, bn, bk;
    int pipe;   // 1 = software-pipelined kernel (aligned Cin only); 3 = pool_gemm_kernel (pooled-input 1x1 layers only; bk is its
                // k-group of 8).  2 is taken: vq_tsn_layer_tiles reports the Winograd form with it
};

// One kernel launch of a forward: a single layer, or the Winograd convolutions of one graph level together.
struct LaunchItem {
    int kind = 0;              // 0 = one layer, 1 = Winograd group
    std::vector<int> layers;   // kind 1: longest K loop first (its workgroups are dispatched first; the short ones fill the tail)
    int max_crops = 0;         // crops one launch may cover: every slot it touches stays below 2^31 bytes (32-bit offsets)
};

struct vq_tsn {
    std::mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    int cus = 256;
    int max_crops = 0;
    int in_channels = 0;
    vq_input_desc input = {0, 0, 0, -1, 0};
    std::vector<vq_tensor_desc> tensors;
    std::vector<vq_layer_desc> layers;
    std::vector<LaunchItem> items;        // the launch sequence (a topological order of the layer graph by levels)
    std::vector<int> item_of_layer;
    int consensus_layer = -1;             // the global-pool layer that writes the feature slot: runs fused with the consensus
    std::map<int, std::vector<int>> tuned;   // tile_key(n_crops, paired) -> per-layer index into kTiles / Winograd variant
    std::map<int, int> split_pref;           // batch size -> 0: this size runs faster on ONE stream than as sub-batches (vq_tsn_set_split); absent: split
    std::set<int> tuned_borrowed;            // keys whose table was copied from a neighbouring size by ensure_tuned (not measured: never persisted)
    bool autotune = true;                 // VQ_TSN_AUTOTUNE != 0 (read at creation)
    bool group_wino = true;               // VQ_TSN_GROUP != 0 (read at creation): independent Winograd layers share a launch
    bool group_pool = true;               // VQ_TSN_GROUP_POOL != 0 (read at creation): a level's pooling rides in its Winograd launch
    bool poison = false;                  // VQ_TSN_POISON=1 (read at creation): NaN-fill every activation slot before a forward (debug:
                                          // a layer that reads what no earlier layer of THIS forward wrote then yields NaN features)
    int forced_tile = -1;                 // VQ_TSN_TILE = "BMxBN[xBK[xP]]" (read at creation): every direct conv uses this tiling
    std::vector<float*> slots;            // device activations, max_crops each
    std::vector<size_t> slot_bytes;
    size_t blob_bytes = 0;
    float* zeros = nullptr;               // 256 bytes of zeros (load target of masked lanes)
    ConvSeg* seg_table = nullptr;         // destination tables of all conv layers, back to back
    std::vector<int> seg_table_off;       // per layer: first entry
    std::vector<int> ksplit;              // per layer: K slices of a direct convolution (1 = not split; layer geometry only)
    float* split_scratch = nullptr;       // [slice][max_crops * Ho * Wo][Cout] partial sums of the split layer in flight
    size_t split_slice_floats = 0;        // floats between slices (= max_crops * split_crop_floats)
    size_t split_crop_floats = 0;         // scratch floats one crop owns per slice: largest Ho*Wo*Cout of a split layer + one row of slack
    float* zero_bias = nullptr;           // zeros for the slices' epilogues (the bias is added once, by the combine pass)
    float* blob = nullptr;                // weights + biases
    int64_t blob_floats = 0;
    int feature_slot = -1, D = 0;
    uint8_t* crops_dev = nullptr;
    size_t crops_cap = 0;
    float* mean_dev = nullptr;
    std::vector<float> mean_cached;       // what mean_dev holds (uploaded only when the caller's mean changes)
    double* feat_dev = nullptr;           // [max_crops][D] (B <= max_crops)
    double flops_per_crop = 0;
    int last_crops = 0;
    int cur_T = 1;                        // snippets per clip of the forward in flight
    bool fused_consensus = false;         // this forward: the global pool launch also forms the consensus (whole clips per launch, T <= kMaxFusedT)
    int profile_depth = 0;                // > 0: HIP events around every launch (bench roofline accounting)
    int profile_count = 0;                // profiled forwards so far (ring of profile_depth event sets)
    int profile_every = 1;                // while profiling: every n-th forward carries the events, all run on one stream
    int profile_tick = 0;                 // forwards since vq_tsn_set_profile
    bool profile_split = false;           // while profiling: the un-sampled forwards run split over the sub-batch streams (vq_tsn_set_profile_split)
    std::vector<hipEvent_t> events;       // profile_depth x n_items x {start, stop}
    int crop_off = 0;                     // first crop the next launch works on (sub-batches, 32-bit offset chunks)
    int n_split = 1;                      // VQ_TSN_SPLIT: sub-batches of one forward run on separate streams
    std::vector<int> split_parts;         // VQ_TSN_SPLIT=a,b,..: relative sizes of the sub-batches (default equal)
    hipStream_t ls = nullptr;             // stream the next launch goes to
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // profiling: events of the next launch
    std::vector<hipStream_t> split_streams;       // [n_split]; entry 0 unused (caller's stream)
    hipEvent_t fork_ev = nullptr;
    std::vector<hipEvent_t> join_ev;              // per extra stream
};

// Activation slots, weights and split-K scratch of a closed extractor are kept (up to VQ_DEVICE_POOL_GB, default 40; 0 = off) for the
// next one of the same shape in the process: on some hosts a hipMalloc of tens of GB right behind the hipFree of as much waits ~1 s for
// the driver to reclaim the memory (seen as 1.2 s extractor builds in back-to-back command-line runs of one process).  A block is only
// reused at its exact size on its device; whatever a forward reads it has written before.
// bookkeeping in csrc/host/vq_block_pool.cc (host-only, sanitizer-built); the HIP calls stay here
static vq::BlockPool& device_pool() {
    static vq::BlockPool p(vq::BlockPool::cap_from_env());
    return p;
}
void vq::device_pool_trim() {
    for (void* blk : device_pool().drain()) (void)hipFree(blk);
}
static hipError_t pool_alloc(int device, size_t bytes, void** out) {
    if (void* blk = device_pool().take(device, bytes)) {
        *out = blk;
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {                       // give the pooled blocks back and try once more
        (void)hipGetLastError();
        vq::device_pool_trim();
        e = hipMalloc(out, bytes);
    }
    return e;
}
static void pool_free(int device, void* p, size_t bytes) {
    if (!p) return;
    if (!device_pool().give(device, p, bytes)) (void)hipFree(p);
}

static void tsn_free(vq_tsn* net) {
    for (hipEvent_t e : net->events) (void)hipEventDestroy(e);
    net->events.clear();
    for (hipEvent_t e : net->join_ev)
        if (e) (void)hipEventDestroy(e);
    if (net->fork_ev) (void)hipEventDestroy(net->fork_ev);
    for (hipStream_t st : net->split_streams)
        if (st) (void)hipStreamDestroy(st);
    for (size_t i = 0; i < net->slots.size(); ++i) pool_free(net->device, net->slots[i], i < net->slot_bytes.size() ? net->slot_bytes[i] : 0);
    pool_free(net->device, net->blob, net->blob_bytes);
    if (net->zeros) (void)hipFree(net->zeros);
    if (net->seg_table) (void)hipFree(net->seg_table);
    pool_free(net->device, net->split_scratch, net->split_slice_floats * 4 * sizeof(float));
    if (net->zero_bias) (void)hipFree(net->zero_bias);
    if (net->crops_dev) (void)hipFree(net->crops_dev);
    if (net->mean_dev) (void)hipFree(net->mean_dev);
    if (net->feat_dev) (void)hipFree(net->feat_dev);
}

template <int BM, int BN, int WM, int WN, int BK, bool SMALL>
static int launch_conv_t(vq_tsn* net, ConvArgs& a) {
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.Cout, BN);
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, BK, SMALL>;
    const size_t lds = sizeof(ConvSmem<BM, BN, BK>);
    VQ_DYN_LDS(kern, lds);            // per instantiation and device
    VQ_LAUNCH(kern, a.tiles_m * a.tiles_n * a.ksplit, 256, lds, net->ls, net->ev_start, net->ev_stop, a);
    VQ_CHECK_LAUNCH();
    return VQ_OK;
}

// Candidate tilings (BM x BN x BK).  Every candidate sums each output element's K terms in the same order, so
// the choice never changes a result bit; it is made per layer and batch size by timing (autotune) or, with
// VQ_TSN_AUTOTUNE=0, by the occupancy heuristic below.
static const ConvTile kTiles[] = {
    {128, 128, 32, 0}, {128, 128, 16, 0}, {128, 96, 32, 0}, {128, 96, 16, 0}, {128, 64, 32, 0}, {128, 64, 16, 0},
    {64, 128, 32, 0},  {64, 128, 16, 0},  {64, 64, 32, 0},  {64, 64, 16, 0},  {128, 32, 32, 0}, {128, 32, 16, 0},
    {32, 128, 32, 0},  {32, 128, 16, 0},
    {128, 128, 32, 1}, {128, 128, 16, 1}, {128, 96, 32, 1}, {128, 96, 16, 1}, {128, 64, 32, 1}, {128, 64, 16, 1},
    {64, 128, 32, 1},  {64, 128, 16, 1},  {64, 64, 32, 1},  {64, 64, 16, 1},  {128, 32, 32, 1}, {128, 32, 16, 1},
    {32, 128, 32, 1},  {32, 128, 16, 1},
    // (round 6: 256 x 64 tilings -- four waves of 64 x 64 one above the other, for the 64-channel stem with its short K -- were built, tested
    // against the oracle and offered to the sweep: never chosen, conv1 stays at 0.233-0.238 ms on 128 x 64; removed again.)
    // pooled-input 1x1 layers only: ONE column tile for up to 256 output columns (every pooling window is read once), and the
    // two-phase kernel
    {64, 256, 16, 0}, {64, 256, 8, 3}, {64, 64, 8, 3}, {128, 64, 8, 3}};
constexpr int kNumTiles = (int)(sizeof(kTiles) / sizeof(kTiles[0]));

// The pipelined kernel decodes pixels with host-made reciprocals on 24-bit multiplies: what the shapes must satisfy
static bool pixel_walk_ok(const ConvArgs& a) {
    const long long n_img = cdiv(a.M, a.Ho * a.Wo);
    return n_img * a.H * a.W < (1 << 23) && a.Cs_in < (1 << 23) && a.Kp < (1 << 23) && a.Cout < (1 << 23) && a.stride < (1 << 23) &&
           recip22_ok((unsigned)a.Wo, (unsigned)a.Wo + 128u) && recip22_ok((unsigned)a.Ho, (unsigned)a.Ho + 130u);
}

template <int BM, int BN, int WM, int WN, int BK, bool SMALL>
static int launch_conv_pipe_t(vq_tsn* net, ConvArgs& a) {
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.Cout, BN);
    VQ_REQUIRE(pixel_walk_ok(a) && (unsigned long long)a.tiles_m * a.tiles_n * a.tiles_n < 0x100000000ull,
               "pipelined convolution: shape outside the range of its reciprocal divisions");
    a.m_tiles_n = a.tiles_n == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)a.tiles_n) + 1u;
    a.d_howo = full_div_for((unsigned)(a.Ho * a.Wo));
    a.d_wo = full_div_for((unsigned)a.Wo);
    a.s_wo = recip22((unsigned)a.Wo);
    a.s_ho = recip22((unsigned)a.Ho);
    auto kern = conv_igemm_pipe_kernel<BM, BN, WM, WN, BK, SMALL>;
    const size_t lds = sizeof(ConvSmem<BM, BN, BK>);
    VQ_DYN_LDS(kern, lds);            // per instantiation and device
    VQ_LAUNCH(kern, a.tiles_m * a.tiles_n * a.ksplit, 256, lds, net->ls, net->ev_start, net->ev_stop, a);
    VQ_CHECK_LAUNCH();
    return VQ_OK;
}

template <bool SMALL>
static int launch_conv_pipe(vq_tsn* net, ConvArgs& a, const ConvTile& t) {
#define P_(BM_, BN_, WM_, WN_, BK_) \
    if (t.bm == BM_ && t.bn == BN_ && t.bk == BK_) return launch_conv_pipe_t<BM_, BN_, WM_, WN_, BK_, SMALL>(net, a);
    P_(128, 128, 2, 2, 32) P_(128, 128, 2, 2, 16) P_(128, 96, 4, 1, 32) P_(128, 96, 4, 1, 16) P_(128, 64, 2, 2, 32)
    P_(128, 64, 2, 2, 16) P_(64, 128, 2, 2, 32) P_(64, 128, 2, 2, 16) P_(64, 64, 2, 2, 32) P_(64, 64, 2, 2, 16)
    P_(128, 32, 4, 1, 32) P_(128, 32, 4, 1, 16) P_(32, 128, 1, 4, 32) P_(32, 128, 1, 4, 16)
#undef P_
    return fail(VQ_E_INVALID, "no pipelined kernel for tile %dx%dx%d", t.bm, t.bn, t.bk);
}

// The pooled loader keeps 8 extra window taps per staged chunk in registers: tilings that stage at most two chunks per thread.
static bool pool_only_tile(const ConvTile& t) { return t.bn > 128 || t.pipe == 3; }      // instantiated for pooled-input layers only
constexpr size_t kPoolGemmMaxLds = 64 * 1024;                  // pool_gemm_kernel: BM x (K + 4) floats, two or three workgroups per unit
static size_t pool_gemm_lds(int bm, int K) { return std::max((size_t)bm * (K + 4) * sizeof(float), (size_t)4 * 32 * 36 * sizeof(float)); }
static bool pool_tile_ok(const ConvTile& t, int K) {
    if (t.pipe == 3) return pool_gemm_lds(t.bm, K) <= kPoolGemmMaxLds;
    return !t.pipe && t.bm >= 64 && t.bm <= 128 && (t.bk == 16 || (t.bm == 64 && t.bn <= 128));
}

template <int BM, int BN, int WM, int WN, int BK>
static int launch_conv_pool_t(vq_tsn* net, ConvArgs& a) {
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.Cout, BN);
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, BK, false, true>;
    const size_t lds = sizeof(ConvSmem<BM, BN, BK>);
    VQ_DYN_LDS(kern, lds);            // per instantiation and device
    VQ_LAUNCH(kern, a.tiles_m * a.tiles_n, 256, lds, net->ls, net->ev_start, net->ev_stop, a);
    VQ_CHECK_LAUNCH();
    return VQ_OK;
}

template <int BM, int BN, int WM, int WN>
static int launch_pool_gemm_t(vq_tsn* net, ConvArgs& a) {
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.Cout, BN);
    auto kern = pool_gemm_kernel<BM, BN, WM, WN>;
    VQ_DYN_LDS(kern, kPoolGemmMaxLds);            // per instantiation and device; a launch asks for what its K needs
    VQ_LAUNCH(kern, a.tiles_m * a.tiles_n, 256, pool_gemm_lds(BM, a.Cin), net->ls, net->ev_start, net->ev_stop, a);
    VQ_CHECK_LAUNCH();
    return VQ_OK;
}

static int launch_conv_pool(vq_tsn* net, ConvArgs& a, const ConvTile& t) {
    if (t.pipe == 3) {
        VQ_REQUIRE(a.pool_k == 3 && a.Cin % 32 == 0 && a.Kp == a.Cin && pool_tile_ok(t, a.Cin), "pool_gemm_kernel: a 3x3 window over a multiple of 32 channels");
        if (t.bm == 64 && t.bn == 256) return launch_pool_gemm_t<64, 256, 1, 4>(net, a);
        if (t.bm == 64 && t.bn == 64) return launch_pool_gemm_t<64, 64, 2, 2>(net, a);
        if (t.bm == 128 && t.bn == 64) return launch_pool_gemm_t<128, 64, 2, 2>(net, a);
        return fail(VQ_E_INVALID, "no pool_gemm kernel for tile %dx%d", t.bm, t.bn);
    }
#define T_(BM_, BN_, WM_, WN_, BK_) \
    if (t.bm == BM_ && t.bn == BN_ && t.bk == BK_) return launch_conv_pool_t<BM_, BN_, WM_, WN_, BK_>(net, a);
    T_(128, 128, 2, 2, 16) T_(128, 96, 4, 1, 16) T_(128, 64, 2, 2, 16) T_(64, 128, 2, 2, 32) T_(64, 128, 2, 2, 16)
    T_(64, 64, 2, 2, 32) T_(64, 64, 2, 2, 16) T_(128, 32, 4, 1, 16) T_(64, 256, 1, 4, 16)
#undef T_
    return fail(VQ_E_INVALID, "no pooled-input kernel for tile %dx%dx%d", t.bm, t.bn, t.bk);
}

template <bool SMALL>
static int launch_conv(vq_tsn* net, ConvArgs& a, const ConvTile& t) {
#define T_(BM_, BN_, WM_, WN_, BK_) \
    if (t.bm == BM_ && t.bn == BN_ && t.bk == BK_) return launch_conv_t<BM_, BN_, WM_, WN_, BK_, SMALL>(net, a);
    T_(128, 128, 2, 2, 32) T_(128, 128, 2, 2, 16) T_(128, 96, 4, 1, 32) T_(128, 96, 4, 1, 16) T_(128, 64, 2, 2, 32)
    T_(128, 64, 2, 2, 16) T_(64, 128, 2, 2, 32) T_(64, 128, 2, 2, 16) T_(64, 64, 2, 2, 32) T_(64, 64, 2, 2, 16)
    T_(128, 32, 4, 1, 32) T_(128, 32, 4, 1, 16) T_(32, 128, 1, 4, 32) T_(32, 128, 1, 4, 16)
#undef T_
    return fail(VQ_E_INVALID, "no kernel for tile %dx%dx%d", t.bm, t.bn, t.bk);
}

static int heuristic_tile(int M, int N, int cus) {
    int best = 0;
    double bs = -1;
    for (int i = 0; i < kNumTiles; ++i) {
        const ConvTile& t = kTiles[i];
        if (t.bk != 32 || t.pipe || pool_only_tile(t)) continue;
        const long long tiles = (long long)cdiv(M, t.bm) * cdiv(N, t.bn);
        const double per_cu = (double)tiles / cus;
        const double balance = per_cu / std::ceil(per_cu);                      // tail quantisation
        const double padding = ((double)M * N) / ((double)tiles * t.bm * t.bn); // ragged edges
        const double area = (double)t.bm * t.bn;
        const double intrinsic = area >= 128 * 128 ? 1.0 : area >= 128 * 96 ? 0.97 : area >= 128 * 64 ? 0.92 : area >= 64 * 64 ? 0.8 : 0.7;
        const double sc = balance * padding * intrinsic;
        if (sc > bs) {
            bs = sc;
            best = i;
        }
    }
    return best;
}

static void fill_conv_args(vq_tsn* net, int li, int n_crops, ConvArgs& a) {
    const vq_layer_desc& L = net->layers[li];
    const vq_tensor_desc& ts = net->tensors[L.src];
    const vq_tensor_desc& td = net->tensors[L.dst];
    a.in = net->slots[L.src] + (size_t)net->crop_off * ts.h * ts.w * ts.c;
    a.out = net->slots[L.dst];
    a.out_row0 = net->crop_off * td.h * td.w;
    a.w = net->blob + L.w_off;
    a.bias = net->blob + L.b_off;
    a.zeros = net->zeros;
    a.segs = net->seg_table + net->seg_table_off[li];
    a.H = ts.h;
    a.W = ts.w;
    a.Cs_in = ts.c;
    a.coff_in = L.src_coff;
    a.Cin = L.cin;
    a.Ho = td.h;
    a.Wo = td.w;
    a.Cs_out = td.c;
    a.coff_out = L.dst_coff;
    a.Cout = L.cout;
    a.k = L.k;
    a.kw = L.k;
    a.col_off = 4;
    a.c_step = 0;        // 0: the kernel's BK
    a.stride = L.stride;
    a.pad = L.pad;
    a.M = n_crops * td.h * td.w;
    a.Kp = (L.k * L.k * L.cin + KPAD - 1) / KPAD * KPAD;
    a.relu = L.relu;
    a.pool_k = L.pre_pool_k;
    a.pool_s = L.pre_pool_stride;
    a.ksplit = 1;
    a.k_per = 0;
    a.seg_stride = 0;
    a.tiles_m = a.tiles_n = 0;
    a.in_bytes = (unsigned)((size_t)n_crops * ts.h * ts.w * ts.c * sizeof(float));   // < 2^31: LaunchItem::max_crops
    a.w_bytes = (unsigned)((size_t)L.cout * a.Kp * sizeof(float));
}

static int launch_conv_layer(vq_tsn* net, int li, int n_crops, int tile_idx) {
    ConvArgs a;
    fill_conv_args(net, li, n_crops, a);
    const vq_layer_desc& L = net->layers[li];
    // A convolution without padding that reads ALL channels of its slot finds the kw taps of a kernel row side by side
    // in memory ([kw][Cin] is one contiguous run of the NHWC row, and the packed weights have the same order): fold the
    // row into the channel axis -- k rows of one tap with kw*Cin channels.  The space-to-depth stem (4x4 over 12
    // channels) becomes 4 taps of 48 contiguous floats and, with BK = 16, runs on the aligned path: no per-chunk tap
    // decoding in its K loop.
    const bool stem_rows = L.src == 0 && net->input.s2d_pad >= 0 && net->input.s2d_order == 1;
    if (stem_rows) {
        // The x-major space-to-depth stem (vq_input_desc.s2d_order = 1): a kernel row is 7 x-taps x (2 rows x c channels) = 14 c contiguous
        // floats of the slot (RGB: 42, + 2 of the next pixel against zero weights = 44; flow stack: 140), and the weights are packed
        // [Cout][steps][4 rows][4]: chunk column r of a staged row holds kernel row r, every K-step advances 4 floats along all four
        // rows at once.  K = 176 (RGB) / 560 (flow) instead of the 192 / 640 of the (p,q,c) order, whose zero taps are scattered
        // through every 16-float step -- with the address arithmetic of the aligned path (a per-lane row offset fixed for the whole
        // loop, a uniform scalar offset per step).
        a.k = a.kw = 1;
        a.Cin = 4 * L.k * L.cin;                       // never reached: the walk stays inside its one "tap"
        a.col_off = net->tensors[L.src].w * net->tensors[L.src].c;
        a.c_step = 4;
        a.Kp = stem_rows_kp(net->input.c);
        a.w_bytes = (unsigned)((size_t)L.cout * a.Kp * sizeof(float));
    } else if (L.pad == 0 && L.k > 1 && L.cin == net->tensors[L.src].c && L.src_coff == 0) {
        a.kw = 1;
        a.Cin = L.k * L.cin;
    }
    if (L.pre_pool_k > 0) {
        // max-pool folded into the loader: any tiling gives the same bits, so an unsupported choice (heuristic, VQ_TSN_TILE)
        // is replaced by the BK = 16 tiling of the same shape
        ConvTile t = kTiles[tile_idx];
        if (!pool_tile_ok(t, a.Cin)) t = ConvTile{std::min(t.bm, 128), std::min(t.bn, 128), 16, 0};
        return launch_conv_pool(net, a, t);
    }
    if (stem_rows) {                                   // four chunk columns = four kernel rows: BK = 16 tilings only (same bits for all)
        ConvTile t = kTiles[tile_idx];
        t.bk = 16;
        return t.pipe && pixel_walk_ok(a) ? launch_conv_pipe<false>(net, a, t) : launch_conv<false>(net, a, t);
    }
    const bool small = (a.Cin % kTiles[tile_idx].bk) != 0;
    if (net->ksplit[li] > 1) {
        // K slices into the scratch planes (zero bias, no ReLU: the slice tables say so), then the combine pass into the
        // layer's destination.  The profiling events bracket the pair.
        const hipEvent_t e0 = net->ev_start, e1 = net->ev_stop;
        a.ksplit = net->ksplit[li];
        a.k_per = a.Kp / a.ksplit;
        a.seg_stride = (L.cout + 31) / 32;
        a.bias = net->zero_bias;
        // Sub-batches of one forward run on separate streams and may be at DIFFERENT split layers at the same moment: every
        // crop owns a fixed stretch of each scratch plane (split_crop_floats, room for the largest split layer), and a
        // launch writes its [M][Cout] partial sums at the first whole row inside the stretch of its first crop.
        a.out_row0 = (int)cdiv((int64_t)net->crop_off * (int64_t)net->split_crop_floats, (int64_t)L.cout);
        net->ev_stop = nullptr;
        int rc = kTiles[tile_idx].pipe == 1 && pixel_walk_ok(a) ? launch_conv_pipe<false>(net, a, kTiles[tile_idx])
                                                                 : launch_conv<false>(net, a, kTiles[tile_idx]);
        net->ev_stop = e1;
        if (rc != VQ_OK) return rc;
        net->ev_start = nullptr;
        const vq_tensor_desc& td = net->tensors[L.dst];
        float* out = net->slots[L.dst] + (size_t)net->crop_off * td.h * td.w * td.c + L.dst_coff;
        const int64_t work = (int64_t)a.M * (L.cout / 4);
        VQ_LAUNCH(splitk_combine_kernel, (unsigned)cdiv(work, 256), 256, 0, net->ls, net->ev_start, net->ev_stop,
                  net->split_scratch + (size_t)a.out_row0 * L.cout, a.ksplit, net->split_slice_floats, a.M, L.cout, net->blob + L.b_off, L.relu, out, td.c);
        net->ev_start = e0;
        VQ_CHECK_LAUNCH();
        return VQ_OK;
    }
    // (shapes outside the pipelined kernel's reciprocal range run the plain kernel of the same tiling: same bits)
    if (kTiles[tile_idx].pipe == 1 && pixel_walk_ok(a))
        return small ? launch_conv_pipe<true>(net, a, kTiles[tile_idx]) : launch_conv_pipe<false>(net, a, kTiles[tile_idx]);
    return small ? launch_conv<true>(net, a, kTiles[tile_idx]) : launch_conv<false>(net, a, kTiles[tile_idx]);
}

static void fill_wino_job(vq_tsn* net, int li, int n_crops, WinoJob& a, int variant) {
    const vq_layer_desc& L = net->layers[li];
    const vq_tensor_desc& ts = net->tensors[L.src];
    const vq_tensor_desc& td = net->tensors[L.dst];
    memset(&a, 0, sizeof a);
    a.in = net->slots[L.src] + (size_t)net->crop_off * ts.h * ts.w * ts.c;
    a.u = net->blob + L.w_off;
    a.bias = net->blob + L.b_off;
    a.out = net->slots[L.dst] + (size_t)net->crop_off * td.h * td.w * td.c;
    a.H = ts.h;
    a.W = ts.w;
    a.Cs_in = ts.c;
    a.coff_in = L.src_coff;
    a.Cin = L.cin;
    a.Cs_out = td.c;
    a.coff_out = L.dst_coff;
    a.Cout = L.cout;
    a.th = (ts.h + 1) / 2;
    a.tw = (ts.w + 1) / 2;
    a.P = n_crops * a.th * a.tw;
    a.relu = L.relu;
    // VQ_OP_CONV_WINOGRAD16: both filter layouts, the 32-tile one first; variants 2, 3 run the units of 16 tiles on the second
    a.t16 = L.op == VQ_OP_CONV_WINOGRAD16 && variant >= 2;
    if (a.t16) a.u += (size_t)16 * L.cout * L.cin;
    a.in_bytes = (unsigned)((size_t)n_crops * ts.h * ts.w * ts.c * sizeof(float));
    a.out_bytes = (unsigned)((size_t)n_crops * td.h * td.w * td.c * sizeof(float));
    a.u_bytes = (unsigned)((size_t)16 * L.cout * L.cin * sizeof(float));
}

static void fill_pool_args(vq_tsn* net, int li, int n_crops, PoolArgs& a) {
    const vq_layer_desc& L = net->layers[li];
    const vq_tensor_desc& ts = net->tensors[L.src];
    const vq_tensor_desc& td = net->tensors[L.dst];
    memset(&a, 0, sizeof a);
    a.in = net->slots[L.src] + (size_t)net->crop_off * ts.h * ts.w * ts.c;
    a.out = net->slots[L.dst] + (size_t)net->crop_off * td.h * td.w * td.c;
    a.H = ts.h;
    a.W = ts.w;
    a.Cs_in = ts.c;
    a.coff_in = L.src_coff;
    a.C = L.cin;
    a.Ho = td.h;
    a.Wo = td.w;
    a.Cs_out = td.c;
    a.coff_out = L.dst_coff;
    a.k = L.k;
    a.stride = L.stride;
    a.pad = L.pad;
    a.is_max = L.op == VQ_OP_MAXPOOL;
    a.total = (int64_t)n_crops * td.h * td.w * (L.cin / 4);
    a.bias = (L.op == VQ_OP_AVGPOOL && L.has_bias) ? net->blob + L.b_off : nullptr;
    a.relu = (L.op == VQ_OP_AVGPOOL) ? L.relu : 0;
}

// The Winograd layers among `members` (independent of each other) plus the pooling layers among them as one launch.
static int launch_wino_layers(vq_tsn* net, const std::vector<int>& members, int n_crops, int variant) {
    WinoGroup g;
    memset(&g, 0, sizeof g);
    for (int li : members) {
        if (is_wino(net->layers[li].op)) {
            VQ_REQUIRE(g.n_jobs < kWinoMaxJobs, "a Winograd launch carries at most %d convolutions", kWinoMaxJobs);
            fill_wino_job(net, li, n_crops, g.job[g.n_jobs++], variant);
        } else {
            VQ_REQUIRE(g.n_pools < kWinoMaxPools, "a Winograd launch carries at most %d pooling layers", kWinoMaxPools);
            fill_pool_args(net, li, n_crops, g.pool[g.n_pools++]);
        }
    }
    bool both = true;
    for (int li : members)
        if (net->layers[li].op == VQ_OP_CONV_WINOGRAD) both = false;
    if (variant >= 2 && !both) {                         // a table written for another layout: the same channel blocks on 32 tiles (same bits)
        variant -= 2;
        for (int q = 0; q < g.n_jobs; ++q) VQ_REQUIRE(!g.job[q].t16, "internal: mixed Winograd layouts in one launch");
    }
    return launch_wino_group(g, variant, net->ls, net->ev_start, net->ev_stop);
}

static int run_layer(vq_tsn* net, int li, int n_crops, int tune_key);

// Launch one item on crops [crop0, crop0 + n_crops) of the batch; tune_key = the batch size whose tuning table applies.
// A launch addresses its slots with 32-bit byte offsets, so an item whose slots exceed 2^31 bytes at this batch size
// runs as several launches over crop ranges (the results do not depend on the cut: every crop is independent).
static int run_item(vq_tsn* net, const LaunchItem& it, int crop0, int n_crops, int tune_key) {
    const hipEvent_t e0 = net->ev_start, e1 = net->ev_stop;
    for (int done = 0; done < n_crops;) {
        const int n = std::min(n_crops - done, it.max_crops);
        net->crop_off = crop0 + done;
        // profiling: the first chunk carries the start event, the last one the stop event
        net->ev_start = done == 0 ? e0 : nullptr;
        net->ev_stop = done + n == n_crops ? e1 : nullptr;
        int rc;
        if (it.kind == 1) {
            auto tn = net->tuned.find(tune_key);
            rc = launch_wino_layers(net, it.layers, n, tn != net->tuned.end() ? tn->second[it.layers[0]] : 0);
        } else {
            rc = run_layer(net, it.layers[0], n, tune_key);
        }
        if (rc != VQ_OK) {
            net->crop_off = 0;
            net->ev_start = net->ev_stop = nullptr;
            return rc;
        }
        done += n;
    }
    net->crop_off = 0;
    net->ev_start = net->ev_stop = nullptr;
    return VQ_OK;
}

// Time every candidate tiling of every conv launch at this batch size (activations hold whatever the slots
// contain; only durations matter) and keep the fastest.
// The chip's clock follows its load: a sweep that starts on an idle GPU times its first candidates at a lower clock than its last ones
// (round 5: a 96-crop table tuned at the start of a test, behind seconds of host work, gave the first three launches of the forward -- conv1,
// conv2/3x3_reduce, the inception_3a group -- the LAST candidates of the list, one of them 1.8x slower than the best: 2.91 ms per step
// instead of 2.73, kept by the tiling cache).  So the sweep starts behind ~30 ms of the forward's own launches, and the fastest
// candidates of a launch are timed again, back to back, before the winner is taken.
// ``paired``: the size is that of a SUB-BATCH of the default forward (two sub-batches on two streams, side by side): a candidate is timed
// the way it will run -- the same launch on all sub-batch streams at once, on the crop ranges of the sub-batches -- because the best tiling of a
// launch that has the chip to itself (few, large workgroups fill it badly) is not the best one beside its twin.
// A table lives under tile_key(size, paired): a size tuned side by side is another table than the same size tuned alone.
constexpr int kPairedKey = 1 << 24;
static inline int tile_key(int n_crops, bool paired) { return n_crops + (paired ? kPairedKey : 0); }

struct EventSet {          // the sweep's events go on every way out
    std::vector<hipEvent_t> ev;
    ~EventSet() {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

static int autotune(vq_tsn* net, int n_crops, bool paired) {
    const int key = tile_key(n_crops, paired);
    std::vector<int>& choice = net->tuned[key];
    choice.assign(net->layers.size(), 0);
    net->tuned_borrowed.erase(key);
    net->ls = net->stream;
    const int ways = paired ? std::max(1, std::min(net->n_split, net->max_crops / n_crops)) : 1;     // streams that run the launch side by side
    EventSet events;
    events.ev.assign(2 + ways, nullptr);
    for (int l = 0; l < 2 + ways; ++l)
        if (l != 2) VQ_HIP(hipEventCreate(&events.ev[l]));
    const hipEvent_t e0 = events.ev[0], e1 = events.ev[1];
    hipEvent_t* const f = events.ev.data() + 2;               // f[1 .. ways-1]
    auto timed = [&](const LaunchItem& it, int reps, float* ms) -> int {
        VQ_HIP(hipEventRecord(e0, net->stream));
        int rc = run_item(net, it, 0, n_crops, key);           // warm
        if (rc != VQ_OK) return rc;
        VQ_HIP(hipEventRecord(e1, net->stream));
        VQ_HIP(hipEventSynchronize(e1));
        float warm = 0.f;
        VQ_HIP(hipEventElapsedTime(&warm, e0, e1));
        // long launches need fewer repetitions (the command line's first run on a machine tunes two networks at 2 x 400 crops) -- but not
        // one look at a 0.3 ms launch: with that, the 224-crop tables of cfg 3 came out 9 % slower (2 345 against 2 580 clips/s)
        // (and with two looks at launches above 0.5 ms still 2 480: only launches of more than 2 ms are timed once)
        if (warm > 2.0f) reps = 1;
        VQ_HIP(hipEventRecord(e0, net->stream));
        for (int l = 1; l < ways; ++l) VQ_HIP(hipStreamWaitEvent(net->split_streams[l], e0, 0));
        for (int r = 0; r < reps; ++r)
            for (int l = 0; l < ways; ++l) {
                net->ls = l > 0 ? net->split_streams[l] : net->stream;
                rc = run_item(net, it, l * n_crops, n_crops, key);
                net->ls = net->stream;
                if (rc != VQ_OK) return rc;
            }
        VQ_HIP(hipEventRecord(e1, net->stream));
        for (int l = 1; l < ways; ++l) VQ_HIP(hipEventRecord(f[l], net->split_streams[l]));
        VQ_HIP(hipEventSynchronize(e1));
        VQ_HIP(hipEventElapsedTime(ms, e0, e1));
        for (int l = 1; l < ways; ++l) {
            float other = 0.f;
            VQ_HIP(hipEventSynchronize(f[l]));
            VQ_HIP(hipEventElapsedTime(&other, e0, f[l]));
            *ms = std::max(*ms, other);
        }
        return VQ_OK;
    };
    {   // bring the clock up: the heuristic tilings (choice = 0 entries are replaced below) through the whole launch list
        std::vector<int> saved = choice;
        net->tuned.erase(key);                                // run_layer falls back to the heuristic without a table
        for (int r = 0; r < 10; ++r)
            for (const LaunchItem& it : net->items) {
                const int rc = run_item(net, it, 0, n_crops, key);
                if (rc != VQ_OK) return rc;
            }
        VQ_HIP(hipStreamSynchronize(net->stream));
        net->tuned[key] = saved;
    }
    std::vector<int>& pick = net->tuned[key];
    for (const LaunchItem& it : net->items) {
        const int li = it.layers[0];
        if (!is_conv(net->layers[li].op)) continue;
        const bool wino = is_wino(net->layers[li].op);
        std::vector<std::pair<float, int>> seen;
        const int n_wino = net->layers[li].op == VQ_OP_CONV_WINOGRAD16 ? 2 * kWinoVariants : kWinoVariants;     // + the 16-tile units
        for (int t = 0; t < (wino ? n_wino : kNumTiles); ++t) {
            if (!wino && net->layers[li].pre_pool_k > 0 && !pool_tile_ok(kTiles[t], net->layers[li].cin)) continue;
            if (!wino && net->layers[li].pre_pool_k == 0 && pool_only_tile(kTiles[t])) continue;
            // a column tile that is mostly padding cannot win: do not let a timing fluke choose it
            if (!wino && kTiles[t].bn >= 2 * ((net->layers[li].cout + 63) / 64 * 64)) continue;
            for (int m : it.layers) pick[m] = t;               // a grouped launch runs one variant for all its members
            float ms = 0.f;
            const int rc = timed(it, 3, &ms);
            if (rc != VQ_OK) return rc;
            seen.emplace_back(ms, t);
        }
        std::sort(seen.begin(), seen.end());
        float best = 1e30f;
        int win = seen.empty() ? 0 : seen[0].second;
        for (size_t k = 0; k < std::min<size_t>(3, seen.size()) && seen.size() > 1; ++k) {      // the finalists, back to back
            for (int m : it.layers) pick[m] = seen[k].second;
            float ms = 0.f;
            const int rc = timed(it, 5, &ms);
            if (rc != VQ_OK) return rc;
            if (ms < best) {
                best = ms;
                win = seen[k].second;
            }
        }
        for (int m : it.layers) pick[m] = win;
    }
    return VQ_OK;
}

// Make sure a tiling table exists for this batch size.  The first size seen is autotuned; a later size within 1.5x of
// a tuned one borrows that table (the ragged last batch of a video must not cost 2 s of tuning launches -- every
// tiling gives the same bits, only the speed differs); anything further away is tuned itself.
static int ensure_tuned(vq_tsn* net, int n_crops, bool paired) {
    const int key = tile_key(n_crops, paired);
    if (net->forced_tile >= 0 || net->tuned.find(key) != net->tuned.end()) return VQ_OK;
    // the nearest size timed the same way (side by side / alone); failing that the nearest one timed the other way
    int nearest = 0, nearest_key = 0;
    for (int same = 1; same >= 0 && nearest == 0; --same)
        for (const auto& kv : net->tuned) {
            const bool kp = kv.first >= kPairedKey;
            const int size = kv.first - (kp ? kPairedKey : 0);
            if ((kp == paired) != (same == 1) || net->tuned_borrowed.count(kv.first)) continue;
            if (10 * std::max(size, n_crops) > 16 * std::min(size, n_crops)) continue;            // further than 1.6x away
            if (nearest == 0 || std::abs(size - n_crops) < std::abs(nearest - n_crops)) {
                nearest = size;
                nearest_key = kv.first;
            }
        }
    if (nearest > 0) {
        net->tuned[key] = net->tuned[nearest_key];
        net->tuned_borrowed.insert(key);
        return VQ_OK;
    }
    if (!net->autotune) return VQ_OK;             // VQ_TSN_AUTOTUNE=0: the occupancy heuristic for sizes without a table
    return autotune(net, n_crops, paired);
}

static int run_layer(vq_tsn* net, int li, int n_crops, int tune_key) {
    const vq_layer_desc& L = net->layers[li];
    const vq_tensor_desc& ts = net->tensors[L.src];
    const vq_tensor_desc& td = net->tensors[L.dst];
    if (L.op == VQ_OP_CONV) {
        int t = net->forced_tile;   // VQ_TSN_TILE (read at creation): test / tuning aid
        if (t < 0) {
            auto it = net->tuned.find(tune_key);
            t = it != net->tuned.end() ? it->second[li] : heuristic_tile(n_crops * td.h * td.w, L.cout, net->cus);
        }
        return launch_conv_layer(net, li, n_crops, t);
    }
    if (is_wino(L.op)) {
        auto it = net->tuned.find(tune_key);
        return launch_wino_layers(net, std::vector<int>{li}, n_crops, it != net->tuned.end() ? it->second[li] : 0);
    }
    if (L.op == VQ_OP_MAXPOOL || L.op == VQ_OP_AVGPOOL) {
        PoolArgs a;
        fill_pool_args(net, li, n_crops, a);
        const int64_t blocks = (a.total + 255) / 256;
        if (L.op == VQ_OP_MAXPOOL)
            VQ_LAUNCH(pool_kernel<true>, (unsigned)blocks, 256, 0, net->ls, net->ev_start, net->ev_stop, a);
        else
            VQ_LAUNCH(pool_kernel<false>, (unsigned)blocks, 256, 0, net->ls, net->ev_start, net->ev_stop, a);
        VQ_CHECK_LAUNCH();
        return VQ_OK;
    }
    if (L.op == VQ_OP_GLOBAL_AVGPOOL) {
        const float* gin = net->slots[L.src] + (size_t)net->crop_off * ts.h * ts.w * ts.c;
        float* gout = net->slots[L.dst] + (size_t)net->crop_off * td.c;
        const int T = net->cur_T;
        if (li == net->consensus_layer && net->fused_consensus) {
            // the feature blob: global pool and segment consensus in one pass (the launch covers whole clips)
            const int B = n_crops / T;
            VQ_LAUNCH(gavgpool_consensus_kernel, dim3(cdiv(L.cin, 64), B), dim3(64, T), 0, net->ls, net->ev_start, net->ev_stop, gin, gout,
                      net->feat_dev + (size_t)(net->crop_off / T) * net->D, T, ts.h * ts.w, ts.c, L.src_coff, L.cin, td.c, L.dst_coff);
        } else {
            VQ_LAUNCH(gavgpool_kernel, cdiv((int64_t)n_crops * L.cin, 256), 256, 0, net->ls, net->ev_start, net->ev_stop, gin, gout, n_crops,
                      ts.h * ts.w, ts.c, L.src_coff, L.cin, td.c, L.dst_coff);
        }
        VQ_CHECK_LAUNCH();
        return VQ_OK;
    }
    return fail(VQ_E_INVALID, "layer %d: unknown op %d", li, L.op);
}

static int pool_out_size(int size, int k, int s, int p) {
    int out = (size + 2 * p - k + s - 1) / s + 1;
    if (p > 0 && (out - 1) * s >= size + p) --out;
    return out;
}

// Channel range of one slot that a layer reads or writes.
struct SlotRange {
    int slot, c0, c1;
};
static bool overlaps(const std::vector<SlotRange>& x, const std::vector<SlotRange>& y) {
    for (const SlotRange& p : x)
        for (const SlotRange& q : y)
            if (p.slot == q.slot && p.c0 < q.c1 && q.c0 < p.c1) return true;
    return false;
}

// The launch sequence.  Layer i depends on an earlier layer j when i reads what j wrote, overwrites what j read, or
// writes the same channels (slots are never recycled, so in a valid plan only the first kind occurs, but all three are
// honoured).  Layers are levelled (level = 1 + deepest dependency) and launched level by level, which is a topological
// order; inside a level every layer is independent of every other, so the level's Winograd convolutions -- the 3x3 and
// the first double-3x3 arm of an inception module -- share ONE launch (vq_wino.hip).
static void build_items(vq_tsn* net, const vq_conv_segment* segments) {
    const int n = (int)net->layers.size();
    std::vector<std::vector<SlotRange>> rd(n), wr(n);
    for (int i = 0; i < n; ++i) {
        const vq_layer_desc& L = net->layers[i];
        const bool whole = L.op == VQ_OP_CONV && L.cin % KPAD != 0;
        rd[i].push_back(whole ? SlotRange{L.src, 0, net->tensors[L.src].c} : SlotRange{L.src, L.src_coff, L.src_coff + L.cin});
        if (L.op == VQ_OP_CONV && L.seg_count > 0)
            for (int q = 0; q < L.seg_count; ++q) {
                const vq_conv_segment& sg = segments[L.seg_first + q];
                wr[i].push_back(SlotRange{sg.dst, sg.dst_coff, sg.dst_coff + sg.cout});
            }
        else
            wr[i].push_back(SlotRange{L.dst, L.dst_coff, L.dst_coff + L.cout});
    }
    std::vector<int> level(n, 0), floor_level(n, 0);
    int n_levels = 0;
    auto levelise = [&]() {
        n_levels = 0;
        for (int i = 0; i < n; ++i) {
            level[i] = floor_level[i];
            for (int j = 0; j < i; ++j)
                if (overlaps(wr[j], rd[i]) || overlaps(rd[j], wr[i]) || overlaps(wr[j], wr[i])) level[i] = std::max(level[i], level[j] + 1);
            n_levels = std::max(n_levels, level[i] + 1);
        }
    };
    levelise();
    // A pooling layer that reads a module's input is ready one level before the module's Winograd convolutions (it sits
    // beside the 1x1 reductions).  Nothing needs it that early: hold it back one level so it can ride in their launch.
    if (net->group_wino && net->group_pool) {
        for (int i = 0; i < n; ++i) {
            const int op = net->layers[i].op;
            if (op != VQ_OP_MAXPOOL && op != VQ_OP_AVGPOOL) continue;
            bool here = false, next = false;
            for (int j = 0; j < n; ++j)
                if (is_wino(net->layers[j].op)) {
                    here |= level[j] == level[i];
                    next |= level[j] == level[i] + 1;
                }
            if (!here && next) floor_level[i] = level[i] + 1;
        }
        levelise();
    }
    auto slot_bytes_per_crop = [&](int slot) {
        const vq_tensor_desc& t = net->tensors[slot];
        return (size_t)t.h * t.w * t.c * sizeof(float);
    };
    auto item_limit = [&](const std::vector<int>& members) {
        size_t worst = 1;
        for (int li : members) {
            const vq_layer_desc& L = net->layers[li];
            worst = std::max(worst, slot_bytes_per_crop(L.src));
            if (L.op == VQ_OP_CONV && L.seg_count > 0)
                for (int q = 0; q < L.seg_count; ++q) worst = std::max(worst, slot_bytes_per_crop(segments[L.seg_first + q].dst));
            else
                worst = std::max(worst, slot_bytes_per_crop(L.dst));
        }
        size_t limit = std::max<size_t>(1, (size_t)0x7FFFFFF0u / worst);
        // the Winograd kernel multiplies pixel indices of a slot on 24 bits (vq_wino.hip: launch_t requires crops x H x W < 2^23 for the
        // source AND the destination): with >= 64 channels per pixel the byte limit above implies it, a narrower slot at a large batch
        // needs the cap itself (the launch then covers the batch in several crop ranges, like any other item)
        for (int li : members) {
            const vq_layer_desc& L = net->layers[li];
            if (!is_wino(L.op)) continue;
            for (int slot : {L.src, L.dst}) {
                const vq_tensor_desc& t = net->tensors[slot];
                limit = std::min(limit, std::max<size_t>(1, ((size_t)(1u << 23) - 1) / ((size_t)t.h * t.w)));
            }
            // ... and decodes a workgroup's first tile with a multiply-high division that is exact while (tiles + 32) x tiles per image < 2^32
            const vq_tensor_desc& ts = net->tensors[L.src];
            const size_t tpi = (size_t)((ts.h + 1) / 2) * ((ts.w + 1) / 2);
            if ((1ull << 32) / tpi > 64) limit = std::min(limit, std::max<size_t>(1, ((size_t)((1ull << 32) / tpi) - 64) / tpi));
        }
        return (int)limit;
    };
    net->items.clear();
    net->item_of_layer.assign(n, -1);
    for (int lv = 0; lv < n_levels; ++lv) {
        std::vector<int> wino, pools;
        for (int i = 0; i < n; ++i)
            if (level[i] == lv && is_wino(net->layers[i].op) && net->group_wino) wino.push_back(i);
        for (int i = 0; i < n; ++i) {
            if (level[i] != lv || (is_wino(net->layers[i].op) && net->group_wino)) continue;
            const int op = net->layers[i].op;
            // the level's pooling rides in its Winograd launch (few short workgroups that fill the tail)
            if (!wino.empty() && (op == VQ_OP_MAXPOOL || op == VQ_OP_AVGPOOL) && (int)pools.size() < kWinoMaxPools && net->group_pool) {
                pools.push_back(i);
                continue;
            }
            LaunchItem it;
            it.kind = 0;
            it.layers = {i};
            it.max_crops = item_limit(it.layers);
            net->item_of_layer[i] = (int)net->items.size();
            net->items.push_back(it);
        }
        // longest K loop first: the hardware hands out workgroups in index order, so the short ones fill the tail
        // (a launch carries layers of one filter layout: the 32-tile form first, then the 16-tile form; a level of BN-Inception has one)
        std::stable_sort(wino.begin(), wino.end(), [&](int x, int y) {
            if (net->layers[x].op != net->layers[y].op) return net->layers[x].op < net->layers[y].op;
            return net->layers[x].cin > net->layers[y].cin;
        });
        for (size_t q = 0; q < wino.size();) {
            size_t end = q;
            while (end < wino.size() && end - q < (size_t)kWinoMaxJobs && net->layers[wino[end]].op == net->layers[wino[q]].op) ++end;
            LaunchItem it;
            it.kind = 1;
            it.layers.assign(wino.begin() + q, wino.begin() + end);
            const bool first = q == 0;
            q = end;
            if (first) it.layers.insert(it.layers.end(), pools.begin(), pools.end());
            it.max_crops = item_limit(it.layers);
            for (int m : it.layers) net->item_of_layer[m] = (int)net->items.size();
            net->items.push_back(it);
        }
    }
}

extern "C" {

int vq_tsn_create(const vq_tensor_desc* tensors, int32_t n_tensors, const vq_layer_desc* layers, int32_t n_layers,
                  const vq_conv_segment* segments, int32_t n_segments, const float* blob_host, int64_t blob_floats,
                  const vq_input_desc* input, int32_t feature_slot, int32_t max_crops, int32_t device, vq_tsn** out) {
    VQ_REQUIRE(out, "out is NULL");
    *out = nullptr;
    VQ_REQUIRE(tensors && layers && blob_host && input, "NULL argument");
    const int in_channels = input->c;
    VQ_REQUIRE(n_tensors > 0 && n_layers > 0 && blob_floats > 0 && max_crops > 0, "sizes must be positive");
    VQ_REQUIRE(n_segments >= 0 && (n_segments == 0 || segments), "bad segment table");
    VQ_REQUIRE(feature_slot > 0 && feature_slot < n_tensors, "feature_slot out of range");
    VQ_REQUIRE(tensors[feature_slot].h == 1 && tensors[feature_slot].w == 1, "feature slot must be 1x1xD");
    VQ_REQUIRE(tensors[0].c % 4 == 0, "input slot channels must be padded to a multiple of 4 (got %d)", tensors[0].c);
    VQ_REQUIRE(input->h > 0 && input->w > 0 && in_channels > 0, "input crops must be h x w x c with positive sizes");
    if (input->s2d_pad < 0) {
        VQ_REQUIRE(tensors[0].h == input->h && tensors[0].w == input->w, "input slot is %dx%d but the crops are %dx%d", tensors[0].h,
                   tensors[0].w, input->h, input->w);
        VQ_REQUIRE(in_channels <= tensors[0].c && tensors[0].c - in_channels < 4, "in_channels %d does not fit the %d-channel input slot",
                   in_channels, tensors[0].c);
    } else {
        VQ_REQUIRE(tensors[0].c == 4 * in_channels, "space-to-depth input slot needs 4 x %d channels (got %d)", in_channels, tensors[0].c);
        VQ_REQUIRE(input->s2d_pad <= 64, "space-to-depth shift out of range");
        VQ_REQUIRE(input->s2d_order == 0 || input->s2d_order == 1, "s2d_order must be 0 or 1");
    }
    // validate every layer against the tensor table BEFORE anything is launched: a mismatch here would be
    // an out-of-bounds access on the device
    double macs = 0;
    for (int i = 0; i < n_layers; ++i) {
        const vq_layer_desc& L = layers[i];
        const bool multi = L.op == VQ_OP_CONV && L.seg_count > 0;
        VQ_REQUIRE(L.src >= 0 && L.src < n_tensors && L.dst > 0 && L.dst < n_tensors && L.src != L.dst,
                   "layer %d: bad tensor slots %d -> %d", i, L.src, L.dst);
        const vq_tensor_desc& ts = tensors[L.src];
        const vq_tensor_desc& td = tensors[L.dst];
        // one crop of any slot must stay addressable with a signed 32-bit byte offset (the kernels' buffer offsets);
        // larger batches are cut into crop ranges per launch (LaunchItem::max_crops)
        VQ_REQUIRE((size_t)ts.h * ts.w * ts.c * sizeof(float) <= 0x7FFFFFF0u && (size_t)td.h * td.w * td.c * sizeof(float) <= 0x7FFFFFF0u,
                   "layer %d: one crop of a tensor slot exceeds 2 GiB", i);
        VQ_REQUIRE(L.src_coff >= 0 && L.cin > 0 && L.src_coff + L.cin <= ts.c, "layer %d: reads channels [%d,%d) of a %d-channel slot",
                   i, L.src_coff, L.src_coff + L.cin, ts.c);
        VQ_REQUIRE(multi || (L.dst_coff >= 0 && L.cout > 0 && L.dst_coff + L.cout <= td.c),
                   "layer %d: writes channels [%d,%d) of a %d-channel slot", i, L.dst_coff, L.dst_coff + L.cout, td.c);
        if (multi) {
            VQ_REQUIRE(L.seg_first >= 0 && L.seg_first + L.seg_count <= n_segments, "layer %d: segments outside the table", i);
            int sum = 0;
            for (int q = 0; q < L.seg_count; ++q) {
                const vq_conv_segment& sg = segments[L.seg_first + q];
                VQ_REQUIRE(sg.dst > 0 && sg.dst < n_tensors && sg.dst != L.src, "layer %d segment %d: bad slot", i, q);
                const vq_tensor_desc& t2 = tensors[sg.dst];
                VQ_REQUIRE(sg.cout > 0 && sg.cout % 32 == 0 && sg.dst_coff >= 0 && sg.dst_coff % 4 == 0 && sg.dst_coff + sg.cout <= t2.c,
                           "layer %d segment %d: channels [%d,%d) do not fit a %d-channel slot (cout must be a multiple of 32)", i, q,
                           sg.dst_coff, sg.dst_coff + sg.cout, t2.c);
                VQ_REQUIRE(t2.h == td.h && t2.w == td.w && t2.c % 4 == 0, "layer %d segment %d: spatial size differs", i, q);
                sum += sg.cout;
            }
            VQ_REQUIRE(sum == L.cout, "layer %d: segments cover %d of %d output channels", i, sum, L.cout);
        }
        VQ_REQUIRE(L.k >= 1 && L.stride >= 1 && L.pad >= 0 && L.pad < L.k, "layer %d: bad kernel/stride/pad", i);
        VQ_REQUIRE(ts.c % 4 == 0 && td.c % 4 == 0 && L.src_coff % 4 == 0 && L.dst_coff % 4 == 0 && L.cin % 4 == 0 && L.cout % 4 == 0,
                   "layer %d: channel counts and offsets must be multiples of 4", i);
        if (L.op == VQ_OP_CONV) {
            VQ_REQUIRE(L.k <= 8, "layer %d: conv kernels up to 8x8 (the tap mask is 64 bits)", i);
            if (L.pre_pool_k > 0) {
                VQ_REQUIRE(L.pre_pool_k == 3 && L.pre_pool_stride >= 1 && L.pre_pool_stride <= 3,
                           "layer %d: the pooled-input form takes a 3x3 max window with stride 1..3", i);
                VQ_REQUIRE(L.k == 1 && L.stride == 1 && L.pad == 0 && L.cin % 32 == 0, "layer %d: the pooled-input form is a 1x1 convolution over a multiple of 32 channels", i);
                VQ_REQUIRE(td.h == pool_out_size(ts.h, 3, L.pre_pool_stride, 0) && td.w == pool_out_size(ts.w, 3, L.pre_pool_stride, 0),
                           "layer %d: pooled-input size mismatch (Caffe ceil rule)", i);
            } else
            VQ_REQUIRE(td.h == (ts.h + 2 * L.pad - L.k) / L.stride + 1 && td.w == (ts.w + 2 * L.pad - L.k) / L.stride + 1,
                       "layer %d: conv output size mismatch", i);
            int64_t kp = (int64_t)(L.k * L.k * L.cin + KPAD - 1) / KPAD * KPAD;
            if (L.src == 0 && input->s2d_pad >= 0 && input->s2d_order == 1) {      // x-major stem: [Cout][steps][4][4], see launch_conv_layer
                VQ_REQUIRE(L.k == 4 && L.cin == 4 * in_channels && L.stride == 1 && L.pad == 0 && L.src_coff == 0 && L.pre_pool_k == 0 &&
                               L.seg_count == 0 && input->s2d_kernel == 7,
                           "layer %d: the x-major space-to-depth stem is a 7x7 / stride-2 convolution reading the whole input slot", i);
                kp = stem_rows_kp(in_channels);
            }
            VQ_REQUIRE(L.w_off >= 0 && L.w_off % 4 == 0 && L.w_off + (int64_t)L.cout * kp <= blob_floats,
                       "layer %d: weights outside the blob", i);
            VQ_REQUIRE(L.b_off >= 0 && L.b_off + L.cout <= blob_floats, "layer %d: bias outside the blob", i);
            VQ_REQUIRE(L.cin % KPAD == 0 || L.src_coff == 0, "layer %d: small-Cin convolution must read a whole slot", i);
            VQ_REQUIRE(L.cin % KPAD == 0 || L.cin == ts.c, "layer %d: small-Cin convolution must read a whole slot", i);
            macs += (double)td.h * td.w * L.cout * first_layer_k2c(*input, L);   // algorithmic, un-padded
        } else if (is_wino(L.op)) {
            VQ_REQUIRE(L.k == 3 && L.stride == 1 && L.pad == 1 && L.seg_count == 0, "layer %d: Winograd form is 3x3 / stride 1 / pad 1, one destination", i);
            VQ_REQUIRE(L.cin % (L.op == VQ_OP_CONV_WINOGRAD16 ? 16 : 8) == 0 && L.cout % 32 == 0,
                       "layer %d: Winograd form needs Cin %% 8 == 0 (16-tile units: %% 16) and Cout %% 32 == 0", i);
            VQ_REQUIRE(td.h == ts.h && td.w == ts.w, "layer %d: conv output size mismatch", i);
            VQ_REQUIRE(L.w_off >= 0 && L.w_off % 4 == 0 &&
                           L.w_off + (int64_t)(L.op == VQ_OP_CONV_WINOGRAD16 ? 2 : 1) * 16 * L.cout * L.cin <= blob_floats,
                       "layer %d: transformed filters outside the blob", i);
            VQ_REQUIRE(L.b_off >= 0 && L.b_off % 4 == 0 && L.b_off + L.cout <= blob_floats, "layer %d: bias outside the blob", i);
            macs += (double)td.h * td.w * L.cout * L.cin * 9;   // algorithmic (direct-form) count
        } else if (L.op == VQ_OP_MAXPOOL || L.op == VQ_OP_AVGPOOL) {
            VQ_REQUIRE(L.cin == L.cout, "layer %d: pooling keeps the channel count", i);
            VQ_REQUIRE(!(L.op == VQ_OP_AVGPOOL && L.has_bias) || (L.b_off >= 0 && L.b_off % 4 == 0 && L.b_off + L.cout <= blob_floats),
                       "layer %d: pooling bias outside the blob", i);
            VQ_REQUIRE(td.h == pool_out_size(ts.h, L.k, L.stride, L.pad) && td.w == pool_out_size(ts.w, L.k, L.stride, L.pad),
                       "layer %d: pooling output size mismatch (Caffe ceil rule)", i);
        } else if (L.op == VQ_OP_GLOBAL_AVGPOOL) {
            VQ_REQUIRE(L.cin == L.cout && td.h == 1 && td.w == 1, "layer %d: global pool must write a 1x1 slot", i);
        } else {
            return fail(VQ_E_INVALID, "layer %d: unknown op %d", i, L.op);
        }
    }
    int ndev = 0;
    VQ_HIP(hipGetDeviceCount(&ndev));
    VQ_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard g(device);
    auto* net = new vq_tsn;
    net->device = device;
    net->max_crops = max_crops;
    net->in_channels = in_channels;
    net->input = *input;
    net->tensors.assign(tensors, tensors + n_tensors);
    net->layers.assign(layers, layers + n_layers);
    net->feature_slot = feature_slot;
    net->D = tensors[feature_slot].c;
    net->blob_floats = blob_floats;
    net->flops_per_crop = 2.0 * macs;
    for (int i = 0; i < n_layers; ++i)
        if (layers[i].op == VQ_OP_GLOBAL_AVGPOOL && layers[i].dst == feature_slot && layers[i].dst_coff == 0 && layers[i].cout == net->D)
            net->consensus_layer = i;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) net->cus = prop.multiProcessorCount;
    net->slots.assign(n_tensors, nullptr);
    auto bail = [&](const char* what, hipError_t e) {
        tsn_free(net);
        delete net;
        return fail(e == hipErrorOutOfMemory ? VQ_E_NOMEM : VQ_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    net->slot_bytes.assign(n_tensors, 0);
    for (int i = 0; i < n_tensors; ++i) {
        const size_t bytes = (size_t)max_crops * tensors[i].h * tensors[i].w * tensors[i].c * sizeof(float);
        hipError_t e = pool_alloc(device, bytes, (void**)&net->slots[i]);
        if (e != hipSuccess) return bail("hipMalloc(activation slot)", e);
        net->slot_bytes[i] = bytes;
    }
    net->blob_bytes = (size_t)blob_floats * sizeof(float);
    hipError_t e = pool_alloc(device, net->blob_bytes, (void**)&net->blob);
    if (e != hipSuccess) return bail("hipMalloc(weights)", e);
    e = hipMemcpy(net->blob, blob_host, (size_t)blob_floats * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail("hipMemcpy(weights)", e);
    {   // Which direct convolutions run split over K: single-destination layers with aligned channels on maps of at most
        // 7 x 7 (M = 49 x crops: at any usual batch fewer output tiles than the chip has room for, each a long serial K chain
        // on one wave per SIMD), cut into slices of at least 16 K-steps of 32 -- at most 4 slices.  On the 14 x 14 maps the
        // same cut LOSES (3c/3x3 0.058 -> 0.076 ms at 96 crops: enough tiles already, the scratch round trip costs more).
        // A function of the layer alone -- never of the batch actually run -- so a crop's features do not depend on the batch
        // it travels in.  VQ_TSN_SPLITK=0 turns it off (A/B measurements).
        net->ksplit.assign(n_layers, 1);
        const char* sk = getenv("VQ_TSN_SPLITK");
        size_t scratch = 0;
        int max_cout = 4;
        for (int i = 0; i < n_layers && !(sk && atoi(sk) == 0); ++i) {
            const vq_layer_desc& L = layers[i];
            const vq_tensor_desc& td = tensors[L.dst];
            if (L.op != VQ_OP_CONV || L.seg_count > 0 || L.pre_pool_k > 0 || L.cin % KPAD != 0 || td.h * td.w > 49) continue;
            const int steps = L.k * L.k * L.cin / KPAD;
            int ks = std::min(4, steps / 16);
            while (ks > 1 && steps % ks != 0) --ks;
            if (ks <= 1) continue;
            net->ksplit[i] = ks;
            scratch = std::max(scratch, (size_t)td.h * td.w * L.cout);
            max_cout = std::max(max_cout, L.cout);
        }
        if (scratch > 0) {
            net->split_crop_floats = scratch + (size_t)max_cout;
            net->split_slice_floats = (size_t)max_crops * net->split_crop_floats;
            e = pool_alloc(device, net->split_slice_floats * 4 * sizeof(float), (void**)&net->split_scratch);      // up to 4 slices
            if (e != hipSuccess) return bail("hipMalloc(split-K scratch)", e);
            e = hipMalloc((void**)&net->zero_bias, (size_t)max_cout * sizeof(float));
            if (e != hipSuccess) return bail("hipMalloc(zero bias)", e);
            e = hipMemset(net->zero_bias, 0, (size_t)max_cout * sizeof(float));
            if (e != hipSuccess) return bail("hipMemset(zero bias)", e);
        }
    }
    {   // destination tables: one entry per 32 output columns of every convolution
        std::vector<ConvSeg> table;
        net->seg_table_off.assign(n_layers, 0);
        for (int i = 0; i < n_layers; ++i) {
            const vq_layer_desc& L = layers[i];
            if (L.op != VQ_OP_CONV) continue;
            net->seg_table_off[i] = (int)table.size();
            if (net->ksplit[i] > 1) {       // slice q of a split layer writes plane q of the scratch, raw
                for (int q = 0; q < net->ksplit[i]; ++q)
                    for (int b = 0; b < (L.cout + 31) / 32; ++b)
                        table.push_back(ConvSeg{net->split_scratch + (size_t)q * net->split_slice_floats, L.cout, 0});
                continue;
            }
            if (L.seg_count > 0) {
                int col0 = 0;
                for (int q = 0; q < L.seg_count; ++q) {
                    const vq_conv_segment& sg = segments[L.seg_first + q];
                    for (int b = 0; b < sg.cout / 32; ++b)
                        table.push_back(ConvSeg{net->slots[sg.dst] + sg.dst_coff - col0, tensors[sg.dst].c, sg.relu});
                    col0 += sg.cout;
                }
            } else {
                for (int b = 0; b < (L.cout + 31) / 32; ++b)
                    table.push_back(ConvSeg{net->slots[L.dst] + L.dst_coff, tensors[L.dst].c, L.relu});
            }
        }
        e = vq::malloc_trim((void**)&net->seg_table, std::max<size_t>(table.size(), 1) * sizeof(ConvSeg));
        if (e != hipSuccess) return bail("vq::malloc_trim(destination tables)", e);
        e = hipMemcpy(net->seg_table, table.data(), table.size() * sizeof(ConvSeg), hipMemcpyHostToDevice);
        if (e != hipSuccess) return bail("hipMemcpy(destination tables)", e);
    }
    {
        if (const char* at = getenv("VQ_TSN_AUTOTUNE")) net->autotune = atoi(at) != 0;
        if (const char* gr = getenv("VQ_TSN_GROUP")) net->group_wino = atoi(gr) != 0;
        if (const char* gp = getenv("VQ_TSN_GROUP_POOL")) net->group_pool = atoi(gp) != 0;
        if (const char* po = getenv("VQ_TSN_POISON")) net->poison = atoi(po) != 0;
        if (const char* force = getenv("VQ_TSN_TILE")) {
            int bm = 0, bn = 0, bk = 32, pipe = 0;   // "BMxBN", "BMxBNxBK" or "BMxBNxBKxP" (P = 1: pipelined kernel)
            if (sscanf(force, "%dx%dx%dx%d", &bm, &bn, &bk, &pipe) >= 2)
                for (int i = 0; i < kNumTiles; ++i)
                    if (kTiles[i].bm == bm && kTiles[i].bn == bn && kTiles[i].bk == bk && kTiles[i].pipe == pipe) net->forced_tile = i;
        }
        const char* sp = getenv("VQ_TSN_SPLIT");
        if (sp && strchr(sp, ',')) {                   // "2,1": sub-batches of 2/3 and 1/3 of the crops
            for (const char* q = sp; *q;) {
                net->split_parts.push_back(std::max(atoi(q), 1));
                q = strchr(q, ',');
                if (!q) break;
                ++q;
            }
            net->n_split = std::min<int>((int)net->split_parts.size(), 8);
            net->split_parts.resize(net->n_split);
        } else {
            net->n_split = std::min(std::max(sp ? atoi(sp) : 2, 1), 8);
            net->split_parts.assign(net->n_split, 1);
        }
        build_items(net, segments);
        net->split_streams.assign(net->n_split, nullptr);
        net->join_ev.assign(net->n_split, nullptr);
        for (int l = 1; l < net->n_split; ++l) {
            e = hipStreamCreateWithFlags(&net->split_streams[l], hipStreamNonBlocking);
            if (e != hipSuccess) return bail("hipStreamCreate(sub-batch)", e);
            e = hipEventCreateWithFlags(&net->join_ev[l], hipEventDisableTiming);
            if (e != hipSuccess) return bail("hipEventCreate(sub-batch join)", e);
        }
        e = hipEventCreateWithFlags(&net->fork_ev, hipEventDisableTiming);
        if (e != hipSuccess) return bail("hipEventCreate(fork)", e);
    }
    e = vq::malloc_trim((void**)&net->zeros, 256);
    if (e != hipSuccess) return bail("vq::malloc_trim(zero page)", e);
    e = hipMemset(net->zeros, 0, 256);
    if (e != hipSuccess) return bail("hipMemset(zero page)", e);
    e = vq::malloc_trim((void**)&net->mean_dev, (size_t)in_channels * sizeof(float));
    if (e != hipSuccess) return bail("vq::malloc_trim(mean)", e);
    e = vq::malloc_trim((void**)&net->feat_dev, (size_t)max_crops * net->D * sizeof(double));
    if (e != hipSuccess) return bail("vq::malloc_trim(features)", e);
    *out = net;
    return VQ_OK;
}

int vq_tsn_destroy(vq_tsn* net) {
    if (!net) return VQ_OK;
    {
        DeviceGuard g(net->device);
        // everything that may still read or write the blocks must be done before the pool hands them to the next extractor: the
        // handle's stream, its sub-batch streams, a stream set earlier (the hipFree this replaced waited for the whole device too)
        (void)hipDeviceSynchronize();
        tsn_free(net);
    }
    delete net;
    return VQ_OK;
}

int vq_tsn_set_stream(vq_tsn* net, void* s) {
    VQ_REQUIRE(net, "net is NULL");
    std::lock_guard<std::mutex> lk(net->mu);
    net->stream = (hipStream_t)s;
    return VQ_OK;
}

// The launches of one forward, in order, on net->stream (and the sub-batch streams): preprocess, the launch items, the consensus.
static int forward_launches(vq_tsn* net, const uint8_t* src, int n_crops, int T, int n_split, const std::vector<int>& sub,
                            const std::vector<int>& sub_off, hipEvent_t* ev) {
    const vq_tensor_desc& t0 = net->tensors[0];
    const int in_c = net->in_channels;
    const int n_items = (int)net->items.size();
    // uint8 crops [crop0, crop0 + n) -> the fp32 input slot, on stream st
    auto preprocess = [&](int crop0, int n, hipStream_t st) -> int {
        const uint8_t* from = src + (size_t)crop0 * net->input.h * net->input.w * in_c;
        float* to = net->slots[0] + (size_t)crop0 * t0.h * t0.w * t0.c;
        if (net->input.s2d_pad < 0) {
            const int64_t nchunk = (int64_t)n * net->input.h * net->input.w * (t0.c / 4);
            preprocess_kernel<<<cdiv(nchunk, 256), 256, 0, st>>>(from, to, nchunk, in_c, t0.c, net->mean_dev);
        } else {
            const int64_t nchunk = (int64_t)n * t0.h * t0.w * in_c;
            if (in_c == 3 && t0.c == 12)
                preprocess_s2d3_kernel<<<cdiv(nchunk / 3, 256), 256, 0, st>>>(from, to, nchunk / 3, net->input.h, net->input.w, t0.h, t0.w,
                                                                               net->input.s2d_pad, net->mean_dev, net->input.s2d_order);
            else
                preprocess_s2d_kernel<<<cdiv(nchunk, 256), 256, 0, st>>>(from, to, nchunk, net->input.h, net->input.w, in_c, t0.h, t0.w,
                                                                        net->input.s2d_pad, net->mean_dev, net->input.s2d_order);
        }
        VQ_CHECK_LAUNCH();
        return VQ_OK;
    };
    if (n_split > 1) {
        // Batch split: the sub-batches are independent, so each runs the whole launch list -- its share of the preprocessing included
        // -- on its own stream with no synchronisation in between; one sub-batch's launch ramp and tail overlap the other's steady
        // state.  (Measured in round 5, tools/trace_prod.sh: the two streams run in step, the GPU is idle 0.007 of 2.6 ms; letting
        // sub-batch b start k launches behind sub-batch b-1, so that memory-bound and compute-bound launches meet, LOSES:
        // 12 090 -> 11 950 / 11 870 / 11 670 clips/s for k = 1 / 2 / 3: the lone tail costs more than the mixing gains.)
        VQ_HIP(hipEventRecord(net->fork_ev, net->stream));
        for (int l = 1; l < n_split; ++l) VQ_HIP(hipStreamWaitEvent(net->split_streams[l], net->fork_ev, 0));
        for (int sb = 0; sb < n_split; ++sb) {
            const int rc = preprocess(sub_off[sb], sub[sb], sb > 0 ? net->split_streams[sb] : net->stream);
            if (rc != VQ_OK) return rc;
        }
        for (const LaunchItem& it : net->items)
            for (int sb = 0; sb < n_split; ++sb) {
                net->ls = sb > 0 ? net->split_streams[sb] : net->stream;
                const int rc = run_item(net, it, sub_off[sb], sub[sb], tile_key(sub[sb], true));
                if (rc != VQ_OK) return rc;
            }
        net->ls = net->stream;
        for (int l = 1; l < n_split; ++l) {
            VQ_HIP(hipEventRecord(net->join_ev[l], net->split_streams[l]));
            VQ_HIP(hipStreamWaitEvent(net->stream, net->join_ev[l], 0));
        }
    } else {
        const int prc = preprocess(0, n_crops, net->stream);
        if (prc != VQ_OK) return prc;
        net->ls = net->stream;
        for (int q = 0; q < n_items; ++q) {
            if (ev) {
                net->ev_start = ev[2 * q];
                net->ev_stop = ev[2 * q + 1];
            }
            const int rc = run_item(net, net->items[q], 0, n_crops, tile_key(n_crops, false));
            if (rc != VQ_OK) return rc;
        }
    }
    if (ev) ++net->profile_count;
    const int B = n_crops / T;
    if (!net->fused_consensus) {                  // consensus not fused into the global-pool launch: separate pass
        consensus_kernel<<<cdiv((int64_t)B * net->D, 256), 256, 0, net->stream>>>(net->slots[net->feature_slot], net->feat_dev, B, T,
                                                                                  net->D, net->D);
        VQ_CHECK_LAUNCH();
    }
    return VQ_OK;
}

int vq_tsn_forward(vq_tsn* net, const uint8_t* crops, int32_t crops_on_device, int32_t n_crops, int32_t T,
                   const float* mean_host, double* feat_host, float* per_snippet_host) {
    VQ_REQUIRE(net && crops && mean_host, "NULL argument");
    VQ_REQUIRE(n_crops > 0 && n_crops <= net->max_crops, "n_crops %d outside (0,%d]", n_crops, net->max_crops);
    VQ_REQUIRE(T > 0 && n_crops % T == 0, "n_crops (%d) must be a multiple of T (%d)", n_crops, T);
    std::lock_guard<std::mutex> lk(net->mu);
    DeviceGuard g(net->device);
    const int in_c = net->in_channels;
    const int64_t npix = (int64_t)n_crops * net->input.h * net->input.w;
    const uint8_t* src = crops;
    if (!crops_on_device) {
        const size_t bytes = (size_t)npix * in_c;
        if (net->crops_cap < bytes) {
            if (net->crops_dev) VQ_HIP(hipFree(net->crops_dev));
            net->crops_dev = nullptr;
            net->crops_cap = 0;
            VQ_HIP(vq::malloc_trim((void**)&net->crops_dev, bytes));
            net->crops_cap = bytes;
        }
        VQ_HIP(hipMemcpyAsync(net->crops_dev, crops, bytes, hipMemcpyHostToDevice, net->stream));
        src = net->crops_dev;
    }
    if (net->mean_cached.size() != (size_t)in_c || memcmp(net->mean_cached.data(), mean_host, in_c * sizeof(float)) != 0) {
        net->mean_cached.assign(mean_host, mean_host + in_c);     // pageable copy from a buffer that outlives this call
        VQ_HIP(hipMemcpyAsync(net->mean_dev, net->mean_cached.data(), in_c * sizeof(float), hipMemcpyHostToDevice, net->stream));
        VQ_HIP(hipStreamSynchronize(net->stream));
    }
    if (net->poison)
        for (size_t i = 0; i < net->slots.size(); ++i) {
            const vq_tensor_desc& t = net->tensors[i];
            VQ_HIP(hipMemsetAsync(net->slots[i], 0xFF, (size_t)net->max_crops * t.h * t.w * t.c * sizeof(float), net->stream));
        }
    net->cur_T = T;
    const int n_items = (int)net->items.size();
    hipEvent_t* ev = nullptr;
    const bool profiling = net->profile_depth > 0;
    if (profiling && net->profile_tick++ % net->profile_every == 0)
        ev = net->events.data() + (size_t)(net->profile_count % net->profile_depth) * (2 * n_items);
    // A sampled forward stays on the caller's stream (each duration is then the launch alone); so do the forwards between two
    // sampled ones (they issue exactly the same launches) unless vq_tsn_set_profile_split asked for the product's own mode there.
    int parts_sum = 0;
    for (int v : net->split_parts) parts_sum += v;
    bool one_stream = profiling && (ev || !net->profile_split);
    if (!one_stream && !net->split_pref.empty()) {       // a measured preference of the nearest batch size within 1.6x (default: split)
        int best = 0, pref = 1;
        for (const auto& kv : net->split_pref)
            if (10 * std::max(kv.first, n_crops) <= 16 * std::min(kv.first, n_crops) && (best == 0 || std::abs(kv.first - n_crops) < std::abs(best - n_crops))) {
                best = kv.first;
                pref = kv.second;
            }
        if (pref == 0) one_stream = true;
    }
    int n_split = (!one_stream && net->n_split > 1 && n_crops % parts_sum == 0) ? net->n_split : 1;
    std::vector<int> sub(n_split, n_crops), sub_off(n_split, 0);
    if (n_split > 1) {
        for (int sb = 0, o = 0; sb < n_split; ++sb) {
            sub[sb] = n_crops / parts_sum * net->split_parts[sb];
            sub_off[sb] = o;
            o += sub[sb];
        }
    }
    {   // the consensus rides in the global-pool launch when every launch of that layer covers whole clips
        bool whole = net->consensus_layer >= 0 && T <= kMaxFusedT;
        if (whole) {
            const int cap = net->items[net->item_of_layer[net->consensus_layer]].max_crops;
            for (int sb = 0; sb < n_split; ++sb)
                if (sub[sb] % T != 0 || sub_off[sb] % T != 0 || (sub[sb] > cap && cap % T != 0)) whole = false;
        }
        net->fused_consensus = whole;
    }
    for (int sb = 0; sb < n_split; ++sb) {      // n_split == 1: sub[0] is the whole batch
        const int rc = ensure_tuned(net, sub[sb], n_split > 1);
        if (rc != VQ_OK) return rc;
    }
    const int frc = forward_launches(net, src, n_crops, T, n_split, sub, sub_off, ev);
    if (frc != VQ_OK) return frc;
    net->last_crops = n_crops;
    const int B = n_crops / T;
    if (feat_host)
        VQ_HIP(hipMemcpyAsync(feat_host, net->feat_dev, (size_t)B * net->D * sizeof(double), hipMemcpyDeviceToHost, net->stream));
    if (per_snippet_host)
        VQ_HIP(hipMemcpyAsync(per_snippet_host, net->slots[net->feature_slot], (size_t)n_crops * net->D * sizeof(float),
                              hipMemcpyDeviceToHost, net->stream));
    if (feat_host || per_snippet_host || !crops_on_device) VQ_HIP(hipStreamSynchronize(net->stream));
    return VQ_OK;
}

int vq_tsn_feat_devptr(vq_tsn* net, void** feat_dev, void** per_snippet_dev) {
    VQ_REQUIRE(net, "net is NULL");
    if (feat_dev) *feat_dev = net->feat_dev;
    if (per_snippet_dev) *per_snippet_dev = net->slots[net->feature_slot];
    return VQ_OK;
}

int vq_tsn_read_tensor(vq_tsn* net, int32_t slot, int32_t n_crops, float* host) {
    VQ_REQUIRE(net && host, "NULL argument");
    VQ_REQUIRE(slot >= 0 && slot < (int)net->tensors.size(), "slot out of range");
    VQ_REQUIRE(n_crops > 0 && n_crops <= net->max_crops, "n_crops out of range");
    std::lock_guard<std::mutex> lk(net->mu);
    DeviceGuard g(net->device);
    const vq_tensor_desc& t = net->tensors[slot];
    VQ_HIP(hipMemcpyAsync(host, net->slots[slot], (size_t)n_crops * t.h * t.w * t.c * sizeof(float), hipMemcpyDeviceToHost,
                          net->stream));
    VQ_HIP(hipStreamSynchronize(net->stream));
    return VQ_OK;
}

int vq_tsn_set_profile(vq_tsn* net, int32_t depth) {
    VQ_REQUIRE(net, "net is NULL");
    VQ_REQUIRE(depth >= 0 && depth <= 1024, "profile depth must be in [0,1024]");
    std::lock_guard<std::mutex> lk(net->mu);
    DeviceGuard g(net->device);
    VQ_HIP(hipStreamSynchronize(net->stream));
    for (hipEvent_t e : net->events) (void)hipEventDestroy(e);
    net->events.clear();
    net->events.resize((size_t)depth * 2 * net->items.size());
    for (hipEvent_t& e : net->events) VQ_HIP(hipEventCreate(&e));
    net->profile_depth = depth;
    net->profile_count = 0;
    net->profile_tick = 0;
    return VQ_OK;
}

int vq_tsn_set_profile_every(vq_tsn* net, int32_t every) {
    VQ_REQUIRE(net && every >= 1, "every must be >= 1");
    std::lock_guard<std::mutex> lk(net->mu);
    net->profile_every = every;
    net->profile_tick = 0;
    return VQ_OK;
}

// Share of a grouped launch's time attributed to one member, as a rough time estimate: matrix-core work of a
// convolution (16 multiplies per tile and channel pair) at 100 TFLOP/s, bytes of a pooling layer at 4 TB/s.
static double wino_weight(const vq_tsn* net, int li) {
    const vq_layer_desc& L = net->layers[li];
    const vq_tensor_desc& td = net->tensors[L.dst];
    if (!is_wino(L.op)) {
        const vq_tensor_desc& ts = net->tensors[L.src];
        return ((double)ts.h * ts.w + (double)td.h * td.w) * L.cin * 4.0 / 4e12;
    }
    return 2.0 * 16.0 * ((td.h + 1) / 2) * ((td.w + 1) / 2) * (double)L.cin * L.cout / 100e12;
}

int vq_tsn_set_profile_split(vq_tsn* net, int32_t unsampled_split) {
    VQ_REQUIRE(net, "net is NULL");
    std::lock_guard<std::mutex> lk(net->mu);
    net->profile_split = unsampled_split != 0;
    return VQ_OK;
}

int vq_tsn_layer_times(vq_tsn* net, float* ms, double* flops, int32_t n_layers) {
    VQ_REQUIRE(net && ms, "NULL argument");
    VQ_REQUIRE(n_layers == (int)net->layers.size(), "n_layers must be %d", (int)net->layers.size());
    std::lock_guard<std::mutex> lk(net->mu);
    if (net->profile_depth == 0 || net->profile_count == 0) return fail(VQ_E_STATE, "no profiled forward has run");
    DeviceGuard g(net->device);
    VQ_HIP(hipStreamSynchronize(net->stream));
    const int sets = std::min(net->profile_count, net->profile_depth);
    const int n_items = (int)net->items.size();
    for (int i = 0; i < n_layers; ++i) ms[i] = 0.f;
    for (int sidx = 0; sidx < sets; ++sidx) {
        hipEvent_t* ev = net->events.data() + (size_t)sidx * (2 * n_items);
        for (int q = 0; q < n_items; ++q) {
            float t = 0.f;
            VQ_HIP(hipEventElapsedTime(&t, ev[2 * q], ev[2 * q + 1]));
            const LaunchItem& it = net->items[q];
            double wsum = 0;
            for (int m : it.layers) wsum += it.kind == 1 ? wino_weight(net, m) : 1.0;
            for (int m : it.layers)                               // a grouped launch: split by matrix-core work
                ms[m] += (float)(t / sets * ((it.kind == 1 ? wino_weight(net, m) : 1.0) / wsum));
        }
    }
    if (flops) {
        for (int i = 0; i < n_layers; ++i) {
            const vq_layer_desc& L = net->layers[i];
            const vq_tensor_desc& td = net->tensors[L.dst];
            flops[i] = is_conv(L.op) ? 2.0 * net->last_crops * td.h * td.w * L.cout * first_layer_k2c(net->input, L)
                                          : 0.0;
        }
    }
    return VQ_OK;
}

int vq_tsn_launch_items(vq_tsn* net, int32_t* item_of_layer, int32_t n_layers, int32_t* n_items) {
    VQ_REQUIRE(net && item_of_layer && n_items, "NULL argument");
    VQ_REQUIRE(n_layers == (int)net->layers.size(), "n_layers must be %d", (int)net->layers.size());
    for (int i = 0; i < n_layers; ++i) item_of_layer[i] = net->item_of_layer[i];
    *n_items = (int)net->items.size();
    return VQ_OK;
}

int vq_tsn_tuned_sizes(vq_tsn* net, int32_t* sizes, int32_t cap, int32_t* n) {
    VQ_REQUIRE(net && n && (cap == 0 || sizes), "NULL argument");
    std::lock_guard<std::mutex> lk(net->mu);
    std::set<int> distinct;
    for (const auto& kv : net->tuned) distinct.insert(kv.first >= kPairedKey ? kv.first - kPairedKey : kv.first);
    int k = 0;
    for (int v : distinct) {
        if (k < cap) sizes[k] = v;
        ++k;
    }
    *n = k;
    return VQ_OK;
}

int vq_device_pool_trim(void) {
    vq::device_pool_trim();
    return VQ_OK;
}

int vq_tsn_set_split(vq_tsn* net, int32_t n_crops, int32_t split) {
    VQ_REQUIRE(net && n_crops > 0, "bad argument");
    std::lock_guard<std::mutex> lk(net->mu);
    if (split < 0)
        net->split_pref.erase(n_crops);
    else
        net->split_pref[n_crops] = split ? 1 : 0;
    return VQ_OK;
}

int vq_tsn_tune(vq_tsn* net, int32_t n_crops, int32_t paired) {
    VQ_REQUIRE(net, "net is NULL");
    VQ_REQUIRE(n_crops > 0 && n_crops * (paired ? std::min(net->n_split, 2) : 1) <= net->max_crops, "n_crops out of range");
    std::lock_guard<std::mutex> lk(net->mu);
    DeviceGuard g(net->device);
    const int rc = autotune(net, n_crops, paired != 0);
    if (rc != VQ_OK) net->tuned.erase(tile_key(n_crops, paired != 0));
    VQ_HIP(hipStreamSynchronize(net->stream));
    return rc;
}

int vq_tsn_tile_tables(vq_tsn* net, int32_t* sizes, int32_t* flags, int32_t cap, int32_t* n) {
    VQ_REQUIRE(net && n && (cap == 0 || (sizes && flags)), "NULL argument");
    std::lock_guard<std::mutex> lk(net->mu);
    int k = 0;
    for (const auto& kv : net->tuned) {
        if (k < cap) {
            const bool paired = kv.first >= kPairedKey;
            sizes[k] = paired ? kv.first - kPairedKey : kv.first;
            flags[k] = (paired ? 1 : 0) | (net->tuned_borrowed.count(kv.first) ? 2 : 0);
        }
        ++k;
    }
    *n = k;
    return VQ_OK;
}

int vq_tsn_get_tiles(vq_tsn* net, int32_t n_crops, int32_t paired, int32_t* tiles, int32_t n_layers) {
    VQ_REQUIRE(net && tiles, "NULL argument");
    VQ_REQUIRE(n_layers == (int)net->layers.size(), "n_layers must be %d", (int)net->layers.size());
    std::lock_guard<std::mutex> lk(net->mu);
    auto it = net->tuned.find(tile_key(n_crops, paired != 0));
    for (int i = 0; i < n_layers; ++i) {
        tiles[4 * i] = tiles[4 * i + 1] = tiles[4 * i + 2] = tiles[4 * i + 3] = 0;
        if (is_wino(net->layers[i].op)) {   // 32 tiles (128 pixels) x 32 (v+1) channels, 8 channels per step
            const int lead = net->items[net->item_of_layer[i]].layers[0];   // a grouped launch runs ONE variant: its first member's
            const int v = it != net->tuned.end() ? it->second[lead] : 0;
            tiles[4 * i] = v >= 2 ? 64 : 128;             // pixels per unit: 16 or 32 tiles of 2 x 2
            tiles[4 * i + 1] = 32 * ((v & 1) + 1);
            tiles[4 * i + 2] = v >= 2 ? 16 : 8;           // channels per step
            tiles[4 * i + 3] = 2;
            continue;
        }
        if (net->layers[i].op != VQ_OP_CONV) continue;
        const vq_tensor_desc& td = net->tensors[net->layers[i].dst];
        const int t = it != net->tuned.end() ? it->second[i] : heuristic_tile(n_crops * td.h * td.w, net->layers[i].cout, net->cus);
        tiles[4 * i] = kTiles[t].bm;
        tiles[4 * i + 1] = kTiles[t].bn;
        tiles[4 * i + 2] = kTiles[t].bk;
        tiles[4 * i + 3] = kTiles[t].pipe;
    }
    return VQ_OK;
}

int vq_tsn_layer_tiles(vq_tsn* net, int32_t n_crops, int32_t* tiles, int32_t n_layers) {
    VQ_REQUIRE(net, "NULL argument");
    bool paired = false;
    {
        std::lock_guard<std::mutex> lk(net->mu);
        paired = net->tuned.find(tile_key(n_crops, false)) == net->tuned.end() && net->tuned.find(tile_key(n_crops, true)) != net->tuned.end();
    }
    return vq_tsn_get_tiles(net, n_crops, paired ? 1 : 0, tiles, n_layers);
}

int vq_tsn_set_tiles(vq_tsn* net, int32_t n_crops, int32_t paired, const int32_t* tiles, int32_t n_layers) {
    VQ_REQUIRE(net && tiles, "NULL argument");
    VQ_REQUIRE(n_layers == (int)net->layers.size(), "n_layers must be %d", (int)net->layers.size());
    VQ_REQUIRE(n_crops > 0 && n_crops <= net->max_crops, "n_crops out of range");
    std::vector<int> choice(net->layers.size(), 0);
    for (int i = 0; i < n_layers; ++i) {
        if (is_wino(net->layers[i].op)) {
            const int bn = tiles[4 * i + 1], bk = tiles[4 * i + 2];
            const bool t16 = tiles[4 * i] == 64 && bk == 16;
            VQ_REQUIRE(tiles[4 * i + 3] == 2 && (bn == 32 || bn == 64) && (t16 || (tiles[4 * i] == 128 && bk == 8)) &&
                           (!t16 || net->layers[i].op == VQ_OP_CONV_WINOGRAD16),
                       "layer %d: no Winograd variant for tile %dx%dx%d", i, tiles[4 * i], bn, bk);
            choice[i] = bn / 32 - 1 + (t16 ? 2 : 0);
            continue;
        }
        if (net->layers[i].op != VQ_OP_CONV) continue;
        int found = -1;
        for (int t = 0; t < kNumTiles; ++t)
            if (kTiles[t].bm == tiles[4 * i] && kTiles[t].bn == tiles[4 * i + 1] && kTiles[t].bk == tiles[4 * i + 2] &&
                kTiles[t].pipe == tiles[4 * i + 3])
                found = t;
        VQ_REQUIRE(found >= 0, "layer %d: no kernel for tile %dx%dx%d pipe=%d", i, tiles[4 * i], tiles[4 * i + 1], tiles[4 * i + 2],
                   tiles[4 * i + 3]);
        choice[i] = found;
    }
    std::lock_guard<std::mutex> lk(net->mu);
    const int key = tile_key(n_crops, paired != 0);
    net->tuned[key] = choice;
    net->tuned_borrowed.erase(key);
    return VQ_OK;
}

// "The tiling of a forward of n_crops, however it runs": the one-stream table of that size and the paired tables of the sub-batch
// sizes the default forward cuts it into.
int vq_tsn_set_layer_tiles(vq_tsn* net, int32_t n_crops, const int32_t* tiles, int32_t n_layers) {
    int rc = vq_tsn_set_tiles(net, n_crops, 0, tiles, n_layers);
    if (rc != VQ_OK) return rc;
    int parts_sum = 0;
    for (int v : net->split_parts) parts_sum += v;
    if (net->n_split > 1 && parts_sum > 0 && n_crops % parts_sum == 0)
        for (int v : net->split_parts) {
            rc = vq_tsn_set_tiles(net, n_crops / parts_sum * v, 1, tiles, n_layers);
            if (rc != VQ_OK) return rc;
        }
    return VQ_OK;
}

int vq_tsn_flops_per_crop(vq_tsn* net, double* flops) {
    VQ_REQUIRE(net && flops, "NULL argument");
    *flops = net->flops_per_crop;
    return VQ_OK;
}

}  // extern "C"
