// Hot path A on gfx950: the frozen TSN BN-Inception forward pass + segment consensus.
//
// What it replaces (paths relative to the reference checkout):
//   src/features_GPU_compute/calcSig_wOF.py:88-113  per-snippet CaffeNet forward (rgb / flow)
//   src/features_GPU_compute/calcSig_wOF.py:82      np.array(frame_features).mean(axis=0)  (consensus)
//   src/features_GPU_compute/models/ucf101/tsn_bn_inception_{rgb,flow}_deploy.prototxt  (the layers)
//
// Layout: activations NHWC fp32 in HBM, one tensor slot per blob; a Concat output is one slot and
// its producers write at their channel offset (zero-copy concat).  Convolution + frozen BN + ReLU is a
// single implicit-GEMM kernel: M = crops*Ho*Wo pixels, N = Cout, K = k*k*Cin, computed on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, 256 FLOP/clk/CU) from LDS-staged,
// register-prefetched tiles.  All B*T crops of a batch go through each layer in one launch.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#include "vq_common.h"

using namespace vq;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// preprocess: uint8 NHWC crops -> fp32 NHWC (channels padded to a multiple of 4), minus mean
// ------------------------------------------------------------------------------------------------
__global__ void preprocess_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t npix, int C, int Cpad,
                                  const float* __restrict__ mean) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    for (int c = 0; c < Cpad; ++c) dst[p * Cpad + c] = c < C ? (float)src[p * C + c] - mean[c] : 0.0f;
}

// ------------------------------------------------------------------------------------------------
// convolution (+ folded BN + ReLU) as implicit GEMM on v_mfma_f32_32x32x2_f32
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* in;
    float* out;
    const float* w;      // [Cout][Kp], k index = (kh*k + kw)*Cin + c, zero padded to Kp
    const float* bias;   // [Cout]
    const float* zeros;  // >= 16 bytes of zeros: where masked-off lanes load from (no branches around loads)
    int H, W, Cs_in, coff_in, Cin;
    int Ho, Wo, Cs_out, coff_out, Cout;
    int k, stride, pad;
    int M, Kp, relu;
    int tiles_m, tiles_n;
};

constexpr int BK = 32;         // K elements staged per step
constexpr int LDS_LD = BK + 4; // padded row: 16 consecutive rows hit 16 distinct 16-byte bank slots

template <int BM, int BN>
struct ConvSmem {
    float a[2][BM][LDS_LD];
    float b[2][BN][LDS_LD];
};

// XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2); give each XCD a contiguous run of
// tiles, with the N tiles of one M tile adjacent, so the gathered activation tile is re-read from L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int BM, int BN, int WM, int WN, bool SMALL_CIN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;   // 32x32 accumulator tiles per wave
    constexpr int RA = BM / 32, RB = BN / 32;             // rows staged per thread
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ConvSmem<BM, BN>& sm = *reinterpret_cast<ConvSmem<BM, BN>*>(smem_raw);

    const int tid = threadIdx.x;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / a.tiles_n) * BM;
    const int n0 = (tile % a.tiles_n) * BN;

    // ---- staging roles: thread -> (row lrow + 32 i, 16-byte column lcol)
    const int lrow = tid >> 3, lcol = tid & 7;
    // Rows past M get an input row far outside the image, so the ordinary bounds test routes them to the
    // zero page; every pointer below stays derived from a kernel argument (global address space -- a null
    // alternative would demote the loads to flat_load, which also counts against lgkmcnt).
    const float* a_img[RA];
    int a_ih0[RA], a_iw0[RA];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + lrow + 32 * i;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int n_img = mm / HoWo, rem = mm - n_img * HoWo;
        const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
        a_img[i] = a.in + (size_t)n_img * a.H * a.W * a.Cs_in + a.coff_in;
        a_ih0[i] = ok ? oh * a.stride - a.pad : -(1 << 20);
        a_iw0[i] = ow * a.stride - a.pad;
    }
    const float* b_row[RB];
    bool b_ok[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + lrow + 32 * i;
        b_ok[i] = n < a.Cout;
        b_row[i] = a.w + (size_t)(b_ok[i] ? n : 0) * a.Kp + lcol * 4;
    }

    floatx4 ra[RA], rb[RB];   // ext-vector values (a HIP float4 struct copy becomes a memcpy the optimiser leaves in scratch)
    int kh = 0, kw = 0, c0 = 0;   // aligned mode: walk (tap, channel chunk) without divisions
    // Every lane always issues its loads: lanes that fall into the zero padding (or past M / Cout) read the
    // zero page instead, so there is no branch and no exec-mask juggling around the global loads.
    // (Macros, not lambdas: by-reference captures of the staging arrays ended up in scratch memory.)
#define VQ_LOAD_TILES(KC)                                                                                          \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                                           \
            const float* src = a.zeros;                                                                            \
            if (SMALL_CIN) {                                                                                       \
                const int kidx = ((KC) * 8 + lcol) * 4;                                                            \
                const int tap = kidx / a.Cin, c = kidx - tap * a.Cin;                                              \
                const int th = tap / a.k, tw = tap - th * a.k;                                                     \
                const int ih = a_ih0[i] + th, iw = a_iw0[i] + tw;                                                  \
                if (tap < a.k * a.k && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)   \
                    src = a_img[i] + ((size_t)ih * a.W + iw) * a.Cs_in + c;                                        \
            } else {                                                                                               \
                const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;                                                  \
                if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)                                  \
                    src = a_img[i] + ((size_t)ih * a.W + iw) * a.Cs_in + c0 + lcol * 4;                            \
            }                                                                                                      \
            ra[i] = *reinterpret_cast<const floatx4*>(src);                                                        \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                             \
            rb[i] = *reinterpret_cast<const floatx4*>(b_ok[i] ? b_row[i] + (size_t)(KC) * BK : a.zeros);           \
        if (!SMALL_CIN) {                                                                                          \
            c0 += BK;                                                                                              \
            if (c0 >= a.Cin) {                                                                                     \
                c0 = 0;                                                                                            \
                if (++kw == a.k) {                                                                                 \
                    kw = 0;                                                                                        \
                    ++kh;                                                                                          \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
    }
#define VQ_STORE_TILES(BUF)                                                                                        \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < RA; ++i)                                                             \
            *reinterpret_cast<floatx4*>(&sm.a[BUF][lrow + 32 * i][lcol * 4]) = ra[i];                              \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                             \
            *reinterpret_cast<floatx4*>(&sm.b[BUF][lrow + 32 * i][lcol * 4]) = rb[i];                              \
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
    float bias_v[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + 32 * j + l31;
        bias_v[j] = a.bias[n < a.Cout ? n : 0];
    }

    const int nk = a.Kp / BK;
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

    // ---- epilogue: + bias (folded BN), ReLU, store at the channel offset of the destination slot.
    // C/D map of the 32x32 MFMA: column (N) = lane & 31, row (M) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5):
    // every store instruction writes two 128-byte runs (32 consecutive channels of two pixels).
#define VQ_STORE_OUT(GUARD)                                                                          \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                 \
        const int n = n0 + wn * (BN / WN) + 32 * j + l31;                                            \
        if (n < a.Cout) {                                                                            \
            const float bv = bias_v[j];                                                              \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                         \
                const int mb = m0 + wm * (BM / WM) + 32 * i + 4 * half;                              \
                float* o = a.out + (size_t)mb * a.Cs_out + a.coff_out + n;                           \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                     \
                    const int dm = (r & 3) + 8 * (r >> 2);                                           \
                    float v = acc[i][j][r] + bv;                                                     \
                    if (a.relu) v = fmaxf(v, 0.f);                                                   \
                    if (!(GUARD) || mb + dm < a.M) o[(size_t)dm * a.Cs_out] = v;                     \
                }                                                                                    \
            }                                                                                        \
        }                                                                                            \
    }
    if (m0 + BM <= a.M) {   // workgroup-uniform: only the last M tile needs row guards
        VQ_STORE_OUT(false)
    } else {
        VQ_STORE_OUT(true)
    }
#undef VQ_LOAD_TILES
#undef VQ_STORE_TILES
#undef VQ_STORE_OUT
}

// ------------------------------------------------------------------------------------------------
// pooling (Caffe semantics: ceil-mode output size; MAX ignores padding; AVE divides by the window
// clipped to the padded extent and accumulates h-major in fp32)
// ------------------------------------------------------------------------------------------------
struct PoolArgs {
    const float* in;
    float* out;
    int H, W, Cs_in, coff_in, C;
    int Ho, Wo, Cs_out, coff_out;
    int k, stride, pad;
    int64_t total;   // n * Ho * Wo * C/4
};

template <bool IS_MAX>
__global__ __launch_bounds__(256) void pool_kernel(PoolArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    const int c4n = a.C >> 2;
    const int c4 = (int)(i % c4n);
    int64_t pix = i / c4n;
    const int pw = (int)(pix % a.Wo);
    pix /= a.Wo;
    const int ph = (int)(pix % a.Ho);
    const int64_t n = pix / a.Ho;
    int hs = ph * a.stride - a.pad, ws = pw * a.stride - a.pad;
    int he = min(hs + a.k, a.H + a.pad), we = min(ws + a.k, a.W + a.pad);
    const float pool_size = (float)((he - hs) * (we - ws));
    hs = max(hs, 0);
    ws = max(ws, 0);
    he = min(he, a.H);
    we = min(we, a.W);
    const float* base = a.in + (size_t)n * a.H * a.W * a.Cs_in + a.coff_in + c4 * 4;
    float4 acc = IS_MAX ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) {
            const float4 v = *reinterpret_cast<const float4*>(base + ((size_t)h * a.W + w) * a.Cs_in);
            if (IS_MAX) {
                acc.x = fmaxf(acc.x, v.x);
                acc.y = fmaxf(acc.y, v.y);
                acc.z = fmaxf(acc.z, v.z);
                acc.w = fmaxf(acc.w, v.w);
            } else {
                acc.x += v.x;
                acc.y += v.y;
                acc.z += v.z;
                acc.w += v.w;
            }
        }
    if (!IS_MAX) {
        acc.x /= pool_size;
        acc.y /= pool_size;
        acc.z /= pool_size;
        acc.w /= pool_size;
    }
    float* o = a.out + (((size_t)n * a.Ho + ph) * a.Wo + pw) * a.Cs_out + a.coff_out + c4 * 4;
    *reinterpret_cast<float4*>(o) = acc;
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
struct ConvTile {
    int bm, bn;
};

struct vq_tsn {
    std::mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    int cus = 256;
    int max_crops = 0;
    int in_channels = 0;
    std::vector<vq_tensor_desc> tensors;
    std::vector<vq_layer_desc> layers;
    std::vector<ConvTile> tiles;          // per layer (conv only)
    std::vector<float*> slots;            // device activations, max_crops each
    float* zeros = nullptr;               // 256 bytes of zeros (load target of masked lanes)
    float* blob = nullptr;                // weights + biases
    int64_t blob_floats = 0;
    int feature_slot = -1, D = 0;
    uint8_t* crops_dev = nullptr;
    size_t crops_cap = 0;
    float* mean_dev = nullptr;
    double* feat_dev = nullptr;           // [max_crops][D] (B <= max_crops)
    double flops_per_crop = 0;
    int last_crops = 0;
    int profile_depth = 0;                // > 0: HIP events around every layer launch (bench roofline accounting)
    int profile_count = 0;                // profiled forwards so far (ring of profile_depth event sets)
    std::vector<hipEvent_t> events;       // profile_depth x (n_layers + 1)
};

static void tsn_free(vq_tsn* net) {
    for (hipEvent_t e : net->events) (void)hipEventDestroy(e);
    net->events.clear();
    for (float* p : net->slots)
        if (p) (void)hipFree(p);
    if (net->blob) (void)hipFree(net->blob);
    if (net->zeros) (void)hipFree(net->zeros);
    if (net->crops_dev) (void)hipFree(net->crops_dev);
    if (net->mean_dev) (void)hipFree(net->mean_dev);
    if (net->feat_dev) (void)hipFree(net->feat_dev);
}

template <int BM, int BN, int WM, int WN, bool SMALL>
static int launch_conv_t(vq_tsn* net, ConvArgs& a) {
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.Cout, BN);
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, SMALL>;
    const size_t lds = sizeof(ConvSmem<BM, BN>);
    VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<a.tiles_m * a.tiles_n, 256, lds, net->stream>>>(a);
    VQ_CHECK_LAUNCH();
    return VQ_OK;
}

// Candidate tilings (BM x BN, wave grid).  The chooser below scores them per layer.
static const ConvTile kTiles[] = {{128, 128}, {128, 96}, {128, 64}, {128, 32}, {64, 64}, {64, 128}};

static double tile_score(const ConvTile& t, int M, int N, int cus) {
    const long long tiles = (long long)cdiv(M, t.bm) * cdiv(N, t.bn);
    const long long rounds = (tiles + cus - 1) / cus;          // MFMA-bound: a CU's tiles run back to back
    const double useful = (double)M * N;
    const double spent = (double)rounds * cus * t.bm * t.bn;
    // larger tiles move fewer bytes per MAC and amortise the prologue/epilogue
    const double intrinsic = t.bm * t.bn >= 128 * 128 ? 1.0 : t.bm * t.bn >= 128 * 96 ? 0.97 : t.bm * t.bn >= 128 * 64 ? 0.93 : 0.85;
    return useful / spent * intrinsic;
}

static ConvTile choose_tile(int M, int N, int cus) {
    ConvTile best = kTiles[0];
    double bs = -1;
    for (const ConvTile& t : kTiles) {
        const double s = tile_score(t, M, N, cus);
        if (s > bs) {
            bs = s;
            best = t;
        }
    }
    return best;
}

template <bool SMALL>
static int launch_conv(vq_tsn* net, ConvArgs& a, ConvTile t) {
    if (t.bm == 128 && t.bn == 128) return launch_conv_t<128, 128, 2, 2, SMALL>(net, a);
    if (t.bm == 128 && t.bn == 96) return launch_conv_t<128, 96, 4, 1, SMALL>(net, a);
    if (t.bm == 128 && t.bn == 64) return launch_conv_t<128, 64, 2, 2, SMALL>(net, a);
    if (t.bm == 128 && t.bn == 32) return launch_conv_t<128, 32, 4, 1, SMALL>(net, a);
    if (t.bm == 64 && t.bn == 128) return launch_conv_t<64, 128, 2, 2, SMALL>(net, a);
    return launch_conv_t<64, 64, 2, 2, SMALL>(net, a);
}

static int run_layer(vq_tsn* net, int li, int n_crops) {
    const vq_layer_desc& L = net->layers[li];
    const vq_tensor_desc& ts = net->tensors[L.src];
    const vq_tensor_desc& td = net->tensors[L.dst];
    if (L.op == VQ_OP_CONV) {
        ConvArgs a;
        a.in = net->slots[L.src];
        a.out = net->slots[L.dst];
        a.w = net->blob + L.w_off;
        a.bias = net->blob + L.b_off;
        a.zeros = net->zeros;
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
        a.stride = L.stride;
        a.pad = L.pad;
        a.M = n_crops * td.h * td.w;
        a.Kp = (L.k * L.k * L.cin + BK - 1) / BK * BK;
        a.relu = L.relu;
        const bool small = (L.cin % BK) != 0;
        const char* force = getenv("VQ_TSN_TILE");   // tuning aid: "BMxBN"
        ConvTile t = choose_tile(a.M, a.Cout, net->cus);
        if (force) {
            int bm = 0, bn = 0;
            if (sscanf(force, "%dx%d", &bm, &bn) == 2)
                for (const ConvTile& c : kTiles)
                    if (c.bm == bm && c.bn == bn) t = c;
        }
        return small ? launch_conv<true>(net, a, t) : launch_conv<false>(net, a, t);
    }
    if (L.op == VQ_OP_MAXPOOL || L.op == VQ_OP_AVGPOOL) {
        PoolArgs a;
        a.in = net->slots[L.src];
        a.out = net->slots[L.dst];
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
        a.total = (int64_t)n_crops * td.h * td.w * (L.cin / 4);
        const int64_t blocks = (a.total + 255) / 256;
        if (L.op == VQ_OP_MAXPOOL)
            pool_kernel<true><<<(unsigned)blocks, 256, 0, net->stream>>>(a);
        else
            pool_kernel<false><<<(unsigned)blocks, 256, 0, net->stream>>>(a);
        VQ_CHECK_LAUNCH();
        return VQ_OK;
    }
    if (L.op == VQ_OP_GLOBAL_AVGPOOL) {
        gavgpool_kernel<<<cdiv((int64_t)n_crops * L.cin, 256), 256, 0, net->stream>>>(
            net->slots[L.src], net->slots[L.dst], n_crops, ts.h * ts.w, ts.c, L.src_coff, L.cin, td.c, L.dst_coff);
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

extern "C" {

int vq_tsn_create(const vq_tensor_desc* tensors, int32_t n_tensors, const vq_layer_desc* layers, int32_t n_layers,
                  const float* blob_host, int64_t blob_floats, int32_t in_channels, int32_t feature_slot, int32_t max_crops,
                  int32_t device, vq_tsn** out) {
    VQ_REQUIRE(out, "out is NULL");
    *out = nullptr;
    VQ_REQUIRE(tensors && layers && blob_host, "NULL argument");
    VQ_REQUIRE(n_tensors > 0 && n_layers > 0 && blob_floats > 0 && max_crops > 0, "sizes must be positive");
    VQ_REQUIRE(feature_slot > 0 && feature_slot < n_tensors, "feature_slot out of range");
    VQ_REQUIRE(tensors[feature_slot].h == 1 && tensors[feature_slot].w == 1, "feature slot must be 1x1xD");
    VQ_REQUIRE(tensors[0].c % 4 == 0, "input slot channels must be padded to a multiple of 4 (got %d)", tensors[0].c);
    VQ_REQUIRE(in_channels > 0 && in_channels <= tensors[0].c && tensors[0].c - in_channels < 4, "in_channels %d does not fit the %d-channel input slot",
               in_channels, tensors[0].c);
    // validate every layer against the tensor table BEFORE anything is launched: a mismatch here would be
    // an out-of-bounds access on the device
    double macs = 0;
    for (int i = 0; i < n_layers; ++i) {
        const vq_layer_desc& L = layers[i];
        VQ_REQUIRE(L.src >= 0 && L.src < n_tensors && L.dst > 0 && L.dst < n_tensors && L.src != L.dst,
                   "layer %d: bad tensor slots %d -> %d", i, L.src, L.dst);
        const vq_tensor_desc& ts = tensors[L.src];
        const vq_tensor_desc& td = tensors[L.dst];
        VQ_REQUIRE(L.src_coff >= 0 && L.cin > 0 && L.src_coff + L.cin <= ts.c, "layer %d: reads channels [%d,%d) of a %d-channel slot",
                   i, L.src_coff, L.src_coff + L.cin, ts.c);
        VQ_REQUIRE(L.dst_coff >= 0 && L.cout > 0 && L.dst_coff + L.cout <= td.c, "layer %d: writes channels [%d,%d) of a %d-channel slot",
                   i, L.dst_coff, L.dst_coff + L.cout, td.c);
        VQ_REQUIRE(L.k >= 1 && L.stride >= 1 && L.pad >= 0 && L.pad < L.k, "layer %d: bad kernel/stride/pad", i);
        VQ_REQUIRE(ts.c % 4 == 0 && td.c % 4 == 0 && L.src_coff % 4 == 0 && L.dst_coff % 4 == 0 && L.cin % 4 == 0,
                   "layer %d: channel counts and offsets must be multiples of 4", i);
        if (L.op == VQ_OP_CONV) {
            VQ_REQUIRE(td.h == (ts.h + 2 * L.pad - L.k) / L.stride + 1 && td.w == (ts.w + 2 * L.pad - L.k) / L.stride + 1,
                       "layer %d: conv output size mismatch", i);
            const int64_t kp = (int64_t)(L.k * L.k * L.cin + BK - 1) / BK * BK;
            VQ_REQUIRE(L.w_off >= 0 && L.w_off % 4 == 0 && L.w_off + (int64_t)L.cout * kp <= blob_floats,
                       "layer %d: weights outside the blob", i);
            VQ_REQUIRE(L.b_off >= 0 && L.b_off + L.cout <= blob_floats, "layer %d: bias outside the blob", i);
            VQ_REQUIRE(L.cin % BK == 0 || L.src_coff == 0, "layer %d: small-Cin convolution must read a whole slot", i);
            VQ_REQUIRE(L.cin % BK == 0 || L.cin == ts.c, "layer %d: small-Cin convolution must read a whole slot", i);
            macs += (double)td.h * td.w * L.cout * (L.src == 0 ? in_channels : L.cin) * L.k * L.k;   // algorithmic, un-padded
        } else if (L.op == VQ_OP_MAXPOOL || L.op == VQ_OP_AVGPOOL) {
            VQ_REQUIRE(L.cin == L.cout, "layer %d: pooling keeps the channel count", i);
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
    net->tensors.assign(tensors, tensors + n_tensors);
    net->layers.assign(layers, layers + n_layers);
    net->feature_slot = feature_slot;
    net->D = tensors[feature_slot].c;
    net->blob_floats = blob_floats;
    net->flops_per_crop = 2.0 * macs;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) net->cus = prop.multiProcessorCount;
    net->slots.assign(n_tensors, nullptr);
    auto bail = [&](const char* what, hipError_t e) {
        tsn_free(net);
        delete net;
        return fail(e == hipErrorOutOfMemory ? VQ_E_NOMEM : VQ_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    for (int i = 0; i < n_tensors; ++i) {
        const size_t bytes = (size_t)max_crops * tensors[i].h * tensors[i].w * tensors[i].c * sizeof(float);
        hipError_t e = hipMalloc((void**)&net->slots[i], bytes);
        if (e != hipSuccess) return bail("hipMalloc(activation slot)", e);
    }
    hipError_t e = hipMalloc((void**)&net->blob, (size_t)blob_floats * sizeof(float));
    if (e != hipSuccess) return bail("hipMalloc(weights)", e);
    e = hipMemcpy(net->blob, blob_host, (size_t)blob_floats * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail("hipMemcpy(weights)", e);
    e = hipMalloc((void**)&net->zeros, 256);
    if (e != hipSuccess) return bail("hipMalloc(zero page)", e);
    e = hipMemset(net->zeros, 0, 256);
    if (e != hipSuccess) return bail("hipMemset(zero page)", e);
    e = hipMalloc((void**)&net->mean_dev, (size_t)in_channels * sizeof(float));
    if (e != hipSuccess) return bail("hipMalloc(mean)", e);
    e = hipMalloc((void**)&net->feat_dev, (size_t)max_crops * net->D * sizeof(double));
    if (e != hipSuccess) return bail("hipMalloc(features)", e);
    *out = net;
    return VQ_OK;
}

int vq_tsn_destroy(vq_tsn* net) {
    if (!net) return VQ_OK;
    {
        DeviceGuard g(net->device);
        (void)hipStreamSynchronize(net->stream);
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

int vq_tsn_forward(vq_tsn* net, const uint8_t* crops, int32_t crops_on_device, int32_t n_crops, int32_t T,
                   const float* mean_host, double* feat_host, float* per_snippet_host) {
    VQ_REQUIRE(net && crops && mean_host, "NULL argument");
    VQ_REQUIRE(n_crops > 0 && n_crops <= net->max_crops, "n_crops %d outside (0,%d]", n_crops, net->max_crops);
    VQ_REQUIRE(T > 0 && n_crops % T == 0, "n_crops (%d) must be a multiple of T (%d)", n_crops, T);
    std::lock_guard<std::mutex> lk(net->mu);
    DeviceGuard g(net->device);
    const vq_tensor_desc& t0 = net->tensors[0];
    const int in_c = net->in_channels;
    const int64_t npix = (int64_t)n_crops * t0.h * t0.w;
    const uint8_t* src = crops;
    if (!crops_on_device) {
        const size_t bytes = (size_t)npix * in_c;
        if (net->crops_cap < bytes) {
            if (net->crops_dev) VQ_HIP(hipFree(net->crops_dev));
            net->crops_dev = nullptr;
            net->crops_cap = 0;
            VQ_HIP(hipMalloc((void**)&net->crops_dev, bytes));
            net->crops_cap = bytes;
        }
        VQ_HIP(hipMemcpyAsync(net->crops_dev, crops, bytes, hipMemcpyHostToDevice, net->stream));
        src = net->crops_dev;
    }
    VQ_HIP(hipMemcpyAsync(net->mean_dev, mean_host, in_c * sizeof(float), hipMemcpyHostToDevice, net->stream));
    preprocess_kernel<<<cdiv(npix, 256), 256, 0, net->stream>>>(src, net->slots[0], npix, in_c, t0.c, net->mean_dev);
    VQ_CHECK_LAUNCH();
    hipEvent_t* ev = nullptr;
    if (net->profile_depth > 0) {
        ev = net->events.data() + (size_t)(net->profile_count % net->profile_depth) * (net->layers.size() + 1);
        VQ_HIP(hipEventRecord(ev[0], net->stream));
    }
    for (int li = 0; li < (int)net->layers.size(); ++li) {
        const int rc = run_layer(net, li, n_crops);
        if (rc != VQ_OK) return rc;
        if (ev) VQ_HIP(hipEventRecord(ev[li + 1], net->stream));
    }
    if (ev) ++net->profile_count;
    const int B = n_crops / T;
    consensus_kernel<<<cdiv((int64_t)B * net->D, 256), 256, 0, net->stream>>>(net->slots[net->feature_slot], net->feat_dev, B, T,
                                                                              net->D, net->D);
    VQ_CHECK_LAUNCH();
    net->last_crops = n_crops;
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
    net->events.resize((size_t)depth * (net->layers.size() + 1));
    for (hipEvent_t& e : net->events) VQ_HIP(hipEventCreate(&e));
    net->profile_depth = depth;
    net->profile_count = 0;
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
    for (int i = 0; i < n_layers; ++i) ms[i] = 0.f;
    for (int sidx = 0; sidx < sets; ++sidx) {
        hipEvent_t* ev = net->events.data() + (size_t)sidx * (n_layers + 1);
        for (int i = 0; i < n_layers; ++i) {
            float t = 0.f;
            VQ_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
            ms[i] += t / sets;                                   // mean over the profiled forwards
        }
    }
    if (flops) {
        for (int i = 0; i < n_layers; ++i) {
            const vq_layer_desc& L = net->layers[i];
            const vq_tensor_desc& td = net->tensors[L.dst];
            flops[i] = L.op == VQ_OP_CONV ? 2.0 * net->last_crops * td.h * td.w * L.cout * (L.src == 0 ? net->in_channels : L.cin) * L.k * L.k
                                          : 0.0;
        }
    }
    return VQ_OK;
}

int vq_tsn_flops_per_crop(vq_tsn* net, double* flops) {
    VQ_REQUIRE(net && flops, "NULL argument");
    *flops = net->flops_per_crop;
    return VQ_OK;
}

}  // extern "C"
