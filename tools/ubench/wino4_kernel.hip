// EXPERIMENT (round 3), not part of the library: 3x3 / stride-1 / pad-1 convolution + folded BN + ReLU as Winograd
// F(4x4, 3x3) on the fp32 matrix cores, for the large maps (conv2/3x3 on 56x56, inception_3a..3c on 28x28 of
// src/features_GPU_compute/models/ucf101/tsn_bn_inception_rgb_deploy.prototxt).  Correct (1e-6 .. 5e-6 max|y| against a direct
// fp64 convolution: tools/ubench/wino4_check.hip) but NOT faster than the F(2x2,3x3) kernel of csrc/vq_wino.hip on this chip,
// so the product keeps F(2x2,3x3); numbers and the reasons are in profiles/README.md (round 3, "F(4x4,3x3)").
//
//     Y = A^T [ (G g G^T) . (B^T d B) ] A          d: 6x6 input patch, g: 3x3 filter, Y: 4x4 outputs
// with the interpolation points 0, +-1, +-2, inf (Lavin & Gray 2016):
//     B^T = [ 4  0 -5  0  1  0 ]   G = [ 1/4    0     0  ]   A^T = [ 1  1  1  1  1  0 ]
//           [ 0 -4 -4  1  1  0 ]       [-1/6  -1/6  -1/6 ]         [ 0  1 -1  2 -2  0 ]
//           [ 0  4 -4 -1  1  0 ]       [-1/6   1/6  -1/6 ]         [ 0  1  1  4  4  0 ]
//           [ 0 -2 -1  2  1  0 ]       [ 1/24  1/12  1/6 ]         [ 0  1 -1  8 -8  1 ]
//           [ 0  2 -1 -2  1  0 ]       [ 1/24 -1/12  1/6 ]
//           [ 0  4  0 -5  0  1 ]       [ 0      0     1  ]
// 36 multiplies per (tile of 16 outputs, cin, cout) instead of 144: 36 independent GEMMs
//     M_xi[cout][tile] = sum_cin U_xi[cout][cin] * V_xi[cin][tile],   xi = 6 i + j
// -- 4x fewer matrix-core cycles than the direct form, 1.78x fewer than F(2x2,3x3).
//
// This (third) shape: a workgroup of FOUR waves (one per SIMD -- six-wave workgroups do not pack two to a compute unit: the
// dispatcher starts every workgroup's waves at SIMD 0) owns 16 tiles x 32 output channels x all 36 positions on
// v_mfma_f32_16x16x4_f32 (72 accumulator registers per wave, three workgroups per compute unit); wave W owns the half rows
// 3 W .. 3 W + 2 of the 6x6 position grid (wave index = template parameter, so rows and coefficients are immediates).
//   activations: a loader task = one patch row of a tile, 2 of the stage's 8 channels (6 buffer_load_dwordx2, four lanes
//                cover a pixel's 32 bytes); transform along the row (12 packed instructions), h[r][j] to LDS (XOR-swizzled
//                k pairs: conflict-free 8-byte fragment reads); the transform along the columns when a wave reads its
//                fragment: V[i][j] = (h[A][j] + c1 h[B][j]) + c2 (h[C][j] + c3 h[D][j]).
//   filters:     U = G g G^T from the host (fp64, rounded once), [Cin/8][36][Cout][8] = fragment order, straight from L2
//                into registers a stage ahead.
//   K loop:      36 MFMAs per wave and stage, each followed by one pinned slice of the other work.
//   epilogue:    A^T along j in registers per half row, eight slots through LDS for the sum along i (two passes of two
//                output columns), ReLU, 16-byte NHWC stores.  Bias: column 1 of A^T is all ones -> accumulator of (1,1).
#include "vq_common.h"
#include "vq_tsn_kernels.h"

using namespace vq;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
typedef unsigned uintx2 __attribute__((__vector_size__(2 * sizeof(unsigned))));

namespace {

// Packed fp32 arithmetic for values that go to LDS or memory (inline asm: see vq_wino.hip for why the fragment side must not).
__device__ __forceinline__ floatx2 pk_add2(floatx2 x, floatx2 y) {
    floatx2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ floatx2 pk_sub2(floatx2 x, floatx2 y) {
    floatx2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ floatx2 pk_fma2(floatx2 x, floatx2 c, floatx2 z) {   // x * c + z
    floatx2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(c), "v"(z));
    return r;
}
__device__ __forceinline__ floatx4 pk_add(floatx4 x, floatx4 y) {
    const floatx2 lo = pk_add2(x.xy, y.xy), hi = pk_add2(x.zw, y.zw);
    return (floatx4){lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ floatx4 pk_sub(floatx4 x, floatx4 y) {
    const floatx2 lo = pk_sub2(x.xy, y.xy), hi = pk_sub2(x.zw, y.zw);
    return (floatx4){lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ floatx4 pk_fma(floatx4 x, floatx2 c, floatx4 z) {
    const floatx2 lo = pk_fma2(x.xy, c, z.xy), hi = pk_fma2(x.zw, c, z.zw);
    return (floatx4){lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

#ifdef VQ_WINO_PHASES   // tools/ubench/wino4_check.hip: where a workgroup's time goes
#define VQ_PHASE(K) \
    if (threadIdx.x == 0) a.phases[(size_t)blockIdx.x * 6 + (K)] = (long long)__builtin_readcyclecounter();
#else
#define VQ_PHASE(K)
#endif

constexpr int BP = 16;                        // tiles per workgroup = the columns of v_mfma_f32_16x16x4_f32
constexpr int KS = 8;                         // channels per stage (two MFMA k groups of 4)
constexpr int POS = BP * KS;                  // floats of one position's image h[r][j][tile][k]
constexpr int HS_ROW = 6 * POS;
constexpr int HS_STAGE = 6 * HS_ROW;
constexpr int EP_ROW = 36;                    // epilogue image: 32 channels of a tile + 4 floats of padding
constexpr int EP_SLOTS = 8;                   // rows 0, 1 (j < 3), 1 (j >= 3), 2, 3, 4 (j < 3), 4 (j >= 3), 5
constexpr int EP_FLOATS = EP_SLOTS * 2 * BP * EP_ROW;   // [slot][x of the pass][tile][EP_ROW]
constexpr int LDS_FLOATS = EP_FLOATS > 2 * HS_STAGE ? EP_FLOATS : 2 * HS_STAGE;

struct TileAt {
    int img, ty, tx;
};
__device__ __forceinline__ TileAt tile_at(int t, int img0, int ty0, int tx0, int th, int tw, unsigned s_th, unsigned s_tw) {
    const unsigned lin = (unsigned)(tx0 + t);
    const unsigned q1 = (lin * s_tw) >> 16;
    const unsigned ly = (unsigned)ty0 + q1;
    const unsigned q2 = (ly * s_th) >> 16;
    return TileAt{img0 + (int)q2, (int)(ly - q2 * (unsigned)th), (int)(lin - q1 * (unsigned)tw)};
}

// Row i of B^T as V_i = (h[a] + c1 h[b]) + c2 (h[c] + c3 h[d]); rows 0 and 5 have three terms (no d).
struct RowMix {
    int a, b, c, d;
    float c1, c3, c2;
    bool three;
};
__device__ constexpr RowMix row_mix(int i) {
    return i == 0   ? RowMix{4, 2, 0, 0, -5.f, 0.f, 4.f, true}
           : i == 1 ? RowMix{4, 2, 3, 1, -4.f, -4.f, 1.f, false}
           : i == 2 ? RowMix{4, 2, 3, 1, -4.f, -4.f, -1.f, false}
           : i == 3 ? RowMix{4, 2, 3, 1, -1.f, -1.f, 2.f, false}
           : i == 4 ? RowMix{4, 2, 3, 1, -1.f, -1.f, -2.f, false}
                    : RowMix{5, 3, 1, 1, -5.f, 0.f, 4.f, true};
}
// Wave W owns the half rows 3 W .. 3 W + 2 of the position grid (half row hr = row hr / 2, columns 3 (hr % 2) .. + 2):
// position p = 3 c + q of the wave is (row, column) below.
__device__ constexpr int pos_i(int W, int p) { return (3 * W + p / 3) >> 1; }
__device__ constexpr int pos_j(int W, int p) { return 3 * ((3 * W + p / 3) & 1) + p % 3; }

template <int W>
__device__ __forceinline__ void wino4_unit(const WinoJob& a, int unit, float* hs) {
    VQ_PHASE(0)
    const int tid = threadIdx.x;
    const int lane = tid & 63, t16 = lane & 15, kq = lane >> 4;
    // ---- which tiles, which channels: all on the scalar unit ---------------------------------------------------
    const int tile = xcd_remap(unit, a.n_units);
    const int pb = (int)magic_div((unsigned)tile, a.m_tiles_n);
    const int p0 = pb * BP;
    const int n0 = (tile - pb * a.tiles_n) * 32;
    const int tpi = a.th * a.tw;
    const int img0 = (int)magic_div((unsigned)p0, a.m_tpi);
    const int rem0 = p0 - img0 * tpi;
    const int ty0 = (int)magic_div((unsigned)rem0, a.m_tw);
    const int tx0 = rem0 - ty0 * a.tw;

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);

    // ---- loader role: task n = patch row W + 4 n of tile lt, channels 2 c2, 2 c2 + 1 of the stage's 8 (waves 0 and 1 have
    //      two tasks, waves 2 and 3 one); four lanes cover the 32 contiguous bytes of a (tile, pixel) --------------------
    constexpr int NT = W < 2 ? 2 : 1;
    unsigned pbase[NT], pfirst[NT], plast[NT];             // byte offsets for pixels 1..4, for pixel 0 and for pixel 5; 0xFFFFFFFF = zero padding
    int hs_store[NT];
    {
        const int lt = (lane >> 2), c2 = lane & 3;
        const TileAt ta = tile_at(lt, img0, ty0, tx0, a.th, a.tw, a.s_th, a.s_tw);
        const int x0 = 4 * ta.tx - 1;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int r = W + 4 * n;
            const int y = 4 * ta.ty - 1 + r;
            const bool row_ok = p0 + lt < a.P && (unsigned)y < (unsigned)a.H;
            // offset of pixel 1 of the row (x = 4 tx: never negative -- the range check sees the lane offset alone, the
            // scalar offset that selects the pixel and the stage is added behind it)
            const int base = (((ta.img * a.H + y) * a.W + x0 + 1) * a.Cs_in + a.coff_in + c2 * 2) * 4;
            pbase[n] = row_ok ? (unsigned)base : 0xFFFFFFFFu;
            pfirst[n] = (row_ok && x0 >= 0) ? (unsigned)(base - a.Cs_in * 4) : 0xFFFFFFFFu;
            plast[n] = (row_ok && x0 + 5 < a.W) ? (unsigned)base : 0xFFFFFFFFu;
            // LDS image of a position: [tile][k pair][2]; the k pair is XORed with (tile / 4) % 4 so that the 16 lanes of a
            // fragment read (one k pair, 16 tiles, 8 bytes at a stride of 32) touch all 32 banks
            hs_store[n] = r * HS_ROW + lt * KS + ((c2 ^ ((lt >> 2) & 3)) * 2);   // + j * POS (+ stage)
        }
    }
    const int px = a.Cs_in * 4;

    // ---- consumer role -------------------------------------------------------------------------------------------------
    const int f_lane = t16 * KS + ((kq ^ ((t16 >> 2) & 3)) * 2);
    const unsigned uvoff = (unsigned)((t16 * KS + kq * 2) * 4);            // lane part of the filter fragment address
    const unsigned u_step = (unsigned)(36 * a.Cout * KS * 4);               // bytes between consecutive 8-channel groups
    const unsigned u_pos = (unsigned)(a.Cout * KS * 4);                     // bytes between positions
    const unsigned u_n0 = (unsigned)(n0 * KS * 4);

    // Accumulators: acc[p][blk][r] of a lane is output channel 16 blk + 4 kq + r of tile t16.  Position (1,1) -- wave 0,
    // p = 7 -- starts at the bias.
    floatx4 acc[9][2];
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) acc[p][blk] = (floatx4){0.f, 0.f, 0.f, 0.f};
    if (W == 0) {
        const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, (unsigned)a.Cout * 4u, 0x00020000);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
            acc[7][blk] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (unsigned)((n0 + 16 * blk + 4 * kq) * 4), 0, 0));
    }

#ifdef VQ_EXP_NOMFMA      // timing experiments (tools/ubench/wino4_check.hip): drop one ingredient of the K loop
#define VQ_EXP_MFMA(ACC, A_, B_) ACC[0] += (A_) * (B_);
#else
#define VQ_EXP_MFMA(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(A_, B_, ACC, 0, 0, 0);
#endif
#ifdef VQ_EXP_NOREAD
#define VQ_EXP_READ false
#else
#define VQ_EXP_READ true
#endif
#ifdef VQ_EXP_NOLOAD
#define VQ_EXP_LOAD false
#else
#define VQ_EXP_LOAD true
#endif
#ifdef VQ_EXP_NOU
#define VQ_EXP_U false
#else
#define VQ_EXP_U true
#endif
    floatx2 d[NT][6];      // patch rows in flight (next stage)
    floatx2 bq[9][2];      // filter fragments of the current stage; reloaded for the next stage right after use

#define VQ_W_LOAD_PATCH(KST, N)                                                                                 \
    _Pragma("unroll") for (int c = 0; c < 6; ++c)                                                               \
        d[N][c] = __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(                             \
            in_rsrc, c == 0 ? pfirst[N] : (c == 5 ? plast[N] : pbase[N]), (KST) * (KS * 4) + (c == 0 ? 0 : c - 1) * px, 0));
#define VQ_W_LOAD_U(KST, P, BLK)                                                                                \
    bq[P][BLK] = __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(                              \
        u_rsrc, uvoff, (KST) * u_step + (unsigned)(6 * pos_i(W, P) + pos_j(W, P)) * u_pos + u_n0 + (BLK) * (16 * KS * 4), 0));
// transform along the row, h = d B, then to LDS stage ST -- in three slices that ride behind MFMAs of the running stage
#define VQ_W_H_PART0(N)                                                                                         \
    {                                                                                                           \
        const floatx2 m4 = {-4.f, -4.f};                                                                        \
        e1[N] = pk_fma2(d[N][2], m4, d[N][4]), e2[N] = pk_sub2(d[N][4], d[N][2]);                               \
        o1[N] = pk_fma2(d[N][1], m4, d[N][3]), o2[N] = pk_sub2(d[N][3], d[N][1]);                               \
    }
#define VQ_W_H_PART1(ST, N)                                                                                     \
    {                                                                                                           \
        float* dst = hs + (ST) * HS_STAGE + hs_store[N];                                                        \
        const floatx2 p2 = {2.f, 2.f}, n2 = {-2.f, -2.f};                                                       \
        *reinterpret_cast<floatx2*>(dst + 1 * POS) = pk_add2(e1[N], o1[N]);                                     \
        *reinterpret_cast<floatx2*>(dst + 2 * POS) = pk_sub2(e1[N], o1[N]);                                     \
        *reinterpret_cast<floatx2*>(dst + 3 * POS) = pk_fma2(o2[N], p2, e2[N]);                                 \
        *reinterpret_cast<floatx2*>(dst + 4 * POS) = pk_fma2(o2[N], n2, e2[N]);                                 \
    }
#define VQ_W_H_PART2(ST, N)                                                                                     \
    {                                                                                                           \
        float* dst = hs + (ST) * HS_STAGE + hs_store[N];                                                        \
        const floatx2 m5 = {-5.f, -5.f}, p4 = {4.f, 4.f};                                                       \
        *reinterpret_cast<floatx2*>(dst + 0 * POS) = pk_fma2(d[N][0], p4, pk_fma2(d[N][2], m5, d[N][4]));       \
        *reinterpret_cast<floatx2*>(dst + 5 * POS) = pk_fma2(d[N][1], p4, pk_fma2(d[N][3], m5, d[N][5]));       \
    }
#define VQ_W_STORE_H(ST, N)                                                                                     \
    { VQ_W_H_PART0(N) VQ_W_H_PART1(ST, N) VQ_W_H_PART2(ST, N) }
// fragment of position P: the h rows its grid row combines (three or four ds_read_b64), then V = (hA + c1 hB) + c2 (hC + c3 hD)
#define VQ_W_READ_A(ST, P)                                                                                      \
    {                                                                                                           \
        const RowMix rm = row_mix(pos_i(W, P));                                                                 \
        const float* hb = hs + (ST) * HS_STAGE + pos_j(W, P) * POS + f_lane;                                    \
        xa = *reinterpret_cast<const floatx2*>(hb + rm.a * HS_ROW);                                             \
        xb = *reinterpret_cast<const floatx2*>(hb + rm.b * HS_ROW);                                             \
        xc = *reinterpret_cast<const floatx2*>(hb + rm.c * HS_ROW);                                             \
        if (!rm.three) xd = *reinterpret_cast<const floatx2*>(hb + rm.d * HS_ROW);                              \
    }
#define VQ_W_AV(P)                                                                                              \
    {                                                                                                           \
        const RowMix rm = row_mix(pos_i(W, P));                                                                 \
        const floatx2 k1 = {rm.c1, rm.c1}, k2 = {rm.c2, rm.c2}, k3 = {rm.c3, rm.c3};                            \
        const floatx2 pp = __builtin_elementwise_fma(xb, k1, xa);                                               \
        const floatx2 qq = rm.three ? xc : __builtin_elementwise_fma(xd, k3, xc);                               \
        av = __builtin_elementwise_fma(qq, k2, pp);                                                             \
    }
// One stage on LDS buffer ST: 9 positions x 2 channel blocks x 2 k groups = 36 MFMAs, every one followed by a pinned slice
// of the other work, so that a wave keeps the matrix pipe fed by itself: the h rows of position p+1 are read behind the first
// MFMA of position p and combined behind its last; a position's filter registers are refilled for stage KST+1 as soon as
// their last MFMA is issued; the patch rows of stage KST+1 are requested behind the first positions and transformed / stored
// to the other LDS buffer behind the last ones (about two thirds of a stage later).
#define VQ_W_STAGE(ST, KST, NEXT)                                                                               \
    {                                                                                                           \
        floatx2 xa, xb, xc, xd, av;                                                                             \
        floatx2 e1[NT], e2[NT], o1[NT], o2[NT];                                                                 \
        VQ_W_READ_A(ST, 0)                                                                                      \
        VQ_W_AV(0)                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int m = 0; m < 36; ++m) {                                                        \
            const int p = m >> 2, sq = m & 3, blk = sq & 1, kg = sq >> 1;                                       \
            VQ_EXP_MFMA(acc[p][blk], bq[p][blk][kg], av[kg])                                                    \
            if (sq == 0 && p + 1 < 9 && VQ_EXP_READ) VQ_W_READ_A(ST, p + 1 < 9 ? p + 1 : 0)                     \
            if (NEXT) {                                                                                         \
                if (sq == 1 && p < NT && VQ_EXP_LOAD) VQ_W_LOAD_PATCH((KST) + 1, p < NT ? p : 0)                \
                if (sq == 1 && p == 5) VQ_W_H_PART0(0)                                                          \
                if (sq == 1 && p == 6) VQ_W_H_PART1((ST) ^ 1, 0)                                                \
                if (sq == 1 && p == 7) VQ_W_H_PART2((ST) ^ 1, 0)                                                \
                if (NT == 2 && sq == 1 && p == 6) VQ_W_H_PART0(NT - 1)                                          \
                if (NT == 2 && sq == 1 && p == 7) VQ_W_H_PART1((ST) ^ 1, NT - 1)                                \
                if (NT == 2 && sq == 1 && p == 8) VQ_W_H_PART2((ST) ^ 1, NT - 1)                                \
                if (sq >= 2 && VQ_EXP_U) VQ_W_LOAD_U((KST) + 1, p, blk)                                         \
            }                                                                                                   \
            if (sq == 3 && p + 1 < 9) VQ_W_AV(p + 1 < 9 ? p + 1 : 0)                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
        }                                                                                                       \
    }

    const int nk = a.Cin / KS;
#pragma unroll
    for (int n = 0; n < NT; ++n) VQ_W_LOAD_PATCH(0, n)
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        VQ_W_LOAD_U(0, p, 0)
        VQ_W_LOAD_U(0, p, 1)
    }
    {
        floatx2 e1[NT], e2[NT], o1[NT], o2[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) VQ_W_STORE_H(0, n)
    }
    __syncthreads();
    VQ_PHASE(1)
    int kc = 0;
    for (; kc + 1 < nk; ++kc) {
        const int st = kc & 1;
        VQ_W_STAGE(st, kc, true)
        __syncthreads();                          // buffer st^1 complete; everybody is done reading buffer st
    }
    VQ_W_STAGE(kc & 1, kc, false)
    VQ_PHASE(2)
#undef VQ_W_LOAD_PATCH
#undef VQ_W_LOAD_U
#undef VQ_W_STORE_H
#undef VQ_W_STAGE
#undef VQ_W_READ_A
#undef VQ_W_AV
#undef VQ_W_H_PART0
#undef VQ_W_H_PART1
#undef VQ_W_H_PART2

    // ---- epilogue: Y = A^T M A ---------------------------------------------------------------------------------------
    // Along j in registers, per half row or whole row of the wave:  s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4,
    //   t0 = m0 + s1 + s2,  t1 = d1 + 2 d2,  t2 = s1 + 4 s2,  t3 = d1 + 8 d2 + m5      (a half row keeps its own terms).
    // Wave W fills slots 2 W and 2 W + 1 of the exchange image ep[slot][x & 1][tile][channel] (pass p: columns x = 2p, 2p+1):
    // even waves a whole row (positions 0..5) and the j < 3 half of the next (6..8), odd waves the j >= 3 half of a row
    // (0..2) and the whole next row (3..8).  A reader item = (tile, x, 4 channels) adds the two halves of rows 1 and 4, sums
    // along i the same way and stores the four output rows of its column.
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
    float* ep_w = hs + ((2 * W) * 2 * BP + t16) * EP_ROW + 4 * kq;      // + (slot & 1) * 2 * BP * EP_ROW + xl * BP * EP_ROW + 16 blk
    const floatx2 two = {2.f, 2.f}, four = {4.f, 4.f}, eight = {8.f, 8.f};
    constexpr bool even = (W & 1) == 0;
    constexpr int full0 = even ? 0 : 3;      // first position of the wave's whole row
    constexpr int part0 = even ? 6 : 0;      // first position of its half row (j < 3 for even waves, j >= 3 for odd ones)
    __syncthreads();            // all waves are done with the K-loop image of the LDS
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            floatx4 tf[2], tp[2];                         // whole row / half row, columns 2 ps and 2 ps + 1
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
#define VQ_M(P) ((floatx2){acc[P][blk][2 * h2], acc[P][blk][2 * h2 + 1]})
                {
                    const floatx2 s1 = pk_add2(VQ_M(full0 + 1), VQ_M(full0 + 2)), d1 = pk_sub2(VQ_M(full0 + 1), VQ_M(full0 + 2));
                    const floatx2 s2 = pk_add2(VQ_M(full0 + 3), VQ_M(full0 + 4)), d2 = pk_sub2(VQ_M(full0 + 3), VQ_M(full0 + 4));
                    const floatx2 u0 = ps == 0 ? pk_add2(pk_add2(VQ_M(full0), s1), s2) : pk_fma2(s2, four, s1);
                    const floatx2 u1 = ps == 0 ? pk_fma2(d2, two, d1) : pk_add2(pk_fma2(d2, eight, d1), VQ_M(full0 + 5));
                    tf[0][2 * h2] = u0.x, tf[0][2 * h2 + 1] = u0.y;
                    tf[1][2 * h2] = u1.x, tf[1][2 * h2 + 1] = u1.y;
                }
                if (even) {                               // j = 0, 1, 2
                    const floatx2 s1 = pk_add2(VQ_M(part0 + 1), VQ_M(part0 + 2)), d1 = pk_sub2(VQ_M(part0 + 1), VQ_M(part0 + 2));
                    const floatx2 u0 = ps == 0 ? pk_add2(VQ_M(part0), s1) : s1, u1 = d1;
                    tp[0][2 * h2] = u0.x, tp[0][2 * h2 + 1] = u0.y;
                    tp[1][2 * h2] = u1.x, tp[1][2 * h2 + 1] = u1.y;
                } else {                                  // j = 3, 4, 5
                    const floatx2 s2 = pk_add2(VQ_M(part0), VQ_M(part0 + 1)), d2 = pk_sub2(VQ_M(part0), VQ_M(part0 + 1));
                    const floatx2 zero = {0.f, 0.f};
                    const floatx2 u0 = ps == 0 ? s2 : pk_fma2(s2, four, zero);
                    const floatx2 u1 = ps == 0 ? pk_fma2(d2, two, zero) : pk_fma2(d2, eight, VQ_M(part0 + 2));
                    tp[0][2 * h2] = u0.x, tp[0][2 * h2 + 1] = u0.y;
                    tp[1][2 * h2] = u1.x, tp[1][2 * h2 + 1] = u1.y;
                }
#undef VQ_M
            }
            // slot 2 W: even waves their whole row, odd waves their half row; slot 2 W + 1 the other one
            float* w0 = ep_w + 16 * blk;
            float* w1 = w0 + 2 * BP * EP_ROW;
            *reinterpret_cast<floatx4*>(w0) = even ? tf[0] : tp[0];
            *reinterpret_cast<floatx4*>(w0 + BP * EP_ROW) = even ? tf[1] : tp[1];
            *reinterpret_cast<floatx4*>(w1) = even ? tp[0] : tf[0];
            *reinterpret_cast<floatx4*>(w1 + BP * EP_ROW) = even ? tp[1] : tf[1];
        }
        __syncthreads();
        {
            const int et = tid >> 4, xl = (tid >> 3) & 1, ec = tid & 7;
            const TileAt te = tile_at(et, img0, ty0, tx0, a.th, a.tw, a.s_th, a.s_tw);
            const int oy = 4 * te.ty, ox = 4 * te.tx + 2 * ps + xl;
            const bool col_ok = p0 + et < a.P && ox < a.W;
            const int obase = (((te.img * a.H + oy) * a.W + ox) * a.Cs_out + a.coff_out + n0 + ec * 4) * 4;
            const int orow = a.W * a.Cs_out * 4;
            const float* ep_r = hs + (xl * BP + et) * EP_ROW + ec * 4;                 // + slot * 2 * BP * EP_ROW
#define VQ_SLOT(S) (*reinterpret_cast<const floatx4*>(ep_r + (S) * 2 * BP * EP_ROW))
            const floatx4 v0 = VQ_SLOT(0), v1 = pk_add(VQ_SLOT(1), VQ_SLOT(2)), v2 = VQ_SLOT(3);
            const floatx4 v3 = VQ_SLOT(4), v4 = pk_add(VQ_SLOT(5), VQ_SLOT(6)), v5 = VQ_SLOT(7);
#undef VQ_SLOT
            const floatx4 s1 = pk_add(v1, v2), d1 = pk_sub(v1, v2), s2 = pk_add(v3, v4), d2 = pk_sub(v3, v4);
            floatx4 y[4];
            y[0] = pk_add(pk_add(v0, s1), s2);
            y[1] = pk_fma(d2, two, d1);
            y[2] = pk_fma(s2, four, s1);
            y[3] = pk_add(pk_fma(d2, eight, d1), v5);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (a.relu) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) y[r][c] = relu1(y[r][c]);
                }
                const unsigned off = (col_ok && oy + r < a.H) ? (unsigned)(obase + r * orow) : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, y[r]), out_rsrc, off, 0, 0);
            }
        }
        if (ps == 0) __syncthreads();
    }
    VQ_PHASE(3)
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void wino_f4x4_3x3_kernel(WinoGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* hs = reinterpret_cast<float*>(smem_raw);        // [2][6][6][16][8] (K loop); [8][2][16][36] (an epilogue pass)
    const int b = blockIdx.x;
    if (b >= g.pool_unit0) {                             // the level's pooling layers: plain streaming work, no LDS
        int q = 0;
        if (g.n_pools > 1 && b >= g.pool[1].unit0) q = 1;
        const PoolArgs& pa = g.pool[q];
        const int unit = b - pa.unit0;
        if (unit >= pa.n_units) return;
        const int64_t first = (int64_t)unit * kPoolPerWG, last = first + kPoolPerWG < pa.total ? first + kPoolPerWG : pa.total;
        for (int64_t i = first + threadIdx.x; i < last; i += 256) {
            if (pa.is_max)
                pool_one<true>(pa, i);
            else
                pool_one<false>(pa, i);
        }
        return;
    }
    int j = 0;
#pragma unroll
    for (int q = 1; q < kWinoMaxJobs; ++q)
        if (q < g.n_jobs && b >= g.job[q].unit0) j = q;
    const WinoJob& a = g.job[j];
    const int unit = b - a.unit0;
    if (unit >= a.n_units) return;                       // padding workgroups between jobs
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0)
        wino4_unit<0>(a, unit, hs);
    else if (wave == 1)
        wino4_unit<1>(a, unit, hs);
    else if (wave == 2)
        wino4_unit<2>(a, unit, hs);
    else
        wino4_unit<3>(a, unit, hs);
}

unsigned magic_u32(unsigned d) { return d == 1 ? 0u : (unsigned)(0x100000000ull / d) + 1u; }   // 0 = "divide by one"

}  // namespace

namespace vq {

// The F(4x4,3x3) jobs of one graph level (+ the level's pooling layers) as one launch.  th / tw / P of a job count 4x4 tiles,
// its filters are laid out [Cin/4][36][Cout][4].
int launch_wino4_group(WinoGroup& g, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    VQ_REQUIRE(g.n_jobs >= 1 && g.n_jobs <= kWinoMaxJobs, "a Winograd launch carries 1..%d jobs", kWinoMaxJobs);
    VQ_REQUIRE(g.n_pools >= 0 && g.n_pools <= kWinoMaxPools, "a Winograd launch carries at most %d pooling layers", kWinoMaxPools);
    int units = 0;
    for (int q = 0; q < g.n_jobs; ++q) {
        WinoJob& a = g.job[q];
        VQ_REQUIRE(a.Cin % KS == 0 && a.W % 4 == 0 && a.Cout % 32 == 0 && a.Cs_out % 4 == 0 && a.coff_out % 4 == 0 && a.Cs_in % 2 == 0 && a.coff_in % 2 == 0,
                   "F(4x4,3x3) convolution needs Cin %% 8 == 0, W %% 4 == 0, Cout %% 32 == 0, 16-byte aligned output and 8-byte aligned input channel offsets");
        VQ_REQUIRE(a.in_bytes <= 0x7FFFFFF0u && a.out_bytes <= 0x7FFFFFF0u,
                   "Winograd convolution: a launch addresses its slots with signed 32-bit byte offsets (split the batch)");
        VQ_REQUIRE(a.th == (a.H + 3) / 4 && a.tw == (a.W + 3) / 4 && a.P % (a.th * a.tw) == 0, "F(4x4,3x3) job %d: tile counts do not match the map", q);
        a.tiles_n = a.Cout / 32;
        const long long n = (long long)cdiv(a.P, BP) * a.tiles_n;
        VQ_REQUIRE(n > 0 && n < (1 << 24), "Winograd job %d: %lld workgroups", q, n);
        a.unit0 = units;
        a.n_units = (int)n;
        units += ((int)n + 7) & ~7;
        const unsigned tpi = (unsigned)(a.th * a.tw);
        VQ_REQUIRE((unsigned long long)n * a.tiles_n < 0x100000000ull && (unsigned long long)(a.P + BP) * tpi < 0x100000000ull &&
                       a.tw <= 1024 && a.th <= 1024,
                   "Winograd job %d: shape outside the reciprocal-division range", q);
        a.m_tiles_n = magic_u32((unsigned)a.tiles_n);
        a.m_tpi = magic_u32(tpi);
        a.m_tw = magic_u32((unsigned)a.tw);
        a.s_tw = 65536u / (unsigned)a.tw + 1u;
        a.s_th = 65536u / (unsigned)a.th + 1u;
    }
    g.pool_unit0 = units;
    for (int q = 0; q < g.n_pools; ++q) {
        PoolArgs& pa = g.pool[q];
        const long long n = (pa.total + kPoolPerWG - 1) / kPoolPerWG;
        VQ_REQUIRE(n > 0 && n < (1 << 24), "pooling job %d: %lld workgroups", q, n);
        pa.unit0 = units;
        pa.n_units = (int)n;
        units += (int)n;
    }
    g.total_units = units;
    const size_t lds = LDS_FLOATS * sizeof(float);
    VQ_LAUNCH(wino_f4x4_3x3_kernel, units, 256, lds, stream, ev_start, ev_stop, g);
    VQ_CHECK_LAUNCH();
    return VQ_OK;
}

}  // namespace vq
