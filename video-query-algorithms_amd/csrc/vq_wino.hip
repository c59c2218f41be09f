// 3x3 / stride-1 / pad-1 convolution + folded BN + ReLU as Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
// The 3x3 stride-1 layers are 64.6 % of the BN-Inception MACs (from
// src/features_GPU_compute/models/ucf101/tsn_bn_inception_rgb_deploy.prototxt).  For a 2x2 output tile
//     Y = A^T [ (G g G^T) . (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter, "." elementwise
// turns 36 multiplies per (tile, cin, cout) into 16: the layer becomes 16 independent GEMMs
//     M_xi[tile][cout] = sum_cin V_xi[tile][cin] * U_xi[cin][cout],   xi = 4 i + j  (position in the 4x4 patch)
// that run on v_mfma_f32_32x32x2_f32, 2.25x fewer matrix-core cycles than the direct form.  Everything is fused in
// one kernel: the input transform happens on the way into LDS / out of LDS, the output transform in the epilogue.
//
// Work split.  A workgroup (4 waves) owns 32 tiles x BN output channels x all 16 positions; wave i owns row i of
// the 4x4 position grid (4 GEMMs, accumulators 4 x (BN/32) x 16 registers).  Cin is walked 8 channels per step.
//   activations: wave r loads patch row r of every tile (4 pixels x 8 channels), applies the column half of the
//                transform (h = d B, 4 adds per channel) and stores h[r][j] to LDS; the row half is applied when
//                wave i reads its A fragment: V[i][j] = h[ra][j] +- h[rb][j]  (two ds_read_b128 and 4 FMAs for 4 k's).
//   filters:     U is pre-transformed on the host (fp64, rounded once) and laid out [Cin/8][16][Cout][8], which
//                is exactly the MFMA B-fragment order: each wave loads its own fragments straight from L2 into
//                registers, a full step ahead, with no LDS traffic and no sharing between waves.
//   epilogue:    wave i reduces its 4 positions along j in registers (A^T along columns), the 4 waves exchange
//                through LDS for the reduction along i, then bias + ReLU and 16-byte NHWC stores.
// The k order of every output element is fixed (channels ascending in groups of 8), so the BN variants produce
// identical bits.  Rounding differs from the direct kernel (Winograd F(2,3) error is ~1-2.5x the direct fp32
// error, measured against the fp64 oracle: tests/test_tsn_gpu.py).
#include "vq_common.h"
#include "vq_tsn_kernels.h"

using namespace vq;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

// The VALU and the matrix pipe of a SIMD do not overlap well (tools/ubench/mfma_coissue.hip: every v_fma_f32 placed
// behind an MFMA adds ~3-4 cycles even with four waves per SIMD), so the transform arithmetic is written as packed
// fp32 instructions (v_pk_add_f32 / v_pk_fma_f32: two floats per lane per instruction; inline asm, because the compiler
// scalarises float4 subtraction).  Only used where the result goes to LDS (see pk_fma below).
typedef float floatx2 __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ floatx4 pk_add(floatx4 x, floatx4 y) {
    const floatx2 lo = pk_add2(x.xy, y.xy), hi = pk_add2(x.zw, y.zw);
    return (floatx4){lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ floatx4 pk_sub(floatx4 x, floatx4 y) {
    const floatx2 lo = pk_sub2(x.xy, y.xy), hi = pk_sub2(x.zw, y.zw);
    return (floatx4){lo.x, lo.y, hi.x, hi.y};
}
// x * s + z on the A-fragment side.  NOT inline asm: the result feeds an MFMA, and the compiler's hazard recogniser
// only inserts the VALU -> MFMA wait states for instructions it can see (an asm VALU in front of an MFMA read stale
// registers); the elementwise builtin lets it emit v_pk_fma_f32 where it wants and keep the hazards right.
__device__ __forceinline__ floatx4 pk_fma(floatx4 x, floatx2 s, floatx4 z) {
    const floatx2 lo = __builtin_elementwise_fma(x.xy, s, z.xy), hi = __builtin_elementwise_fma(x.zw, s, z.zw);
    return (floatx4){lo.x, lo.y, hi.x, hi.y};
}

#ifdef VQ_WINO_PHASES   // tools/ubench/wino_phases.hip: where a workgroup's time goes
#define VQ_PHASE(K)                                                                                               \
    if (threadIdx.x == 0) {                                                                                       \
        a.phases[(size_t)blockIdx.x * 6 + (K)] = (long long)__builtin_readcyclecounter();                         \
        if ((K) == 0) {                                                                                           \
            a.phases[(size_t)blockIdx.x * 6 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  /* HW_ID */         \
            a.phases[(size_t)blockIdx.x * 6 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20); /* XCC_ID */       \
        }                                                                                                         \
    }
#else
#define VQ_PHASE(K)
#endif

constexpr int BP = 32;    // tiles per workgroup
constexpr int KC = 8;     // channels per step
constexpr int HS_STAGE = 4 * 4 * BP * KC;   // floats: h[r][j][tile][k]

template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB == 1 ? 4 : 2, NB == 1 ? 4 : 2)))
void wino_f2x2_3x3_kernel(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* hs = reinterpret_cast<float*>(smem_raw);        // [2][4][4][32][8] (K loop); [4][2][32][32] (epilogue)

    VQ_PHASE(0)
    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int p0 = (tile / a.tiles_n) * BP;
    const int n0 = (tile % a.tiles_n) * (32 * NB);
    const int tpi = a.th * a.tw;                           // tiles per image

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);

    // ---- loader role: patch row r = wave of tile t, channels c4*4 .. +4 of the step's 8 ------------------------
    unsigned poff[4];                                      // byte offset of pixel c of the row; 0xFFFFFFFF = zero padding
    {
        const int t = lane >> 1, c4 = lane & 1;
        const int p = p0 + t;
        const bool ok = p < a.P;
        const int pp = ok ? p : 0;
        const int n_img = pp / tpi, rem = pp - n_img * tpi;
        const int ty = rem / a.tw, tx = rem - ty * a.tw;
        const int y = 2 * ty - 1 + wave, x0 = 2 * tx - 1;
        const bool row_ok = ok && (unsigned)y < (unsigned)a.H;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int x = x0 + c;
            poff[c] = (row_ok && (unsigned)x < (unsigned)a.W)
                          ? (unsigned)((((n_img * a.H + y) * a.W + x) * a.Cs_in + a.coff_in + c4 * 4) * 4)
                          : 0xFFFFFFFFu;
        }
    }
    const int hs_store = (wave * 4 * BP * KC) + lane * 4;  // + j * BP * KC (+ stage)

    // ---- consumer role: positions (i = wave, j = 0..3) ------------------------------------------------------------
    // V[i][j] = h[ra][j] + sgn * h[rb][j]   (B^T along rows: i=0: h0-h2, 1: h1+h2, 2: h2-h1, 3: h1-h3)
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float sgn = wave == 1 ? 1.0f : -1.0f;
    const floatx2 sgn2 = {sgn, sgn};
    const int fa_off = ra * 4 * BP * KC + l31 * KC + half * 4;
    const int fb_off = rb * 4 * BP * KC + l31 * KC + half * 4;
    unsigned uvoff[NB];                                    // lane part of the filter fragment address
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) uvoff[nb] = (n0 + 32 * nb < a.Cout) ? (unsigned)((l31 * 8 + half * 4) * 4) : 0xFFFFFFFFu;
    const unsigned u_step = (unsigned)(16 * a.Cout * 8 * 4);          // bytes between consecutive 8-channel groups
    const unsigned u_wave = (unsigned)(((wave * 4) * a.Cout + n0) * 8 * 4);
    const unsigned u_pos = (unsigned)(a.Cout * 8 * 4);                  // bytes between positions

    floatx16 acc[4][NB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][nb][e] = 0.f;

    constexpr int G = NB == 1 ? 2 : 1, NGRP = 4 / G;   // positions per MFMA group, groups per step
    floatx4 d[4];          // patch row in flight (next step)
    floatx4 bq[4][NB];     // filter fragments of the current step; reloaded for the next step right after use

#define VQ_W_LOAD_PATCH(KSTEP)                                                                                  \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                               \
        d[c] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, poff[c], (KSTEP) * (KC * 4), 0));
#define VQ_W_LOAD_U(KSTEP, J)                                                                                   \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                           \
        bq[J][nb] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(                          \
            u_rsrc, uvoff[nb], (KSTEP) * u_step + u_wave + (J) * u_pos + nb * (32 * 8 * 4), 0));
// column half of the input transform, h = d B: (d0 - d2, d1 + d2, d2 - d1, d1 - d3), then to LDS stage ST
#define VQ_W_STORE_H(ST)                                                                                        \
    {                                                                                                           \
        float* dst = hs + (ST) * HS_STAGE + hs_store;                                                           \
        *reinterpret_cast<floatx4*>(dst + 0 * BP * KC) = pk_sub(d[0], d[2]);                                    \
        *reinterpret_cast<floatx4*>(dst + 1 * BP * KC) = pk_add(d[1], d[2]);                                    \
        *reinterpret_cast<floatx4*>(dst + 2 * BP * KC) = pk_sub(d[2], d[1]);                                    \
        *reinterpret_cast<floatx4*>(dst + 3 * BP * KC) = pk_sub(d[1], d[3]);                                    \
    }
// One step on LDS stage ST: 4 positions x 4 k-pairs x NB MFMAs, in groups of G positions chosen so that two
// accumulators alternate (consecutive MFMAs never chain on one accumulator).  The A fragments of group g+1 are read
// under the MFMAs of group g; with NEXT, a group's filter registers are refilled for step KSTEP+1 as soon as its
// MFMAs are issued (the loads then have a whole step to land).  sched_barrier pins that order: left alone, the
// compiler sinks every global load to the end of the step and waits for it on the spot.
#define VQ_W_READ_A(ST, GRP, SET)                                                                               \
    _Pragma("unroll") for (int q = 0; q < G; ++q) {                                                             \
        a0[SET][q] = *reinterpret_cast<const floatx4*>(hs + (ST) * HS_STAGE + fa_off + (G * (GRP) + q) * BP * KC); \
        a1[SET][q] = *reinterpret_cast<const floatx4*>(hs + (ST) * HS_STAGE + fb_off + (G * (GRP) + q) * BP * KC); \
    }
#define VQ_W_STEP(ST, KSTEP, NEXT)                                                                              \
    {                                                                                                           \
        floatx4 a0[2][G], a1[2][G];                                                                             \
        VQ_W_READ_A(ST, 0, 0)                                                                                   \
        _Pragma("unroll") for (int g = 0; g < NGRP; ++g) {                                                      \
            if (g + 1 < NGRP) VQ_W_READ_A(ST, g + 1, (g + 1) & 1)                                               \
            floatx4 av[G];                                                                                      \
            _Pragma("unroll") for (int q = 0; q < G; ++q)                                                       \
                av[q] = pk_fma(a1[g & 1][q], sgn2, a0[g & 1][q]);                                                  \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                       \
                _Pragma("unroll") for (int q = 0; q < G; ++q)                                                   \
                    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                           \
                        acc[G * g + q][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][e], bq[G * g + q][nb][e], acc[G * g + q][nb], 0, 0, 0); \
            if (NEXT) {                                                                                         \
                _Pragma("unroll") for (int q = 0; q < G; ++q) VQ_W_LOAD_U((KSTEP) + 1, G * g + q)               \
            }                                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
        }                                                                                                       \
    }

    const int nk = a.Cin / KC;
    VQ_W_LOAD_PATCH(0)
#pragma unroll
    for (int j = 0; j < 4; ++j) VQ_W_LOAD_U(0, j)
    VQ_W_STORE_H(0)
    __syncthreads();
    VQ_PHASE(1)
    int kc = 0;
    for (; kc + 1 < nk; ++kc) {
        const int st = kc & 1;
        VQ_W_LOAD_PATCH(kc + 1)
        __builtin_amdgcn_sched_barrier(0);
        VQ_W_STEP(st, kc, true)
        VQ_W_STORE_H(st ^ 1)
        __syncthreads();                          // stage st^1 complete; everybody is done reading stage st
    }
    VQ_W_STEP(kc & 1, kc, false)
    VQ_PHASE(2)
#undef VQ_W_LOAD_PATCH
#undef VQ_W_LOAD_U
#undef VQ_W_STORE_H
#undef VQ_W_STEP
#undef VQ_W_READ_A

    // ---- epilogue: Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]] ---------------------------------------------------
    // This thread finishes output pixel (a_, b_) of tile p0 + (tid >> 3), channels (tid & 7)*4 .. +4 of each block.
    const int et = tid >> 3, ec = tid & 7;
    const int ep = p0 + et;
    const bool ep_ok = ep < a.P;
    const int epp = ep_ok ? ep : 0;
    const int e_img = epp / tpi, e_rem = epp - e_img * tpi;
    const int ety = e_rem / a.tw, etx = e_rem - ety * a.tw;
    __syncthreads();            // all waves are done with the K-loop image of the LDS
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int nbase = n0 + 32 * nb;
        // along j, in registers; in the MFMA C/D layout register e of a lane is tile row (e&3) + 8 (e>>2) + 4 half,
        // column (= output channel) l31
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            const float t0 = (acc[0][nb][e] + acc[1][nb][e]) + acc[2][nb][e];
            const float t1 = (acc[1][nb][e] - acc[2][nb][e]) - acc[3][nb][e];
            hs[((wave * 2 + 0) * 32 + row) * 32 + l31] = t0;
            hs[((wave * 2 + 1) * 32 + row) * 32 + l31] = t1;
        }
        __syncthreads();
        if (nbase < a.Cout) {
            const floatx4 bias = *reinterpret_cast<const floatx4*>(a.bias + nbase + ec * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a_ = u >> 1, b_ = u & 1;
                const float* src = hs + (b_ * 32 + et) * 32 + ec * 4;
                const floatx4 v1 = *reinterpret_cast<const floatx4*>(src + 1 * 2048);
                const floatx4 v2 = *reinterpret_cast<const floatx4*>(src + 2 * 2048);
                floatx4 y;
                if (a_ == 0) {
                    const floatx4 v0 = *reinterpret_cast<const floatx4*>(src + 0 * 2048);
                    y = (v0 + v1) + v2;
                } else {
                    const floatx4 v3 = *reinterpret_cast<const floatx4*>(src + 3 * 2048);
                    y = (v1 - v2) - v3;
                }
                y += bias;
                if (a.relu) {
                    y[0] = fmaxf(y[0], 0.f);
                    y[1] = fmaxf(y[1], 0.f);
                    y[2] = fmaxf(y[2], 0.f);
                    y[3] = fmaxf(y[3], 0.f);
                }
                const int oy = 2 * ety + a_, ox = 2 * etx + b_;
                if (ep_ok && oy < a.H && ox < a.W)
                    *reinterpret_cast<floatx4*>(a.out + ((size_t)(e_img * a.H + oy) * a.W + ox) * a.Cs_out + a.coff_out + nbase + ec * 4) = y;
            }
        }
        if (nb + 1 < NB) __syncthreads();
    }
    VQ_PHASE(3)
}

template <int NB>
int launch_t(const WinoArgs& a0, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    WinoArgs a = a0;
    a.tiles_m = cdiv(a.P, BP);
    a.tiles_n = cdiv(a.Cout, 32 * NB);
    auto kern = wino_f2x2_3x3_kernel<NB>;
    const size_t lds = 2 * HS_STAGE * sizeof(float);
    VQ_LAUNCH(kern, a.tiles_m * a.tiles_n, 256, lds, stream, ev_start, ev_stop, a);
    VQ_CHECK_LAUNCH();
    return VQ_OK;
}

}  // namespace

namespace vq {

int launch_wino(const WinoArgs& a, int variant, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    VQ_REQUIRE(a.Cin % KC == 0 && a.Cout % 32 == 0 && a.Cs_out % 4 == 0 && a.coff_out % 4 == 0 && a.Cs_in % 4 == 0 && a.coff_in % 4 == 0,
               "Winograd convolution needs Cin %% 8 == 0, Cout %% 32 == 0 and 16-byte aligned channel offsets");
    if (variant == 0) return launch_t<1>(a, stream, ev_start, ev_stop);
    if (variant == 1) return launch_t<2>(a, stream, ev_start, ev_stop);
    return fail(VQ_E_INVALID, "no Winograd kernel variant %d", variant);
}

}  // namespace vq
