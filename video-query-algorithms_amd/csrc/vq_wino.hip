// 3x3 / stride-1 / pad-1 convolution + folded BN + ReLU as Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
// The 3x3 stride-1 layers are 64.6 % of the BN-Inception MACs (from
// src/features_GPU_compute/models/ucf101/tsn_bn_inception_rgb_deploy.prototxt).  For a 2x2 output tile
//     Y = A^T [ (G g G^T) . (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter, "." elementwise
// turns 36 multiplies per (tile, cin, cout) into 16: the layer becomes 16 independent GEMMs
//     M_xi[cout][tile] = sum_cin U_xi[cout][cin] * V_xi[cin][tile],   xi = 4 i + j  (position in the 4x4 patch)
// that run on v_mfma_f32_32x32x2_f32, 2.25x fewer matrix-core cycles than the direct form.  Everything is fused in
// one kernel: the input transform happens on the way into LDS / out of LDS, the output transform in the epilogue.
//
// Work split.  A workgroup (4 waves) owns 32 tiles x BN output channels x all 16 positions; wave i owns row i of
// the 4x4 position grid (4 GEMMs, accumulators 4 x (BN/32) x 16 registers).  Cin is walked 8 channels per step.
//   activations: wave r loads patch row r of every tile (4 pixels x 8 channels), applies the column half of the
//                transform (h = d B, 4 adds per channel) and stores h[r][j] to LDS; the row half is applied when
//                wave i reads its fragment: V[i][j] = h[ra][j] +- h[rb][j]  (two ds_read_b128 and 4 FMAs for 4 k's).
//   filters:     U is pre-transformed on the host (fp64, rounded once) and laid out [Cin/8][16][Cout][8], which
//                is exactly the MFMA fragment order: each wave loads its own fragments straight from L2 into
//                registers, a full step ahead, with no LDS traffic and no sharing between waves.
//   MFMA roles:  the FILTER fragment is the A operand (rows = output channels), the activation fragment the B
//                operand (columns = tiles).  A lane then holds 4 CONSECUTIVE output channels of one tile in 4
//                consecutive accumulator registers, so everything behind the K loop moves 16 bytes per instruction.
//   epilogue:    wave i reduces its 4 positions along j in registers (A^T along columns), the 4 waves exchange
//                through LDS (ds_write_b128) for the reduction along i, then ReLU and 16-byte NHWC buffer stores.
//                The bias costs nothing: A^T e_11 A = all ones, so starting the accumulator of position (1,1) at the
//                bias adds it to all four outputs of the tile.
//
// The fp32 matrix instructions execute on the SIMD's fp32 lanes -- every VALU instruction of any co-resident wave takes
// matrix-pipe time (tools/ubench/mfma_coissue.hip).  The code around the K loop is therefore written for few VALU
// instructions: the tile -> (image, row, column) decoding of a workgroup is done once on the scalar unit with
// multiply-high reciprocals (the per-lane part is a walk of at most 31 tiles from there), transforms use packed fp32
// instructions, addresses are 32-bit buffer offsets whose out-of-range value doubles as the padding / tail mask.
//
// A launch carries up to kWinoMaxJobs independent convolutions (WinoGroup: the 3x3 and the first double-3x3 arm of an
// inception module): their workgroups fill each other's tail rounds and one kernel boundary disappears.
// Layers on maps of at most 14 x 14 carry a second filter layout and can run in units of SIXTEEN tiles on v_mfma_f32_16x16x4_f32
// (wino16_unit below; the tiling table picks per launch): where a launch has too few 32-tile units for 256 compute units.
// The k order of every output element is fixed (channels ascending in groups of 8), so all variants and groupings
// produce identical bits.  Rounding differs from the direct kernel (Winograd F(2,3) error is ~1-2.5x the direct fp32
// error, measured against the fp64 oracle: tests/test_tsn_gpu.py).
#include "vq_common.h"
#include "vq_tsn_kernels.h"

using namespace vq;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((__vector_size__(4 * sizeof(unsigned))));

namespace {

// Packed fp32 arithmetic (v_pk_add_f32 / v_pk_fma_f32: two floats per lane per instruction; inline asm, because the
// compiler scalarises float4 subtraction).  Only used where the result goes to LDS or memory (see pk_fma below).
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
// x * s + z on the fragment side.  NOT inline asm: the result feeds an MFMA, and the compiler's hazard recogniser
// only inserts the VALU -> MFMA wait states for instructions it can see (an asm VALU in front of an MFMA read stale
// registers); the elementwise builtin lets it emit v_pk_fma_f32 where it wants and keep the hazards right.
__device__ __forceinline__ floatx4 pk_fma(floatx4 x, floatx2 s, floatx4 z) {
    const floatx2 lo = __builtin_elementwise_fma(x.xy, s, z.xy), hi = __builtin_elementwise_fma(x.zw, s, z.zw);
    return (floatx4){lo.x, lo.y, hi.x, hi.y};
}

// max(x, 0) as ONE instruction: fmaxf() on a value that comes out of inline asm costs a canonicalising v_max(x, x)
// first (the compiler cannot know it is not a signalling NaN); the result only goes to memory.
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
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

// (s_setprio 1 / 0 around every MFMA group -- cdna_hip_programming.md T5 -- was measured in round 4: 9 300 against 11 650 clips/s at cfg 2.
// With four waves per SIMD the raised waves starve the ones that stage the next step; the order pinned by sched_barrier stays as it is.
// Also measured there: the first K step peeled off so that the first MFMA of 3 of the 4 positions takes a literal 0 as C (`v_mfma ..., 0`)
// and 48 x NB accumulator clears go: 11 437-11 534 against 11 487-11 582 clips/s in alternating builds on one box -- the clears sat in the
// shadow of the prologue's first loads, the peeled step is a third more code.)
constexpr int BP = 32;    // tiles per workgroup
constexpr int KC = 8;     // channels per step
constexpr int HS_STAGE = 4 * 4 * BP * KC;     // floats: h[r][j][tile][k]
constexpr int EP_ROW = 36;                    // epilogue image: 32 channels of a tile + 4 floats of padding
constexpr int EP_FLOATS = 2 * 4 * BP * EP_ROW;   // [i][x][tile][EP_ROW]
constexpr int LDS_FLOATS = EP_FLOATS > 2 * HS_STAGE ? EP_FLOATS : 2 * HS_STAGE;

// Image, tile row and tile column of the t-th tile behind the workgroup's first one (t < 32): a walk, not a division.
struct TileAt {
    int img, ty, tx;
};
__device__ __forceinline__ TileAt tile_at(int t, int img0, int ty0, int tx0, int th, int tw, unsigned s_th, unsigned s_tw) {
    // every factor is below 2^24 (launch_t checks the shapes): v_mul_u32_u24 runs at the full rate, v_mul_lo_u32 at a quarter of it
    const unsigned lin = (unsigned)(tx0 + t);
    const unsigned q1 = umul24(lin, s_tw) >> kRecipShift;
    const unsigned ly = (unsigned)ty0 + q1;
    const unsigned q2 = umul24(ly, s_th) >> kRecipShift;
    return TileAt{img0 + (int)q2, (int)(ly - umul24(q2, (unsigned)th)), (int)(lin - umul24(q1, (unsigned)tw))};
}

template <int NB>
__device__ __forceinline__ void wino_unit(const WinoJob& a, int unit, float* hs) {
    VQ_PHASE(0)
    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- which tiles, which channels: all on the scalar unit ---------------------------------------------------
    const int tile = xcd_remap(unit, a.n_units);
    const int pb = (int)magic_div((unsigned)tile, a.m_tiles_n);
    const int p0 = pb * BP;
    const int n0 = (tile - pb * a.tiles_n) * (32 * NB);
    const int tpi = a.th * a.tw;
    const int img0 = (int)magic_div((unsigned)p0, a.m_tpi);
    const int rem0 = p0 - img0 * tpi;
    const int ty0 = (int)magic_div((unsigned)rem0, a.m_tw);
    const int tx0 = rem0 - ty0 * a.tw;

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);

    // ---- loader role: patch row r = wave of tile t, channels c4*4 .. +4 of the step's 8 ------------------------
    unsigned poff[4];                                      // byte offset of pixel c of the row; 0xFFFFFFFF = zero padding
    {
        const int t = lane >> 1, c4 = lane & 1;
        const TileAt ta = tile_at(t, img0, ty0, tx0, a.th, a.tw, a.s_th, a.s_tw);
        const int y = 2 * ta.ty - 1 + wave, x0 = 2 * ta.tx - 1;
        const bool row_ok = p0 + t < a.P && (unsigned)y < (unsigned)a.H;
        const int base = (imul24(imul24(ta.img * a.H + y, a.W) + x0, a.Cs_in) + a.coff_in + c4 * 4) * 4;
        const int px = a.Cs_in * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            poff[c] = (row_ok && (unsigned)(x0 + c) < (unsigned)a.W) ? (unsigned)(base + c * px) : 0xFFFFFFFFu;
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

    // Accumulators: register e of a lane is output channel (e & 3) + 8 (e >> 2) + 4 half of tile l31.  Position (1,1)
    // starts at the bias (see the header), everything else at zero.
    floatx16 acc[4][NB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][nb][e] = 0.f;
    if (wave == 1) {
        const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, (unsigned)a.Cout * 4u, 0x00020000);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 bv = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                   b_rsrc, (unsigned)((n0 + 32 * nb + 8 * g + 4 * half) * 4), 0, 0));
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[1][nb][4 * g + r] = bv[r];
            }
    }

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
// accumulators alternate (consecutive MFMAs never chain on one accumulator).  The fragments of group g+1 are read
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
                        acc[G * g + q][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[G * g + q][nb][e], av[q][e], acc[G * g + q][nb], 0, 0, 0); \
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
    // Exchange image: ep[i][x][tile][channel], x = the two sums along j of position row i.  This thread then finishes
    // the four output pixels of tile et, channels ec*4 .. +4 of each 32-channel block.
    const int et = tid >> 3, ec = tid & 7;
    const TileAt te = tile_at(et, img0, ty0, tx0, a.th, a.tw, a.s_th, a.s_tw);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
    const int oy = 2 * te.ty, ox = 2 * te.tx;
    const int obase = (imul24(imul24(te.img * a.H + oy, a.W) + ox, a.Cs_out) + a.coff_out + n0 + ec * 4) * 4;
    const bool tile_ok = p0 + et < a.P;
    unsigned ooff[4];                                      // (a_, b_) = (u >> 1, u & 1); 0xFFFFFFFF: the store is dropped
#pragma unroll
    for (int u = 0; u < 4; ++u)
        ooff[u] = (tile_ok && oy + (u >> 1) < a.H && ox + (u & 1) < a.W) ? (unsigned)(obase + ((u >> 1) * a.W + (u & 1)) * a.Cs_out * 4)
                                                                           : 0xFFFFFFFFu;
    float* ep_w = hs + ((wave * 2) * BP + l31) * EP_ROW + 4 * half;      // + x * BP * EP_ROW + 8 g
    const float* ep_r = hs + et * EP_ROW + ec * 4;                        // + (i * 2 + b_) * BP * EP_ROW
    __syncthreads();            // all waves are done with the K-loop image of the LDS
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        // along j, in registers, two channels per instruction: t0 = (m0 + m1) + m2, t1 = (m1 - m2) - m3
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            floatx4 t0, t1;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int e = 4 * g + 2 * h2;
                const floatx2 m0 = {acc[0][nb][e], acc[0][nb][e + 1]}, m1 = {acc[1][nb][e], acc[1][nb][e + 1]};
                const floatx2 m2 = {acc[2][nb][e], acc[2][nb][e + 1]}, m3 = {acc[3][nb][e], acc[3][nb][e + 1]};
                const floatx2 s0 = pk_add2(pk_add2(m0, m1), m2), s1 = pk_sub2(pk_sub2(m1, m2), m3);
                t0[2 * h2] = s0.x;
                t0[2 * h2 + 1] = s0.y;
                t1[2 * h2] = s1.x;
                t1[2 * h2 + 1] = s1.y;
            }
            *reinterpret_cast<floatx4*>(ep_w + 8 * g) = t0;
            *reinterpret_cast<floatx4*>(ep_w + BP * EP_ROW + 8 * g) = t1;
        }
        __syncthreads();
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_) {
            const floatx4 v0 = *reinterpret_cast<const floatx4*>(ep_r + (0 * 2 + b_) * BP * EP_ROW);
            const floatx4 v1 = *reinterpret_cast<const floatx4*>(ep_r + (1 * 2 + b_) * BP * EP_ROW);
            const floatx4 v2 = *reinterpret_cast<const floatx4*>(ep_r + (2 * 2 + b_) * BP * EP_ROW);
            const floatx4 v3 = *reinterpret_cast<const floatx4*>(ep_r + (3 * 2 + b_) * BP * EP_ROW);
            floatx4 y0 = pk_add(pk_add(v0, v1), v2);          // a_ = 0
            floatx4 y1 = pk_sub(pk_sub(v1, v2), v3);          // a_ = 1
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y0[r] = relu1(y0[r]);
                    y1[r] = relu1(y1[r]);
                }
            }
            const bool blk_ok = n0 + 32 * nb < a.Cout;                  // a 32-channel block past Cout: stores dropped
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, y0), out_rsrc, blk_ok ? ooff[b_] : 0xFFFFFFFFu, nb * (32 * 4), 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, y1), out_rsrc, blk_ok ? ooff[2 + b_] : 0xFFFFFFFFu, nb * (32 * 4), 0);
        }
        if (nb + 1 < NB) __syncthreads();
    }
    VQ_PHASE(3)
}

// ------------------------------------------------------------------------------------------------------------------
// The SIXTEEN-tile unit (round 6; layers on maps of at most 14 x 14, VQ_OP_CONV_WINOGRAD16): 16 tiles x 32 NB output channels x
// all 16 positions on v_mfma_f32_16x16x4_f32, 16 channels per step.
// Why: at 96 crops a 14 x 14 layer has 147 blocks of 32 tiles and a 7 x 7 layer 48, i.e. 336-1 176 units per launch for the 256
// compute units: the last round of units leaves most of the chip idle (a 14 x 14 layer reaches 164 algorithmic TFLOP/s at 96 crops and
// 237 at 448; a 7 x 7 layer 115 and 174).  Half as many tiles per unit = twice the units of half the length: the rounds quantise half
// as coarsely.  Filter traffic per multiply doubles (a fragment serves 16 tiles instead of 32); the activations' halves.
// Same bits as the 32-tile kernel: an fp32 MFMA is a k-ordered chain of fused multiply-adds (one rounding per product), and the
// channel that k index kq of step-MFMA g multiplies is chosen so that every output element meets its channels in the 32-tile
// kernel's order -- within 16 channels 0,4,1,5,2,6,3,7,8,12,9,13,10,14,11,15: slot p = 4 kq + g of a lane's 16-byte fragment holds
// channel kPerm16[p] = {0,2,8,10, 4,6,12,14, 1,3,9,11, 5,7,13,15}.  The host lays the filters out [Cin/16][16 positions][Cout][16
// slots] in that order (tsn/net.py: winograd_filters16); the loader scatters a pixel's four consecutive channels 4 q .. 4 q + 3 as two
// 8-byte pairs (channels 4q, 4q+2 -> slots s, s+1; 4q+1, 4q+3 -> s+8, s+9; s = {0,4,2,6}[q]).
//   lanes:    tile l & 15, k quarter l >> 4 (A = filters: row = output channel l & 15 of a 16-channel block; B = activations);
//             a lane's 4 accumulator registers of a (position, block) are output channels 4 (l >> 4) + r of tile l & 15.
//   LDS:      h[r][j][tile][16 slots]: a wave instruction reads or writes 1 KB of contiguous memory (no bank conflicts).
//   epilogue: as the 32-tile kernel's (j in registers, i through LDS, 16-byte stores), on 16 tiles.
constexpr int BP16 = 16;                                // tiles per unit
constexpr int KC16 = 16;                                // channels per step
constexpr int HS16_STAGE = 4 * 4 * BP16 * KC16;         // floats
constexpr int EP16_FLOATS = 2 * 4 * BP16 * EP_ROW;      // [i][x][tile][EP_ROW]
constexpr int LDS16_FLOATS = EP16_FLOATS > 2 * HS16_STAGE ? EP16_FLOATS : 2 * HS16_STAGE;

template <int NB>
__device__ __forceinline__ void wino16_unit(const WinoJob& a, int unit, float* hs) {
    constexpr int NBLK = 2 * NB;                        // 16-channel blocks per unit
    const int tid = threadIdx.x;
    const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(unit, a.n_units);
    const int pb = (int)magic_div((unsigned)tile, a.m_tiles_n);
    const int p0 = pb * BP16;
    const int n0 = (tile - pb * a.tiles_n) * (32 * NB);
    const int tpi = a.th * a.tw;
    const int img0 = (int)magic_div((unsigned)p0, a.m_tpi);
    const int rem0 = p0 - img0 * tpi;
    const int ty0 = (int)magic_div((unsigned)rem0, a.m_tw);
    const int tx0 = rem0 - ty0 * a.tw;

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);

    // ---- loader: patch row r = wave of tile t = lane >> 2, channels 4 q .. 4 q + 3 (q = lane & 3) of the step's 16 ---------------
    unsigned poff[4];
    const int lq = lane & 3;
    {
        const int t = lane >> 2;
        const TileAt ta = tile_at(t, img0, ty0, tx0, a.th, a.tw, a.s_th, a.s_tw);
        const int y = 2 * ta.ty - 1 + wave, x0 = 2 * ta.tx - 1;
        const bool row_ok = p0 + t < a.P && (unsigned)y < (unsigned)a.H;
        const int base = (imul24(imul24(ta.img * a.H + y, a.W) + x0, a.Cs_in) + a.coff_in + lq * 4) * 4;
        const int px = a.Cs_in * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            poff[c] = (row_ok && (unsigned)(x0 + c) < (unsigned)a.W) ? (unsigned)(base + c * px) : 0xFFFFFFFFu;
    }
    // slots of the pair (channels 4q, 4q+2): {0, 4, 2, 6}[q]; the pair (4q+1, 4q+3) sits 8 slots further
    const int slot_a = ((lq & 1) << 2) | (lq & 2);
    const int hs_store = (wave * 4 * BP16 + (lane >> 2)) * KC16 + slot_a;       // + j * BP16 * KC16 (+ stage)

    // ---- consumer: positions (i = wave, j = 0..3) ------------------------------------------------------------------------------
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float sgn = wave == 1 ? 1.0f : -1.0f;
    const floatx2 sgn2 = {sgn, sgn};
    const int fa_off = (ra * 4 * BP16 + l15) * KC16 + kq * 4;                   // + j * BP16 * KC16
    const int fb_off = (rb * 4 * BP16 + l15) * KC16 + kq * 4;
    unsigned uvoff[NBLK];                                  // lane part of the filter fragment address: output channel l15 of block b
#pragma unroll
    for (int b = 0; b < NBLK; ++b) uvoff[b] = (n0 + 16 * b < a.Cout) ? (unsigned)(((16 * b + l15) * 16 + kq * 4) * 4) : 0xFFFFFFFFu;
    const unsigned u_step = (unsigned)(16 * a.Cout * 16 * 4);         // bytes between consecutive 16-channel groups
    const unsigned u_wave = (unsigned)(((wave * 4) * a.Cout + n0) * 16 * 4);
    const unsigned u_pos = (unsigned)(a.Cout * 16 * 4);               // bytes between positions

    floatx4 acc[4][NBLK];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[j][b] = (floatx4){0.f, 0.f, 0.f, 0.f};
    if (wave == 1) {                                                  // position (1,1) starts at the bias
        const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, (unsigned)a.Cout * 4u, 0x00020000);
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
            acc[1][b] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (unsigned)((n0 + 16 * b + 4 * kq) * 4), 0, 0));
    }

    floatx4 d[4];              // patch row in flight (next step)
    floatx4 bq[4][NBLK];       // filter fragments of the current step (slot g of a fragment = the A operand of step-MFMA g)

#define VQ_W16_LOAD_PATCH(KSTEP)                                                                                \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                               \
        d[c] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, poff[c], (KSTEP) * (KC16 * 4), 0));
#define VQ_W16_LOAD_U(KSTEP, J)                                                                                 \
    _Pragma("unroll") for (int b = 0; b < NBLK; ++b)                                                            \
        bq[J][b] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(                           \
            u_rsrc, uvoff[b], (KSTEP) * u_step + u_wave + (J) * u_pos, 0));
// column half of the input transform, then the two channel pairs of the quad to their slots
#define VQ_W16_PUT(J, V)                                                                                        \
    {                                                                                                           \
        const floatx4 v_ = (V);                                                                                 \
        *reinterpret_cast<floatx2*>(dst + (J) * BP16 * KC16) = (floatx2){v_.x, v_.z};                           \
        *reinterpret_cast<floatx2*>(dst + (J) * BP16 * KC16 + 8) = (floatx2){v_.y, v_.w};                       \
    }
#define VQ_W16_STORE_H(ST)                                                                                      \
    {                                                                                                           \
        float* dst = hs + (ST) * HS16_STAGE + hs_store;                                                         \
        VQ_W16_PUT(0, pk_sub(d[0], d[2]))                                                                       \
        VQ_W16_PUT(1, pk_add(d[1], d[2]))                                                                       \
        VQ_W16_PUT(2, pk_sub(d[2], d[1]))                                                                       \
        VQ_W16_PUT(3, pk_sub(d[1], d[3]))                                                                       \
    }
#define VQ_W16_READ_A(ST, GRP, SET)                                                                             \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                             \
        a0[SET][q] = *reinterpret_cast<const floatx4*>(hs + (ST) * HS16_STAGE + fa_off + (2 * (GRP) + q) * BP16 * KC16); \
        a1[SET][q] = *reinterpret_cast<const floatx4*>(hs + (ST) * HS16_STAGE + fb_off + (2 * (GRP) + q) * BP16 * KC16); \
    }
// One step: 4 positions x 4 step-MFMAs x NBLK blocks, in two groups of two positions; consecutive MFMAs never chain on one accumulator
// (v_mfma_f32_16x16x4: 32-cycle issue, 40-cycle dependent latency).
#define VQ_W16_STEP(ST, KSTEP, NEXT)                                                                            \
    {                                                                                                           \
        floatx4 a0[2][2], a1[2][2];                                                                             \
        VQ_W16_READ_A(ST, 0, 0)                                                                                 \
        _Pragma("unroll") for (int g2 = 0; g2 < 2; ++g2) {                                                      \
            if (g2 + 1 < 2) VQ_W16_READ_A(ST, g2 + 1, (g2 + 1) & 1)                                             \
            floatx4 av[2];                                                                                      \
            _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                       \
                av[q] = pk_fma(a1[g2 & 1][q], sgn2, a0[g2 & 1][q]);                                             \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                       \
                _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                   \
                    _Pragma("unroll") for (int b = 0; b < NBLK; ++b)                                            \
                        acc[2 * g2 + q][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[2 * g2 + q][b][e], av[q][e], acc[2 * g2 + q][b], 0, 0, 0); \
            if (NEXT) {                                                                                         \
                _Pragma("unroll") for (int q = 0; q < 2; ++q) VQ_W16_LOAD_U((KSTEP) + 1, 2 * g2 + q)            \
            }                                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
        }                                                                                                       \
    }

    const int nk = a.Cin / KC16;
    VQ_W16_LOAD_PATCH(0)
#pragma unroll
    for (int j = 0; j < 4; ++j) VQ_W16_LOAD_U(0, j)
    VQ_W16_STORE_H(0)
    __syncthreads();
    int kc = 0;
    for (; kc + 1 < nk; ++kc) {
        const int st = kc & 1;
        VQ_W16_LOAD_PATCH(kc + 1)
        __builtin_amdgcn_sched_barrier(0);
        VQ_W16_STEP(st, kc, true)
        VQ_W16_STORE_H(st ^ 1)
        __syncthreads();
    }
    VQ_W16_STEP(kc & 1, kc, false)
#undef VQ_W16_LOAD_PATCH
#undef VQ_W16_LOAD_U
#undef VQ_W16_PUT
#undef VQ_W16_STORE_H
#undef VQ_W16_STEP
#undef VQ_W16_READ_A

    // ---- epilogue: Y = A^T M A ------------------------------------------------------------------------------------------------------
    // last stage: thread (tile et, channel quad ec, output column b_ = bsel) finishes two output pixels of every 32-channel block
    const int et = (tid >> 3) & 15, ec = tid & 7, bsel = tid >> 7;
    const TileAt te = tile_at(et, img0, ty0, tx0, a.th, a.tw, a.s_th, a.s_tw);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
    const int oy = 2 * te.ty, ox = 2 * te.tx;
    const int obase = (imul24(imul24(te.img * a.H + oy, a.W) + ox, a.Cs_out) + a.coff_out + n0 + ec * 4) * 4;
    const bool tile_ok = p0 + et < a.P;
    unsigned ooff[2];                                      // a_ = 0, 1 at column b_ = bsel; 0xFFFFFFFF: the store is dropped
#pragma unroll
    for (int u = 0; u < 2; ++u)
        ooff[u] = (tile_ok && oy + u < a.H && ox + bsel < a.W) ? (unsigned)(obase + (u * a.W + bsel) * a.Cs_out * 4) : 0xFFFFFFFFu;
    float* ep_w = hs + ((wave * 2) * BP16 + l15) * EP_ROW + 4 * kq;      // + x * BP16 * EP_ROW + 16 (block & 1)
    const float* ep_r = hs + et * EP_ROW + ec * 4;                        // + (i * 2 + b_) * BP16 * EP_ROW
    __syncthreads();            // all waves are done with the K-loop image of the LDS
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const int b = 2 * nb + bb;
            floatx4 t0, t1;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const floatx2 m0 = {acc[0][b][2 * h2], acc[0][b][2 * h2 + 1]}, m1 = {acc[1][b][2 * h2], acc[1][b][2 * h2 + 1]};
                const floatx2 m2 = {acc[2][b][2 * h2], acc[2][b][2 * h2 + 1]}, m3 = {acc[3][b][2 * h2], acc[3][b][2 * h2 + 1]};
                const floatx2 s0 = pk_add2(pk_add2(m0, m1), m2), s1 = pk_sub2(pk_sub2(m1, m2), m3);
                t0[2 * h2] = s0.x;
                t0[2 * h2 + 1] = s0.y;
                t1[2 * h2] = s1.x;
                t1[2 * h2 + 1] = s1.y;
            }
            *reinterpret_cast<floatx4*>(ep_w + 16 * bb) = t0;
            *reinterpret_cast<floatx4*>(ep_w + BP16 * EP_ROW + 16 * bb) = t1;
        }
        __syncthreads();
        {
            const floatx4 v0 = *reinterpret_cast<const floatx4*>(ep_r + (0 * 2 + bsel) * BP16 * EP_ROW);
            const floatx4 v1 = *reinterpret_cast<const floatx4*>(ep_r + (1 * 2 + bsel) * BP16 * EP_ROW);
            const floatx4 v2 = *reinterpret_cast<const floatx4*>(ep_r + (2 * 2 + bsel) * BP16 * EP_ROW);
            const floatx4 v3 = *reinterpret_cast<const floatx4*>(ep_r + (3 * 2 + bsel) * BP16 * EP_ROW);
            floatx4 y0 = pk_add(pk_add(v0, v1), v2);          // a_ = 0
            floatx4 y1 = pk_sub(pk_sub(v1, v2), v3);          // a_ = 1
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y0[r] = relu1(y0[r]);
                    y1[r] = relu1(y1[r]);
                }
            }
            const bool blk_ok = n0 + 32 * nb < a.Cout;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, y0), out_rsrc, blk_ok ? ooff[0] : 0xFFFFFFFFu, nb * (32 * 4), 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, y1), out_rsrc, blk_ok ? ooff[1] : 0xFFFFFFFFu, nb * (32 * 4), 0);
        }
        if (nb + 1 < NB) __syncthreads();
    }
}

template <int NB, int TB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB == 1 ? 4 : 2, NB == 1 ? 4 : 2)))
void wino_f2x2_3x3_kernel(WinoGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* hs = reinterpret_cast<float*>(smem_raw);        // [2][4][4][32][8] (K loop); [4][2][32][36] (epilogue)
    const int b = blockIdx.x;
    if (b >= g.pool_unit0) {                             // the level's pooling layers: plain streaming work, no LDS
        int q = 0;
        if (g.n_pools > 1 && b >= g.pool[1].unit0) q = 1;
        const PoolArgs& pa = g.pool[q];
        const int unit = b - pa.unit0;
        if (unit >= pa.n_units) return;
#pragma unroll
        for (int r = 0; r < kPoolPerWG / 256; ++r) {
            const int64_t i = (int64_t)unit * kPoolPerWG + r * 256 + threadIdx.x;
            if (i < pa.total) {
                if (pa.is_max)
                    pool_one<true>(pa, i);
                else
                    pool_one<false>(pa, i);
            }
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
    if constexpr (TB == 16)
        wino16_unit<NB>(a, unit, hs);
    else
        wino_unit<NB>(a, unit, hs);
}

unsigned magic_u32(unsigned d) { return d == 1 ? 0u : (unsigned)(0x100000000ull / d) + 1u; }   // 0 = "divide by one"

template <int NB, int TB>
int launch_t(WinoGroup& g, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    int units = 0;
    for (int q = 0; q < g.n_jobs; ++q) {
        WinoJob& a = g.job[q];
        a.tiles_n = cdiv(a.Cout, 32 * NB);
        const long long n = (long long)cdiv(a.P, TB) * a.tiles_n;
        VQ_REQUIRE(n > 0 && n < (1 << 24), "Winograd job %d: %lld workgroups", q, n);
        a.unit0 = units;
        a.n_units = (int)n;
        units += ((int)n + 7) & ~7;
        const unsigned tpi = (unsigned)(a.th * a.tw);
        // exactness of the multiply-high divisions (vq_tsn_kernels.h): dividend < 2^32 / divisor
        VQ_REQUIRE((unsigned long long)n * a.tiles_n < 0x100000000ull && (unsigned long long)(a.P + BP) * tpi < 0x100000000ull &&
                       recip22_ok((unsigned)a.tw, (unsigned)a.tw + BP) && recip22_ok((unsigned)a.th, (unsigned)a.th + BP + 2),
                   "Winograd job %d: shape outside the reciprocal-division range", q);
        // 24-bit multiplies in the address arithmetic (tile_at, the pixel index times the slot's channel count)
        VQ_REQUIRE((unsigned long long)cdiv(a.P, (int)tpi) * a.H * a.W < (1ull << 23) && a.Cs_in < (1 << 23) && a.Cs_out < (1 << 23),
                   "Winograd job %d: more than 2^23 pixels in a slot (split the batch)", q);
        a.m_tiles_n = magic_u32((unsigned)a.tiles_n);
        a.m_tpi = magic_u32(tpi);
        a.m_tw = magic_u32((unsigned)a.tw);
        a.s_tw = recip22((unsigned)a.tw);
        a.s_th = recip22((unsigned)a.th);
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
    auto kern = wino_f2x2_3x3_kernel<NB, TB>;
    const size_t lds = (TB == 16 ? LDS16_FLOATS : LDS_FLOATS) * sizeof(float);
    VQ_LAUNCH(kern, units, 256, lds, stream, ev_start, ev_stop, g);
    VQ_CHECK_LAUNCH();
    return VQ_OK;
}

}  // namespace

namespace vq {

int launch_wino_group(WinoGroup& g, int variant, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    VQ_REQUIRE(g.n_jobs >= 1 && g.n_jobs <= kWinoMaxJobs, "a Winograd launch carries 1..%d jobs", kWinoMaxJobs);
    VQ_REQUIRE(g.n_pools >= 0 && g.n_pools <= kWinoMaxPools, "a Winograd launch carries at most %d pooling layers", kWinoMaxPools);
    for (int q = 0; q < g.n_jobs; ++q) {
        const WinoJob& a = g.job[q];
        VQ_REQUIRE(a.Cin % KC == 0 && a.Cout % 32 == 0 && a.Cs_out % 4 == 0 && a.coff_out % 4 == 0 && a.Cs_in % 4 == 0 && a.coff_in % 4 == 0,
                   "Winograd convolution needs Cin %% 8 == 0, Cout %% 32 == 0 and 16-byte aligned channel offsets");
        VQ_REQUIRE(a.in_bytes <= 0x7FFFFFF0u && a.out_bytes <= 0x7FFFFFF0u,
                   "Winograd convolution: a launch addresses its slots with signed 32-bit byte offsets (split the batch)");
    }
    VQ_REQUIRE((variant >= 2) == (g.job[0].t16 != 0), "Winograd variant %d on the other filter layout", variant);
    if (variant == 0) return launch_t<1, 32>(g, stream, ev_start, ev_stop);
    if (variant == 1) return launch_t<2, 32>(g, stream, ev_start, ev_stop);
    if (variant == 2) return launch_t<1, 16>(g, stream, ev_start, ev_stop);
    if (variant == 3) return launch_t<2, 16>(g, stream, ev_start, ev_stop);
    return fail(VQ_E_INVALID, "no Winograd kernel variant %d", variant);
}

}  // namespace vq
