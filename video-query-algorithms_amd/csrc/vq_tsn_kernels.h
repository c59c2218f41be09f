// Kernel entry points shared between the TSN executor (vq_tsn.hip) and separately compiled kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace vq {

// XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2); give each XCD a contiguous run of
// tiles, with the N tiles of one M tile adjacent, so the gathered activation tile is re-read from L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// x / d as one multiply-high: m = floor(2^32 / d) + 1 is exact while x * (m d - 2^32) < 2^32, i.e. certainly for
// x < 2^32 / d (the host checks that bound where it builds the constants: magic_u32 in vq_wino.hip).  d = 1 has no such
// constant; it is encoded as m = 0.  On uniform values the compiler keeps all of it on the scalar unit (s_mul_hi_u32),
// which costs the matrix pipe nothing.
__device__ __forceinline__ unsigned magic_div(unsigned x, unsigned m) { return m ? __umulhi(x, m) : x; }

// x / d for EVERY 32-bit x (Granlund & Montgomery): l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1,
// t = mulhi(m, x), q = (t + ((x - t) >> min(l, 1))) >> max(l - 1, 0).  Five scalar instructions on a uniform x.
struct FullDiv {
    unsigned m, sh1, sh2;
};
__device__ __forceinline__ unsigned full_div(unsigned x, const FullDiv& d) {
    const unsigned t = __umulhi(d.m, x);
    return (t + ((x - t) >> d.sh1)) >> d.sh2;
}
static inline FullDiv full_div_for(unsigned d) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    FullDiv f;
    f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 1 ? l - 1 : 0;
    return f;
}
// x / d = (x * s) >> 22 with s = 2^22 / d + 1 on 24-bit multiplies (v_mul_u32_u24: full rate; v_mul_lo / v_mul_hi_u32 run at a quarter
// of it): exact while x (s d - 2^22) < 2^22, i.e. certainly for x d < 2^22, and x s < 2^32 (recip22_ok).
constexpr int kRecipShift = 22;
// 24-bit multiplies as inline asm: through __umul24 / __mul24 the compiler reasons about the operand masks, finds them dead where only the
// low bits of a product are used further on, and is back at a plain 32-bit multiply (v_mul_lo_u32, v_mad_u64_u32).
// The second factor is UNIFORM (a kernel argument or scalar arithmetic on some): it stays in a scalar register.
__device__ __forceinline__ unsigned umul24(unsigned x, unsigned y_uniform) {
    unsigned r;
    asm("v_mul_u32_u24 %0, %2, %1" : "=v"(r) : "v"(x), "s"(y_uniform));
    return r;
}
__device__ __forceinline__ int imul24(int x, int y_uniform) {
    int r;
    asm("v_mul_i32_i24 %0, %2, %1" : "=v"(r) : "v"(x), "s"(y_uniform));
    return r;
}
static inline unsigned recip22(unsigned d) { return (1u << kRecipShift) / d + 1u; }
static inline bool recip22_ok(unsigned d, unsigned xmax) {
    return d >= 1 && (unsigned long long)xmax * d < (1ull << kRecipShift) && (unsigned long long)xmax * recip22(d) < (1ull << 32);
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
    int is_max;
    int64_t total;   // n * Ho * Wo * C/4
    const float* bias;   // AVE only: added after the division (finishes a commuted 1x1 projection), may be null
    int relu;
    int unit0, n_units;  // grouped launch: workgroups [unit0, unit0 + n_units), kPoolPerWG outputs each
};
constexpr int kPoolPerWG = 1024;   // float4 outputs per pooling workgroup of a grouped launch

// Output i (4 consecutive channels of one pooled pixel).
template <bool IS_MAX>
__device__ __forceinline__ void pool_one(const PoolArgs& a, int64_t i) {
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
        if (a.bias) {
            const float4 b = *reinterpret_cast<const float4*>(a.bias + c4 * 4);
            acc.x += b.x;
            acc.y += b.y;
            acc.z += b.z;
            acc.w += b.w;
        }
        if (a.relu) {
            acc.x = fmaxf(acc.x, 0.f);
            acc.y = fmaxf(acc.y, 0.f);
            acc.z = fmaxf(acc.z, 0.f);
            acc.w = fmaxf(acc.w, 0.f);
        }
    }
    float* o = a.out + (((size_t)n * a.Ho + ph) * a.Wo + pw) * a.Cs_out + a.coff_out + c4 * 4;
    *reinterpret_cast<float4*>(o) = acc;
}

// One 3x3 / stride 1 / pad 1 convolution (+ folded BN + ReLU) in Winograd F(2x2, 3x3) form: see vq_wino.hip.
struct WinoJob {
    const float* in;     // NHWC slot (first crop of this launch)
    const float* u;      // transformed filters [Cin/8][16][Cout][8]: U = G g G^T, position xi = 4 i + j
    const float* bias;   // [Cout]
    float* out;          // NHWC slot, same H x W (first crop of this launch)
    int H, W, Cs_in, coff_in, Cin;
    int Cs_out, coff_out, Cout;
    int th, tw, P;       // 2x2 output tiles per image (rows, columns) and in the whole launch
    int relu;
    int t16;             // 1: u is the second layout of a VQ_OP_CONV_WINOGRAD16 layer -- [Cin/16][16][Cout][16 slots] -- and the units have 16 tiles
    int tiles_n;         // 32*NB-channel blocks of Cout
    int unit0, n_units;  // workgroups [unit0, unit0 + n_units) of the launch belong to this job (unit0 % 8 == 0)
    unsigned in_bytes, u_bytes, out_bytes;   // extents for the buffer descriptors (out-of-range offsets read as zero / are dropped)
    unsigned m_tiles_n, m_tpi, m_tw;         // magic_div constants for tiles_n, th * tw and tw
    unsigned s_tw, s_th;                     // recip22 constants for the per-lane tile walk (at most 31 tiles from the workgroup's first)
#ifdef VQ_WINO_PHASES
    long long* phases;   // tools/ubench/wino_phases.hip: [workgroup][6] s_memtime stamps
#endif
};

constexpr int kWinoMaxJobs = 4;
// Independent layers of one graph level (the 3x3 and the first double-3x3 arm of an inception module, and the module's
// pooling arm) as ONE launch: their workgroups fill each other's tail rounds and kernel boundaries disappear.
constexpr int kWinoMaxPools = 2;
struct WinoGroup {
    int n_jobs;
    int n_pools;         // pooling layers of the same graph level: their (few, short) workgroups come last and fill the tail
    int total_units;     // grid size (jobs padded to multiples of 8 workgroups so every job keeps its XCD mapping)
    int pool_unit0;      // first pooling workgroup
    WinoJob job[kWinoMaxJobs];
    PoolArgs pool[kWinoMaxPools];
};

constexpr int kWinoVariants = 2;   // output-channel blocks of 32 per workgroup: variant v -> (v & 1) + 1; v >= 2: units of 16 tiles (layers that carry both layouts)

// Fill the derived fields of the jobs (unit ranges, magic constants) for `variant` and launch on `stream`.
// ev_start / ev_stop (both or neither): events that receive the kernel's own begin / end timestamps
// (hipExtLaunchKernelGGL) for per-launch profiling.  Returns a VQ_* status.
int launch_wino_group(WinoGroup& g, int variant, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop);

// A launch whose begin / end timestamps land in two events when they are given: the timestamps come from the
// dispatch packet itself, so the measured interval is the kernel alone (what rocprofv3 reports) and no extra
// marker packets sit between consecutive layers.
#define VQ_LAUNCH(KERN, GRID, BLOCK, LDS, STREAM, EV_START, EV_STOP, ...)                                     \
    do {                                                                                                      \
        if ((EV_START) || (EV_STOP))                                                                          \
            hipExtLaunchKernelGGL(KERN, dim3(GRID), dim3(BLOCK), (std::uint32_t)(LDS), STREAM, EV_START, EV_STOP, 0, __VA_ARGS__); \
        else                                                                                                  \
            KERN<<<GRID, BLOCK, LDS, STREAM>>>(__VA_ARGS__);                                                  \
    } while (0)

}  // namespace vq
