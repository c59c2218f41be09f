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
    int tiles_n;         // 32*NB-channel blocks of Cout
    int unit0, n_units;  // workgroups [unit0, unit0 + n_units) of the launch belong to this job (unit0 % 8 == 0)
    unsigned in_bytes, u_bytes, out_bytes;   // extents for the buffer descriptors (out-of-range offsets read as zero / are dropped)
    unsigned m_tiles_n, m_tpi, m_tw;         // magic_div constants for tiles_n, th * tw and tw
    unsigned s_tw, s_th;                     // 16-bit reciprocals for the per-lane tile walk: x / tw = (x * s_tw) >> 16, x < 1024
#ifdef VQ_WINO_PHASES
    long long* phases;   // tools/ubench/wino_phases.hip: [workgroup][6] s_memtime stamps
#endif
};

constexpr int kWinoMaxJobs = 4;
// Independent layers of one graph level (the 3x3 and the first double-3x3 arm of an inception module) as ONE launch:
// their workgroups fill each other's tail rounds and one kernel boundary disappears.
struct WinoGroup {
    int n_jobs;
    int total_units;     // grid size (jobs padded to multiples of 8 workgroups so every job keeps its XCD mapping)
    WinoJob job[kWinoMaxJobs];
};

constexpr int kWinoVariants = 2;   // output-channel blocks of 32 per workgroup: variant v -> v + 1

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
