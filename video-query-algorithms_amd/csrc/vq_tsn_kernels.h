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

// 3x3 / stride 1 / pad 1 convolution (+ folded BN + ReLU) as Winograd F(2x2, 3x3): see vq_wino.hip.
struct WinoArgs {
    const float* in;     // NHWC slot
    const float* u;      // transformed filters [Cin/8][16][Cout][8]: U = G g G^T, position xi = 4 i + j
    const float* bias;   // [Cout]
    float* out;          // NHWC slot, same H x W
    int H, W, Cs_in, coff_in, Cin;
    int Cs_out, coff_out, Cout;
    int th, tw, P;       // 2x2 output tiles per image (rows, columns) and in the whole batch
    int relu;
    int tiles_m, tiles_n;
    unsigned in_bytes, u_bytes;   // extents for the buffer descriptors (out-of-range offsets read as zero)
#ifdef VQ_WINO_PHASES
    long long* phases;   // tools/ubench/wino_phases.hip: [workgroup][4] s_memtime stamps
#endif
};

constexpr int kWinoVariants = 2;   // output-channel blocks of 32 per workgroup: variant v -> v + 1

// Launch on `stream`; variant in [0, kWinoVariants).  ev_start / ev_stop (both or neither): events that receive the
// kernel's own begin / end timestamps (hipExtLaunchKernelGGL) for per-layer profiling.  Returns a VQ_* status.
int launch_wino(const WinoArgs& a, int variant, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop);

// A launch whose begin / end timestamps land in two events when they are given: the timestamps come from the
// dispatch packet itself, so the measured interval is the kernel alone (what rocprofv3 reports) and no extra
// marker packets sit between consecutive layers.
#define VQ_LAUNCH(KERN, GRID, BLOCK, LDS, STREAM, EV_START, EV_STOP, ...)                                     \
    do {                                                                                                      \
        if (EV_START)                                                                                         \
            hipExtLaunchKernelGGL(KERN, dim3(GRID), dim3(BLOCK), (std::uint32_t)(LDS), STREAM, EV_START, EV_STOP, 0, __VA_ARGS__); \
        else                                                                                                  \
            KERN<<<GRID, BLOCK, LDS, STREAM>>>(__VA_ARGS__);                                                  \
    } while (0)

}  // namespace vq
