// Micro-benchmark: what v_mfma_f32_32x32x2_f32 sustains on this chip, alone and beside LDS operand reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float sm[2][128][36];
    for (int i = threadIdx.x; i < 2 * 128 * 36; i += 256) (&sm[0][0][0])[i] = seed * (i % 7);
    __syncthreads();
    floatx16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    floatx4 a = {seed, seed * 2, seed * 3, seed * 4}, b = {1.f, 2.f, 3.f, 4.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            floatx4 af = a, bf[NACC];
            if (LDS) {
                af = *reinterpret_cast<const floatx4*>(&sm[0][wave * 32 + l31][kk * 8 + half * 4]);
#pragma unroll
                for (int j = 0; j < NACC; ++j) bf[j] = *reinterpret_cast<const floatx4*>(&sm[1][(j * 32 + l31) & 127][kk * 8 + half * 4]);
            } else {
#pragma unroll
                for (int j = 0; j < NACC; ++j) bf[j] = b;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[j][s], acc[j], 0, 0, 0);
        }
    }
    float v = 0;
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) v += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks_per_cu) {
    int cus = 256;
    float* out;
    hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC, LDS><<<cus * blocks_per_cu, 256>>>(out, 10, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, LDS><<<cus * blocks_per_cu, 256>>>(out, iters, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)cus * blocks_per_cu * 4 /*waves*/ * iters * 16.0 * NACC * (2.0 * 32 * 32 * 2);
    printf("%-28s blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<1, false>("regs, 1 acc", 1);
    run<1, false>("regs, 1 acc", 2);
    run<3, false>("regs, 3 acc", 1);
    run<3, false>("regs, 3 acc", 2);
    run<4, false>("regs, 4 acc", 2);
    run<1, true>("lds operands, 1 acc", 1);
    run<1, true>("lds operands, 1 acc", 2);
    run<1, true>("lds operands, 1 acc", 4);
    run<3, true>("lds operands, 3 acc", 1);
    run<3, true>("lds operands, 3 acc", 2);
    run<4, true>("lds operands, 4 acc", 2);
    return 0;
}
