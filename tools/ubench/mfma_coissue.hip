// Micro-benchmark: can a wave overlap its own VALU / LDS instructions with its own MFMAs?  NV independent v_fma (or one
// ds_read_b128) are placed behind every v_mfma_f32_32x32x2_f32 of a dependent chain; the time per MFMA should stay at
// 64 cycles if they co-issue.  Run with 1, 2 and 4 workgroups (= waves per SIMD) per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NV, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = seed * (i % 7);
    __syncthreads();
    floatx16 acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1);
    floatx4 l = {0.f, 0.f, 0.f, 0.f};
    const float a = seed, b = 1.0f + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m & 1)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
            else
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j & 7]) : "v"(b));
            if (LDS) {
                floatx4 t = *reinterpret_cast<const floatx4*>(&sm[((threadIdx.x & 63) * 4 + m * 256) & 4095]);
                asm volatile("" ::"v"(t));
                l = t;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + l[0];
}

template <int NV, bool LDS>
void run(int blocks_per_cu) {
    const int cus = 256, iters = 2000;
    float* out;
    (void)hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<NV, LDS><<<cus * blocks_per_cu, 256>>>(out, 10, 1e-3f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NV, LDS><<<cus * blocks_per_cu, 256>>>(out, iters, 1e-3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = (double)blocks_per_cu * iters * 16;
    printf("%d v_fma%s per MFMA, %d wave(s)/SIMD: %.3f ms -> %.1f ns per MFMA per SIMD (64 cycles = %.1f ns at 2.4 GHz)\n", NV,
           LDS ? " + 1 ds_read_b128" : "", blocks_per_cu, ms, ms * 1e6 / mfmas_per_simd, 64 / 2.4);
    (void)hipFree(out);
}

int main() {
    run<0, false>(1);
    run<2, false>(1);
    run<4, false>(1);
    run<8, false>(1);
    run<0, true>(1);
    run<4, true>(1);
    run<4, false>(2);
    run<8, false>(2);
    run<8, false>(4);
    run<4, true>(2);
    run<4, true>(4);
    return 0;
}
