// Micro-benchmark: 4 MFMA waves (LDS operands) + 4 loader waves (LDS-DMA or register staging) per workgroup,
// no synchronisation between them: how much does the staging stream slow the matrix waves down?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// MODE 0: loaders idle; 1: LDS-DMA, light address math; 2: LDS-DMA + ~20 VALU per DMA; 3: global->VGPR->ds_write
// PRIO: consumers raise their priority
template <int MODE, int PRIO, int NDMA>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sm = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < 3 * 256 * 32; i += 512) sm[i] = seed * (i % 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    if (wave < 4) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        floatx16 acc[4];
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const char* stage = smem + (it % 3) * 32768;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                floatx4 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = (wave >> 1) * 64 + 32 * i + l31;
                    fa[i] = *reinterpret_cast<const floatx4*>(stage + r * 128 + (((kk * 2 + half) ^ ((r >> 1) & 7)) << 4));
                    const int rb = (wave & 1) * 64 + 32 * i + l31;
                    fb[i] = *reinterpret_cast<const floatx4*>(stage + 16384 + rb * 128 + (((kk * 2 + half) ^ ((rb >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i * 2 + j], 0, 0, 0);
            }
        }
        float v = 0;
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) v += acc[j][r];
        out[blockIdx.x * 256 + threadIdx.x] = v;
    } else if (MODE != 0) {
        const int L = wave - 4;
        const float* base = src + (size_t)blockIdx.x * 65536 + lane * 4;
        unsigned h = lane;
        for (int it = 0; it < iters; ++it) {
            char* stage = smem + ((it + 2) % 3) * 32768;
#pragma unroll
            for (int t = 0; t < NDMA; ++t) {
                const int q = L + 4 * t;
                size_t off = (size_t)((it * 32 + q) & 255) * 256;
                if (MODE == 2) {   // emulate im2col address arithmetic
#pragma unroll
                    for (int z = 0; z < 10; ++z) h = h * 1664525u + 1013904223u + (unsigned)off;
                    off += (h >> 31);   // 0 or 1: keeps the chain alive without leaving the buffer
                }
                if (MODE == 3) {
                    floatx4 v = *reinterpret_cast<const floatx4*>(base + off);
                    *reinterpret_cast<floatx4*>(stage + q * 1024 + lane * 16) = v;
                } else {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                                     (__attribute__((address_space(3))) void*)(stage + q * 1024), 16, 0, 0);
                }
            }
            if (MODE != 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        }
    }
}

template <int MODE, int PRIO, int NDMA>
void run(const char* name, const float* src) {
    const int cus = 256, iters = 1500;
    float* out;
    hipMalloc(&out, (size_t)cus * 256 * 4);
    auto kern = k<MODE, PRIO, NDMA>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<cus, 512, 98304>>>(src, out, 10, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<cus, 512, 98304>>>(src, out, iters, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)cus * 4 * iters * 64.0 * (2.0 * 32 * 32 * 2);
    printf("%-52s %.3f ms  %.1f TFLOP/s (MFMA waves)  staging %.2f TB/s\n", name, ms, flops / ms / 1e9,
           MODE ? (double)cus * 4 * NDMA * 1024.0 * iters / ms / 1e9 : 0.0);
    hipFree(out);
}

int main() {
    float* src;
    hipMalloc(&src, (size_t)256 * 65536 * 4 + 4096);
    hipMemset(src, 0, (size_t)256 * 65536 * 4 + 4096);
    run<0, 0, 8>("loaders idle", src);
    run<1, 0, 8>("LDS-DMA 8/wave/step, light address math", src);
    run<1, 1, 8>("LDS-DMA 8/wave/step, light, consumers prio 3", src);
    run<2, 0, 8>("LDS-DMA 8/wave/step, ~20 VALU per DMA", src);
    run<2, 1, 8>("LDS-DMA 8/wave/step, ~20 VALU, consumers prio 3", src);
    run<3, 0, 8>("global->VGPR->ds_write 8/wave/step", src);
    run<3, 1, 8>("global->VGPR->ds_write, consumers prio 3", src);
    run<1, 0, 4>("LDS-DMA 4/wave/step, light", src);
    run<2, 1, 4>("LDS-DMA 4/wave/step, ~20 VALU, prio 3", src);
    return 0;
}
