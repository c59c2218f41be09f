// csrc/vq_flow_math.h against the compiler's own `a / b` and sqrtf(): bit for bit on pseudo-random operand PATTERNS (every exponent, denormals,
// zeros, infinities, NaNs occur) and on operands of the magnitudes the TV-L1 iteration produces.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Ivideo-query-algorithms_amd/csrc tools/ubench/flow_math_check.hip -o tools/ubench/build/flow_math_check
//   tools/ubench/build/flow_math_check [log2 of the number of operand pairs per test, default 28]      prints "ok" or the first mismatches
#include <cstdio>
#include <cstdlib>

#include "vq_flow_math.h"

__device__ __forceinline__ unsigned mix(unsigned long long i, unsigned salt) {      // splitmix-style: a different 32-bit pattern per index
    unsigned long long z = (i + salt) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)((z ^ (z >> 31)) >> 16);
}
__device__ __forceinline__ bool same(float a, float b) {
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);      // NaNs: any payload
}

// mode 0: raw patterns; mode 1: numerator tiny / zero / normal against denominators in [1, 2^20) (the dual step); mode 2: like 0 with every
// 64th numerator an exact zero of either sign
__global__ void check_div(unsigned long long n, int mode, unsigned long long* bad, float* first) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f2 a = {__uint_as_float(mix(4 * i, 1)), __uint_as_float(mix(4 * i + 1, 1))};
    f2 b = {__uint_as_float(mix(4 * i + 2, 1)), __uint_as_float(mix(4 * i + 3, 1))};
    if (mode == 1) {
        b.x = 1.0f + fabsf(__uint_as_float((mix(4 * i + 2, 7) & 0x007FFFFFu) | ((127u + (mix(i, 9) % 20u)) << 23)));
        b.y = 1.0f + fabsf(__uint_as_float((mix(4 * i + 3, 7) & 0x007FFFFFu) | ((127u + (mix(i, 11) % 20u)) << 23)));
        const unsigned ex = mix(i, 13) % 160u, ey = mix(i, 15) % 160u;                   // numerators from 2^-150 (-> 0, denormal) to 2^9
        a.x = __uint_as_float((__float_as_uint(a.x) & 0x807FFFFFu) | (ex << 23));
        a.y = __uint_as_float((__float_as_uint(a.y) & 0x807FFFFFu) | (ey << 23));
    }
    if (mode == 2 && (i & 63) == 0) {
        a.x = __uint_as_float(__float_as_uint(a.x) & 0x80000000u);
        a.y = 0.0f;
    }
    const f2 q = div_ieee(a, b);
    const float rx = a.x / b.x, ry = a.y / b.y;
    if (!same(q.x, rx) || !same(q.y, ry)) {
        if (atomicAdd(bad, 1ull) == 0) {
            first[0] = a.x, first[1] = b.x, first[2] = q.x, first[3] = rx, first[4] = a.y, first[5] = b.y, first[6] = q.y, first[7] = ry;
        }
    }
}

// mode 0: every non-negative pattern class; mode 1: sums of two squares of gradients between 2^-60 and 2^4 (incl. the scaled path below 2^-96)
__global__ void check_sqrt(unsigned long long n, int mode, unsigned long long* bad, float* first) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f2 x = {__uint_as_float(mix(2 * i, 3) & 0x7FFFFFFFu), __uint_as_float(mix(2 * i + 1, 3) & 0x7FFFFFFFu)};
    if (mode == 1) {
        const float ux = __uint_as_float((mix(2 * i, 5) & 0x807FFFFFu) | ((67u + mix(i, 17) % 64u) << 23));
        const float uy = __uint_as_float((mix(2 * i + 1, 5) & 0x807FFFFFu) | ((67u + mix(i, 19) % 64u) << 23));
        x.x = ux * ux + uy * uy;
        x.y = (i & 7) == 0 ? 0.0f : uy * uy + ux * ux * 0.25f;
    }
    const f2 s = sqrt_ieee(x);
    const float rx = sqrtf(x.x), ry = sqrtf(x.y);
    if (!same(s.x, rx) || !same(s.y, ry)) {
        if (atomicAdd(bad, 1ull) == 0) {
            first[0] = x.x, first[1] = 0, first[2] = s.x, first[3] = rx, first[4] = x.y, first[5] = 0, first[6] = s.y, first[7] = ry;
        }
    }
}

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            return 2;                                                               \
        }                                                                           \
    } while (0)

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 28;
    const unsigned long long n = 1ull << lg;
    unsigned long long* bad;
    float* first;
    CK(hipMalloc(&bad, sizeof *bad));
    CK(hipMalloc(&first, 8 * sizeof(float)));
    int rc = 0;
    for (int test = 0; test < 5; ++test) {
        CK(hipMemset(bad, 0, sizeof *bad));
        const unsigned blocks = (unsigned)((n + 255) / 256);
        if (test < 3)
            check_div<<<blocks, 256>>>(n, test, bad, first);
        else
            check_sqrt<<<blocks, 256>>>(n, test - 3, bad, first);
        CK(hipDeviceSynchronize());
        unsigned long long h = 0;
        float f[8];
        CK(hipMemcpy(&h, bad, sizeof h, hipMemcpyDeviceToHost));
        CK(hipMemcpy(f, first, sizeof f, hipMemcpyDeviceToHost));
        printf("%s mode %d: %llu operand pairs x 2, %llu mismatches\n", test < 3 ? "division" : "square root", test < 3 ? test : test - 3, n, h);
        if (h) {
            printf("  first: %a / %a -> %a, library %a;  %a / %a -> %a, library %a\n", f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
            rc = 1;
        }
    }
    printf(rc ? "MISMATCH\n" : "ok\n");
    return rc;
}
