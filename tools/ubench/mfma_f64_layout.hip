// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, checked against a host matmul:
//   A[i][k]: lane l holds i = l % 16, k = l / 16        B[k][j]: lane l holds k = l / 16, j = l % 16
//   D[i][j]: lane l, register r holds i = l / 16 + 4 * r, j = l % 16
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_f64_layout.hip -o tools/ubench/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double doublex4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D, long long* cyc) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    doublex4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];   // raw: [lane][register]
    // issue rate: 256 dependent-free MFMAs on 4 accumulators
    doublex4 c0 = acc, c1 = acc, c2 = acc, c3 = acc;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 64; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[0] = t1 - t0;
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678) D[0] = 0;
}
int main() {
    double hA[64], hB[64], hD[256], ref[256] = {0};
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = (double)((i * 4 + kk) * 2654435761u % 1000003u) / 997.0;
    for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = (double)((kk * 16 + j) * 40503u % 999983u) / 991.0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int kk = 0; kk < 4; ++kk) ref[i * 16 + j] += hA[i * 4 + kk] * hB[kk * 16 + j];
    double *dA, *dB, *dD; long long* dc; long long hc = 0;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dc, 8);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD, dc);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int wi = l / 16 + 4 * r, wj = l % 16;
            const double want = ref[wi * 16 + wj], got = hD[l * 4 + r];
            if (!(got > want * (1 - 1e-14) && got < want * (1 + 1e-14))) {       // the k summation order may differ from the host's
                if (bad < 8) printf("lane %d reg %d: %.17g, expected D[%d][%d] = %.17g\n", l, r, got, wi, wj, want);
                ++bad;
            }
        }
    printf("layout %s (%d mismatches); 256 MFMAs in %lld cycles = %.1f cycles each\n", bad ? "WRONG" : "confirmed", bad, hc, hc / 256.0);
    return bad != 0;
}
