// Target bootstrapping on gfx950: the closed-form query update of the reference from user-validated clips.
//
// What it replaces (paths relative to the reference checkout):
//   src/models/target_clip.py:161-198  _bootstrap_valid_matches       w = X (X^T X)^-1 1
//   src/models/target_clip.py:200-261  _bootstrap_valid_plus_invalid  M = I + s Y^T Y (1024 x 1024), two explicit inverses
// with X = the m validated matches and Y = the n validated non-matches of one (stream, split), s = mu / tr(Y Y^T).
//
// The reference inverts a D x D matrix per (stream, split, bag) -- 18 inverses of 1024^3 per query round, seconds on a
// host.  Both closed forms only ever apply that inverse to vectors in span{X, Y}: with the Gram matrix
// G = [X; Y][X; Y]^T = [[Gxx, Gxy], [Gyx, Gyy]] (Woodbury, M^-1 = I - s Y^T (I + s Gyy)^-1 Y)
//     K' = (I + s Gyy)^-1,  C = s K' Gyx,  gamma = s K' 1
//     B  = Gxx - Gxy C,     beta = B^-1 1,  delta = B^-1 Gxy gamma
//     w  = X^T (beta - delta) + Y^T (gamma - C (beta - delta))
// (n = 0 or mu = 0:  w = X^T Gxx^-1 1, the first form).  One workgroup per problem: Gram matrix in fp64 (a wave per
// row pair, 64-lane strided dot + shuffle reduction), the two small solves by Gauss-Jordan with partial pivoting in a
// global scratch area that stays in L2, then the combination of the rows.  A few hundred KB of traffic per problem.
#include <vector>

#include "vq_common.h"

using namespace vq;

namespace {

constexpr int BOOT_MAX_ROWS = 256;     // validated clips per problem (m + n)

struct BootArgs {
    const void* base;          // rows live at base + row_off[p * stride + k] elements
    const int64_t* row_off;    // [P][stride]: the m valid rows first, then the n invalid rows
    const int32_t* n_valid;    // [P]
    const int32_t* n_invalid;  // [P]
    int stride, D;
    double mu;
    double* ws;                // [P][ws_stride] scratch
    int64_t ws_stride;
    double* out;               // [P][D]
    int32_t* status;           // [P]: 0 ok, 1 singular system
};

// Solve A Z = R in place on the augmented matrix aug[dim][dim + nrhs] (row stride ld): Gauss-Jordan with partial
// pivoting, whole workgroup.  Returns false on a zero / non-finite pivot.
__device__ bool gauss_jordan(double* aug, int dim, int nrhs, int ld, int* piv_shared) {
    const int tid = threadIdx.x, nt = blockDim.x, width = dim + nrhs;
    for (int k = 0; k < dim; ++k) {
        if (tid == 0) {
            int best = k;
            double bv = fabs(aug[(size_t)k * ld + k]);
            for (int r = k + 1; r < dim; ++r) {
                const double v = fabs(aug[(size_t)r * ld + k]);
                if (v > bv) {
                    bv = v;
                    best = r;
                }
            }
            *piv_shared = (bv > 0.0 && bv < 1.0e300) ? best : -1;
        }
        __syncthreads();
        const int p = *piv_shared;
        if (p < 0) return false;
        if (p != k)
            for (int c = tid; c < width; c += nt) {
                const double t = aug[(size_t)k * ld + c];
                aug[(size_t)k * ld + c] = aug[(size_t)p * ld + c];
                aug[(size_t)p * ld + c] = t;
            }
        __syncthreads();
        const double inv = 1.0 / aug[(size_t)k * ld + k];
        __syncthreads();
        for (int c = tid; c < width; c += nt) aug[(size_t)k * ld + c] *= inv;
        __syncthreads();
        // eliminate column k from every other row; the factor is read before anybody overwrites column k
        for (int idx = tid; idx < dim * (width - k - 1); idx += nt) {
            const int r = idx / (width - k - 1), c = k + 1 + idx % (width - k - 1);
            if (r != k) aug[(size_t)r * ld + c] -= aug[(size_t)r * ld + k] * aug[(size_t)k * ld + c];
        }
        __syncthreads();
        for (int r = tid; r < dim; r += nt)
            if (r != k) aug[(size_t)r * ld + k] = 0.0;
        __syncthreads();
    }
    return true;
}

template <typename T>
__global__ __launch_bounds__(256) void bootstrap_kernel(BootArgs a) {
    __shared__ int piv;
    __shared__ double sh_scale;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = a.n_valid[p], n = a.n_invalid[p], r = m + n, D = a.D;
    const T* base = static_cast<const T*>(a.base);
    const int64_t* off = a.row_off + (size_t)p * a.stride;
    double* G = a.ws + (size_t)p * a.ws_stride;            // [r][r]
    double* A1 = G + (size_t)r * r;                         // [n][n + m + 1]
    double* Bm = A1 + (size_t)n * (n + m + 1);              // [m][m + 2]
    double* coef = Bm + (size_t)m * (m + 2);                // [r]: a (m), b (n)

    // Gram matrix, upper triangle then mirrored: one wave per (i, j)
    for (int idx = wave; idx < r * r; idx += 4) {
        const int i = idx / r, j = idx - i * r;
        if (j < i) continue;
        const T* xi = base + off[i];
        const T* xj = base + off[j];
        double s = 0.0;
        for (int d = lane; d < D; d += 64) s += (double)xi[d] * (double)xj[d];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) {
            G[(size_t)i * r + j] = s;
            G[(size_t)j * r + i] = s;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double tr = 0.0;
        for (int j = 0; j < n; ++j) tr += G[(size_t)(m + j) * r + (m + j)];
        sh_scale = (n > 0 && a.mu != 0.0) ? a.mu / tr : 0.0;     // target_clip.py:248-249
    }
    __syncthreads();
    const double scale = sh_scale;
    const bool with_y = n > 0 && scale != 0.0;
    bool ok = true;
    if (with_y) {
        const int w1 = n + m + 1;
        for (int idx = tid; idx < n * w1; idx += 256) {
            const int i = idx / w1, c = idx - i * w1;
            double v;
            if (c < n)
                v = (i == c ? 1.0 : 0.0) + scale * G[(size_t)(m + i) * r + (m + c)];     // I + s Gyy
            else if (c < n + m)
                v = G[(size_t)(m + i) * r + (c - n)];                                        // Gyx
            else
                v = 1.0;
            A1[idx] = v;
        }
        __syncthreads();
        ok = gauss_jordan(A1, n, m + 1, w1, &piv);      // columns n..n+m-1: K' Gyx, column n+m: K' 1
        if (ok) {
            // B = Gxx - Gxy C,  C = s K' Gyx;  right-hand sides 1 and Gxy gamma, gamma = s K' 1
            for (int idx = tid; idx < m * (m + 2); idx += 256) {
                const int i = idx / (m + 2), c = idx - i * (m + 2);
                double v;
                if (c < m) {
                    double s = 0.0;
                    for (int j = 0; j < n; ++j) s += G[(size_t)i * r + (m + j)] * (scale * A1[(size_t)j * w1 + n + c]);
                    v = G[(size_t)i * r + c] - s;
                } else if (c == m) {
                    v = 1.0;
                } else {
                    double s = 0.0;
                    for (int j = 0; j < n; ++j) s += G[(size_t)i * r + (m + j)] * (scale * A1[(size_t)j * w1 + n + m]);
                    v = s;
                }
                Bm[idx] = v;
            }
            __syncthreads();
            ok = gauss_jordan(Bm, m, 2, m + 2, &piv);
        }
        if (ok) {
            for (int i = tid; i < m; i += 256) coef[i] = Bm[(size_t)i * (m + 2) + m] - Bm[(size_t)i * (m + 2) + m + 1];   // a = beta - delta
            __syncthreads();
            for (int j = tid; j < n; j += 256) {                                                                      // b = gamma - C a
                double s = 0.0;
                for (int i = 0; i < m; ++i) s += (scale * A1[(size_t)j * w1 + n + i]) * coef[i];
                coef[m + j] = scale * A1[(size_t)j * w1 + n + m] - s;
            }
        }
    } else {
        for (int idx = tid; idx < m * (m + 2); idx += 256) {
            const int i = idx / (m + 2), c = idx - i * (m + 2);
            Bm[idx] = c < m ? G[(size_t)i * r + c] : (c == m ? 1.0 : 0.0);
        }
        __syncthreads();
        ok = gauss_jordan(Bm, m, 2, m + 2, &piv);
        if (ok) {
            for (int i = tid; i < m; i += 256) coef[i] = Bm[(size_t)i * (m + 2) + m];
            for (int j = tid; j < n; j += 256) coef[m + j] = 0.0;
        }
    }
    __syncthreads();
    if (tid == 0) a.status[p] = ok ? 0 : 1;
    if (!ok) return;
    for (int d = tid; d < D; d += 256) {
        double s = 0.0;
        for (int k = 0; k < r; ++k) s += coef[k] * (double)base[off[k] + d];
        a.out[(size_t)p * D + d] = s;
    }
}

// Shared launcher.  row_off_host: [P][stride] element offsets into base_dev (valid rows first).
int run_bootstrap(const void* base_dev, int dtype, const std::vector<int64_t>& row_off_host, int stride, const int32_t* n_valid,
                  const int32_t* n_invalid, int P, int D, double mu, hipStream_t stream, double* targets_host, double* targets_dev_copy) {
    int max_r = 0;
    for (int p = 0; p < P; ++p) {
        VQ_REQUIRE(n_valid[p] >= 1, "problem %d: bootstrapping needs at least one validated match", p);
        VQ_REQUIRE(n_invalid[p] >= 0 && n_valid[p] + n_invalid[p] <= BOOT_MAX_ROWS, "problem %d: %d + %d validated clips (limit %d)", p,
                   n_valid[p], n_invalid[p], BOOT_MAX_ROWS);
        max_r = std::max(max_r, n_valid[p] + n_invalid[p]);
    }
    VQ_REQUIRE(stride >= max_r, "row offset stride too small");
    const int64_t ws_stride = (int64_t)max_r * max_r + (int64_t)max_r * (max_r + 1) + (int64_t)max_r * (max_r + 2) + max_r;
    int64_t* d_off = nullptr;
    int32_t* d_cnt = nullptr;      // n_valid [P], n_invalid [P], status [P]
    double *d_ws = nullptr, *d_out = nullptr;
    auto cleanup = [&]() {
        if (d_off) (void)hipFree(d_off);
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_ws) (void)hipFree(d_ws);
        if (d_out) (void)hipFree(d_out);
    };
#define VQ_BOOT_HIP(call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            cleanup();                                                                         \
            return fail(e_ == hipErrorOutOfMemory ? VQ_E_NOMEM : VQ_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
        }                                                                                      \
    } while (0)
    VQ_BOOT_HIP(vq::malloc_trim((void**)&d_off, row_off_host.size() * sizeof(int64_t)));
    VQ_BOOT_HIP(vq::malloc_trim((void**)&d_cnt, (size_t)3 * P * sizeof(int32_t)));
    VQ_BOOT_HIP(vq::malloc_trim((void**)&d_ws, (size_t)P * ws_stride * sizeof(double)));
    VQ_BOOT_HIP(vq::malloc_trim((void**)&d_out, (size_t)P * D * sizeof(double)));
    VQ_BOOT_HIP(hipMemcpyAsync(d_off, row_off_host.data(), row_off_host.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    VQ_BOOT_HIP(hipMemcpyAsync(d_cnt, n_valid, (size_t)P * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    VQ_BOOT_HIP(hipMemcpyAsync(d_cnt + P, n_invalid, (size_t)P * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    BootArgs a;
    a.base = base_dev;
    a.row_off = d_off;
    a.n_valid = d_cnt;
    a.n_invalid = d_cnt + P;
    a.stride = stride;
    a.D = D;
    a.mu = mu;
    a.ws = d_ws;
    a.ws_stride = ws_stride;
    a.out = d_out;
    a.status = d_cnt + 2 * P;
    if (dtype == VQ_F32)
        bootstrap_kernel<float><<<P, 256, 0, stream>>>(a);
    else
        bootstrap_kernel<double><<<P, 256, 0, stream>>>(a);
    {
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) {
            cleanup();
            return fail(VQ_E_HIP, "bootstrap kernel launch failed: %s", hipGetErrorString(e_));
        }
    }
    std::vector<int32_t> status(P);
    VQ_BOOT_HIP(hipMemcpyAsync(status.data(), d_cnt + 2 * P, (size_t)P * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    if (targets_host) VQ_BOOT_HIP(hipMemcpyAsync(targets_host, d_out, (size_t)P * D * sizeof(double), hipMemcpyDeviceToHost, stream));
    if (targets_dev_copy) VQ_BOOT_HIP(hipMemcpyAsync(targets_dev_copy, d_out, (size_t)P * D * sizeof(double), hipMemcpyDeviceToDevice, stream));
    VQ_BOOT_HIP(hipStreamSynchronize(stream));
#undef VQ_BOOT_HIP
    cleanup();
    for (int p = 0; p < P; ++p)
        if (status[p] != 0)
            return fail(VQ_E_INVALID, "problem %d: singular system (linearly dependent validated clips) -- numpy.linalg.inv raises here too", p);
    return VQ_OK;
}

}  // namespace

namespace vq {

int bootstrap_from_device_rows(const void* base_dev, int dtype, const std::vector<int64_t>& row_off, int stride, const int32_t* n_valid,
                               const int32_t* n_invalid, int P, int D, double mu, hipStream_t stream, double* targets_host,
                               double* targets_dev_copy) {
    return run_bootstrap(base_dev, dtype, row_off, stride, n_valid, n_invalid, P, D, mu, stream, targets_host, targets_dev_copy);
}

}  // namespace vq

extern "C" int vq_bootstrap_targets(const void* rows_host, int32_t dtype, int32_t n_problems, const int32_t* n_valid,
                                    const int32_t* n_invalid, int32_t dim, double mu, int32_t device, double* targets_host) {
    VQ_REQUIRE(rows_host && n_valid && n_invalid && targets_host, "NULL argument");
    VQ_REQUIRE(dtype == VQ_F32 || dtype == VQ_F64, "dtype must be VQ_F32 or VQ_F64");
    VQ_REQUIRE(n_problems > 0 && dim > 0, "n_problems and dim must be positive");
    int ndev = 0;
    VQ_HIP(hipGetDeviceCount(&ndev));
    VQ_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard g(device);
    int stride = 0;
    int64_t total = 0;
    for (int p = 0; p < n_problems; ++p) {
        VQ_REQUIRE(n_valid[p] >= 0 && n_invalid[p] >= 0, "negative row count");
        stride = std::max(stride, n_valid[p] + n_invalid[p]);
        total += n_valid[p] + n_invalid[p];
    }
    VQ_REQUIRE(stride > 0, "no rows");
    std::vector<int64_t> off((size_t)n_problems * stride, 0);
    int64_t row = 0;
    for (int p = 0; p < n_problems; ++p)
        for (int k = 0; k < n_valid[p] + n_invalid[p]; ++k) off[(size_t)p * stride + k] = (row++) * dim;
    const size_t esz = dtype == VQ_F64 ? 8 : 4;
    void* d_rows = nullptr;
    VQ_HIP(vq::malloc_trim(&d_rows, (size_t)total * dim * esz));
    hipError_t e = hipMemcpy(d_rows, rows_host, (size_t)total * dim * esz, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d_rows);
        return fail(VQ_E_HIP, "hipMemcpy(rows) failed: %s", hipGetErrorString(e));
    }
    const int rc = run_bootstrap(d_rows, dtype, off, stride, n_valid, n_invalid, n_problems, dim, mu, nullptr, targets_host, nullptr);
    (void)hipFree(d_rows);
    return rc;
}
