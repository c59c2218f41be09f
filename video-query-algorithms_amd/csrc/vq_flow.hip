// TV-L1 optical flow on gfx950: the arithmetic behind the reference's flow frames.
//
// What it replaces (paths relative to the reference checkout):
//   src/features_GPU_compute/build_wof_clips.py:55-76   run_warp_optical_flow(): os.system("<TSN_ROOT>/lib/dense_flow/build/
//       extract_warp_gpu -f <video> -x flow_x -y flow_y -b 20 -t 1 -d <gpu> -s 1 -o dir") -- a third-party binary (OpenCV CUDA
//       TV-L1; not in the reference tree, no pinned version).  "-t 1" = TV-L1, "-b 20" = clamp to +-20 px and quantise to 8 bits.
// PARITY UNPINNED: the reference holds neither frames nor flow images nor the binary.  The kernels follow oracle/
// tvl1_oracle.py -- the PUBLISHED algorithm (Zach, Pock & Bischof 2007 as formulated in IPOL 2013, Algorithm 1) with OpenCV's
// default parameters and interpolation choices -- operation for operation in fp32 (contraction off, correctly rounded
// division and square root), so device and oracle agree to rounding.  The feature-matching half of dense_flow's camera-motion
// "warp" (SURF + RANSAC homography) is not built; vq_flow_warp_homography applies a GIVEN homography to the second frame.
//
// Shape of the work: a BATCH of independent frame pairs (one 340 x 256 pair is only 87 k pixels).  Per pyramid level and warp:
// one warp kernel, then per inner iteration two stencil kernels over all pairs -- the primal step needs every neighbour's dual
// variable, the dual step every neighbour's new primal value, so the two cannot share a launch without halo recomputation
// (two launches per iteration; the dual launch also closes the iteration):
//   tvl1_primal_kernel: thresholding step + u = v + theta div p, per-pair squared update summed in fp64 (one atomic per wave)
//   tvl1_dual_kernel:   p = (p + tau/theta grad u) / (1 + tau/theta |grad u|)
// A pair that has converged (mean squared update <= epsilon^2, or the iteration cap) is switched off on the device and its
// workgroups exit at once; the host looks at the number of live pairs every few iterations only.  Everything is HBM / L2
// streaming of fp32 planes (about 90 bytes per pixel and iteration): no LDS, no MFMA -- a bandwidth-bound stencil.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "vq_common.h"

using namespace vq;

namespace {

constexpr float kGradIsZero = 1e-10f;

struct Level {
    int h, w;
    size_t off;          // float offset of this level inside a per-plane pyramid buffer (per pair: see plane())
};

// bilinear sample positions of cv::resize INTER_LINEAR (fp64 coordinate, fp32 weights -- oracle.resize_bilinear)
__global__ void resize_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int ih, int iw, int oh, int ow, float gain) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * oh * ow) return;
    const int x = (int)(i % ow), y = (int)((i / ow) % oh);
    const int64_t p = i / ((int64_t)oh * ow);
    double ys = ((double)y + 0.5) * (double)ih / (double)oh - 0.5, xs = ((double)x + 0.5) * (double)iw / (double)ow - 0.5;
    ys = fmin(fmax(ys, 0.0), (double)(ih - 1));
    xs = fmin(fmax(xs, 0.0), (double)(iw - 1));
    const int y0 = (int)floor(ys), x0 = (int)floor(xs);
    const int y1 = min(y0 + 1, ih - 1), x1 = min(x0 + 1, iw - 1);
    const float wy = (float)(ys - (double)y0), wx = (float)(xs - (double)x0);
    const float* a = src + p * (int64_t)ih * iw;
    const float top = a[y0 * iw + x0] * (1.0f - wx) + a[y0 * iw + x1] * wx;
    const float bot = a[y1 * iw + x0] * (1.0f - wx) + a[y1 * iw + x1] * wx;
    dst[i] = (top * (1.0f - wy) + bot * wy) * gain;
}

__global__ void u8_to_float_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) dst[i] = (float)src[i];
}

// centered gradient of I1: 0.5 (I[x+1] - I[x-1]), the missing neighbour at the border replaced by the pixel itself
__global__ void gradient_kernel(const float* __restrict__ img, float* __restrict__ gx, float* __restrict__ gy, int n, int h, int w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * h * w) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const float* a = img + (i - (int64_t)y * w - x);
    gx[i] = 0.5f * (a[y * w + min(x + 1, w - 1)] - a[y * w + max(x - 1, 0)]);
    gy[i] = 0.5f * (a[min(y + 1, h - 1) * w + x] - a[max(y - 1, 0) * w + x]);
}

__device__ __forceinline__ float sample_bilinear(const float* __restrict__ a, int h, int w, float xs, float ys) {
    xs = fminf(fmaxf(xs, 0.0f), (float)(w - 1));
    ys = fminf(fmaxf(ys, 0.0f), (float)(h - 1));
    const int x0 = (int)floorf(xs), y0 = (int)floorf(ys);
    const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    const float wx = xs - (float)x0, wy = ys - (float)y0;
    const float top = a[y0 * w + x0] * (1.0f - wx) + a[y0 * w + x1] * wx;
    const float bot = a[y1 * w + x0] * (1.0f - wx) + a[y1 * w + x1] * wx;
    return top * (1.0f - wy) + bot * wy;
}

struct PairState {
    double err;          // sum of squared primal updates of the iteration in flight
    int stop_iter;       // iterations >= stop_iter of the current warp do not run (INT_MAX while the inner loop is live)
    int iters;           // inner iterations run in the current warp
};
constexpr int kNoStop = 0x7FFFFFFF;

// Start of a warp: I1 and its gradient sampled at x + u, |grad|^2, the constant part of rho; the pair becomes active.
__global__ void tvl1_warp_kernel(const float* __restrict__ i0, const float* __restrict__ i1, const float* __restrict__ i1x,
                                 const float* __restrict__ i1y, const float* __restrict__ u1, const float* __restrict__ u2,
                                 float* __restrict__ i1wx, float* __restrict__ i1wy, float* __restrict__ grad, float* __restrict__ rho_c,
                                 PairState* __restrict__ st, int n, int h, int w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * h * w) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const int64_t p = i / ((int64_t)h * w), base = p * (int64_t)h * w;
    const float a = u1[i], b = u2[i];
    const float xs = (float)x + a, ys = (float)y + b;
    const float w0 = sample_bilinear(i1 + base, h, w, xs, ys);
    const float wx = sample_bilinear(i1x + base, h, w, xs, ys);
    const float wy = sample_bilinear(i1y + base, h, w, xs, ys);
    i1wx[i] = wx;
    i1wy[i] = wy;
    grad[i] = wx * wx + wy * wy;
    rho_c[i] = w0 - wx * a - wy * b - i0[i];
    if (x == 0 && y == 0) {
        st[p].err = 0.0;
        st[p].stop_iter = kNoStop;
        st[p].iters = 0;
    }
}

struct IterArgs {
    const float *i1wx, *i1wy, *grad, *rho_c;
    float *u1, *u2, *p11, *p12, *p21, *p22;
    PairState* st;
    int n, h, w;
    int k;               // index of this inner iteration inside the warp
    float l_t, theta, taut;
};

// Primal step of one pair per blockIdx.y; blockIdx.x strides over its pixels.
__global__ __launch_bounds__(256) void tvl1_primal_kernel(IterArgs a) {
    const int p = blockIdx.y;
    if (a.st[p].stop_iter <= a.k) return;
    const int hw = a.h * a.w;
    const int64_t base = (int64_t)p * hw;
    double local = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const int x = i % a.w, y = i / a.w;
        const int64_t g = base + i;
        const float ux = a.u1[g], uy = a.u2[g], gx = a.i1wx[g], gy = a.i1wy[g], gr = a.grad[g];
        const float rho = a.rho_c[g] + (gx * ux + gy * uy);
        float d1, d2;
        if (rho < -a.l_t * gr) {
            d1 = a.l_t * gx;
            d2 = a.l_t * gy;
        } else if (rho > a.l_t * gr) {
            d1 = -a.l_t * gx;
            d2 = -a.l_t * gy;
        } else if (gr > kGradIsZero) {
            const float fi = -rho / gr;
            d1 = fi * gx;
            d2 = fi * gy;
        } else {
            d1 = d2 = 0.0f;
        }
        // divergence of the dual variables: backward differences, p[-1] = 0
        const float div1 = (x > 0 ? a.p11[g] - a.p11[g - 1] : a.p11[g]) + (y > 0 ? a.p12[g] - a.p12[g - a.w] : a.p12[g]);
        const float div2 = (x > 0 ? a.p21[g] - a.p21[g - 1] : a.p21[g]) + (y > 0 ? a.p22[g] - a.p22[g - a.w] : a.p22[g]);
        const float n1 = (ux + d1) + a.theta * div1, n2 = (uy + d2) + a.theta * div2;
        const float e = (n1 - ux) * (n1 - ux) + (n2 - uy) * (n2 - uy);
        local += (double)e;
        a.u1[g] = n1;
        a.u2[g] = n2;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&a.st[p].err, local);
}

// Dual step.  Its first thread also closes the iteration: the squared update of the primal step just finished (complete: it
// ran in the previous launch) decides whether iteration k + 1 runs.  Every workgroup of this launch tests stop_iter > k,
// which holds for the old value (no stop) and the new one (k + 1) alike, so the write cannot split the pair.
__global__ __launch_bounds__(256) void tvl1_dual_kernel(IterArgs a, double eps2, int max_iters, int* n_active) {
    const int p = blockIdx.y;
    if (a.st[p].stop_iter <= a.k) return;
    const int hw = a.h * a.w;
    const int64_t base = (int64_t)p * hw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const int x = i % a.w, y = i / a.w;
        const int64_t g = base + i;
        const float c1 = a.u1[g], c2 = a.u2[g];
        const float u1x = x + 1 < a.w ? a.u1[g + 1] - c1 : 0.0f, u1y = y + 1 < a.h ? a.u1[g + a.w] - c1 : 0.0f;
        const float u2x = x + 1 < a.w ? a.u2[g + 1] - c2 : 0.0f, u2y = y + 1 < a.h ? a.u2[g + a.w] - c2 : 0.0f;
        const float ng1 = 1.0f + a.taut * sqrtf(u1x * u1x + u1y * u1y);
        const float ng2 = 1.0f + a.taut * sqrtf(u2x * u2x + u2y * u2y);
        a.p11[g] = (a.p11[g] + a.taut * u1x) / ng1;
        a.p12[g] = (a.p12[g] + a.taut * u1y) / ng1;
        a.p21[g] = (a.p21[g] + a.taut * u2x) / ng2;
        a.p22[g] = (a.p22[g] + a.taut * u2y) / ng2;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double mean = a.st[p].err / (double)hw;
        a.st[p].err = 0.0;
        a.st[p].iters = a.k + 1;
        if (!(mean > eps2) || a.k + 1 >= max_iters) {
            a.st[p].stop_iter = a.k + 1;
            atomicSub(n_active, 1);
        }
    }
}

__global__ void flow_to_image_kernel(const float* __restrict__ flow, uint8_t* __restrict__ img, int64_t total, float bound) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double v = ((double)flow[i] + (double)bound) * (255.0 / (2.0 * (double)bound));
    img[i] = (uint8_t)fmin(fmax(floor(v + 0.5), 0.0), 255.0);
}

__global__ void homography_warp_kernel(const float* __restrict__ src, float* __restrict__ dst, const double* __restrict__ hinv, int n, int h,
                                       int w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * h * w) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const int64_t p = i / ((int64_t)h * w);
    const double* m = hinv + p * 9;
    const double den = m[6] * x + m[7] * y + m[8];
    const float sx = (float)((m[0] * x + m[1] * y + m[2]) / den), sy = (float)((m[3] * x + m[4] * y + m[5]) / den);
    // oracle.warp_homography: bilinear sample at (x + (sx - x), y + (sy - y)) -- the same fp32 additions
    dst[i] = sample_bilinear(src + p * (int64_t)h * w, h, w, (float)x + (sx - (float)x), (float)y + (sy - (float)y));
}

}  // namespace

struct vq_flow {
    std::mutex mu;
    int device = 0, max_pairs = 0, h = 0, w = 0;
    vq_tvl1_params prm;
    std::vector<Level> levels;
    size_t pyr_floats = 0;                 // floats of one pair's pyramid
    float *pyr0 = nullptr, *pyr1 = nullptr;      // [level][pair][h_l][w_l]
    float* plane[12] = {nullptr};          // i1x, i1y, i1wx, i1wy, grad, rho_c, u1, u2 / p11, p12, p21, p22 at the current level ...
    float* tmp[2] = {nullptr, nullptr};    // flow of the coarser level while it is resized
    uint8_t* frames_dev[2] = {nullptr, nullptr};
    uint8_t* img_dev[2] = {nullptr, nullptr};
    PairState* st = nullptr;
    int* n_active = nullptr;
    int* iters_log = nullptr;              // [levels][warps][pairs]
    double* hinv_dev = nullptr;
};

static void flow_free(vq_flow* f) {
    for (float* p : f->plane)
        if (p) (void)hipFree(p);
    for (float* p : f->tmp)
        if (p) (void)hipFree(p);
    if (f->pyr0) (void)hipFree(f->pyr0);
    if (f->pyr1) (void)hipFree(f->pyr1);
    for (int k = 0; k < 2; ++k) {
        if (f->frames_dev[k]) (void)hipFree(f->frames_dev[k]);
        if (f->img_dev[k]) (void)hipFree(f->img_dev[k]);
    }
    if (f->st) (void)hipFree(f->st);
    if (f->n_active) (void)hipFree(f->n_active);
    if (f->iters_log) (void)hipFree(f->iters_log);
    if (f->hinv_dev) (void)hipFree(f->hinv_dev);
}

extern "C" {

int vq_tvl1_default_params(vq_tvl1_params* p) {
    VQ_REQUIRE(p, "NULL argument");
    p->tau = 0.25f;
    p->lambda = 0.15f;
    p->theta = 0.3f;
    p->epsilon = 0.01f;
    p->scale_step = 0.8f;
    p->nscales = 5;
    p->warps = 5;
    p->iterations = 300;
    p->bound = 20.0f;
    return VQ_OK;
}

int vq_flow_create(int32_t max_pairs, int32_t h, int32_t w, const vq_tvl1_params* params, int32_t device, vq_flow** out) {
    VQ_REQUIRE(out, "out is NULL");
    *out = nullptr;
    VQ_REQUIRE(max_pairs > 0 && h >= 16 && w >= 16 && (int64_t)max_pairs * h * w < (1ll << 31), "bad batch shape %d x %d x %d", max_pairs, h, w);
    vq_tvl1_params prm;
    vq_tvl1_default_params(&prm);
    if (params) prm = *params;
    VQ_REQUIRE(prm.nscales >= 1 && prm.nscales <= 16 && prm.warps >= 1 && prm.warps <= 64 && prm.iterations >= 1 && prm.theta > 0 &&
                   prm.scale_step > 0 && prm.scale_step < 1 && prm.bound > 0 && prm.epsilon >= 0,
               "TV-L1 parameters out of range");
    int ndev = 0;
    VQ_HIP(hipGetDeviceCount(&ndev));
    VQ_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard g(device);
    auto* f = new vq_flow;
    f->device = device;
    f->max_pairs = max_pairs;
    f->h = h;
    f->w = w;
    f->prm = prm;
    // level sizes, finest first: round(previous * scale_step), stop before 16 pixels (oracle.pyramid_sizes)
    size_t off = 0;
    int lh = h, lw = w;
    for (int s = 0; s < prm.nscales; ++s) {
        if (s > 0) {
            const int nh = (int)std::nearbyint((double)lh * (double)prm.scale_step), nw = (int)std::nearbyint((double)lw * (double)prm.scale_step);
            if (nh < 16 || nw < 16) break;
            lh = nh;
            lw = nw;
        }
        f->levels.push_back(Level{lh, lw, off});
        off += (size_t)max_pairs * lh * lw;
    }
    f->pyr_floats = off;
    auto bail = [&](const char* what, hipError_t e) {
        flow_free(f);
        delete f;
        return fail(e == hipErrorOutOfMemory ? VQ_E_NOMEM : VQ_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    const size_t full = (size_t)max_pairs * h * w;
    hipError_t e;
    if ((e = hipMalloc((void**)&f->pyr0, off * sizeof(float))) != hipSuccess) return bail("hipMalloc(pyramid)", e);
    if ((e = hipMalloc((void**)&f->pyr1, off * sizeof(float))) != hipSuccess) return bail("hipMalloc(pyramid)", e);
    for (float*& p : f->plane)
        if ((e = hipMalloc((void**)&p, full * sizeof(float))) != hipSuccess) return bail("hipMalloc(plane)", e);
    for (float*& p : f->tmp)
        if ((e = hipMalloc((void**)&p, full * sizeof(float))) != hipSuccess) return bail("hipMalloc(plane)", e);
    for (int k = 0; k < 2; ++k) {
        if ((e = hipMalloc((void**)&f->frames_dev[k], full)) != hipSuccess) return bail("hipMalloc(frames)", e);
        if ((e = hipMalloc((void**)&f->img_dev[k], full)) != hipSuccess) return bail("hipMalloc(images)", e);
    }
    if ((e = hipMalloc((void**)&f->st, (size_t)max_pairs * sizeof(PairState))) != hipSuccess) return bail("hipMalloc(state)", e);
    if ((e = hipMalloc((void**)&f->n_active, sizeof(int))) != hipSuccess) return bail("hipMalloc(state)", e);
    if ((e = hipMalloc((void**)&f->iters_log, (size_t)f->levels.size() * prm.warps * max_pairs * sizeof(int))) != hipSuccess)
        return bail("hipMalloc(log)", e);
    if ((e = hipMalloc((void**)&f->hinv_dev, (size_t)max_pairs * 9 * sizeof(double))) != hipSuccess) return bail("hipMalloc(homographies)", e);
    *out = f;
    return VQ_OK;
}

int vq_flow_destroy(vq_flow* f) {
    if (!f) return VQ_OK;
    {
        DeviceGuard g(f->device);
        (void)hipDeviceSynchronize();
        flow_free(f);
    }
    delete f;
    return VQ_OK;
}

int vq_flow_levels(vq_flow* f, int32_t* n_levels, int32_t* sizes_hw, int32_t cap) {
    VQ_REQUIRE(f && n_levels, "NULL argument");
    *n_levels = (int)f->levels.size();
    for (int s = 0; s < (int)f->levels.size() && s < cap && sizes_hw; ++s) {
        sizes_hw[2 * s] = f->levels[s].h;
        sizes_hw[2 * s + 1] = f->levels[s].w;
    }
    return VQ_OK;
}

int vq_flow_tvl1(vq_flow* f, const uint8_t* frames0, const uint8_t* frames1, int32_t frames_on_device, int32_t n_pairs,
                 const double* homographies_host, float* u1_host, float* u2_host, uint8_t* flow_x_host, uint8_t* flow_y_host,
                 int32_t* iters_host, void* hip_stream) {
    VQ_REQUIRE(f && frames0 && frames1, "NULL argument");
    VQ_REQUIRE(n_pairs > 0 && n_pairs <= f->max_pairs, "n_pairs %d outside (0,%d]", n_pairs, f->max_pairs);
    std::lock_guard<std::mutex> lk(f->mu);
    DeviceGuard g(f->device);
    hipStream_t st = (hipStream_t)hip_stream;
    const vq_tvl1_params& P = f->prm;
    const int h = f->h, w = f->w;
    const int64_t full = (int64_t)n_pairs * h * w;
    const uint8_t *d0 = frames0, *d1 = frames1;
    if (!frames_on_device) {
        VQ_HIP(hipMemcpyAsync(f->frames_dev[0], frames0, (size_t)full, hipMemcpyHostToDevice, st));
        VQ_HIP(hipMemcpyAsync(f->frames_dev[1], frames1, (size_t)full, hipMemcpyHostToDevice, st));
        d0 = f->frames_dev[0];
        d1 = f->frames_dev[1];
    }
    float *i1x = f->plane[0], *i1y = f->plane[1], *i1wx = f->plane[2], *i1wy = f->plane[3], *grad = f->plane[4], *rho_c = f->plane[5];
    float *u1 = f->plane[6], *u2 = f->plane[7], *p11 = f->plane[8], *p12 = f->plane[9], *p21 = f->plane[10], *p22 = f->plane[11];
    const int nl = (int)f->levels.size();
    // level 0 of the pyramids: the frames as floats (0..255); the second frame optionally through a homography first
    u8_to_float_kernel<<<cdiv(full, 256), 256, 0, st>>>(d0, f->pyr0, full);
    if (homographies_host) {
        std::vector<double> inv((size_t)n_pairs * 9);
        for (int p = 0; p < n_pairs; ++p) {          // 3x3 inverse by cofactors (fp64), as numpy.linalg.inv does to rounding
            const double* m = homographies_host + (size_t)p * 9;
            const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
            VQ_REQUIRE(std::fabs(det) > 1e-300, "homography %d is singular", p);
            double* o = inv.data() + (size_t)p * 9;
            o[0] = (m[4] * m[8] - m[5] * m[7]) / det;
            o[1] = (m[2] * m[7] - m[1] * m[8]) / det;
            o[2] = (m[1] * m[5] - m[2] * m[4]) / det;
            o[3] = (m[5] * m[6] - m[3] * m[8]) / det;
            o[4] = (m[0] * m[8] - m[2] * m[6]) / det;
            o[5] = (m[2] * m[3] - m[0] * m[5]) / det;
            o[6] = (m[3] * m[7] - m[4] * m[6]) / det;
            o[7] = (m[1] * m[6] - m[0] * m[7]) / det;
            o[8] = (m[0] * m[4] - m[1] * m[3]) / det;
        }
        VQ_HIP(hipMemcpyAsync(f->hinv_dev, inv.data(), inv.size() * sizeof(double), hipMemcpyHostToDevice, st));
        VQ_HIP(hipStreamSynchronize(st));           // `inv` leaves scope
        u8_to_float_kernel<<<cdiv(full, 256), 256, 0, st>>>(d1, f->tmp[0], full);
        homography_warp_kernel<<<cdiv(full, 256), 256, 0, st>>>(f->tmp[0], f->pyr1, f->hinv_dev, n_pairs, h, w);
    } else {
        u8_to_float_kernel<<<cdiv(full, 256), 256, 0, st>>>(d1, f->pyr1, full);
    }
    VQ_CHECK_LAUNCH();
    for (int s = 1; s < nl; ++s) {
        const Level &a = f->levels[s - 1], &b = f->levels[s];
        const int64_t tot = (int64_t)n_pairs * b.h * b.w;
        resize_kernel<<<cdiv(tot, 256), 256, 0, st>>>(f->pyr0 + a.off, f->pyr0 + b.off, n_pairs, a.h, a.w, b.h, b.w, 1.0f);
        resize_kernel<<<cdiv(tot, 256), 256, 0, st>>>(f->pyr1 + a.off, f->pyr1 + b.off, n_pairs, a.h, a.w, b.h, b.w, 1.0f);
    }
    VQ_CHECK_LAUNCH();
    {
        const Level& c = f->levels[nl - 1];
        VQ_HIP(hipMemsetAsync(u1, 0, (size_t)n_pairs * c.h * c.w * sizeof(float), st));
        VQ_HIP(hipMemsetAsync(u2, 0, (size_t)n_pairs * c.h * c.w * sizeof(float), st));
    }
    const double eps2 = (double)P.epsilon * (double)P.epsilon;      // oracle: float(float32(epsilon)) ** 2
    for (int s = nl - 1; s >= 0; --s) {
        const Level& L = f->levels[s];
        const int64_t tot = (int64_t)n_pairs * L.h * L.w;
        const float *i0 = f->pyr0 + L.off, *i1 = f->pyr1 + L.off;
        gradient_kernel<<<cdiv(tot, 256), 256, 0, st>>>(i1, i1x, i1y, n_pairs, L.h, L.w);
        for (float* p : {p11, p12, p21, p22}) VQ_HIP(hipMemsetAsync(p, 0, (size_t)tot * sizeof(float), st));
        IterArgs a;
        a.i1wx = i1wx;
        a.i1wy = i1wy;
        a.grad = grad;
        a.rho_c = rho_c;
        a.u1 = u1;
        a.u2 = u2;
        a.p11 = p11;
        a.p12 = p12;
        a.p21 = p21;
        a.p22 = p22;
        a.st = f->st;
        a.n = n_pairs;
        a.h = L.h;
        a.w = L.w;
        a.l_t = (float)((double)P.lambda * (double)P.theta);      // oracle: float32(lam * theta) on the float32 parameters
        a.theta = P.theta;
        a.taut = (float)((double)P.tau / (double)P.theta);
        const dim3 grid((unsigned)std::min(cdiv((int64_t)L.h * L.w, 256), 64), (unsigned)n_pairs);
        for (int wp = 0; wp < P.warps; ++wp) {
            tvl1_warp_kernel<<<cdiv(tot, 256), 256, 0, st>>>(i0, i1, i1x, i1y, u1, u2, i1wx, i1wy, grad, rho_c, f->st, n_pairs, L.h, L.w);
            int live = n_pairs;
            VQ_HIP(hipMemcpyAsync(f->n_active, &live, sizeof(int), hipMemcpyHostToDevice, st));
            VQ_HIP(hipStreamSynchronize(st));
            // converged pairs switch themselves off on the device; the host only looks every `chunk` iterations
            for (int it = 0; it < P.iterations && live > 0;) {
                const int chunk = std::min(P.iterations - it, it < 16 ? 8 : 16);
                for (int k = 0; k < chunk; ++k) {
                    a.k = it + k;
                    tvl1_primal_kernel<<<grid, 256, 0, st>>>(a);
                    tvl1_dual_kernel<<<grid, 256, 0, st>>>(a, eps2, P.iterations, f->n_active);
                }
                VQ_CHECK_LAUNCH();
                it += chunk;
                VQ_HIP(hipMemcpyAsync(&live, f->n_active, sizeof(int), hipMemcpyDeviceToHost, st));
                VQ_HIP(hipStreamSynchronize(st));
            }
            if (iters_host) {
                std::vector<PairState> hs((size_t)n_pairs);
                VQ_HIP(hipMemcpyAsync(hs.data(), f->st, hs.size() * sizeof(PairState), hipMemcpyDeviceToHost, st));
                VQ_HIP(hipStreamSynchronize(st));
                for (int p = 0; p < n_pairs; ++p) iters_host[((size_t)(nl - 1 - s) * P.warps + wp) * n_pairs + p] = hs[p].iters;
            }
        }
        if (s > 0) {          // to the next finer level: bilinear resize, flow values divided by the scale step
            const Level& F = f->levels[s - 1];
            const int64_t ftot = (int64_t)n_pairs * F.h * F.w;
            const float inv = (float)(1.0 / (double)P.scale_step);
            VQ_HIP(hipMemcpyAsync(f->tmp[0], u1, (size_t)tot * sizeof(float), hipMemcpyDeviceToDevice, st));
            VQ_HIP(hipMemcpyAsync(f->tmp[1], u2, (size_t)tot * sizeof(float), hipMemcpyDeviceToDevice, st));
            resize_kernel<<<cdiv(ftot, 256), 256, 0, st>>>(f->tmp[0], u1, n_pairs, L.h, L.w, F.h, F.w, inv);
            resize_kernel<<<cdiv(ftot, 256), 256, 0, st>>>(f->tmp[1], u2, n_pairs, L.h, L.w, F.h, F.w, inv);
            VQ_CHECK_LAUNCH();
        }
    }
    if (u1_host) VQ_HIP(hipMemcpyAsync(u1_host, u1, (size_t)full * sizeof(float), hipMemcpyDeviceToHost, st));
    if (u2_host) VQ_HIP(hipMemcpyAsync(u2_host, u2, (size_t)full * sizeof(float), hipMemcpyDeviceToHost, st));
    if (flow_x_host || flow_y_host) {
        flow_to_image_kernel<<<cdiv(full, 256), 256, 0, st>>>(u1, f->img_dev[0], full, P.bound);
        flow_to_image_kernel<<<cdiv(full, 256), 256, 0, st>>>(u2, f->img_dev[1], full, P.bound);
        VQ_CHECK_LAUNCH();
        if (flow_x_host) VQ_HIP(hipMemcpyAsync(flow_x_host, f->img_dev[0], (size_t)full, hipMemcpyDeviceToHost, st));
        if (flow_y_host) VQ_HIP(hipMemcpyAsync(flow_y_host, f->img_dev[1], (size_t)full, hipMemcpyDeviceToHost, st));
    }
    VQ_HIP(hipStreamSynchronize(st));
    return VQ_OK;
}

}  // extern "C"
