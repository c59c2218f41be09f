// Frame ingest on gfx950: decoded frames -> the crops the TSN forward consumes.
//
// What it replaces (paths relative to the reference checkout): the resize + over-sample part of
//   src/features_GPU_compute/calcSig_wOF.py:94,111   CaffeNet.predict_single_frame / predict_single_flow_stack(...,
//                                                    frame_size=(340, 256))  ->  crop 0 (top-left 224 x 224, un-mirrored)
// i.e. per frame: cv2.resize(frame, (340, 256)) (INTER_LINEAR), then the top-left crop.  Only the crop x crop pixels that
// survive are computed.  Two rules (vq_amd.h):
//   VQ_RESIZE_CV2_FIXED (default of every caller)  OpenCV's own uint8 rule: float sample positions, 11-bit fixed-point weights,
//       int32 horizontal pass, ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 vertical pass -- integer
//       arithmetic, so host (tsn/frames.py:resize_cv2_fixed), device and oracle (oracle/frames_oracle.py) agree bit for bit;
//   VQ_RESIZE_EXACT  the same sampling grid with exact fp64 weights, rounded half to even (tsn/frames.py:resize_exact,
//       operation for operation, contraction off).
// Both restate cv2 from memory of imgproc/resize.cpp: "parity unpinned" for lack of cv2 and of the reference's frames.
#include "vq_common.h"

using namespace vq;

namespace {

struct ResizeArgs {
    const uint8_t* src;   // [n][h][w][c]
    uint8_t* dst;         // [n][crop][crop][dst_c], this plane at channel dst_c0
    int64_t total;        // n * crop * crop
    int h, w, c, rw, rh, crop, dst_c, dst_c0;
};

// cv::resize INTER_LINEAR, 8-bit: tap position and weights of output coordinate d (frames.py:_cv2_linear_taps)
__device__ inline void cv2_taps(int d, int n_in, int n_out, bool clamp_taps, int& s, int& w0, int& w1) {
    const double scale = 1.0 / ((double)n_out / (double)n_in);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f = f - (float)s;
    if (clamp_taps) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n_in - 1) { f = 0.f; s = n_in - 1; }
    }
    w0 = (int)rintf((1.f - f) * 2048.f);          // cvRound: round half to even
    w1 = (int)rintf(f * 2048.f);
}

__global__ void resize_crop_cv2_kernel(ResizeArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    const int x = (int)(i % a.crop), y = (int)((i / a.crop) % a.crop);
    const int64_t n = i / ((int64_t)a.crop * a.crop);
    int sx, a0, a1, sy, b0, b1;
    cv2_taps(x, a.w, a.rw, true, sx, a0, a1);
    cv2_taps(y, a.h, a.rh, false, sy, b0, b1);      // along y the weights stay, the ROWS are clipped
    const int x1 = min(sx + 1, a.w - 1);
    const int y0 = min(max(sy, 0), a.h - 1), y1 = min(max(sy + 1, 0), a.h - 1);
    const uint8_t* img = a.src + n * (int64_t)a.h * a.w * a.c;
    uint8_t* out = a.dst + (n * a.crop * a.crop + (int64_t)y * a.crop + x) * a.dst_c + a.dst_c0;
    for (int ch = 0; ch < a.c; ++ch) {
        const int p00 = img[((int64_t)y0 * a.w + sx) * a.c + ch], p01 = img[((int64_t)y0 * a.w + x1) * a.c + ch];
        const int p10 = img[((int64_t)y1 * a.w + sx) * a.c + ch], p11 = img[((int64_t)y1 * a.w + x1) * a.c + ch];
        const int s0 = p00 * a0 + p01 * a1, s1 = p10 * a0 + p11 * a1;
        const int v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
        out[ch] = (uint8_t)min(max(v, 0), 255);
    }
}

__global__ void resize_crop_kernel(ResizeArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    const int x = (int)(i % a.crop), y = (int)((i / a.crop) % a.crop);
    const int64_t n = i / ((int64_t)a.crop * a.crop);
    // frames.py: ys = clip((arange(h) + 0.5) * ih / h - 0.5, 0, ih - 1)
    double ys = ((double)y + 0.5) * (double)a.h / (double)a.rh - 0.5;
    double xs = ((double)x + 0.5) * (double)a.w / (double)a.rw - 0.5;
    ys = fmin(fmax(ys, 0.0), (double)(a.h - 1));
    xs = fmin(fmax(xs, 0.0), (double)(a.w - 1));
    const int y0 = (int)floor(ys), x0 = (int)floor(xs);
    const int y1 = min(y0 + 1, a.h - 1), x1 = min(x0 + 1, a.w - 1);
    const double wy = ys - (double)y0, wx = xs - (double)x0;
    const uint8_t* img = a.src + n * (int64_t)a.h * a.w * a.c;
    uint8_t* out = a.dst + (n * a.crop * a.crop + (int64_t)y * a.crop + x) * a.dst_c + a.dst_c0;
    for (int ch = 0; ch < a.c; ++ch) {
        const double a00 = img[((int64_t)y0 * a.w + x0) * a.c + ch], a01 = img[((int64_t)y0 * a.w + x1) * a.c + ch];
        const double a10 = img[((int64_t)y1 * a.w + x0) * a.c + ch], a11 = img[((int64_t)y1 * a.w + x1) * a.c + ch];
        // a[y0][:, x0] * (1 - wy) * (1 - wx) + a[y0][:, x1] * (1 - wy) * wx + a[y1][:, x0] * wy * (1 - wx) + a[y1][:, x1] * wy * wx
        double v = a00 * (1.0 - wy) * (1.0 - wx);
        v = v + a01 * (1.0 - wy) * wx;
        v = v + a10 * wy * (1.0 - wx);
        v = v + a11 * wy * wx;
        v = fmin(fmax(rint(v), 0.0), 255.0);      // np.clip(np.rint(out), 0, 255)
        out[ch] = (uint8_t)v;
    }
}

// A frame that already has the size: the crop is a copy under either rule (weights 1 and 0)
__global__ void crop_copy_kernel(ResizeArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    const int x = (int)(i % a.crop), y = (int)((i / a.crop) % a.crop);
    const int64_t n = i / ((int64_t)a.crop * a.crop);
    const uint8_t* px = a.src + ((n * a.h + y) * (int64_t)a.w + x) * a.c;
    uint8_t* out = a.dst + (n * a.crop * a.crop + (int64_t)y * a.crop + x) * a.dst_c + a.dst_c0;
    for (int ch = 0; ch < a.c; ++ch) out[ch] = px[ch];
}

// All C grey planes of a snippet stack in ONE launch (the flow net's ten): a thread owns two horizontally adjacent output pixels, computes
// their taps once, reads them from every plane and writes its 2 C bytes as whole 32-bit words of the interleaved crop.  The per-plane
// form above costs a launch per plane -- ten passes whose byte stores are C bytes apart (0.53 ms each for 800 crops: 5.3 ms of the GPU per
// command-line batch, beside the networks) -- and frames that already have the size went through the fp64 rule as a copy.
// MODE 0: the cv2 fixed-point rule, 1: exact fp64 weights, 2: the frame has the size already (the crop is a copy: weights 1, 0 under either rule).
struct PlanesArgs {
    const uint8_t* src;   // plane p of frame n at src + p * plane_stride + n * h * w
    uint8_t* dst;         // [n][crop][crop][C]
    int64_t plane_stride, total;   // total = n * crop * crop / 2
    int h, w, rw, rh, crop;
};

template <int C, int MODE>
__global__ void resize_crop_planes_kernel(PlanesArgs a) {
    static_assert((2 * C) % 4 == 0, "a thread's two pixels are whole 32-bit words");
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    const int half = a.crop / 2;
    const int xp = (int)(i % half), y = (int)((i / half) % a.crop);
    const int64_t n = i / ((int64_t)half * a.crop);
    const uint8_t* img = a.src + n * (int64_t)a.h * a.w;
    uint32_t words[2 * C / 4];
#pragma unroll
    for (int q = 0; q < 2 * C / 4; ++q) words[q] = 0u;
    int sy = y, b0 = 0, b1 = 0, y0 = y, y1 = y;
    double wy = 0.0;
    if (MODE == 0) {
        cv2_taps(y, a.h, a.rh, false, sy, b0, b1);
        y0 = min(max(sy, 0), a.h - 1);
        y1 = min(max(sy + 1, 0), a.h - 1);
    } else if (MODE == 1) {
        double ys = ((double)y + 0.5) * (double)a.h / (double)a.rh - 0.5;
        ys = fmin(fmax(ys, 0.0), (double)(a.h - 1));
        y0 = (int)floor(ys);
        y1 = min(y0 + 1, a.h - 1);
        wy = ys - (double)y0;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int x = 2 * xp + e;
        int x0 = x, x1 = x, a0 = 0, a1 = 0;
        double wx = 0.0;
        if (MODE == 0) {
            cv2_taps(x, a.w, a.rw, true, x0, a0, a1);
            x1 = min(x0 + 1, a.w - 1);
        } else if (MODE == 1) {
            double xs = ((double)x + 0.5) * (double)a.w / (double)a.rw - 0.5;
            xs = fmin(fmax(xs, 0.0), (double)(a.w - 1));
            x0 = (int)floor(xs);
            x1 = min(x0 + 1, a.w - 1);
            wx = xs - (double)x0;
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const uint8_t* pl = img + ch * a.plane_stride;
            int v;
            if (MODE == 2) {
                v = pl[(int64_t)y * a.w + x];
            } else if (MODE == 0) {
                const int p00 = pl[(int64_t)y0 * a.w + x0], p01 = pl[(int64_t)y0 * a.w + x1];
                const int p10 = pl[(int64_t)y1 * a.w + x0], p11 = pl[(int64_t)y1 * a.w + x1];
                const int s0 = p00 * a0 + p01 * a1, s1 = p10 * a0 + p11 * a1;
                v = min(max((((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2, 0), 255);
            } else {
                const double a00 = pl[(int64_t)y0 * a.w + x0], a01 = pl[(int64_t)y0 * a.w + x1];
                const double a10 = pl[(int64_t)y1 * a.w + x0], a11 = pl[(int64_t)y1 * a.w + x1];
                double t = a00 * (1.0 - wy) * (1.0 - wx);
                t = t + a01 * (1.0 - wy) * wx;
                t = t + a10 * wy * (1.0 - wx);
                t = t + a11 * wy * wx;
                v = (int)fmin(fmax(rint(t), 0.0), 255.0);
            }
            const int pos = e * C + ch;
            words[pos >> 2] |= (uint32_t)v << (8 * (pos & 3));
        }
    }
    uint32_t* out = reinterpret_cast<uint32_t*>(a.dst + ((n * a.crop + y) * (int64_t)a.crop + 2 * xp) * C);
#pragma unroll
    for (int q = 0; q < 2 * C / 4; ++q) out[q] = words[q];
}

}  // namespace

extern "C" int vq_resize_crop_planes(const uint8_t* planes_dev, int32_t n, int32_t h, int32_t w, int32_t c, int64_t plane_stride, int32_t resize_w,
                                     int32_t resize_h, int32_t crop, int32_t rule, uint8_t* crops_dev, int32_t device, void* stream) {
    VQ_REQUIRE(planes_dev && crops_dev, "NULL argument");
    VQ_REQUIRE(n > 0 && h > 0 && w > 0, "frames must be [n][h][w] with positive sizes");
    VQ_REQUIRE(c == 10, "the one-launch form is built for the 10 planes of a flow stack (got %d): use vq_resize_crop per plane", c);
    VQ_REQUIRE(plane_stride >= (int64_t)n * h * w, "plane_stride is smaller than a plane");
    VQ_REQUIRE(resize_w >= crop && resize_h >= crop && crop > 0 && crop % 2 == 0, "crop %d must be even and fit the %dx%d resized frame", crop, resize_w,
               resize_h);
    VQ_REQUIRE(rule == VQ_RESIZE_CV2_FIXED || rule == VQ_RESIZE_EXACT, "unknown resize rule %d", rule);
    VQ_REQUIRE(((uintptr_t)crops_dev & 3u) == 0, "crops_dev must be 4-byte aligned");
    int ndev = 0;
    VQ_HIP(hipGetDeviceCount(&ndev));
    VQ_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard g(device);
    hipStream_t st = (hipStream_t)stream;
    PlanesArgs a;
    a.src = planes_dev;
    a.dst = crops_dev;
    a.plane_stride = plane_stride;
    a.total = (int64_t)n * crop * (crop / 2);
    a.h = h;
    a.w = w;
    a.rw = resize_w;
    a.rh = resize_h;
    a.crop = crop;
    const unsigned blocks = (unsigned)cdiv(a.total, 256);
    if (h == resize_h && w == resize_w)
        resize_crop_planes_kernel<10, 2><<<blocks, 256, 0, st>>>(a);
    else if (rule == VQ_RESIZE_CV2_FIXED)
        resize_crop_planes_kernel<10, 0><<<blocks, 256, 0, st>>>(a);
    else
        resize_crop_planes_kernel<10, 1><<<blocks, 256, 0, st>>>(a);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(VQ_E_HIP, "kernel launch failed: %s", hipGetErrorString(le));
    return VQ_OK;
}

extern "C" int vq_resize_crop(const uint8_t* frames, int32_t frames_on_device, int32_t n, int32_t h, int32_t w, int32_t c,
                              int32_t resize_w, int32_t resize_h, int32_t crop, int32_t rule, uint8_t* crops_dev, int32_t dst_channels,
                              int32_t dst_channel0, int32_t device, void* stream) {
    VQ_REQUIRE(frames && crops_dev, "NULL argument");
    VQ_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0, "frames must be [n][h][w][c] with positive sizes");
    VQ_REQUIRE(resize_w >= crop && resize_h >= crop && crop > 0, "crop %d does not fit the %dx%d resized frame", crop, resize_w, resize_h);
    VQ_REQUIRE(rule == VQ_RESIZE_CV2_FIXED || rule == VQ_RESIZE_EXACT, "unknown resize rule %d", rule);
    VQ_REQUIRE(dst_channel0 >= 0 && dst_channel0 + c <= dst_channels, "channels [%d,%d) outside the %d-channel crop buffer", dst_channel0,
               dst_channel0 + c, dst_channels);
    int ndev = 0;
    VQ_HIP(hipGetDeviceCount(&ndev));
    VQ_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard g(device);
    hipStream_t st = (hipStream_t)stream;
    const uint8_t* src = frames;
    uint8_t* staged = nullptr;
    if (!frames_on_device) {
        const size_t bytes = (size_t)n * h * w * c;
        VQ_HIP(vq::malloc_trim((void**)&staged, bytes));
        hipError_t e = hipMemcpyAsync(staged, frames, bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            (void)hipFree(staged);
            return fail(VQ_E_HIP, "hipMemcpyAsync(frames) failed: %s", hipGetErrorString(e));
        }
        src = staged;
    }
    ResizeArgs a;
    a.src = src;
    a.dst = crops_dev;
    a.total = (int64_t)n * crop * crop;
    a.h = h;
    a.w = w;
    a.c = c;
    a.rw = resize_w;
    a.rh = resize_h;
    a.crop = crop;
    a.dst_c = dst_channels;
    a.dst_c0 = dst_channel0;
    if (h == resize_h && w == resize_w)         // a frame that already has the size is copied by either rule (weights 1, 0)
        crop_copy_kernel<<<cdiv(a.total, 256), 256, 0, st>>>(a);
    else if (rule == VQ_RESIZE_CV2_FIXED)
        resize_crop_cv2_kernel<<<cdiv(a.total, 256), 256, 0, st>>>(a);
    else
        resize_crop_kernel<<<cdiv(a.total, 256), 256, 0, st>>>(a);
    hipError_t le = hipGetLastError();
    if (staged) {
        (void)hipStreamSynchronize(st);
        (void)hipFree(staged);
    }
    if (le != hipSuccess) return fail(VQ_E_HIP, "kernel launch failed: %s", hipGetErrorString(le));
    return VQ_OK;
}

// ---- device plumbing for a host that has no tensor library loaded ----------------------------------------------------------------
// The single-GPU command line needs four things of a device runtime: a buffer for the crops of a batch, a stream per preparation lane,
// a wait, a read-back.  With these the drop-in `python calcSig_wOF.py ...` runs without importing torch (0.8 s of a 2.3 s process); ranks
// of a multi-GPU run, and anything that holds torch tensors already, keep using torch (tsn/devmem.py decides once per process).
extern "C" int vq_dev_malloc(void** ptr, int64_t bytes, int32_t device) {
    VQ_REQUIRE(ptr && bytes > 0, "bad argument");
    DeviceGuard g(device);
    VQ_HIP(vq::malloc_trim(ptr, (size_t)bytes));
    return VQ_OK;
}

extern "C" int vq_dev_free(void* ptr, int32_t device) {
    if (!ptr) return VQ_OK;
    DeviceGuard g(device);
    VQ_HIP(hipFree(ptr));
    return VQ_OK;
}

extern "C" int vq_stream_create(void** stream, int32_t device) {
    VQ_REQUIRE(stream, "NULL argument");
    DeviceGuard g(device);
    hipStream_t st;
    VQ_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void*)st;
    return VQ_OK;
}

extern "C" int vq_stream_create_priority(void** stream, int32_t device, int32_t priority) {
    VQ_REQUIRE(stream, "NULL argument");
    DeviceGuard g(device);
    int least = 0, greatest = 0;                         // numerically: greatest priority <= least priority
    VQ_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t st;
    VQ_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority < 0 ? least : (priority > 0 ? greatest : (least + greatest) / 2)));
    *stream = (void*)st;
    return VQ_OK;
}

extern "C" int vq_stream_destroy(void* stream, int32_t device) {
    if (!stream) return VQ_OK;
    DeviceGuard g(device);
    VQ_HIP(hipStreamDestroy((hipStream_t)stream));
    return VQ_OK;
}

extern "C" int vq_stream_synchronize(void* stream, int32_t device) {
    DeviceGuard g(device);
    VQ_HIP(hipStreamSynchronize((hipStream_t)stream));          // NULL: the default stream
    return VQ_OK;
}

extern "C" int vq_dev_read(void* host, const void* dev, int64_t bytes, int32_t device) {
    VQ_REQUIRE(host && dev && bytes > 0, "bad argument");
    DeviceGuard g(device);
    VQ_HIP(hipMemcpy(host, dev, (size_t)bytes, hipMemcpyDeviceToHost));
    return VQ_OK;
}
