// See vq_corners.h.  Host-only translation unit: no HIP.
#include "vq_corners.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace vq {

// cv::goodFeaturesToTrack's selection on the host: candidates above quality * max, strongest first (equal strengths: the later
// pixel first, as cv's pointer comparison does), accepted unless a kept corner lies closer than min_distance (cell grid).
int select_corners(const float* peaks, int h, int w, float top, int max_corners, float quality, float min_distance, float* out_xy) {
    const float thresh = top * quality;
    std::vector<int> cand;
    for (int i = 0; i < h * w; ++i)
        if (peaks[i] > thresh && peaks[i] != 0.f) cand.push_back(i);
    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return peaks[a] > peaks[b] || (peaks[a] == peaks[b] && a > b); });
    int kept = 0;
    if (min_distance >= 1.f) {
        const int cell = (int)std::nearbyint(min_distance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        std::vector<int> head((size_t)gw * gh, -1), next, pix;      // per cell: chain of kept corners (indices into pix)
        const float md2 = min_distance * min_distance;
        for (int i : cand) {
            const int y = i / w, x = i - y * w;
            const int cx = x / cell, cy = y / cell;
            bool good = true;
            for (int yy = std::max(cy - 1, 0); yy <= std::min(cy + 1, gh - 1) && good; ++yy)
                for (int xx = std::max(cx - 1, 0); xx <= std::min(cx + 1, gw - 1) && good; ++xx)
                    for (int q = head[(size_t)yy * gw + xx]; q >= 0; q = next[q]) {
                        const int j = pix[q];
                        const float dx = (float)(x - j % w), dy = (float)(y - j / w);
                        if (dx * dx + dy * dy < md2) {
                            good = false;
                            break;
                        }
                    }
            if (!good) continue;
            next.push_back(head[(size_t)cy * gw + cx]);
            pix.push_back(i);
            head[(size_t)cy * gw + cx] = (int)pix.size() - 1;
            out_xy[2 * kept] = (float)x;
            out_xy[2 * kept + 1] = (float)y;
            if (++kept == max_corners) break;
        }
    } else {
        for (int i : cand) {
            out_xy[2 * kept] = (float)(i % w);
            out_xy[2 * kept + 1] = (float)(i / w);
            if (++kept == max_corners) break;
        }
    }
    return kept;
}

void select_corners_batch(const float* peaks, const unsigned* top_bits, int n, int h, int w, int max_corners, float quality, float min_distance,
                          float* corners_xy, int* counts) {
    const int workers = std::max(1, std::min<int>({n, 16, (int)std::thread::hardware_concurrency()}));
    auto work = [&](int first) {
        for (int p = first; p < n; p += workers) {
            float t;
            memcpy(&t, &top_bits[p], sizeof t);
            counts[p] = select_corners(peaks + (size_t)p * h * w, h, w, t, max_corners, quality, min_distance, corners_xy + (size_t)p * max_corners * 2);
        }
    };
    std::vector<std::thread> pool;
    for (int k = 1; k < workers; ++k) pool.emplace_back(work, k);
    work(0);
    for (std::thread& th : pool) th.join();
}

}  // namespace vq
