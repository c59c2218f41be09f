"""TV-L1 flow throughput: a batch of 340 x 256 frame pairs (the frame size the TSN pipeline resizes to) on the GPU, and one
pair through the numpy oracle on the host for scale."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from video_query_algorithms_amd.tsn.flow import Tvl1Flow
from test_flow_oracle import _shifted_pair
import tvl1_oracle as tv


def main(n_pairs=64):
    rng = np.random.default_rng(0)
    pairs = [_shifted_pair(256, 340, float(rng.uniform(-5, 5)), float(rng.uniform(-3, 3)), seed=k % 8, margin=40) for k in range(n_pairs)]
    f0, f1 = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    m = Tvl1Flow(n_pairs, 256, 340)
    r = m.flow(f0, f1, iterations=True)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        r = m.flow(f0, f1, iterations=True)
    dt = (time.perf_counter() - t0) / reps
    its = r["iters"]                                             # [levels, warps, pairs], coarsest level first
    px = np.array([h * w for h, w in m.levels[::-1]], dtype=np.float64)
    pixel_iters = float((its.sum(axis=1) * px[:, None]).sum())   # pixel-iterations actually run
    # algorithmic bytes of one pixel-iteration: primal reads rho_c, I1wx, I1wy, grad, u1, u2, p11, p12, p21, p22 and writes
    # u1, u2; dual reads u1, u2, p11..p22 and writes p11..p22 (neighbour reads hit the caches): 22 floats
    gbs = pixel_iters * 22 * 4 / dt / 1e9
    print("%d pairs of 340x256: %.1f ms per batch -> %.0f pairs/s; %.3g pixel-iterations, %.0f GB/s algorithmic (%.1f%% of 8 TB/s); "
          "mean inner iterations per warp %.1f" % (n_pairs, dt * 1e3, n_pairs / dt, pixel_iters, gbs, gbs / 80.0, its.mean()))
    # the warped flow: first pass, corners + RANSAC homography, second pass on the compensated frame
    m.warped(f0, f1)
    t0 = time.perf_counter()
    for _ in range(reps):
        w = m.warped(f0, f1)
    dw = (time.perf_counter() - t0) / reps
    first = m.flow(f0, f1, images=False)
    t0 = time.perf_counter()
    for _ in range(reps):
        H, matches, inliers = m.camera_motion(f0, first["u1"], first["u2"])
    dc = (time.perf_counter() - t0) / reps
    print("warped flow: %.1f ms per batch -> %.0f pairs/s (camera-motion estimate alone %.1f ms: %d corners and %d inliers per pair on average)"
          % (dw * 1e3, n_pairs / dw, dc * 1e3, matches.mean(), inliers.mean()))
    t0 = time.perf_counter()
    tv.tvl1_flow(f0[0], f1[0])
    print("oracle (numpy, 1 thread): %.2f s per pair" % (time.perf_counter() - t0))
    m.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
