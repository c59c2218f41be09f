"""cfg 2 with the crops handed over as a HOST buffer (the CLI path): H2D copy of 96 x 150 KB + forward + D2H of the
features, wall clock per step.  The number DESIGN.md quotes beside bench.py's resident-input `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import video_query_algorithms_amd  # noqa: F401
from video_query_algorithms_amd.tsn import bn_inception, net

g = bn_inception.bn_inception(3)
m = net.TsnNet(g, net.synthetic_weights(g, seed=2), max_crops=96)
crops = np.random.default_rng(1).integers(0, 256, (96, 224, 224, 3), dtype=np.uint8)
for _ in range(3):
    m.forward(crops, 3, net.RGB_MEAN, want_per_snippet=False)
ts = []
for _ in range(15):
    t0 = time.perf_counter()
    m.forward(crops, 3, net.RGB_MEAN, want_per_snippet=False)
    ts.append(time.perf_counter() - t0)
med = sorted(ts)[len(ts) // 2]
print("host crops (pageable numpy, %.1f MB per step): %.2f ms per step -> %.0f clips/s" % (crops.nbytes / 1e6, med * 1e3, 32 / med))
m.close()
