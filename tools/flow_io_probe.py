import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "oracle"))
import numpy as np
from video_query_algorithms_amd.tsn.flow import Tvl1Flow
from test_flow_oracle import _shifted_pair
n = 64
rng = np.random.default_rng(0)
pairs = [_shifted_pair(256, 340, float(rng.uniform(-5, 5)), float(rng.uniform(-3, 3)), seed=k % 8, margin=40) for k in range(n)]
f0, f1 = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
m = Tvl1Flow(n, 256, 340)
H = np.stack([np.array([[1, 0, 2.0], [0, 1, -1.0], [0, 0, 1.0]])] * n)
for label, kw in (("plain none", dict(images=False, fields=False)), ("plain images", dict(images=True, fields=False)), ("plain fields", dict(images=False, fields=True)),
                  ("homog none", dict(homographies=H, images=False, fields=False)), ("homog images", dict(homographies=H, images=True, fields=False))):
    m.flow(f0, f1, **kw)
    t = time.perf_counter()
    for _ in range(3):
        m.flow(f0, f1, **kw)
    print(label, "%.1f ms" % ((time.perf_counter() - t) / 3 * 1e3), "inner %.1f" % m.last_timing()[0])
t = time.perf_counter()
for _ in range(3):
    m.good_features(f0)
print("good_features %.1f ms" % ((time.perf_counter() - t) / 3 * 1e3))
