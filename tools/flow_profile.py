"""One warm and one measured batch of 64 frame pairs (340 x 256) through vq_flow_tvl1 for rocprofv3 (kernel trace / PMC passes):
    rocprofv3 --kernel-trace --stats -d gpurun_out/flow_prof --output-format csv -- python3 tools/flow_profile.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from video_query_algorithms_amd.tsn.flow import Tvl1Flow
from test_flow_oracle import _shifted_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rng = np.random.default_rng(0)
pairs = [_shifted_pair(256, 340, float(rng.uniform(-5, 5)), float(rng.uniform(-3, 3)), seed=k % 8, margin=40) for k in range(n)]
f0, f1 = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
m = Tvl1Flow(n, 256, 340)
for _ in range(reps):
    r = m.flow(f0, f1, fields=False, iterations=True)
its = r["iters"]
px = np.array([h * w for h, w in m.levels[::-1]], dtype=np.float64)
print("pixel-iterations per batch %.4g; per level (coarsest first) %s; mean inner iterations per warp %.1f"
      % (float((its.sum(axis=1) * px[:, None]).sum()), (its.sum(axis=(1, 2)) * px).tolist(), its.mean()))
m.close()
