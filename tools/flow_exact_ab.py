"""TV-L1: hardware reciprocal / square root (VQ_FLOW_FAST=1, opt-in) against the IEEE form (default): time per batch of 64 pairs and how far
the two flow fields are apart, plain and warped."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from video_query_algorithms_amd.tsn.flow import Tvl1Flow
from test_flow_oracle import _shifted_pair

n = 64
rng = np.random.default_rng(0)
pairs = [_shifted_pair(256, 340, float(rng.uniform(-5, 5)), float(rng.uniform(-3, 3)), seed=k % 8, margin=40) for k in range(n)]
f0, f1 = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
out = {}
for exact in ("1", "0"):
    os.environ["VQ_FLOW_FAST"] = "0" if exact == "1" else "1"
    m = Tvl1Flow(n, 256, 340)
    m.flow(f0, f1, fields=False)
    t0 = time.perf_counter()
    for _ in range(3):
        r = m.flow(f0, f1, fields=True, iterations=True)
    dt = (time.perf_counter() - t0) / 3
    ms, nl = m.last_timing()
    m.warped(f0, f1)
    t0 = time.perf_counter()
    for _ in range(3):
        w = m.warped(f0, f1, fields=True)
    dw = (time.perf_counter() - t0) / 3
    out[exact] = (r, w)
    print("exact=%s: %.2f ms per batch (%.0f pairs/s), inner loops %.2f ms in %d launches, mean iterations %.2f; warped %.2f ms (%.0f pairs/s)"
          % (exact, dt * 1e3, n / dt, ms, nl, r["iters"].mean(), dw * 1e3, n / dw), flush=True)
    m.close()
(re, we), (rf, wf) = out["1"], out["0"]
print("plain : max |du| %.3g px, |dv| %.3g px; images differ on %.4f %% of the pixels (max %d grey levels); iteration counts differ for %d of %d (level, warp, pair)"
      % (np.abs(re["u1"] - rf["u1"]).max(), np.abs(re["u2"] - rf["u2"]).max(), 100.0 * (re["flow_x"] != rf["flow_x"]).mean(),
         int(np.abs(re["flow_x"].astype(int) - rf["flow_x"].astype(int)).max()), int((re["iters"] != rf["iters"]).sum()), re["iters"].size))
print("warped: max |du| %.3g px, |dv| %.3g px" % (np.abs(we["u1"] - wf["u1"]).max(), np.abs(we["u2"] - wf["u2"]).max()))
