"""Does bench.py's end-to-end command-line figure depend on what ran before it in the process?  e2e, then the flow / JPEG / similarity legs, then e2e again."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VQ_DEVICE_POOL_GB", "120")
import bench


def e2e(tag):
    r = bench.bench_e2e_cli(0)
    print(tag, "value %.0f steady %.0f seconds %s" % (r["value"], r["steady_state"]["value"], [round(x, 3) for x in r["seconds_of_the_three_runs"]]), flush=True)


e2e("first")
for leg in sys.argv[1:] or ["flow", "jpeg"]:
    t0 = time.perf_counter()
    if leg == "flow":
        bench.bench_flow(0, True)
    elif leg == "jpeg":
        bench.bench_jpeg(0)
    elif leg == "rounds":
        bench.bench_rounds(0, True)
    elif leg == "sleep":
        time.sleep(20)
    print("leg", leg, "%.1f s" % (time.perf_counter() - t0), flush=True)
    e2e("after " + leg)
