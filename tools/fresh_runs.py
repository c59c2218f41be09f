"""Fresh `python calcSig_wOF.py` processes on the bench's 256-clip tree, timed from outside, with the child's own phase stamps: six runs
with GAP seconds of idle GPU before each (default 3; 0 = back to back: a run's hipMalloc then waits for the driver to reclaim what the
previous process freed -- 1.1-2.0 s instead of 0.95-1.05).  python tools/fresh_runs.py [GAP]"""
import io
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image

root = tempfile.mkdtemp(prefix="vq_fresh_")
rng = np.random.default_rng(21)
ys, xs = np.mgrid[0:256, 0:340]
rgb, grey = [], []
for k in range(8):
    base = np.stack([127 + 90 * np.sin(xs / (7.0 + k) + k) + 20 * np.cos(ys / 3.0), 127 + 80 * np.cos(ys / 9.0) + 30 * np.sin(xs / 2.5), 127 + 70 * np.sin((xs + ys) / 11.0)], -1)
    b = io.BytesIO(); Image.fromarray(np.clip(base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)).save(b, "JPEG", quality=95, subsampling=2); rgb.append(b.getvalue())
    flow = 128 + 20 * np.sin(xs / (23.0 + k)) * np.cos(ys / 31.0) + rng.normal(0, 1.0, (256, 340))
    b = io.BytesIO(); Image.fromarray(np.clip(flow, 0, 255).astype(np.uint8)).save(b, "JPEG", quality=95); grey.append(b.getvalue())
for c in range(256):
    d = os.path.join(root, "frames", "video", "clip_%04d" % (c + 1)); os.makedirs(d)
    for i in range(1, 31):
        for name, blob in (("img", rgb[(c + i) % 8]), ("flow_x", grey[(c + i) % 8]), ("flow_y", grey[(c + 3 * i) % 8])):
            open(os.path.join(d, "%s_%05d.jpg" % (name, i)), "wb").write(blob)
from video_query_algorithms_amd.tsn import bn_inception
for name, ch in (("rgb", 3), ("flow", 10)):
    open(os.path.join(root, name + ".prototxt"), "w").write(bn_inception.to_prototxt(bn_inception.bn_inception(ch)))
cli = os.path.join(ROOT, "video-query-algorithms_amd", "calcSig_wOF.py")
base = [os.path.join(root, "frames"), os.path.join(root, "rgb.prototxt"), "synthetic:2", os.path.join(root, "flow.prototxt"), "synthetic:5",
        "--modelname", "UCF101_split1", "--num_worker", "16", "--gpus", "0", "--device_jpeg"]
env = dict(os.environ, VQ_CLI_TRACE="1")
GAP = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
for rep in range(6):
    time.sleep(GAP)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, cli] + base + ["--outFeatures_dir", os.path.join(root, "out%d" % rep)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    stamps = [l for l in r.stderr.decode().splitlines() if l.startswith("trace:") and ("ready" in l or "last batch" in l or "written" in l or "parsed" in l)]
    print("run %d: %.3f s   %s" % (rep, dt, " | ".join(s[7:] for s in stamps)), flush=True)
import shutil
shutil.rmtree(root, ignore_errors=True)
