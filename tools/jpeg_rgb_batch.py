"""One call of 800 RGB frames (the command line's RGB batch: host entropy decoding), for VQ_JPEG_HOST_STAMPS."""
import io
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from PIL import Image
from video_query_algorithms_amd.tsn.jpeg import JpegDecoder

h, w, nb = 256, 340, int(sys.argv[1]) if len(sys.argv) > 1 else 800
rng = np.random.default_rng(9)
ys, xs = np.mgrid[0:h, 0:w]
base = np.stack([127 + 100 * np.sin(xs / 7.0) + 20 * np.cos(ys / 3.0), 127 + 90 * np.cos(ys / 9.0) + 30 * np.sin(xs / 2.5), 127 + 80 * np.sin((xs + ys) / 11.0)], -1)
blobs = []
for k in range(16):
    buf = io.BytesIO()
    Image.fromarray(np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=95, subsampling=2)
    blobs.append(buf.getvalue())
batch = [blobs[i % 16] for i in range(nb)]
print("mean file %.1f KB" % (sum(map(len, batch)) / nb / 1024))
d = JpegDecoder(nb, h, w)
d.decode_to_device(batch, color=True)
for _ in range(3):
    t0 = time.perf_counter()
    d.decode_to_device(batch, color=True)
    print("%.1f ms" % ((time.perf_counter() - t0) * 1e3))
