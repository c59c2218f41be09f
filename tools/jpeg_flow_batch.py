"""One call of 8 000 grey flow frames (the command line's flow batch) through the device entropy decoder, for rocprofv3 / VQ_JPEG_STAMPS."""
import io
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from PIL import Image
from video_query_algorithms_amd.tsn.jpeg import JpegDecoder

h, w, nb = 256, 340, int(sys.argv[1]) if len(sys.argv) > 1 else 8000
rng = np.random.default_rng(9)
ys, xs = np.mgrid[0:h, 0:w]
blobs = []
for k in range(16):
    buf = io.BytesIO()
    flow = 128 + 20 * np.sin(xs / (23.0 + k)) * np.cos(ys / 31.0) + rng.normal(0, 1.0, (h, w))
    Image.fromarray(np.clip(flow, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=95)
    blobs.append(buf.getvalue())
batch = [blobs[i % 16] for i in range(nb)]
print("mean file %.1f KB" % (sum(map(len, batch)) / nb / 1024))
if len(sys.argv) > 2 and sys.argv[2] == "paths":          # the files by PATH, as the command line hands them over: the library reads them
    import tempfile
    root = tempfile.mkdtemp(prefix="vq_flow_batch_")
    paths = []
    for i, b in enumerate(batch):
        paths.append(os.path.join(root, "flow_%05d.jpg" % i))
        with open(paths[-1], "wb") as f:
            f.write(b)
    batch = paths
d = JpegDecoder(nb, h, w)
d.decode_to_device(batch, color=False)
for _ in range(3):
    t0 = time.perf_counter()
    d.decode_to_device(batch, color=False)
    print("%.1f ms" % ((time.perf_counter() - t0) * 1e3))
