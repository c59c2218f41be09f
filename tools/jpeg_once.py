import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "oracle"))
import numpy as np
from test_jpeg_oracle import encode, picture
from video_query_algorithms_amd.tsn.jpeg import JpegDecoder
n = 256
rng = np.random.default_rng(0)
base = picture(256, 340, 4).astype(np.int16)
files = [encode(np.clip(base + rng.integers(-25, 25, base.shape), 0, 255).astype(np.uint8), quality=95, subsampling=2) for _ in range(n)]
dec = JpegDecoder(n, 256, 340)
for _ in range(2):
    dec.decode_to_device(files)
dec.close()
