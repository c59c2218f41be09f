"""Writes tests/golden/jpeg/*.jpg and the pixels libjpeg-turbo (through Pillow) decodes them to (*.npy): fixtures for the JPEG
oracle and decoder on machines without Pillow.  Run once here: ``python oracle/gen_golden_jpeg.py``."""
import io
import os
import sys

import numpy as np
from PIL import Image, features

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_jpeg_oracle import encode, picture  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "jpeg")
os.makedirs(OUT, exist_ok=True)
cases = {"c420_q95_37x53": (picture(37, 53, 1), dict(quality=95, subsampling=2)),
         "c422_q80_40x61": (picture(40, 61, 2), dict(quality=80, subsampling=1)),
         "c444_q60_24x24": (picture(24, 24, 3), dict(quality=60, subsampling=0)),
         "c420_rst_50x77": (picture(50, 77, 4), dict(quality=85, subsampling=2, restart_marker_blocks=3)),
         "c420_opt_33x70": (picture(33, 70, 5), dict(quality=90, subsampling=2, optimize=True)),
         "grey_q90_45x70": (picture(45, 70, 6)[:, :, 0], dict(quality=90))}
for name, (img, kw) in cases.items():
    data = encode(img, **kw)
    with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
        f.write(data)
    im = Image.open(io.BytesIO(data))
    px = np.asarray(im) if im.mode == "L" else np.asarray(im.convert("RGB"))[:, :, ::-1]
    np.save(os.path.join(OUT, name + ".npy"), np.ascontiguousarray(px))
    print(name, len(data), "bytes", px.shape)
print("Pillow", Image.__version__ if hasattr(Image, "__version__") else "", "libjpeg", features.version("jpg"), "turbo", features.check_feature("libjpeg_turbo"))
