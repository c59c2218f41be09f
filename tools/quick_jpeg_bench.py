"""JPEG decode throughput: batches of 340 x 256 4:2:0 quality-95 frames (what cv2.imwrite leaves) through the library's decoder
(Huffman on host threads, the rest on the GPU, frames stay on the device) next to Pillow's libjpeg-turbo on one host thread."""
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from PIL import Image
from test_jpeg_oracle import encode, picture
from video_query_algorithms_amd.tsn.jpeg import JpegDecoder


def main(n=256):
    rng = np.random.default_rng(0)
    base = picture(256, 340, 4).astype(np.int16)
    files = [encode(np.clip(base + rng.integers(-25, 25, base.shape), 0, 255).astype(np.uint8), quality=95, subsampling=2) for _ in range(n)]
    kb = sum(len(f) for f in files) / n / 1024
    dec = JpegDecoder(n, 256, 340)
    dec.decode_to_device(files)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        dec.decode_to_device(files)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        dec.decode(files)
    dh = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for f in files[:64]:
        np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
    dp = (time.perf_counter() - t0) / 64
    print("%d frames of 340x256 (%.0f KB each): %.1f ms per batch to device memory -> %.0f frames/s (%.0f with the copy back to the host); "
          "Pillow / libjpeg-turbo on one thread: %.0f frames/s; %d host threads for the entropy decoding"
          % (n, kb, dt * 1e3, n / dt, n / dh, 1 / dp, min(16, os.cpu_count() or 1)))
    dec.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
