"""CPU: the host-only translation units of the library (video-query-algorithms_amd/csrc/host/*.cc) under AddressSanitizer +
UndefinedBehaviorSanitizer and under ThreadSanitizer.

vq_jpeg_host.cc parses UNTRUSTED files on worker threads (markers, Huffman tables, restart markers, the host entropy decoder, the
unstuffing pass of the device decoder); vq_corners.cc selects corners on host threads; vq_block_pool.cc is a process-wide pool
shared by every extractor handle; vq_csv.cc writes text into caller-sized buffers.  GPU sanitizers are not available on the
pool, so these units are kept free of HIP and built here by tests/sanitize/Makefile with plain g++; tests/sanitize/san_driver.cc
drives them the way csrc/vq_jpeg.hip / vq_flow.hip / vq_tsn.hip do, with buffers of EXACTLY the sizes the product reserves.
The corpus: the committed JPEG fixtures (4:2:0 / 4:2:2 / 4:4:4 / grey, optimised tables, restart intervals, odd sizes) and
600 seeded mutations of them (overwritten bytes, truncations, 4-byte splices) -- the mutation test of tests/test_jpeg_gpu.py, which
runs un-instrumented on the GPU box."""
import glob
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FINDINGS = ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "SUMMARY: ")


@pytest.fixture(scope="module")
def drivers(tmp_path_factory):
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no g++ / make")
    out = str(tmp_path_factory.mktemp("san"))
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "sanitize"), "OUT=" + out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return {k: os.path.join(out, "san_driver_" + k) for k in ("asan", "tsan")}


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("corpus"))
    seeds = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "jpeg", "*.jpg")))]
    assert len(seeds) >= 5
    for k, s in enumerate(seeds):
        with open(os.path.join(d, "seed_%02d.jpg" % k), "wb") as f:
            f.write(s)
    rng = np.random.default_rng(0)
    for it in range(600):
        base = bytearray(seeds[it % len(seeds)])
        if it % 3 == 0:
            for _ in range(int(rng.integers(1, 6))):
                base[int(rng.integers(2, len(base)))] = int(rng.integers(0, 256))
        elif it % 3 == 1:
            base = base[:int(rng.integers(4, len(base)))]
        else:
            p = int(rng.integers(2, len(base) - 8))
            base[p:p + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
        with open(os.path.join(d, "mut_%03d.jpg" % it), "wb") as f:
            f.write(bytes(base))
    return d


def _run(exe, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=0")
    r = subprocess.run([exe] + list(args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert r.returncode == 0 and not any(tag in r.stdout for tag in FINDINGS), r.stdout[-4000:]
    return r.stdout


def test_every_damaged_file_alone_under_asan_ubsan(drivers, corpus):
    out = _run(drivers["asan"], "single", corpus)
    decoded, refused = (int(x.split()[0]) for x in out.split(":")[1].split(","))
    assert decoded > 100 and refused > 100, out            # both outcomes are exercised, as on the GPU box


def test_threaded_batch_stages_under_asan_and_tsan(drivers, corpus):
    for kind in ("asan", "tsan"):
        out = _run(drivers[kind], "batch", corpus)
        assert "batches decoded" in out and " 0 batches decoded" not in out, out


@pytest.mark.parametrize("mode", ["csv", "corners", "pool"])
def test_formatter_corner_selection_and_block_pool(drivers, mode):
    for kind in ("asan", "tsan"):
        assert mode + ": ok" in _run(drivers[kind], mode)


def test_host_entropy_decoder_gives_the_oracles_coefficients(drivers):
    """No GPU needed for this half of the JPEG path: csrc/host/vq_jpeg_host.cc:decode_scan (the decoder of the command line's RGB batches)
    against oracle/jpeg_oracle.py:decode_coefficients -- which is pinned against libjpeg-turbo -- on every committed fixture, as
    position-weighted sums per component."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import jpeg_oracle as jo
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "jpeg", "*.jpg")))
    assert len(files) >= 5
    try:                                                   # with Pillow here: noise at quality 100 (an FF 00 every ~250 bytes, the longest codes)
        import io
        from PIL import Image
        rng = np.random.default_rng(5)
        for k, (h, w, sub) in enumerate(((40, 56, 2), (33, 47, 1), (24, 24, 0), (64, 80, 2))):
            buf = io.BytesIO()
            kw = {"restart_marker_blocks": 3} if k == 3 else {}
            try:
                Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(buf, "JPEG", quality=100, subsampling=sub, **kw)
            except TypeError:
                continue
            path = os.path.join(os.path.dirname(drivers["asan"]), "noise_%d.jpg" % k)
            with open(path, "wb") as f:
                f.write(buf.getvalue())
            files.append(path)
    except ImportError:
        pass
    for path in files:
        data = open(path, "rb").read()
        try:
            _info, coef = jo.decode_coefficients(data)
        except jo.JpegError:
            continue                                       # a fixture the decoders refuse (progressive, ...)
        out = _run(drivers["asan"], "coef", path)
        lines = [ln for ln in out.splitlines() if ln.startswith("component")]
        assert len(lines) == len(coef), out
        for ln, c in zip(lines, coef):
            flat = c.reshape(-1).astype(np.int64)
            w = np.arange(flat.size, dtype=np.int64) % 65521 + 1
            want = "%d x %d blocks, sums %d %d" % (c.shape[0], c.shape[1], int(flat.sum()), int((flat * w).sum()))
            assert ln.endswith(want), (path, ln, want)
