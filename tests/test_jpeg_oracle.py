"""CPU: oracle/jpeg_oracle.py against libjpeg-turbo itself (through Pillow) -- the one place where half A's ingest has an
independent implementation of the reference's own dependency to be pinned against: cv2.imread (calcSig_wOF.py:92,105-106)
decodes with libjpeg(-turbo) at its defaults, Pillow binds the same library with the same defaults.  Bit for bit."""
import io

import numpy as np
import pytest

import jpeg_oracle as jo

try:
    from PIL import Image
except ImportError:          # the committed fixtures still pin the oracle
    Image = None
needs_pil = pytest.mark.skipif(Image is None, reason="Pillow not installed")


def picture(h, w, seed):
    r = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xs / 7.0 + seed) + 20 * np.cos(ys / 3.0), 127 + 90 * np.cos(ys / 9.0) + 30 * np.sin(xs / 2.5),
                     127 + 80 * np.sin((xs + ys) / 11.0)], -1)
    return np.clip(base + r.normal(0, 12, base.shape), 0, 255).astype(np.uint8)


def encode(a, **kw):
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, "JPEG", **kw)
    return buf.getvalue()


def pil_bgr(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[:, :, ::-1]


@needs_pil
@pytest.mark.parametrize("h,w", [(48, 64), (37, 53), (8, 8), (1, 1), (17, 16), (31, 2), (3, 4), (31, 3), (9, 5), (5, 6)])
@pytest.mark.parametrize("sub", [0, 1, 2])
def test_colour_files_equal_libjpeg_bit_for_bit(h, w, sub):
    for q in (95, 75, 30):
        data = encode(picture(h, w, h * w + q), quality=q, subsampling=sub)
        assert (jo.decode(data) == pil_bgr(data)).all(), (h, w, sub, q)


@needs_pil
def test_grey_files_restart_markers_and_custom_tables():
    g = picture(45, 70, 3)[:, :, 0]
    data = encode(g, quality=90)
    assert (jo.decode(data, color=False) == np.asarray(Image.open(io.BytesIO(data)))).all()
    assert (jo.decode(data, color=True) == np.repeat(jo.decode(data, color=False)[:, :, None], 3, 2)).all()
    a = picture(50, 77, 9)
    for kw in (dict(quality=85, subsampling=2, optimize=True), dict(quality=85, subsampling=2, restart_marker_blocks=3),
               dict(quality=92, subsampling=1, restart_marker_rows=1), dict(quality=60, subsampling=0, restart_marker_blocks=1)):
        data = encode(a, **kw)
        if "restart_marker_blocks" in kw or "restart_marker_rows" in kw:
            assert b"\xff\xdd" in data                                           # the file really carries a restart interval
        assert (jo.decode(data) == pil_bgr(data)).all(), kw
    # a grey read of a colour file is its Y plane (libjpeg's JCS_GRAYSCALE output, what cv2.IMREAD_GRAYSCALE returns)
    data = encode(a, quality=90, subsampling=2)
    im = Image.open(io.BytesIO(data))
    im.draft("L", im.size)                                                       # asks libjpeg for grayscale output
    assert (jo.decode(data, color=False) == np.asarray(im)).all()


@needs_pil
def test_the_video_frame_size_and_what_is_refused():
    data = encode(picture(256, 340, 1), quality=95, subsampling=2)               # cv2.imwrite's defaults: quality 95, 4:2:0
    assert (jo.decode(data) == pil_bgr(data)).all()
    with pytest.raises(jo.JpegError):
        jo.decode(encode(picture(16, 16, 2), progressive=True))
    with pytest.raises(jo.JpegError):
        jo.decode(b"\x89PNG....")


def test_committed_fixtures_decoded_by_libjpeg_turbo():
    """tests/golden/jpeg: files and the pixels Pillow's libjpeg-turbo 6.2 decoded them to (oracle/gen_golden_jpeg.py) -- the pin
    also holds on a machine without Pillow."""
    import glob
    import os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
    names = sorted(glob.glob(os.path.join(root, "*.jpg")))
    assert len(names) == 6
    for path in names:
        want = np.load(path[:-4] + ".npy")
        with open(path, "rb") as f:
            data = f.read()
        assert (jo.decode(data, color=want.ndim == 3) == want).all(), path


def test_library_header_parser_without_a_gpu():
    """vq_jpeg_info is host code: frame header of the committed fixtures, and the refusals, through the C ABI on the CPU."""
    import glob
    import os
    import video_query_algorithms_amd as vqa
    from video_query_algorithms_amd.tsn import jpeg
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
    for path in sorted(glob.glob(os.path.join(root, "*.jpg"))):
        want = np.load(path[:-4] + ".npy")
        with open(path, "rb") as f:
            data = f.read()
        assert jpeg.info(data) == (want.shape[0], want.shape[1], 3 if want.ndim == 3 else 1), path
        with pytest.raises(vqa.VqError):
            jpeg.info(data[:20])
    with pytest.raises(vqa.VqError, match="not a JPEG"):
        jpeg.info(b"GIF89a" + b"\0" * 32)
    if Image is not None:
        with pytest.raises(vqa.VqError, match="progressive"):
            jpeg.info(encode(picture(16, 16, 2), progressive=True))
