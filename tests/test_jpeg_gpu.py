"""GPU: the JPEG decoder (csrc/vq_jpeg.hip: Huffman decoding on host threads in the library, IDCT / upsampling / colour on the
device) against oracle/jpeg_oracle.py and against libjpeg-turbo itself (through Pillow) -- bit for bit.  This is the ingest
step with a REAL pin: cv2.imread (calcSig_wOF.py:92,105-106) decodes with the same library at the same defaults."""
import io

import numpy as np
import pytest

import jpeg_oracle as jo
from test_jpeg_oracle import encode, picture, pil_bgr

pytestmark = pytest.mark.gpu
try:
    from PIL import Image
except ImportError:
    Image = None
needs_pil = pytest.mark.skipif(Image is None, reason="Pillow not installed (the committed fixtures still run)")


@pytest.fixture(scope="module")
def jpeg(gpu):
    from video_query_algorithms_amd.tsn import jpeg
    return jpeg


@needs_pil
@pytest.mark.parametrize("h,w", [(48, 64), (37, 53), (8, 8), (1, 1), (31, 3), (9, 5), (256, 340)])
def test_batches_of_mixed_layouts_equal_libjpeg(jpeg, h, w):
    files = []
    for k, (sub, q) in enumerate(((2, 95), (1, 80), (0, 60), (2, 30))):
        files.append(encode(picture(h, w, 10 * k + h), quality=q, subsampling=sub))
    files.append(encode(picture(h, w, 77)[:, :, 1], quality=88))                  # a one-component file in the same batch
    files.append(encode(picture(h, w, 5), quality=85, subsampling=2, optimize=True))
    if h * w > 64:
        files.append(encode(picture(h, w, 6), quality=85, subsampling=2, restart_marker_blocks=2))
    dec = jpeg.JpegDecoder(8, h, w)
    got = dec.decode(files, color=True)
    grey = dec.decode(files, color=False)
    assert got.shape == (len(files), h, w, 3) and grey.shape == (len(files), h, w)
    for i, data in enumerate(files):
        assert jpeg.info(data)[:2] == (h, w)
        assert (got[i] == jo.decode(data, color=True)).all(), i
        assert (got[i] == pil_bgr(data)).all(), i                                 # libjpeg-turbo's own pixels
        assert (grey[i] == jo.decode(data, color=False)).all(), i
    # one file at a time gives the same pixels as inside the batch
    assert (dec.decode(files[1:2])[0] == got[1]).all()
    dec.close()


@needs_pil
def test_a_clip_of_video_frames_and_the_device_hand_over(jpeg):
    """96 frames of 340 x 256 (cv2.imwrite defaults: quality 95, 4:2:0) decoded in one call; the decoder's device copy goes
    straight into vq_resize_crop and gives the crops the host path gives."""
    import ctypes as C

    import torch
    from video_query_algorithms_amd import _lib
    from video_query_algorithms_amd.tsn import frames
    rng = np.random.default_rng(2)
    base = picture(256, 340, 4).astype(np.int16)
    files = [encode(np.clip(base + rng.integers(-20, 20, base.shape), 0, 255).astype(np.uint8), quality=95, subsampling=2) for _ in range(96)]
    dec = jpeg.JpegDecoder(96, 256, 340)
    got = dec.decode(files)
    for i in (0, 41, 95):
        assert (got[i] == pil_bgr(files[i])).all()
    dev, (n, h, w) = dec.decode_to_device(files)
    crops = torch.empty((n, 224, 224, 3), dtype=torch.uint8, device="cuda")
    _lib.call("vq_resize_crop", C.c_void_p(dev), 1, n, h, w, 3, 340, 256, 224, 0, C.c_void_p(crops.data_ptr()), 3, 0, 0, None)
    torch.cuda.synchronize()
    want = np.stack([frames.crop0(got[i], (340, 256), 224) for i in (0, 95)])
    assert (crops.cpu().numpy()[[0, 95]] == want).all()
    dec.close()


@needs_pil
def test_what_is_refused(jpeg):
    from video_query_algorithms_amd import VqError
    dec = jpeg.JpegDecoder(2, 32, 32)
    good = encode(picture(32, 32, 1), quality=90)
    with pytest.raises(VqError, match="progressive"):
        dec.decode([encode(picture(32, 32, 2), progressive=True)])
    with pytest.raises(VqError, match="not a JPEG"):
        dec.decode([b"\x89PNG\r\n\x1a\n0000"])
    with pytest.raises(VqError, match="decodes 32x32"):
        dec.decode([good, encode(picture(16, 32, 3), quality=90)])
    with pytest.raises(VqError):
        dec.decode([good[:len(good) // 3]])                                       # cut inside the headers
    cut = dec.decode([good[:len(good) - 40]])                                     # cut inside the scan: the missing blocks decode from zero bits
    assert cut.shape == (1, 32, 32, 3)
    with pytest.raises(VqError, match="outside"):
        dec.decode([good, good, good])
    assert (dec.decode([good])[0] == pil_bgr(good)).all()                          # the handle survives all of it
    dec.close()


@needs_pil
def test_files_by_path_and_the_path_list_entry(jpeg, tmp_path):
    """Paths instead of contents: the library's threads read the files (vq_jpeg_decode_path_list: the paths in one buffer, each closed
    by its NUL) -- the same pixels; a list that holds fewer paths than the call names, an unterminated list, an unreadable file and a
    path with a NUL inside are refused."""
    import ctypes as C
    from video_query_algorithms_amd import VqError, _lib
    blobs = [encode(picture(40, 56, k), quality=85, subsampling=2) for k in range(5)]
    paths = []
    for k, b in enumerate(blobs):
        paths.append(str(tmp_path / ("f%d.jpg" % k)))
        with open(paths[-1], "wb") as f:
            f.write(b)
    dec = jpeg.JpegDecoder(8, 40, 56)
    assert (dec.decode(paths) == dec.decode(blobs)).all()
    assert (dec.decode(paths, color=False) == dec.decode(blobs, color=False)).all()
    with pytest.raises(VqError, match="nope.jpg"):
        dec.decode(paths[:2] + [str(tmp_path / "nope.jpg")])
    with pytest.raises(ValueError, match="NUL"):
        dec.decode([paths[0], paths[1] + "\0x"])
    out = np.empty((5, 40, 56, 3), np.uint8)
    buf = b"".join(p.encode() + b"\0" for p in paths)
    args = lambda b, n: (dec._h, b, len(b), n, 1, 40, 56, out.ctypes.data_as(C.c_void_p), None, None)
    with pytest.raises(VqError, match="holds 5 paths"):
        _lib.call("vq_jpeg_decode_path_list", *args(buf, 6))
    with pytest.raises(VqError, match="NUL of its last path"):
        _lib.call("vq_jpeg_decode_path_list", *args(buf[:-1] + b"x", 5))
    _lib.call("vq_jpeg_decode_path_list", *args(buf, 5))
    assert (out == dec.decode(blobs)).all()
    dec.close()


@needs_pil
def test_cli_on_a_jpeg_frame_tree_device_decode_equals_host_decode(jpeg, tmp_path):
    """calcSig_wOF.py on img_/flow_x_/flow_y_ .jpg files as build_wof_clips.py leaves them: --device_jpeg (library decoder,
    frames never on the host) writes the same CSV bytes as the default path (host libjpeg through Pillow, resize on the GPU)
    and as --host_resize (host libjpeg, numpy resize): the frames are 480 x 360, so cv2's fixed-point resize rule runs on the
    device, on the device behind the library's decoder, and on the host, and must agree to the bit in all three.
    --exact_resize (the exact-weight rule) is a different function of the pixels: its files differ."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_tsn_gpu import _write_protos
    from video_query_algorithms_amd import calcSig_wOF
    from video_query_algorithms_amd.tsn import bn_inception as bi
    root = tmp_path / "frames"
    rng = np.random.default_rng(11)
    for clip, n in (("clip_0001", 8), ("clip_0002", 11), ("clip_0003", 6)):
        d = root / "vid" / clip
        d.mkdir(parents=True)
        for i in range(1, n + 1):
            (d / ("img_%05d.jpg" % i)).write_bytes(encode(picture(360, 480, int(rng.integers(1 << 30))), quality=95, subsampling=2))
            for p in ("flow_x", "flow_y"):
                (d / ("%s_%05d.jpg" % (p, i))).write_bytes(encode(picture(360, 480, int(rng.integers(1 << 30)))[:, :, 0], quality=95))
    protos = _write_protos(bi, tmp_path)
    outs = {}
    for tag, extra in (("host", []), ("device", ["--device_jpeg"]), ("host_resize", ["--host_resize"]), ("exact", ["--exact_resize"])):
        out_dir = tmp_path / ("features_" + tag)
        rc = calcSig_wOF.main([str(root), protos["rgb"], "synthetic:2", protos["flow"], "synthetic:5", "--num_frame_per_video", "3",
                               "--outFeatures_dir", str(out_dir), "--modelname", "UCF101_split1", "--batch_clips", "2"] + extra)
        assert rc == 0
        outs[tag] = {f: (out_dir / "vid" / "UCF101_split1" / f).read_bytes()
                     for f in ("rgb_global_pool_features.csv", "warped_optical_flow_global_pool_features.csv")}
    assert outs["host"] == outs["device"] == outs["host_resize"]
    assert outs["exact"]["rgb_global_pool_features.csv"] != outs["host"]["rgb_global_pool_features.csv"]
    assert outs["host"]["rgb_global_pool_features.csv"].count(b"\n") == 4                # header + three clips


def test_committed_libjpeg_fixtures(jpeg):
    """tests/golden/jpeg (files + the pixels libjpeg-turbo decoded them to): the decoder reproduces them on the device."""
    import glob
    import os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
    for path in sorted(glob.glob(os.path.join(root, "*.jpg"))):
        want = np.load(path[:-4] + ".npy")
        dec = jpeg.JpegDecoder(1, want.shape[0], want.shape[1])
        assert (dec.decode([path], color=want.ndim == 3)[0] == want).all(), path
        dec.close()


@needs_pil
@pytest.mark.parametrize("form", ["1", "0"])
def test_damaged_files_never_take_the_process_down(jpeg, monkeypatch, form):
    """600 random mutations (overwritten bytes, truncation, a 4-byte splice) of four valid files: every one is either decoded
    to SOMETHING of the right shape or refused with VqError; the handle keeps working.  Both entropy decoders: on host threads
    (form 1) and on the device (form 0: a lane per stream must stay inside its stream, its ring and its block whatever the bits say)."""
    from video_query_algorithms_amd import VqError
    monkeypatch.setenv("VQ_JPEG_HOST_HUFFMAN", form)
    rng = np.random.default_rng(0)
    dec = jpeg.JpegDecoder(2, 48, 64)
    seeds = [encode(picture(48, 64, 1), quality=90, subsampling=2), encode(picture(48, 64, 2), quality=70, subsampling=1, restart_marker_blocks=2),
             encode(picture(48, 64, 3)[:, :, 0], quality=85), encode(picture(48, 64, 4), quality=95, subsampling=0, optimize=True)]
    decoded = refused = 0
    for it in range(600):
        base = bytearray(seeds[it % 4])
        if it % 3 == 0:
            for _ in range(int(rng.integers(1, 6))):
                base[int(rng.integers(2, len(base)))] = int(rng.integers(0, 256))
        elif it % 3 == 1:
            base = base[:int(rng.integers(4, len(base)))]
        else:
            p = int(rng.integers(2, len(base) - 8))
            base[p:p + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
        try:
            assert dec.decode([bytes(base)]).shape == (1, 48, 64, 3)
            decoded += 1
        except VqError:
            refused += 1
    assert decoded > 100 and refused > 100
    assert (dec.decode([seeds[0]])[0] == pil_bgr(seeds[0])).all()
    dec.close()


@needs_pil
def test_device_entropy_decoding_equals_the_host_decoder(jpeg, monkeypatch):
    """Huffman decoding on the device (one lane per stream: a frame's scan, or one restart interval of it; streams grouped by
    Huffman table set) against the host decoder of rounds 1-2 (VQ_JPEG_HOST_HUFFMAN=1 when the handle is created) and against
    libjpeg-turbo: the same pixels, bit for bit, on a batch that mixes 4:2:0 / 4:2:2 / 4:4:4 / grey files, per-file optimised
    tables (several table sets in one batch: several waves), restart intervals of 1, 2 and 5 MCU rows, odd sizes; more than 64
    streams of one table set (more than one wave) and a batch of one."""
    h, w = 72, 104
    files = []
    for k, (q, sub) in enumerate([(95, 2), (80, 1), (60, 0), (30, 2), (90, 2)]):
        files.append(encode(picture(h, w, 10 * k + 3), quality=q, subsampling=sub))
    files.append(encode(picture(h, w, 77)[:, :, 1], quality=88))
    files.append(encode(picture(h, w, 5), quality=85, subsampling=2, optimize=True))
    files.append(encode(picture(h, w, 6), quality=75, subsampling=0, optimize=True))
    try:
        for rows in (1, 2, 5):
            files.append(encode(picture(h, w, 20 + rows), quality=85, subsampling=2, restart_marker_rows=rows))
        files.append(encode(picture(h, w, 31), quality=85, subsampling=1, restart_marker_blocks=3))
    except TypeError:
        pass                                      # an older Pillow without restart-marker options
    files += [encode(picture(h, w, 100 + k), quality=92, subsampling=2) for k in range(70)]      # > 64 streams of one set
    out = {}
    for form in ("1", "0"):
        monkeypatch.setenv("VQ_JPEG_HOST_HUFFMAN", form)
        dec = jpeg.JpegDecoder(len(files), h, w)
        out[form] = dec.decode(files)
        one = dec.decode(files[8:9])
        assert (one[0] == out[form][8]).all()
        dec.close()
    assert (out["0"] == out["1"]).all()
    for i in (0, 1, 2, 6, 7, len(files) - 1):
        assert (out["0"][i] == pil_bgr(files[i])).all()
    grey = jpeg.JpegDecoder(4, h, w)
    g = grey.decode([files[5], files[0]], color=False)
    assert (g[0] == np.asarray(Image.open(io.BytesIO(files[5])).convert("L"))).all()
    grey.close()


@needs_pil
def test_two_batches_in_preparation_at_once_give_the_crops_of_one_at_a_time(jpeg):
    """The command line keeps two batches in preparation (CaffeNet.crops_from_jpegs on lanes 0 and 1 from two threads, a decoder and a
    stream per lane): RGB and flow crops equal the ones the same calls give one after the other, and a second call on a busy lane waits."""
    from concurrent.futures import ThreadPoolExecutor

    from video_query_algorithms_amd.tsn import bn_inception
    from video_query_algorithms_amd.tsn.caffe_net import CaffeNet
    rng = np.random.default_rng(11)
    base = picture(256, 340, 9).astype(np.int16)

    def rgb_file():
        return encode(np.clip(base + rng.integers(-25, 25, base.shape), 0, 255).astype(np.uint8), quality=95, subsampling=2)

    def grey_file():
        return encode(np.clip(base[:, :, 1] + rng.integers(-25, 25, base.shape[:2]), 0, 255).astype(np.uint8), quality=95)
    for ch, make, per in ((3, rgb_file, 1), (10, grey_file, 10)):
        net = CaffeNet(bn_inception.bn_inception(ch), "synthetic:3", 0, max_crops=6)
        batches = [[make() for _ in range(6 * per)] for _ in range(4)]
        one_by_one = [net.crops_from_jpegs(b).cpu().numpy() for b in batches]
        with ThreadPoolExecutor(max_workers=3) as pool:
            jobs = [pool.submit(net.crops_from_jpegs, b, lane=k % 2) for k, b in enumerate(batches)]       # three threads, two lanes
            together = [j.result().cpu().numpy() for j in jobs]
        for a, b in zip(one_by_one, together):
            assert a.shape == (6, 224, 224, ch) and (a == b).all()
        net.close()


@needs_pil
def test_closed_ingest_objects_leave_their_decoders_for_the_next(jpeg):
    """tsn/ingest.py keeps the decoders of closed FrameIngest objects (their buffers cost tens of milliseconds to free and to allocate): a
    second object takes them over -- for other frame sizes too, as long as they fit --, gives the same crops as a fresh decoder, the pool
    never holds more than its limit, and drain_decoder_pool() closes what is idle."""
    from video_query_algorithms_amd.tsn import ingest
    ingest.drain_decoder_pool()
    big = [encode(picture(256, 340, k), quality=90, subsampling=2) for k in range(4)]
    small = [encode(picture(120, 160, k), quality=90, subsampling=2) for k in range(4)]
    first = ingest.FrameIngest(3, 0)
    want_big = first.crops_from_jpegs(big).cpu().numpy()
    first.crops_from_jpegs(big, lane=1)
    first.close()
    assert len(ingest._idle_decoders[0]) == 2
    kept = list(ingest._idle_decoders[0])
    second = ingest.FrameIngest(3, 0)
    got_small = second.crops_from_jpegs(small).cpu().numpy()              # smaller frames fit the kept decoder
    assert second._lanes[0]["jpeg"] in kept and len(ingest._idle_decoders[0]) == 1
    assert (second.crops_from_jpegs(big).cpu().numpy() == want_big).all()
    second.close()
    ingest.drain_decoder_pool()
    fresh = ingest.FrameIngest(3, 0)
    assert (fresh.crops_from_jpegs(small).cpu().numpy() == got_small).all()
    fresh.close()
    many = [ingest.FrameIngest(3, 0) for _ in range(ingest._POOL_KEEP + 2)]
    for m in many:
        m.crops_from_jpegs(small[:1])
    for m in many:
        m.close()
    assert len(ingest._idle_decoders[0]) == ingest._POOL_KEEP
    ingest.drain_decoder_pool()
    assert not ingest._idle_decoders


@needs_pil
def test_host_decoder_on_streams_full_of_stuffed_bytes(jpeg, monkeypatch):
    """The host entropy decoder refills its bit buffer eight bytes at a time unless one of them is 0xFF (csrc/host/vq_jpeg_host.h): pure
    noise at quality 100 stuffs an FF 00 every ~250 bytes and uses the longest codes and the largest coefficients there are; random
    sizes, samplings, restart intervals and per-file tables on top.  Pixels of libjpeg-turbo, bit for bit -- and the same from the device
    decoder."""
    rng = np.random.default_rng(77)
    for trial in range(12):
        h, w = int(rng.integers(9, 120)), int(rng.integers(9, 150))
        files = []
        for k in range(6):
            noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            kw = dict(quality=int(rng.choice([100, 98, 90, 50])), subsampling=int(rng.integers(0, 3)), optimize=bool(rng.integers(0, 2)))
            if k % 3 == 2:
                kw["restart_marker_blocks"] = int(rng.integers(1, 6))
            try:
                files.append(encode(noise if k % 2 else picture(h, w, trial * 10 + k), **kw))
            except TypeError:                                         # an older Pillow without restart-marker options
                kw.pop("restart_marker_blocks", None)
                files.append(encode(noise, **kw))
        assert any(b"\xff\x00" in f for f in files)
        out = {}
        for form in ("1", "0"):
            monkeypatch.setenv("VQ_JPEG_HOST_HUFFMAN", form)
            dec = jpeg.JpegDecoder(len(files), h, w)
            out[form] = dec.decode(files)
            dec.close()
        for i, data in enumerate(files):
            assert (out["1"][i] == pil_bgr(data)).all(), (trial, i)
        assert (out["0"] == out["1"]).all()


@needs_pil
def test_frames_that_have_the_size_are_decoded_straight_into_crops(jpeg):
    """build_wof_clips.py writes 340 x 256 frames, so cv2.resize(frame, (340, 256)) is the identity and crop 0 is the top-left 224 x 224:
    ``crops_from_jpegs`` then decodes to the component planes only and crops from there (vq_jpeg_decode(color | 2) + vq_jpeg_crops) -- the
    same bytes as libjpeg's whole frames cropped on the host, colour (4:2:0, 4:2:2, 4:4:4 and grey files in one batch) and flow stacks
    (files in stack order x0, y0, x1, ...); frames of another size keep the resize path (same bytes as decode + resize on the host)."""
    import ctypes as C

    import torch
    from video_query_algorithms_amd import _lib
    from video_query_algorithms_amd.tsn import frames
    from video_query_algorithms_amd.tsn.ingest import FrameIngest
    rng = np.random.default_rng(5)
    base = picture(256, 340, 21).astype(np.int16)

    def noisy(img):
        return np.clip(img + rng.integers(-30, 30, img.shape), 0, 255).astype(np.uint8)
    rgb_files = [encode(noisy(base), quality=int(q), subsampling=int(s)) for q, s in ((95, 2), (80, 1), (90, 0), (60, 2), (95, 2))]
    rgb_files.append(encode(noisy(base[:, :, 0]), quality=90))                       # a grey file read as colour: B = G = R = Y
    ing = FrameIngest(3, 0, "cv2")
    got = ing.crops_from_jpegs(rgb_files).cpu().numpy()
    want = np.stack([pil_bgr(f)[:224, :224] for f in rgb_files])
    assert got.shape == (6, 224, 224, 3) and (got == want).all()
    ing.close()
    flow_files = [encode(noisy(base[:, :, k % 3]), quality=95) for k in range(3 * 10)]           # three stacks, stack order
    ing = FrameIngest(10, 0, "cv2")
    got = ing.crops_from_jpegs(flow_files).cpu().numpy()
    planes = np.stack([np.asarray(Image.open(io.BytesIO(f)).convert("L")) for f in flow_files]).reshape(3, 10, 256, 340)
    assert got.shape == (3, 224, 224, 10) and (got == planes[:, :, :224, :224].transpose(0, 2, 3, 1)).all()
    # another frame size: the resize path, as before
    small = [encode(noisy(picture(120, 160, 3 + k)[:, :, 0]), quality=90) for k in range(10)]
    got = ing.crops_from_jpegs(small).cpu().numpy()
    grey = np.stack([np.asarray(Image.open(io.BytesIO(f)).convert("L")) for f in small])
    assert (got[0] == np.stack([frames.crop0(g, (340, 256)) for g in grey], axis=-1)).all()
    ing.close()
    # the entry refuses what it cannot do
    dec = jpeg.JpegDecoder(8, 256, 340, 0)
    out = torch.empty((1, 224, 224, 10), dtype=torch.uint8, device="cuda")
    with pytest.raises(_lib.VqError):
        _lib.call("vq_jpeg_crops", dec._h, 10, 224, C.c_void_p(out.data_ptr()), None)            # nothing decoded yet
    dec.decode_to_device(rgb_files[:2], color=True)
    with pytest.raises(_lib.VqError):
        _lib.call("vq_jpeg_crops", dec._h, 10, 224, C.c_void_p(out.data_ptr()), None)            # colour frames are not grey planes
    dec.close()
