"""CPU: the feature-file layout (writer = calcSig_wOF.py:116-134, reader = api_load_records.py:41-58) and the
frame-ingest helpers."""
import json
import os

import numpy as np
import pytest

import tsn_oracle as to
from video_query_algorithms_amd.tsn import feature_csv, frames

REF_CSV = ("/root/reference/data/features/stock-video-clips_features/DowntownBrooklynDrive_480p/UCF101_split1/"
           "rgb_global_pool_features.csv")


def test_writer_layout_and_number_format(tmp_path):
    feat = {"rgb": np.array([[0.1, 1.0 / 3.0, 2.5e-7, 77.5], [0.0, 1e22, 3.0, 4.000000000000001]]),
            "warped_optical_flow": np.ones((2, 4))}
    files = feature_csv.write_features(str(tmp_path), "vid", "/frames/vid/", "UCF101_split2", "global_pool",
                                       ["clip_0003", "clip_0012"], feat, {"rgb": "w_rgb.caffemodel",
                                                                          "warped_optical_flow": "w_flow.caffemodel"})
    assert [os.path.relpath(f, tmp_path) for f in files] == ["vid/UCF101_split2/rgb_global_pool_features.csv",
                                                             "vid/UCF101_split2/warped_optical_flow_global_pool_features.csv"]
    raw = open(files[0], "rb").read()
    assert b"\r" not in raw and raw.endswith(b"\n")
    lines = raw.decode().split("\n")
    assert lines[0] == "video =vid, video url =/frames/vid/, CNN stream =rgb, feature blob =global_pool, caffe model =w_rgb.caffemodel"
    assert lines[1] == "3,0.1,0.3333333333333333,2.5e-07,77.5"
    assert lines[2] == "12,0.0,1e+22,3.0,4.000000000000001"
    meta, clips, back = feature_csv.read_features(files[0])
    assert meta == {"video": "vid", "dnn_stream": "rgb", "feature_name": "global_pool", "dnn_weights_file_uri": "w_rgb.caffemodel"}
    assert clips.tolist() == [3, 12] and (back == feat["rgb"]).all()          # exact round trip
    nsplit, streams = feature_csv.read_split_dir(os.path.dirname(files[0]))
    assert nsplit == 2 and set(streams) == {"rgb", "warped_optical_flow"}


REF_CSV_G12 = ("/root/reference/data/features/SHRP2_Forward_clips_features/S06NDS_Sample_120406_1451_00186_Forward/"
               "UCF101_split2/warped_optical_flow_global_pool_features.csv")


@pytest.mark.parametrize("path,number_format", [(REF_CSV, "repr"), (REF_CSV_G12, "g12")])
def test_writer_reproduces_a_shipped_reference_file_byte_for_byte(tmp_path, path, number_format):
    """The reference ships files of both numpy float-str generations (feature_csv docstring); each is reproduced
    byte for byte by the matching number_format."""
    if not os.path.exists(path):
        pytest.skip("reference checkout not present (GPU box)")
    meta, clips, feats = feature_csv.read_features(path)
    raw = open(path, "rb").read()
    header = raw.split(b"\n", 1)[0].decode()
    fields = [h.split("=")[-1] for h in header.split(", ")]
    stream = fields[2]
    files = feature_csv.write_features(str(tmp_path), fields[0], fields[1], "UCF101_splitX", fields[3],
                                       ["clip_%04d" % c for c in clips], {stream: feats}, {stream: fields[4]},
                                       number_format=number_format)
    assert open(files[0], "rb").read() == raw


def test_g12_format_truncates_full_precision_values(tmp_path):
    """Full-precision float64 features under the numpy < 1.14 rule: every field equals '%.12g' (+ '.0' when
    integral), as in the shipped S06NDS files; the default format keeps all digits."""
    rng = np.random.default_rng(5)
    feat = np.abs(rng.standard_normal((3, 16))) * np.array([1e-6, 1, 1e3, 1e15] * 4)
    feat[0, :3] = [0.0, 3.0, 1.4111196446412344]
    names = ["clip_0001", "clip_0002", "clip_0003"]
    f12 = feature_csv.write_features(str(tmp_path / "a"), "v", "/p/", "m1", "global_pool", names, {"rgb": feat}, {"rgb": "w"}, "g12")
    full = feature_csv.write_features(str(tmp_path / "b"), "v", "/p/", "m1", "global_pool", names, {"rgb": feat}, {"rgb": "w"})
    rows12 = [l.split(",")[1:] for l in open(f12[0]).read().split("\n")[1:-1]]
    assert rows12[0][:3] == ["0.0", "3.0", "1.41111964464"]
    for got, vals in zip(rows12, feat):
        for tok, v in zip(got, vals):
            want = "%.12g" % v
            assert tok == (want if ("." in want or "e" in want) else want + ".0")
            assert abs(float(tok) - v) <= 5e-12 * abs(v)
    _, _, back = feature_csv.read_features(full[0])
    assert (back == feat).all()


def test_ticks_and_stacks_match_the_oracle():
    for cnt in (150, 151, 60, 10, 5, 26):
        for T in (25, 7, 3, 2):
            for depth in (1, 5):
                assert frames.frame_ticks(cnt, T, depth) == to.frame_ticks(cnt, T, depth)
    assert frames.flow_stack_indices(148, 150, 5) == to.flow_stack_indices(148, 150, 5)


def test_parse_directory_and_image_io(tmp_path):
    rng = np.random.default_rng(0)
    for clip, n in (("clip_0002", 4), ("clip_0010", 3)):
        d = tmp_path / "video" / clip
        d.mkdir(parents=True)
        for i in range(1, n + 1):
            frames.write_pnm(str(d / ("img_%05d.ppm" % i)), rng.integers(0, 256, (20, 30, 3), dtype=np.uint8))
            frames.write_pnm(str(d / ("flow_x_%05d.pgm" % i)), rng.integers(0, 256, (20, 30), dtype=np.uint8))
            frames.write_pnm(str(d / ("flow_y_%05d.pgm" % i)), rng.integers(0, 256, (20, 30), dtype=np.uint8))
    dirs, rgb, flow = frames.parse_directory(str(tmp_path / "video"))
    assert rgb == {"clip_0002": 4, "clip_0010": 3} and flow == rgb and set(dirs) == set(rgb)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    frames.write_pnm(str(tmp_path / "a.ppm"), img)
    assert (frames.imread(str(tmp_path / "a.ppm"), True) == img).all()        # BGR in, BGR out
    (tmp_path / "video" / "clip_0002" / "flow_y_00004.pgm").unlink()
    with pytest.raises(ValueError):
        frames.parse_directory(str(tmp_path / "video"))


def test_host_resize_equals_the_pixel_loop_oracle():
    """tsn/frames.py (the --host_resize path) against oracle/frames_oracle.py, byte for byte, under both rules: "cv2" (the
    default: OpenCV's 11-bit fixed-point INTER_LINEAR as restated from memory of imgproc/resize.cpp -- what the reference's
    dependency computes at calcSig_wOF.py:94,111) and "exact" (fp64 weights, the rule of rounds 1-2).  Down- and up-scaling,
    colour and grey, odd sizes.  PARITY UNPINNED -- no cv2 and no reference frames here."""
    import frames_oracle as fo
    rng = np.random.default_rng(3)
    for shape in ((360, 480, 3), (240, 320), (256, 340, 3), (97, 131), (480, 854), (1080, 1920, 3), (200, 500)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        small = dict(crop=40) if shape[0] > 400 else {}                # the scalar loops are slow: a corner of the big frames
        assert (frames.crop0(img, **small) == fo.crop0(img, **small)).all()
        assert (frames.crop0(img, rule="exact", **small) == fo.crop0(img, rule="exact", **small)).all()
    # the whole resized frame (not only crop 0): right and bottom borders, where the taps are clamped / the rows clipped
    img = rng.integers(0, 256, (45, 61), dtype=np.uint8)
    rows = img.tolist()
    for (w, h) in ((340, 256), (30, 20), (61, 90)):
        got = frames.resize_bilinear(img, (w, h))
        want = np.array([[fo.fixed_point_resize_pixel(rows, y, x, h, w, None) for x in range(w)] for y in range(h)], dtype=np.uint8)
        assert (got == want).all()
    # how far the two rules are apart: at most one grey level
    img = rng.integers(0, 256, (360, 480), dtype=np.uint8)
    d = np.abs(frames.crop0(img).astype(int) - frames.crop0(img, rule="exact").astype(int))
    assert d.max() == 1 and 0.02 < (d > 0).mean() < 0.3


def test_fixed_point_resize_known_answers():
    """Facts about cv2.resize(INTER_LINEAR, uint8) that do not depend on the remembered source: an exact 2x reduction is the
    rounded mean of each 2 x 2 block, ``(a + b + c + d + 2) >> 2`` (OpenCV itself switches to its INTER_AREA code for this
    case because the two coincide); a constant image stays constant; the weights of every tap sum to 2048."""
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    a = img.astype(int)
    want = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
    assert (frames.resize_bilinear(img, (48, 32)) == want).all()
    for v in (0, 1, 37, 254, 255):
        assert (frames.resize_bilinear(np.full((100, 120), v, np.uint8), (340, 256)) == v).all()
        assert (frames.resize_bilinear(np.full((700, 900), v, np.uint8), (340, 256)) == v).all()
    for n_in, n_out in ((480, 256), (854, 340), (100, 256), (1080, 256), (3, 340)):
        for clamp in (True, False):
            s, w0, w1 = frames._cv2_linear_taps(n_in, n_out, clamp)
            assert ((w0 + w1) == 2048).all() and (w0 >= 0).all() and (w1 >= 0).all()


def test_resize_and_crop0():
    img = np.arange(256 * 340 * 3, dtype=np.uint32).reshape(256, 340, 3).astype(np.uint8)
    assert frames.resize_bilinear(img, (340, 256)) is img                      # already 340x256: untouched
    c = frames.crop0(img)
    assert c.shape == (224, 224, 3) and (c == img[:224, :224]).all()           # crop 0 = top-left, un-mirrored
    flat = np.full((100, 120), 37, dtype=np.uint8)
    assert (frames.resize_bilinear(flat, (340, 256)) == 37).all()
    ramp = np.tile(np.arange(0, 200, 2, dtype=np.uint8)[None, :], (50, 1))     # horizontal ramp stays monotone
    r = frames.resize_bilinear(ramp, (340, 256))
    assert r.shape == (256, 340) and (np.diff(r[0].astype(int)) >= 0).all() and r[0, 0] == 0 and r[0, -1] == 198


def test_caffemodel_round_trip(tmp_path):
    """The schema-less .caffemodel reader against files written by our own encoder (no .caffemodel ships with the
    reference: blob ORDER of the BN layer is unpinned, the container format is what is tested here)."""
    from video_query_algorithms_amd.tsn import bn_inception as bi, caffemodel
    from video_query_algorithms_amd.tsn.net import synthetic_weights
    g = bi.bn_inception(10)
    w = synthetic_weights(g, seed=9)
    path = str(tmp_path / "flow.caffemodel")
    caffemodel.write_caffemodel(path, g, w)
    raw = caffemodel.read_caffemodel(path)
    assert raw["conv1/7x7_s2"]["type"] == "Convolution" and raw["conv1/7x7_s2"]["blobs"][0].shape == (64, 10, 7, 7)
    assert raw["conv1/7x7_s2_bn"]["type"] == "BN" and len(raw["conv1/7x7_s2_bn"]["blobs"]) == 4
    back = caffemodel.weights_from_caffemodel(path, g)
    assert set(back) == set(w)
    for layer, d in w.items():
        for k, a in d.items():
            assert back[layer][k].dtype == np.float32 and (back[layer][k] == a).all(), (layer, k)
    # legacy 4-d shape fields + V1 'layers' container
    legacy = (caffemodel._enc_varint(1 << 3) + caffemodel._enc_varint(2) + caffemodel._enc_varint(2 << 3) + caffemodel._enc_varint(3)
              + caffemodel._enc_varint(3 << 3) + caffemodel._enc_varint(1) + caffemodel._enc_varint(4 << 3) + caffemodel._enc_varint(1)
              + caffemodel._enc_ld(5, np.arange(6, dtype="<f4").tobytes()))
    v1 = caffemodel._enc_ld(2, caffemodel._enc_ld(4, b"ip") + caffemodel._enc_varint(5 << 3) + caffemodel._enc_varint(14)
                            + caffemodel._enc_ld(6, legacy))
    (tmp_path / "v1.caffemodel").write_bytes(v1)
    r = caffemodel.read_caffemodel(str(tmp_path / "v1.caffemodel"))
    assert r["ip"]["type"] == 14 and r["ip"]["blobs"][0].shape == (2, 3, 1, 1) and r["ip"]["blobs"][0].ravel().tolist() == [0, 1, 2, 3, 4, 5]
    bad = dict(w)
    del bad["conv2/3x3_bn"]
    caffemodel.write_caffemodel(str(tmp_path / "bad.caffemodel"), g, bad)
    with pytest.raises(KeyError):
        caffemodel.weights_from_caffemodel(str(tmp_path / "bad.caffemodel"), g)


def test_caffemodel_reader_against_the_protobuf_library(tmp_path):
    """The same reader against bytes produced by google.protobuf's own encoder from a schema of the NetParameter subset
    (field numbers as in BVLC caffe.proto, from memory -- the reference ships neither caffe.proto nor a .caffemodel), with
    the things a real file has and our writer does not produce: fields the reader must skip (bottom / top / phase / a
    nested convolution_param / blob diff), an un-packed repeated float, double_data, the legacy 4-d shape, V1 layers."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    from video_query_algorithms_amd.tsn import caffemodel
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="caffe_subset.proto", package="caffe_subset", syntax="proto2")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, extra in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if "type_name" in extra:
                f.type_name = ".caffe_subset." + extra["type_name"]
            if "packed" in extra:
                f.options.packed = extra["packed"]
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("BlobShape", [("dim", 1, F.TYPE_INT64, REP, {"packed": True})])
    msg("BlobProto", [("shape", 7, F.TYPE_MESSAGE, OPT, {"type_name": "BlobShape"}), ("data", 5, F.TYPE_FLOAT, REP, {"packed": True}),
                      ("diff", 6, F.TYPE_FLOAT, REP, {"packed": True}), ("double_data", 8, F.TYPE_DOUBLE, REP, {"packed": True}),
                      ("num", 1, F.TYPE_INT32, OPT, {}), ("channels", 2, F.TYPE_INT32, OPT, {}), ("height", 3, F.TYPE_INT32, OPT, {}),
                      ("width", 4, F.TYPE_INT32, OPT, {})])
    msg("LooseBlob", [("shape", 7, F.TYPE_MESSAGE, OPT, {"type_name": "BlobShape"}), ("data", 5, F.TYPE_FLOAT, REP, {"packed": False})])
    msg("ConvParam", [("num_output", 1, F.TYPE_UINT32, OPT, {}), ("kernel_size", 4, F.TYPE_UINT32, REP, {})])
    msg("LayerParameter", [("name", 1, F.TYPE_STRING, OPT, {}), ("type", 2, F.TYPE_STRING, OPT, {}), ("bottom", 3, F.TYPE_STRING, REP, {}),
                           ("top", 4, F.TYPE_STRING, REP, {}), ("phase", 10, F.TYPE_INT32, OPT, {}),
                           ("blobs", 7, F.TYPE_MESSAGE, REP, {"type_name": "BlobProto"}),
                           ("convolution_param", 106, F.TYPE_MESSAGE, OPT, {"type_name": "ConvParam"})])
    msg("LooseLayer", [("name", 1, F.TYPE_STRING, OPT, {}), ("type", 2, F.TYPE_STRING, OPT, {}),
                       ("blobs", 7, F.TYPE_MESSAGE, REP, {"type_name": "LooseBlob"})])
    msg("V1LayerParameter", [("name", 4, F.TYPE_STRING, OPT, {}), ("type", 5, F.TYPE_INT32, OPT, {}),
                             ("blobs", 6, F.TYPE_MESSAGE, REP, {"type_name": "BlobProto"})])
    msg("NetParameter", [("name", 1, F.TYPE_STRING, OPT, {}), ("layers", 2, F.TYPE_MESSAGE, REP, {"type_name": "V1LayerParameter"}),
                         ("layer", 100, F.TYPE_MESSAGE, REP, {"type_name": "LayerParameter"})])
    msg("LooseNet", [("layer", 100, F.TYPE_MESSAGE, REP, {"type_name": "LooseLayer"})])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    cls = {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("caffe_subset." + n)) for n in ("NetParameter", "LooseNet")}

    rng = np.random.default_rng(17)
    W = rng.standard_normal((8, 3, 3, 3)).astype(np.float32)
    bn = [rng.standard_normal(8).astype(np.float32) for _ in range(4)]
    net = cls["NetParameter"](name="subset")
    data_layer = net.layer.add(name="data", type="Input")
    data_layer.top.append("data")                                              # no blobs: not reported
    conv = net.layer.add(name="conv1", type="Convolution", phase=1)
    conv.bottom.append("data")
    conv.top.append("conv1")
    conv.convolution_param.num_output = 8
    conv.convolution_param.kernel_size.append(3)
    b = conv.blobs.add()
    b.shape.dim.extend(W.shape)
    b.data.extend(W.ravel().tolist())
    b.diff.extend([0.0] * 5)                                                   # skipped
    b2 = conv.blobs.add()
    b2.shape.dim.append(8)
    b2.double_data.extend(np.arange(8, dtype=np.float64).tolist())             # bias stored as doubles
    bnl = net.layer.add(name="conv1_bn", type="BN")
    for a in bn:
        q = bnl.blobs.add(num=1, channels=8, height=1, width=1)                # legacy 4-d shape
        q.data.extend(a.tolist())
    v1 = net.layers.add(name="old_ip", type=14)
    q = v1.blobs.add(num=1, channels=1, height=2, width=3)
    q.data.extend([0.5, 1.5, 2.5, 3.5, 4.5, 5.5])
    (tmp_path / "lib.caffemodel").write_bytes(net.SerializeToString())
    r = caffemodel.read_caffemodel(str(tmp_path / "lib.caffemodel"))
    assert set(r) == {"conv1", "conv1_bn", "old_ip"}
    assert r["conv1"]["type"] == "Convolution" and (r["conv1"]["blobs"][0] == W).all() and r["conv1"]["blobs"][0].shape == W.shape
    assert r["conv1"]["blobs"][1].tolist() == list(range(8))
    assert all(r["conv1_bn"]["blobs"][i].shape == (1, 8, 1, 1) and (r["conv1_bn"]["blobs"][i].ravel() == bn[i]).all() for i in range(4))
    assert r["old_ip"]["type"] == 14 and r["old_ip"]["blobs"][0].shape == (1, 1, 2, 3)
    # the same weights with the float field written element by element (what an encoder without [packed = true] emits)
    loose = cls["LooseNet"]()
    ll = loose.layer.add(name="conv1", type="Convolution")
    lb = ll.blobs.add()
    lb.shape.dim.extend(W.shape)
    lb.data.extend(W.ravel().tolist())
    raw = loose.SerializeToString()
    assert len(raw) > 5 * W.size                                               # really one tag per element
    (tmp_path / "loose.caffemodel").write_bytes(raw)
    r2 = caffemodel.read_caffemodel(str(tmp_path / "loose.caffemodel"))
    assert (r2["conv1"]["blobs"][0] == W).all()
    # and our own writer is readable by the library's decoder
    from video_query_algorithms_amd.tsn import bn_inception as bi
    from video_query_algorithms_amd.tsn.net import synthetic_weights
    g = bi.parse_prototxt('''name: "t" input: "data" input_dim: 1 input_dim: 3 input_dim: 8 input_dim: 8
      layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { num_output: 32 pad: 1 kernel_size: 3 } }
      layer { name: "c_bn" type: "BN" bottom: "c" top: "c_bn" }
      layer { name: "r" type: "ReLU" bottom: "c_bn" top: "c_bn" }
      layer { name: "gp" type: "Pooling" bottom: "c_bn" top: "gp" pooling_param { pool: AVE kernel_size: 8 stride: 1 } }''')
    w = synthetic_weights(g, seed=1)
    caffemodel.write_caffemodel(str(tmp_path / "ours.caffemodel"), g, w)
    back = cls["NetParameter"]()
    back.ParseFromString((tmp_path / "ours.caffemodel").read_bytes())
    by_name = {l.name: l for l in back.layer}
    assert list(by_name["c"].blobs[0].shape.dim) == [32, 3, 3, 3]
    assert (np.array(by_name["c"].blobs[0].data, dtype=np.float32) == w["c"]["W"].ravel()).all() and len(by_name["c_bn"].blobs) == 4


def test_feature_store_round_trip_and_csv_tree_import(tmp_path):
    """feature_store: the [N,S,E,D] block + ids + presence mask on disk; import of a data/features CSV tree with a
    missing file (one stream of one split of one video) and clips that only some files hold."""
    import video_query_algorithms_amd  # noqa: F401
    from video_query_algorithms_amd import feature_store as fs
    from video_query_algorithms_amd.tsn import feature_csv
    rng = np.random.default_rng(3)
    tree = tmp_path / "features"
    vids = {"video_b": [1, 2, 3, 5], "video_a": [1, 2]}
    vals = {}
    for video, clips in vids.items():
        for split in (1, 2, 3):
            feats = {}
            for st in feature_csv.STREAM_MODES:
                if video == "video_a" and split == 2 and st == "rgb":
                    continue                                               # a missing file
                use = [c for c in clips if not (video == "video_b" and c == 5 and split == 3)]   # a clip missing in one split
                f = np.abs(rng.standard_normal((len(use), 1024))).astype(np.float32).astype(np.float64)
                for c, row in zip(use, f):
                    vals[(video, c, st, split)] = row
                feats[st] = (use, f)
            for st, (use, f) in feats.items():
                feature_csv.write_features(str(tree), video, "/v/", "UCF101_split%d" % split, "global_pool",
                                           ["clip_%04d" % c for c in use], {st: f}, {st: "w"})
    out = fs.store_from_csv_tree(str(tree), str(tmp_path / "store"))
    meta, feats, ids, present = fs.open_store(out)
    assert meta["streams"] == list(feature_csv.STREAM_MODES) and meta["splits"] == [1, 2, 3] and feats.shape == (6, 2, 3, 1024)
    assert ids.tolist() == [1, 2, 3, 4, 5, 6]                              # video_a (2 clips) first, then video_b
    clips = json.load(open(os.path.join(out, "clips.json")))
    assert [(c["video"], c["clip"]) for c in clips] == [("video_a", 1), ("video_a", 2), ("video_b", 1), ("video_b", 2),
                                                        ("video_b", 3), ("video_b", 5)]
    for row, c in enumerate(clips):
        for si, st in enumerate(feature_csv.STREAM_MODES):
            for ei, sp in enumerate((1, 2, 3)):
                key = (c["video"], c["clip"], st, sp)
                assert bool(present[row, si, ei]) == (key in vals)
                if key in vals:
                    assert (feats[row, si, ei] == vals[key].astype(np.float32)).all()
    # plain round trip, dense: no presence file is written
    x = rng.standard_normal((5, 2, 3, 8)).astype(np.float32)
    p2 = fs.save_store(str(tmp_path / "s2"), x, [9, 4, 7, 1, 3], ("rgb", "warped_optical_flow"), (1, 2, 3))
    m2, f2, i2, pr2 = fs.open_store(p2)
    assert (np.asarray(f2) == x).all() and i2.tolist() == [9, 4, 7, 1, 3] and pr2 is None and m2["dim"] == 8
    with pytest.raises(ValueError):
        fs.save_store(str(tmp_path / "s3"), x, [1, 1, 2, 3, 4], ("rgb", "warped_optical_flow"), (1, 2, 3))


REF_TREE = "/root/reference/data/features/stock-video-clips_features"


@pytest.mark.skipif(not os.path.isdir(REF_TREE), reason="the reference checkout (build container only) is not here")
def test_store_from_the_reference_csv_tree_matches_the_golden_subset(tmp_path):
    """The shipped data/features tree of the reference -> binary store (fp64): the 24 golden clips (tests/golden/
    real_subset_x.npy, written by oracle/gen_golden.py from the same files) come back value for value."""
    import video_query_algorithms_amd  # noqa: F401
    from video_query_algorithms_amd import feature_store as fs
    out = fs.store_from_csv_tree(REF_TREE, str(tmp_path / "store"), dtype=np.float64)
    meta, feats, ids, present = fs.open_store(out)
    assert meta["splits"] == [1, 2, 3] and meta["dim"] == 1024 and present is None          # dense: 87 clips x 2 x 3
    clips = json.load(open(os.path.join(out, "clips.json")))
    assert len(clips) == 87 and all(c["video"] == "DowntownBrooklynDrive_480p" for c in clips)
    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_subset.json")))
    gx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_subset_x.npy"))
    row_of = {c["clip"]: i for i, c in enumerate(clips)}
    for k, clip in enumerate(golden["clip_ids"]):
        assert (feats[row_of[clip]] == gx[k]).all()


def test_clip_plan_follows_create_clip():
    """build_wof_clips.py:78-128: full 150-frame clips, a short last clip only if it lasts >= 2 s (30 frames at 15 fps)."""
    assert frames.clip_plan(450) == ([(1, 1, 150), (2, 151, 300), (3, 301, 450)], 0)
    assert frames.clip_plan(479) == ([(1, 1, 150), (2, 151, 300), (3, 301, 450)], 29)          # 29 frames < 2 s: dropped
    assert frames.clip_plan(480) == ([(1, 1, 150), (2, 151, 300), (3, 301, 450), (4, 451, 480)], 0)
    assert frames.clip_plan(100) == ([(1, 1, 100)], 0) and frames.clip_plan(20) == ([], 20)
    assert frames.clip_plan(310, frames_per_clip=100, frames_per_second=5) == ([(1, 1, 100), (2, 101, 200), (3, 201, 300), (4, 301, 310)], 0)


def test_packed_weight_cache_skips_the_loader_and_returns_the_same_blob(tmp_path, monkeypatch):
    """TsnNet(cache_key=...): the second handle of the same network takes its packed device blob (BN folded, GEMM / Winograd layouts,
    biases) and its layer table from the cache and never asks for the weights; another key, other packing options or VQ_WEIGHT_CACHE=0
    pack afresh.  The library call is stubbed (no GPU here): what would have been uploaded is what is compared."""
    import ctypes as C

    from video_query_algorithms_amd.tsn import bn_inception
    from video_query_algorithms_amd.tsn import net as tnet
    g = bn_inception.bn_inception(3)
    seen, loads = {}, []

    def fake_call(name, *a):
        if name != "vq_tsn_create":
            raise tnet._lib.VqError(tnet._lib.VQ_E_INVALID if hasattr(tnet._lib, "VQ_E_INVALID") else -1, "stubbed library: " + name)
        seen["blob"] = np.ctypeslib.as_array(C.cast(a[6], C.POINTER(C.c_float)), shape=(a[7],)).copy()
        seen["layers"], seen["segments"] = bytes(a[2]), bytes(a[4])

    def loader(seed):
        def load():
            loads.append(seed)
            return tnet.synthetic_weights(g, seed)
        return load
    monkeypatch.setattr(tnet, "call", fake_call)
    monkeypatch.setenv("VQ_WEIGHT_CACHE", str(tmp_path / "wc"))
    monkeypatch.setenv("VQ_TUNE_CACHE", "0")
    tnet.TsnNet(g, loader(2), cache_key="synthetic:2")
    first = dict(seen)
    tnet.TsnNet(g, loader(2), cache_key="synthetic:2")
    assert loads == [2] and (seen["blob"] == first["blob"]).all() and seen["layers"] == first["layers"] and seen["segments"] == first["segments"]
    direct = tnet.TsnNet(g, tnet.synthetic_weights(g, 2))                        # no key: packed from the weights, same result
    assert (seen["blob"] == first["blob"]).all() and seen["layers"] == first["layers"]
    assert direct.conv_kp == tnet.TsnNet(g, loader(2), cache_key="synthetic:2").conv_kp and loads == [2]
    tnet.TsnNet(g, loader(3), cache_key="synthetic:3")                           # other weights: other entry
    assert loads == [2, 3] and not (seen["blob"] == first["blob"]).all()
    tnet.TsnNet(g, loader(2), cache_key="synthetic:2", winograd=False)           # other packing options: other entry
    assert loads == [2, 3, 2] and seen["blob"].size != first["blob"].size
    monkeypatch.setenv("VQ_WEIGHT_CACHE", "0")
    tnet.TsnNet(g, loader(2), cache_key="synthetic:2")
    assert loads == [2, 3, 2, 2] and (seen["blob"] == first["blob"]).all()
    # a damaged entry is ignored, not trusted
    monkeypatch.setenv("VQ_WEIGHT_CACHE", str(tmp_path / "wc"))
    for f in (tmp_path / "wc").glob("*.npy"):
        f.write_bytes(f.read_bytes()[:1000])
    tnet.TsnNet(g, loader(2), cache_key="synthetic:2")
    assert loads[-1] == 2 and len(loads) == 5 and (seen["blob"] == first["blob"]).all()
    # ... also when the file keeps its size and only a value changed (the stored checksum no longer matches) ...
    tnet.TsnNet(g, loader(2), cache_key="synthetic:2")                           # entry rewritten by the handle above: a hit again
    assert len(loads) == 5
    for f in (tmp_path / "wc").glob("*.npy"):
        raw = bytearray(f.read_bytes())
        raw[-8] ^= 0x40
        f.write_bytes(bytes(raw))
    tnet.TsnNet(g, loader(2), cache_key="synthetic:2")
    assert len(loads) == 6 and (seen["blob"] == first["blob"]).all()
    # ... and an entry written by OTHER packing code is never found: the key carries a digest of the packing sources
    n_entries = len(list((tmp_path / "wc").glob("*.json")))
    monkeypatch.setattr(tnet, "_PACKER_DIGEST", "as if net.py had been edited")
    tnet.TsnNet(g, loader(2), cache_key="synthetic:2")
    assert len(loads) == 7 and len(list((tmp_path / "wc").glob("*.json"))) == n_entries + 1


def test_native_row_formatter_prints_like_str_of_numpy_float64():
    """The feature rows are formatted by the library (csrc/host/vq_csv.cc: shortest round-trip digits laid out by repr's rule, or
    '%.12g' + '.0').  Against Python's own ``repr`` / the numpy < 1.14 rule on values that exercise every branch: the
    fixed / scientific boundaries (1e-4, 1e16), integral values, denormals, the largest double, signed zeros, inf / nan, 17-digit
    values, fp64 means of 25 fp32 numbers (what the command line writes), wide random exponents."""
    from video_query_algorithms_amd.tsn.feature_csv import NUMBER_FORMATS, format_rows
    rng = np.random.default_rng(0)
    special = [0.0, -0.0, 1.0, 100000.0, 1e16, 1e15, 9999999999999998.0, 1e-4, 9.999e-5, 1e-5, 1.5e-7, 123456789012345678.0, 5e-324,
               2.2250738585072014e-308, 1.7976931348623157e308, float("inf"), float("-inf"), float("nan"), 2.0 ** 53, 0.1, 0.30000000000000004,
               1e22, 1e21, 12345.0, 0.5, 999999999999999.9, 0.0001234, 1e100, 1.2345e-100]
    vals = np.concatenate([rng.random(40_000) * 3, 10.0 ** rng.uniform(-320, 308, 40_000), -rng.random(1_000), special,
                           np.float32(rng.random(20_000) * 5).astype(np.float64),
                           np.float32(rng.random((8_000, 25)) * 4).astype(np.float64).mean(axis=1)])
    vals = vals[:vals.size // 8 * 8].reshape(-1, 8)
    nos = np.arange(7, 7 + vals.shape[0])
    for name, fmt in NUMBER_FORMATS.items():
        want = "".join(str(int(c)) + "," + ",".join(map(fmt, row.tolist())) + "\n" for c, row in zip(nos, vals)).encode()
        assert format_rows(vals, nos, name) == want, name
    assert format_rows(np.zeros((0, 4)), np.zeros(0, dtype=np.int64)) == b""
    with pytest.raises(KeyError):
        format_rows(vals, nos, "g7")


def _reference_rgb_and_flow_values():
    """Feature values of the reference's own shipped CSVs in their lossless (17-digit) number format: read from /root/reference
    where it exists (the build container), and always the committed 24-clip extract of the same files (tests/golden/real_subset_x.npy,
    exact fp64, written by oracle/gen_golden.py)."""
    out = {"committed extract": np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_subset_x.npy"), allow_pickle=False).reshape(-1)}
    root = "/root/reference/data/features/stock-video-clips_features"
    if os.path.isdir(root):
        vals = []
        for dirpath, _dirs, files in sorted(os.walk(root)):
            for f in sorted(files):
                if f.endswith("_global_pool_features.csv"):
                    with open(os.path.join(dirpath, f)) as fh:
                        fh.readline()
                        vals += [float(t) for line in fh for t in line.strip().split(",")[1:]]
        assert len(vals) == 6 * 87 * 1024
        out["reference CSVs"] = np.array(vals)
    return out


def test_reference_feature_values_are_fp64_means_of_25_fp32_blobs():
    """The ONE statement about half A's arithmetic that the reference's own fixtures can make: the
    shipped feature files are outputs of calcSig_wOF.py:82 at the script's default of 25 snippets -- the fp64 mean of 25 fp32
    global_pool blobs.  Then 25 x value is the EXACT fp64 sum of 25 float32 numbers, a short dyadic number (24 bits + log2 25 + the
    exponent spread of the addends: 31 bits in the median, never the 53 of an arbitrary double), the values are post-ReLU averages
    (>= 0), and they are NOT float32 numbers (the mean was not taken in fp32).  SURVEY's "<= 29 bits" is the equal-exponent case.
    The GPU twin (tests/test_tsn_gpu.py::test_t25_features_have_the_arithmetic_of_the_reference_files) holds the product to the same."""
    from _helpers import mean_sum_bits
    for where, v in _reference_rgb_and_flow_values().items():
        assert (v >= 0).all(), where
        pos = v[v > 0]
        need = mean_sum_bits(pos, 25)
        assert np.median(need) <= 33 and (need <= 40).mean() >= 0.99 and need.max() <= 52, (where, np.median(need), (need <= 40).mean(), need.max())
        assert (mean_sum_bits(pos[:20000], 24) >= 48).mean() >= 0.95, where         # ... of 25 addends: for another count nothing short fits
        assert (pos.astype(np.float32).astype(np.float64) == pos).mean() <= 0.05, where    # consensus in fp64, not fp32
