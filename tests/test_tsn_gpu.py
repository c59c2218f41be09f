"""GPU: the HIP TSN forward (through the C ABI) against the CPU oracle (oracle/tsn_oracle.py, fp64).

Tolerance (fp32 network, stated once): the convolutions are exact-fp32 FMA chains on the matrix cores
(v_mfma_f32_32x32x2_f32) with BN folded into the weights; against an fp64 evaluation of the un-folded
layer list the expected difference per layer is a few 1e-7 relative to the accumulated magnitude.
  * single conv / pool layers:  |d| <= 2e-5 * max|y|   (K up to 2304, fp32 accumulation)
  * whole network features:     |d| <= 2e-4 * max|y|   (69 layers deep)
  * pooling: bit-exact vs the loop-level fp32 restatement (same operation order)
  * consensus: bit-exact fp64 mean of the device's own fp32 per-snippet blobs.
Parity of the network arithmetic against Caffe itself is UNPINNED (oracle header)."""
import os

import numpy as np
import pytest

import tsn_oracle as to

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tsn(gpu):
    import video_query_algorithms_amd  # noqa: F401
    from video_query_algorithms_amd.tsn import bn_inception, net
    return bn_inception, net


def _mini(bi, c, h, w, cout, k, s, p, pool=None):
    """data -> conv+bn+relu [-> pool] -> global average pool (feature)."""
    g = bi.Graph("mini", "data", (c, h, w))
    g.layers.append(bi.Layer("c", "Convolution", ["data"], ["c"], cout, k, s, p))
    g.layers.append(bi.Layer("c_bn", "BN", ["c"], ["c_bn"]))
    g.layers.append(bi.Layer("c_relu", "ReLU", ["c_bn"], ["c_bn"]))
    top = "c_bn"
    ho, wo = to.conv_out(h, k, s, p), to.conv_out(w, k, s, p)
    if pool:
        kind, pk, ps_, pp = pool
        g.layers.append(bi.Layer("p", "Pooling", [top], ["p"], kernel=pk, stride=ps_, pad=pp, pool=kind))
        top = "p"
        ho, wo = to.pool_out(ho, pk, ps_, pp), to.pool_out(wo, pk, ps_, pp)
    assert ho == wo
    g.layers.append(bi.Layer("gp", "Pooling", [top], ["gp"], kernel=ho, stride=1, pad=0, pool="AVE"))
    return g


def _nchw(a):
    return np.ascontiguousarray(a.transpose(0, 3, 1, 2))


CONV_CASES = [
    # (cin, h, cout, k, s, p, n_crops)        what it stands for
    (64, 14, 64, 1, 1, 0, 3),                 # 1x1 reduce
    (192, 28, 96, 1, 1, 0, 2),                # 1x1, N = 96 (128x96 tile)
    (64, 28, 96, 3, 1, 1, 2),                 # 3x3 s1 p1
    (128, 28, 160, 3, 2, 1, 2),               # 3x3 s2 (3c)
    (96, 14, 128, 3, 1, 1, 3),                # 14x14
    (160, 7, 224, 3, 1, 1, 5),                # 7x7, M = 245 (ragged M tile)
    (1056, 7, 352, 1, 1, 0, 2),               # K = 1056, N = 352 (ragged N tile)
    (3, 32, 64, 7, 2, 3, 2),                  # stem, RGB (small-Cin path, K = 147 -> 224 padded)
    (10, 32, 64, 7, 2, 3, 2),                 # stem, flow stack
    (32, 9, 32, 3, 1, 1, 1),                  # tiny: single partial tile
    (1024, 7, 128, 1, 1, 0, 3),               # 5b/pool_proj: 7x7 map, K = 1024 -> two K slices + combine pass
    (256, 14, 256, 3, 2, 1, 2),               # 4e/double_3x3_2: 7x7 output, K = 2304 -> four K slices, slices start mid-tap
]


@pytest.mark.parametrize("cin,h,cout,k,s,p,n", CONV_CASES)
@pytest.mark.parametrize("tile", [None, "128x128", "128x96", "128x64", "128x32", "64x64", "64x128",
                                  "128x128x16", "128x96x16", "128x64x16", "128x32x16", "64x64x16", "64x128x16",
                                  "128x128x32x1", "128x96x32x1", "128x64x32x1", "128x32x32x1", "64x64x32x1", "64x128x32x1",
                                  "128x128x16x1", "128x96x16x1", "128x64x16x1", "128x32x16x1", "64x64x16x1", "64x128x16x1",
                                  "32x128", "32x128x16", "32x128x32x1", "32x128x16x1"])
def test_conv_bn_relu_layer(tsn, monkeypatch, cin, h, cout, k, s, p, n, tile):
    bi, net = tsn
    if tile:
        monkeypatch.setenv("VQ_TSN_TILE", tile)
    else:
        monkeypatch.delenv("VQ_TSN_TILE", raising=False)
    g = _mini(bi, cin, h, h, cout, k, s, p)
    w = net.synthetic_weights(g, seed=cin * 7 + k)
    rng = np.random.default_rng(cin + h)
    crops = rng.integers(0, 256, (n, h, h, cin), dtype=np.uint8)
    mean = np.linspace(100.0, 130.0, cin).astype(np.float32)
    m = net.TsnNet(g, w, max_crops=n, feature_blob="gp")
    feat, ps = m.forward(crops, 1, mean)
    got = _nchw(m.read_blob("c_bn", n))
    want = to.forward(g.layers, "data", w, to.preprocess(crops, mean), keep=("c_bn", "gp"))
    tol = 2e-5 * np.abs(want["c_bn"]).max()
    assert got.shape == want["c_bn"].shape
    assert np.abs(got - want["c_bn"]).max() <= tol
    assert (got >= 0).all()
    assert np.abs(ps - want["gp"].reshape(n, -1)).max() <= tol
    m.close()


def test_maps_wider_than_the_reciprocal_range_run_the_plain_kernel(tsn, monkeypatch):
    """The pipelined kernel decodes a workgroup's pixels with 22-bit reciprocals (exact for output maps up to ~1 980 pixels wide, a sufficient
    condition checked on the host: csrc/vq_tsn.hip:pixel_walk_ok); a 2 048-pixel-wide map asked for on a pipelined tiling runs the plain
    kernel of the same tiling instead.  A 1x1 convolution + BN + ReLU is a per-pixel function: the oracle is run on the first and the last
    two rows of the map; all tilings give the same bits."""
    bi, net = tsn
    h = 2048
    g = _mini(bi, 8, h, h, 32, 1, 1, 0)
    w = net.synthetic_weights(g, seed=5)
    crops = np.random.default_rng(9).integers(0, 256, (1, h, h, 8), dtype=np.uint8)
    mean = np.linspace(100.0, 130.0, 8).astype(np.float32)
    ref = None
    for tile in ("64x64x16", "64x64x16x1", "128x64x32x1"):
        monkeypatch.setenv("VQ_TSN_TILE", tile)
        m = net.TsnNet(g, w, max_crops=1, feature_blob="gp")
        m.forward(crops, 1, mean)
        got = _nchw(m.read_blob("c_bn", 1))
        m.close()
        if ref is None:
            ref = got
            for rows in (slice(0, 2), slice(h - 2, h)):
                want = to.forward(g.layers[:3], "data", w, to.preprocess(crops[:, rows], mean), keep=("c_bn",))["c_bn"]
                assert np.abs(got[:, :, rows] - want).max() <= 2e-5 * np.abs(want).max()
        else:
            assert (got == ref).all()


def _pooled_mini(bi, c, h, k_mid, n_out):
    """data -> 1x1 conv (k_mid channels) -> 3x3/2 max pool -> 1x1 conv (n_out) -> global average pool: the shape of
    pool1 -> conv2/3x3_reduce and pool2 -> inception_3a's sibling 1x1 group (the pool disappears into the GEMM behind it)."""
    g = bi.Graph("pooled", "data", (c, h, h))
    g.layers.append(bi.Layer("c", "Convolution", ["data"], ["c"], k_mid, 1, 1, 0))
    g.layers.append(bi.Layer("c_bn", "BN", ["c"], ["c_bn"]))
    g.layers.append(bi.Layer("c_relu", "ReLU", ["c_bn"], ["c_bn"]))
    g.layers.append(bi.Layer("p", "Pooling", ["c_bn"], ["p"], kernel=3, stride=2, pad=0, pool="MAX"))
    g.layers.append(bi.Layer("d", "Convolution", ["p"], ["d"], n_out, 1, 1, 0))
    g.layers.append(bi.Layer("d_bn", "BN", ["d"], ["d_bn"]))
    g.layers.append(bi.Layer("d_relu", "ReLU", ["d_bn"], ["d_bn"]))
    ho = to.pool_out(h, 3, 2, 0)
    g.layers.append(bi.Layer("gp", "Pooling", ["d_bn"], ["gp"], kernel=ho, stride=1, pad=0, pool="AVE"))
    return g


@pytest.mark.parametrize("h,k_mid,n_out,n", [(28, 64, 64, 3),        # pool1 -> conv2/3x3_reduce in small
                                             (17, 192, 224, 5),      # pool2 -> the 224-column group of inception_3a; odd size: clipped windows, ragged M
                                             (12, 32, 96, 2)])
def test_pooled_input_gemm_every_tiling(tsn, h, k_mid, n_out, n):
    """A 3x3 max pool read by one 1x1 GEMM lives in that GEMM (bn_inception._fuse).  Every kernel that can run it -- the pooled
    loader of conv_igemm_kernel under each of its tilings, and the two-phase pool_gemm_kernel -- gives the bits of the un-fused
    pair of layers (max is exact, the k order of the GEMM is the same) and agrees with the fp64 oracle."""
    bi, net = tsn
    g = _pooled_mini(bi, 32, h, k_mid, n_out)
    w = net.synthetic_weights(g, seed=h + n_out)
    crops = np.random.default_rng(h).integers(0, 256, (n, h, h, 32), dtype=np.uint8)
    mean = np.linspace(100.0, 130.0, 32).astype(np.float32)
    want = to.forward(g.layers, "data", w, to.preprocess(crops, mean), keep=("d_bn",))["d_bn"]
    plain = net.TsnNet(g, w, max_crops=n, feature_blob="gp", fuse=False)
    plain.forward(crops, 1, mean)
    base = plain.read_blob("d_bn", n)
    plain.close()
    assert np.abs(_nchw(base) - want).max() <= 2e-5 * np.abs(want).max()
    m = net.TsnNet(g, w, max_crops=n, feature_blob="gp")
    li = [i for i, o in enumerate(m.plan.ops) if o.pre_pool]
    assert len(li) == 1 and "p" not in m.plan.blob_loc
    ran = 0
    for tile in [(128, 128, 16, 0), (128, 96, 16, 0), (128, 64, 16, 0), (64, 128, 32, 0), (64, 128, 16, 0), (64, 64, 32, 0), (64, 64, 16, 0),
                 (128, 32, 16, 0), (64, 256, 16, 0), (64, 256, 8, 3), (64, 64, 8, 3), (128, 64, 8, 3)]:
        if tile[3] == 3 and tile[0] * (k_mid + 4) * 4 > 64 * 1024:
            continue                                                    # the pooled image of a tile must fit the LDS budget
        m.forward(crops, 1, mean)                                       # tuned table for this batch size exists now
        tiles = m.layer_tiles(n).copy()
        tiles[li[0]] = tile
        m.set_layer_tiles(n, tiles)
        m.forward(crops, 1, mean)
        assert (m.layer_tiles(n)[li[0]] == tile).all()
        got = m.read_blob("d_bn", n)
        assert (got == base).all(), tile
        ran += 1
    assert ran >= 11
    m.close()


def test_k_split_is_a_property_of_the_layer_not_of_the_batch(tsn, monkeypatch):
    """The 7x7-map layers with long K run as K slices + a combine pass (a different summation order than one chain, so
    it must apply to a layer at EVERY batch size).  Split and unsplit agree to rounding; with the split on, a crop's
    output is the same bits alone, in a batch, and in a batch cut into sub-batches on two streams."""
    bi, net = tsn
    g = _mini(bi, 256, 14, 14, 256, 3, 2, 1)
    w = net.synthetic_weights(g, seed=4)
    crops = np.random.default_rng(6).integers(0, 256, (9, 14, 14, 256), dtype=np.uint8)
    mean = np.full(256, 120.0, np.float32)
    outs = {}
    for sk in ("0", "1"):
        monkeypatch.setenv("VQ_TSN_SPLITK", sk)
        m = net.TsnNet(g, w, max_crops=9, feature_blob="gp")
        m.forward(crops, 1, mean)
        outs[sk] = m.read_blob("c_bn", 9)
        if sk == "1":
            for lo, hi in ((0, 1), (3, 8), (8, 9)):
                m.forward(crops[lo:hi], 1, mean)
                assert (m.read_blob("c_bn", hi - lo) == outs[sk][lo:hi]).all(), (lo, hi)
            monkeypatch.setenv("VQ_TSN_SPLIT", "1")
            m1 = net.TsnNet(g, w, max_crops=9, feature_blob="gp")
            m1.forward(crops, 1, mean)
            assert (m1.read_blob("c_bn", 9) == outs[sk]).all()
            m1.close()
        m.close()
    assert not (outs["0"] == outs["1"]).all()                                   # the split really is another order ...
    assert np.abs(outs["0"] - outs["1"]).max() <= 2e-6 * np.abs(outs["0"]).max()  # ... of the same sum


def test_every_direct_tiling_gives_the_same_bits(tsn, monkeypatch):
    """The tiling of a direct convolution is chosen per layer and batch size by timing: it must never change a result bit
    (every tiling sums an output element's K terms in the same order).  A stride-2 3x3 layer with a ragged M and N edge
    and a poisoned slot, under every tiling the table holds."""
    bi, net = tsn
    monkeypatch.setenv("VQ_TSN_POISON", "1")
    g = _mini(bi, 128, 14, 14, 160, 3, 2, 1)
    w = net.synthetic_weights(g, seed=9)
    crops = np.random.default_rng(3).integers(0, 256, (5, 14, 14, 128), dtype=np.uint8)
    mean = np.full(128, 117.0, np.float32)
    seen = None
    for bm, bn in ((128, 128), (128, 96), (128, 64), (64, 128), (64, 64), (128, 32), (32, 128)):
        for bk in (32, 16):
            for pipe in (0, 1):
                monkeypatch.setenv("VQ_TSN_TILE", "%dx%dx%dx%d" % (bm, bn, bk, pipe))
                m = net.TsnNet(g, w, max_crops=5, feature_blob="gp")
                m.forward(crops, 1, mean)
                got = m.read_blob("c_bn", 5)
                m.close()
                assert np.isfinite(got).all(), (bm, bn, bk, pipe)
                if seen is None:
                    seen = got
                assert (got == seen).all(), (bm, bn, bk, pipe)


WINO_CASES = [
    # (cin, h, cout, n_crops)      3x3 / stride 1 / pad 1 layers: Winograd F(2x2,3x3) form (csrc/vq_wino.hip)
    (64, 28, 96, 2),               # 3a/double_3x3_1: 392 tiles, Cout = 96 (ragged second 64-block)
    (64, 56, 192, 1),              # conv2/3x3
    (96, 14, 128, 3),              # 14x14: 147 tiles (ragged tile block)
    (160, 7, 224, 5),              # 7x7: odd size, 4x4 tiles cover 8x8
    (8, 5, 32, 1),                 # minimum: one K step, 9 tiles
    (256, 9, 320, 2),              # 32 K steps, odd size
    (64, 14, 96, 7),               # 4a/3x3 in small: 343 tiles = 21 units of 16 + 7 tiles, three 32-channel blocks
    (16, 3, 32, 2),                # one 16-channel step, 4 tiles per image
]


@pytest.mark.parametrize("cin,h,cout,n", WINO_CASES)
def test_winograd_conv_layer(tsn, monkeypatch, cin, h, cout, n):
    """Both workgroup shapes of the Winograd kernel against the fp64 oracle (same tolerance as the direct kernel),
    bit-identical to each other, and within the stated tolerance of the direct kernel."""
    bi, net = tsn
    monkeypatch.delenv("VQ_TSN_TILE", raising=False)
    g = _mini(bi, cin, h, h, cout, 3, 1, 1)
    w = net.synthetic_weights(g, seed=cin + cout)
    crops = np.random.default_rng(cin * h).integers(0, 256, (n, h, h, cin), dtype=np.uint8)
    mean = np.linspace(100.0, 130.0, cin).astype(np.float32)
    want = to.forward(g.layers, "data", w, to.preprocess(crops, mean), keep=("c_bn", "gp"))
    tol = 2e-5 * np.abs(want["c_bn"]).max()
    from video_query_algorithms_amd import _lib as vlib
    outs = []
    # 32 / 64 output channels per workgroup, units of 32 tiles; on maps of at most 14 x 14 with Cin % 16 == 0 (VQ_OP_CONV_WINOGRAD16: the
    # layer carries a second filter layout) also units of 16 tiles on v_mfma_f32_16x16x4 -- the channels meet every output element in the
    # 32-tile kernel's order, so all forms give the same bits
    both = cin % 16 == 0 and h <= 14
    for bm, bn, bk in ((128, 32, 8), (128, 64, 8)) + (((64, 32, 16), (64, 64, 16)) if both else ()):
        m = net.TsnNet(g, w, max_crops=n, feature_blob="gp", winograd=True)
        assert m.layer_op(0) == (vlib.VQ_OP_CONV_WINOGRAD16 if both else vlib.VQ_OP_CONV_WINOGRAD)
        tiles = m.layer_tiles(n)
        assert tiles[0, 3] == 2                                      # the conv layer is in Winograd form
        tiles[0] = (bm, bn, bk, 2)
        m.set_layer_tiles(n, tiles)
        feat, ps = m.forward(crops, 1, mean)
        got = _nchw(m.read_blob("c_bn", n))
        assert m.layer_tiles(n)[0].tolist() == [bm, bn, bk, 2]
        assert np.abs(got - want["c_bn"]).max() <= tol
        assert (got >= 0).all()
        assert np.abs(ps - want["gp"].reshape(n, -1)).max() <= tol
        outs.append(got)
        m.close()
    assert all((o == outs[0]).all() for o in outs[1:])
    if not both:                                                      # a layer without the second layout refuses the 16-tile form
        m = net.TsnNet(g, w, max_crops=n, feature_blob="gp", winograd=True)
        tiles = m.layer_tiles(n)
        tiles[0] = (64, 32, 16, 2)
        with pytest.raises(vlib.VqError):
            m.set_layer_tiles(n, tiles)
        m.close()
    monkeypatch.setenv("VQ_TSN_WINO16", "0")                           # the switch: no second layout anywhere
    m = net.TsnNet(g, w, max_crops=n, feature_blob="gp", winograd=True)
    assert m.layer_op(0) == vlib.VQ_OP_CONV_WINOGRAD
    m.forward(crops, 1, mean)
    assert (_nchw(m.read_blob("c_bn", n)) == outs[0]).all()
    m.close()
    monkeypatch.delenv("VQ_TSN_WINO16")
    m = net.TsnNet(g, w, max_crops=n, feature_blob="gp", winograd=False)
    assert m.layer_tiles(n)[0, 3] != 2
    m.forward(crops, 1, mean)
    direct = _nchw(m.read_blob("c_bn", n))
    m.close()
    assert np.abs(direct - outs[0]).max() <= tol


@pytest.mark.parametrize("kind,k,s,p,h,c", [("MAX", 3, 2, 0, 28, 64), ("MAX", 3, 2, 0, 14, 96), ("AVE", 3, 1, 1, 14, 64),
                                          ("MAX", 3, 1, 1, 7, 128), ("AVE", 3, 1, 1, 7, 32), ("MAX", 3, 2, 0, 9, 32)])
def test_pooling_is_bit_exact(tsn, kind, k, s, p, h, c):
    bi, net = tsn
    g = _mini(bi, 32, h, h, c, 1, 1, 0, pool=(kind, k, s, p))
    w = net.synthetic_weights(g, seed=3)
    crops = np.random.default_rng(h).integers(0, 256, (3, h, h, 32), dtype=np.uint8)
    mean = np.full(32, 120.0, dtype=np.float32)
    m = net.TsnNet(g, w, max_crops=3, feature_blob="gp")
    m.forward(crops, 3, mean)
    x = _nchw(m.read_blob("c_bn", 3))                       # the device's own conv output, fp32
    got = _nchw(m.read_blob("p", 3))
    want = to.pool_direct(x, k, s, p, kind)                 # fp32, same accumulation order
    assert got.shape == want.shape and (got == want).all()
    # and the global average pool + consensus on top of it
    feat, ps = m.forward(crops, 3, mean)
    gp = to.pool_direct(got, got.shape[2], 1, 0, "AVE").reshape(3, -1)
    assert (ps == gp).all()
    assert (feat == to.consensus(ps, 3)).all()
    m.close()


@pytest.fixture(scope="module")
def rgb_case(tsn):
    bi, net = tsn
    g = bi.bn_inception(3)
    w = net.synthetic_weights(g, seed=2)
    crops = np.random.default_rng(1).integers(0, 256, (4, 224, 224, 3), dtype=np.uint8)
    keep = ("conv1/7x7_s2_bn", "pool1/3x3_s2", "conv2/3x3_bn", "pool2/3x3_s2", "inception_3a/output",
            "inception_3b/output", "inception_3c/output", "inception_4a/output", "inception_4c/output",
            "inception_4e/output", "inception_5a/output", "inception_5b/output", "global_pool")
    want = to.forward(g.layers, "data", w, to.preprocess(crops, net.RGB_MEAN), keep=keep)
    return g, w, crops, keep, want


def test_bn_inception_rgb_layerwise_and_features(tsn, rgb_case):
    bi, net = tsn
    g, w, crops, keep, want = rgb_case
    m = net.TsnNet(g, w, max_crops=8)
    feat, ps = m.forward(crops, 2, net.RGB_MEAN)
    assert abs(m.flops_per_crop() - 2 * 2_031_576_064) < 1
    worst = {}
    plain = net.TsnNet(g, w, max_crops=4, fuse=False)      # materialises the blobs the fused plan folds away (the stem pools)
    plain.forward(crops, 2, net.RGB_MEAN)
    folded = [name for name in keep[:-1] if name not in m.plan.blob_loc]
    assert folded == ["pool1/3x3_s2"] + (["pool2/3x3_s2"] if bi.FOLD_POOL_MAX_COUT >= 224 else [])
    for name in keep[:-1]:
        got = _nchw((plain if name in folded else m).read_blob(name, 4))
        d = np.abs(got - want[name]).max() / np.abs(want[name]).max()
        worst[name] = d
        assert d <= 2e-4, (name, d)
    ref = want["global_pool"].reshape(4, -1)
    assert ps.shape == (4, 1024) and feat.shape == (2, 1024)
    assert np.abs(ps - ref).max() <= 2e-4 * np.abs(ref).max()
    assert (ps >= 0).all() and np.isfinite(ps).all() and ps.max() > 0.05       # post-ReLU average, O(1) activations
    assert (feat == to.consensus(ps, 2)).all()                                  # fp64 mean of the fp32 blobs
    print("worst relative layer errors:", {k: float("%.2e" % v) for k, v in worst.items()})
    # batch-size / order invariance: each crop's feature does not depend on its neighbours in the batch
    feat1, ps1 = m.forward(crops[[2, 0]], 1, net.RGB_MEAN)
    assert (ps1[0] == ps[2]).all() and (ps1[1] == ps[0]).all()
    m.close()
    plain.close()


def test_fused_and_unfused_graphs_agree(tsn, rgb_case):
    """Merging sibling 1x1 convolutions never changes a bit (same k order per output element); commuting the average
    pool with its projection changes only that branch, at rounding level."""
    bi, net = tsn
    g, w, crops, keep, want = rgb_case
    m0 = net.TsnNet(g, w, max_crops=4, fuse=False)
    m1 = net.TsnNet(g, w, max_crops=4, fuse=True)
    f0, p0 = m0.forward(crops, 2, net.RGB_MEAN)
    f1, p1 = m1.forward(crops, 2, net.RGB_MEAN)
    # the 1x1 GEMM behind pool1 takes the window maximum in its loader: same bits as pool layer + GEMM
    for name in ("conv2/3x3_reduce_bn", "inception_3a/3x3_reduce_bn", "inception_3a/double_3x3_reduce_bn"):
        assert (m0.read_blob(name, 4) == m1.read_blob(name, 4)).all(), name
    a0, a1 = m0.read_blob("inception_3a/output", 4), m1.read_blob("inception_3a/output", 4)
    assert (a0[..., :224] == a1[..., :224]).all()                      # 1x1 | 3x3 | double 3x3 branches: identical bits
    assert np.abs(a0[..., 224:] - a1[..., 224:]).max() <= 1e-5 * np.abs(a0).max()   # pool_proj branch: rounding only
    assert np.abs(p0 - p1).max() <= 2e-5 * np.abs(p0).max()
    ref = want["global_pool"].reshape(4, -1)
    assert np.abs(p1 - ref).max() <= 2e-4 * np.abs(ref).max()
    m0.close()
    m1.close()


def test_bn_inception_flow_features(tsn):
    bi, net = tsn
    g = bi.bn_inception(10)
    w = net.synthetic_weights(g, seed=5)
    crops = np.random.default_rng(2).integers(0, 256, (3, 224, 224, 10), dtype=np.uint8)
    want = to.forward(g.layers, "data", w, to.preprocess(crops, net.FLOW_MEAN), keep=("global_pool",))["global_pool"]
    m = net.TsnNet(g, w, max_crops=3)
    feat, ps = m.forward(crops, 3, net.FLOW_MEAN)
    assert abs(m.flops_per_crop() - 2 * 2_306_941_952) < 1
    ref = want.reshape(3, -1)
    assert np.abs(ps - ref).max() <= 2e-4 * np.abs(ref).max()
    assert (feat == to.consensus(ps, 3)).all()
    m.close()


def test_cfg2_shape_runs_and_is_deterministic(tsn):
    """BASELINE config[1]: B = 32 clips x T = 3 snippets, RGB.  Two runs must agree bit for bit (no atomics,
    fixed reduction order), features finite and non-negative; a sample of the 96 crops of THIS batch (first, last, a clip
    boundary, the middle) against the fp64 CPU evaluation of the layer list at the stated whole-network tolerance, and the
    consensus of every clip recomputed in fp64 from the per-snippet features, bit for bit."""
    bi, net = tsn
    g = bi.bn_inception(3)
    w = net.synthetic_weights(g, seed=2)
    crops = np.random.default_rng(1).integers(0, 256, (96, 224, 224, 3), dtype=np.uint8)
    m = net.TsnNet(g, w, max_crops=96)
    f1, p1 = m.forward(crops, 3, net.RGB_MEAN)
    f2, p2 = m.forward(crops, 3, net.RGB_MEAN)
    assert (f1 == f2).all() and (p1 == p2).all()
    assert f1.shape == (32, 1024) and np.isfinite(f1).all() and (f1 >= 0).all()
    pick = np.array([0, 2, 3, 47, 50, 95])
    ref = to.forward(g.layers, "data", w, to.preprocess(crops[pick], net.RGB_MEAN), keep=("global_pool",))["global_pool"].reshape(len(pick), -1)
    assert np.abs(p1[pick] - ref).max() <= 2e-4 * np.abs(ref).max()
    assert (f1 == to.consensus(p1, 3)).all()
    # same crops in a smaller batch give the same bits (tile choice may differ with M: k-order is fixed)
    m2 = net.TsnNet(g, w, max_crops=6)
    f3, p3 = m2.forward(crops[:6], 3, net.RGB_MEAN)
    assert (p3 == p1[:6]).all() and (f3 == f1[:2]).all()
    m.close()
    m2.close()


def test_t25_features_have_the_arithmetic_of_the_reference_files(tsn):
    """The reference's shipped feature files are fp64 means of 25 fp32 global_pool blobs (calcSig_wOF.py:82 at the script's default
    of 25 snippets; tests/test_feature_files.py::test_reference_feature_values_are_fp64_means_of_25_fp32_blobs measures it on the
    files).  The product's T = 25 features have the same arithmetic: 25 x value is a short exact sum, values >= 0 and not fp32."""
    from _helpers import mean_sum_bits
    bi, net = tsn
    g = bi.bn_inception(3)
    w = net.synthetic_weights(g, seed=2)
    crops = np.random.default_rng(25).integers(0, 256, (50, 224, 224, 3), dtype=np.uint8)
    m = net.TsnNet(g, w, max_crops=50)
    feat, ps = m.forward(crops, 25, net.RGB_MEAN)
    m.close()
    assert feat.shape == (2, 1024) and feat.dtype == np.float64 and (feat >= 0).all()
    assert (feat == to.consensus(ps, 25)).all()
    pos = feat[feat > 0]
    need = mean_sum_bits(pos, 25)
    assert np.median(need) <= 33 and (need <= 40).mean() >= 0.99 and need.max() <= 52, (np.median(need), (need <= 40).mean(), need.max())
    assert (mean_sum_bits(pos, 24) >= 48).mean() >= 0.95
    assert (pos.astype(np.float32).astype(np.float64) == pos).mean() <= 0.05


def test_bad_shapes_are_rejected_before_launch(tsn):
    bi, net = tsn
    import video_query_algorithms_amd as vqa
    g = _mini(bi, 32, 8, 8, 32, 3, 1, 1)
    w = net.synthetic_weights(g, seed=1)
    m = net.TsnNet(g, w, max_crops=2, feature_blob="gp")
    with pytest.raises(ValueError):
        m.forward(np.zeros((2, 8, 8, 31), dtype=np.uint8), 1, np.zeros(32, dtype=np.float32))
    with pytest.raises(vqa.VqError):
        m.forward(np.zeros((4, 8, 8, 32), dtype=np.uint8), 1, np.zeros(32, dtype=np.float32))   # > max_crops
    bad = dict(w)
    bad["c"] = {"W": w["c"]["W"][:, :16], "b": w["c"]["b"]}
    with pytest.raises(ValueError):
        net.TsnNet(g, bad, max_crops=2, feature_blob="gp")
    m.close()


def _write_protos(bi, tmp_path):
    """The two deploy prototxts of the reference's layout, written from the built-in graph (and parsed back)."""
    protos = {}
    for name, c in (("rgb", 3), ("flow", 10)):
        g = bi.bn_inception(c)
        lines = ['name: "BN-Inception"', 'input: "data"', "input_dim: 1", "input_dim: %d" % c, "input_dim: 224", "input_dim: 224"]
        for l in g.layers:
            body = 'layer { name: "%s" type: "%s" %s %s' % (l.name, l.type, " ".join('bottom: "%s"' % b for b in l.bottoms),
                                                            " ".join('top: "%s"' % t for t in l.tops))
            if l.type == "Convolution":
                body += " convolution_param { num_output: %d pad: %d kernel_size: %d stride: %d }" % (l.num_output, l.pad, l.kernel, l.stride)
            elif l.type == "Pooling":
                body += " pooling_param { pool: %s kernel_size: %d stride: %d pad: %d }" % (l.pool, l.kernel, l.stride, l.pad)
            elif l.type == "InnerProduct":
                body += " inner_product_param { num_output: %d }" % l.num_output
            lines.append(body + " }")
        protos[name] = str(tmp_path / ("%s.prototxt" % name))
        open(protos[name], "w").write("\n".join(lines))
        assert bi.load_prototxt(protos[name]).layers == g.layers
    return protos


def test_calcsig_command_line_end_to_end(tsn, tmp_path):
    """The drop-in CLI on a synthetic frame tree: two clips of one video, T = 3, both streams; the CSVs it writes
    must parse with load_db's rules and carry the oracle's features; the reference's own per-snippet loop run on the
    CaffeNet-shaped adapter must give the same numbers."""
    bi, net = tsn
    from video_query_algorithms_amd import calcSig_wOF
    from video_query_algorithms_amd.tsn import caffe_net, feature_csv, frames
    rng = np.random.default_rng(7)
    root = tmp_path / "frames"
    counts = {"clip_0001": 7, "clip_0002": 9}
    for clip, n in counts.items():
        d = root / "myvideo" / clip
        d.mkdir(parents=True)
        for i in range(1, n + 1):
            frames.write_pnm(str(d / ("img_%05d.ppm" % i)), rng.integers(0, 256, (256, 340, 3), dtype=np.uint8))
            # grayscale (P5) content; the reader goes by the magic number, so one extension serves both
            frames.write_pnm(str(d / ("flow_x_%05d.ppm" % i)), rng.integers(0, 256, (256, 340), dtype=np.uint8))
            frames.write_pnm(str(d / ("flow_y_%05d.ppm" % i)), rng.integers(0, 256, (256, 340), dtype=np.uint8))
    protos = _write_protos(bi, tmp_path)
    wfile = {}
    weights = {}
    for name, c, seed in (("rgb", 3, 2), ("flow", 10, 5)):
        weights[name] = net.synthetic_weights(bi.bn_inception(c), seed=seed)
        if name == "rgb":                                  # one stream from a .caffemodel (what the reference passes) ...
            from video_query_algorithms_amd.tsn import caffemodel
            wfile[name] = str(tmp_path / "ucf101_split1_tsn_rgb_bn_inception.caffemodel")
            caffemodel.write_caffemodel(wfile[name], bi.bn_inception(c), weights[name])
        else:                                              # ... the other from the .npz layout
            wfile[name] = str(tmp_path / ("ucf101_split1_tsn_%s_bn.npz" % name))
            caffe_net.save_weights(wfile[name], weights[name])
    out_dir = tmp_path / "features"
    rc = calcSig_wOF.main([str(root), protos["rgb"], wfile["rgb"], protos["flow"], wfile["flow"], "--num_frame_per_video", "3",
                           "--outFeatures_dir", str(out_dir), "--modelname", "UCF101_split1", "--frame_ext", ".ppm",
                           "--batch_clips", "2", "--num_worker", "3"])
    assert rc == 0
    nsplit, streams = feature_csv.read_split_dir(str(out_dir / "myvideo" / "UCF101_split1"))
    assert nsplit == 1 and set(streams) == {"rgb", "warped_optical_flow"}
    for mode, name, c, mean in (("rgb", "rgb", 3, net.RGB_MEAN), ("warped_optical_flow", "flow", 10, net.FLOW_MEAN)):
        clips, feats, meta = streams[mode]
        assert clips.tolist() == [1, 2] and feats.shape == (2, 1024) and meta["video"] == "myvideo"
        assert meta["dnn_weights_file_uri"] == wfile[name]
        g = bi.bn_inception(c)
        crops = []
        for clip, n in counts.items():
            d = str(root / "myvideo" / clip)
            ticks = to.frame_ticks(n, 3, 1 if c == 3 else 5)
            if c == 3:
                crops.append(frames.load_rgb_snippets(d, ticks, ext=".ppm"))
            else:
                crops.append(frames.load_flow_snippets(d, ticks, n, ext=".ppm"))
        crops = np.concatenate(crops)
        ps, cons = to.features(g.layers, "data", weights[name], crops, mean, 3)
        assert np.abs(feats - cons).max() <= 2e-4 * np.abs(cons).max()
        if c == 3:
            # the reference's own loop (calcSig_wOF.py:88-96) on the CaffeNet-shaped adapter
            cn = caffe_net.CaffeNet(protos["rgb"], wfile["rgb"], 0, max_crops=4)
            d = str(root / "myvideo" / "clip_0001")
            frame_features = []
            for tick in to.frame_ticks(7, 3, 1):
                frame = frames.imread(os.path.join(d, "img_%05d.ppm" % tick), True)
                cn.predict_single_frame([frame, ], "fc-action", frame_size=(340, 256))
                frame_features.append(cn._net.blobs["global_pool"].data[0].reshape(1, -1).tolist())
            v_avg_feature = np.array(frame_features).mean(axis=0)[0]
            assert (v_avg_feature == feats[0]).all()          # batched path == per-snippet path, bit for bit
            cn.close()


def test_one_pass_ensemble_on_the_gpu_writes_the_bytes_of_separate_runs(tsn, tmp_path):
    """calcSig_wOF_ensemble.sh:13-37 as ONE command (``--ensemble``): three weight sets fed from one decode on the real extractors --
    two videos (batches straddle them), T = 3 -- must write, per member, the bytes a run of its own writes.  Once with decoded frames
    resized on the GPU (one resize feeding three networks) and, where Pillow can make JPEG files, with ``--device_jpeg`` (the
    work goes batch by batch through both streams there)."""
    bi, net = tsn
    from video_query_algorithms_amd import calcSig_wOF
    from video_query_algorithms_amd.tsn import caffe_net, frames
    rng = np.random.default_rng(17)
    try:
        from PIL import Image
    except ImportError:
        Image = None
    protos = _write_protos(bi, tmp_path)
    members = []
    for k in (1, 2, 3):
        files = {}
        for name, c in (("rgb", 3), ("flow", 10)):
            files[name] = str(tmp_path / ("ucf101_split%d_tsn_%s_bn.npz" % (k, name)))
            caffe_net.save_weights(files[name], net.synthetic_weights(bi.bn_inception(c), seed=10 * k + c))
        members.append(("UCF101_split%d" % k, files["rgb"], files["flow"]))

    def tree(name, ext, write):
        root = tmp_path / name
        for video, clips in (("va", {"clip_0001": 6, "clip_0002": 8, "clip_0004": 6}), ("vb", {"clip_0003": 7})):
            for clip, n in clips.items():
                d = root / video / clip
                d.mkdir(parents=True)
                for i in range(1, n + 1):
                    write(str(d / ("img_%05d%s" % (i, ext))), rng.integers(0, 256, (64, 96, 3), dtype=np.uint8))
                    write(str(d / ("flow_x_%05d%s" % (i, ext))), rng.integers(0, 256, (64, 96), dtype=np.uint8))
                    write(str(d / ("flow_y_%05d%s" % (i, ext))), rng.integers(0, 256, (64, 96), dtype=np.uint8))
        return str(root)

    def csvs(out):
        found = {}
        for dirpath, _, names in os.walk(out):
            for fn in names:
                found[os.path.relpath(os.path.join(dirpath, fn), out)] = open(os.path.join(dirpath, fn), "rb").read()
        return found

    cases = [("ppm", ".ppm", frames.write_pnm, [])]
    if Image is not None:
        cases.append(("jpg", ".jpg", lambda path, a: Image.fromarray(a).save(path, "JPEG", quality=92), ["--device_jpeg"]))
    for label, ext, write, extra in cases:
        root = tree("frames_" + label, ext, write)
        common = ["--num_frame_per_video", "3", "--frame_ext", ext, "--batch_clips", "3", "--num_worker", "4"] + extra
        want = {}
        for name, w_rgb, w_flow in members:
            out = str(tmp_path / ("sep_%s_%s" % (label, name)))
            assert calcSig_wOF.main([root, protos["rgb"], w_rgb, protos["flow"], w_flow, "--outFeatures_dir", out, "--modelname", name] + common) == 0
            want.update(csvs(out))
        assert len(want) == 12 and len(set(want.values())) == 12            # 2 videos x 3 members x 2 streams, all different
        out = str(tmp_path / ("ens_" + label))
        argv = [root, protos["rgb"], members[0][1], protos["flow"], members[0][2], "--outFeatures_dir", out, "--modelname", members[0][0]] + common
        for name, w_rgb, w_flow in members[1:]:
            argv += ["--ensemble", name, w_rgb, w_flow]
        assert calcSig_wOF.main(argv) == 0
        assert csvs(out) == want


def test_grouped_winograd_launches_do_not_change_a_bit(tsn, monkeypatch):
    """The Winograd convolutions of one dependency level (the 3x3 and the first double-3x3 arm of an inception module)
    and the level's pooling layer share one kernel launch.  VQ_TSN_GROUP=0 gives every layer its own launch: same bits
    either way, and the launch table really groups sibling arms while a chain (double_3x3_1 -> double_3x3_2) stays in
    separate launches."""
    bi, net = tsn
    g = bi.bn_inception(3)
    w = net.synthetic_weights(g, seed=2)
    crops = np.random.default_rng(9).integers(0, 256, (3, 224, 224, 3), dtype=np.uint8)
    monkeypatch.setenv("VQ_TSN_SPLIT", "1")
    monkeypatch.setenv("VQ_TSN_GROUP", "0")
    m = net.TsnNet(g, w, max_crops=3)
    items, n_items = m.launch_items()
    assert n_items == len(m.plan.ops) and len(set(items.tolist())) == n_items
    f1, p1 = m.forward(crops, 3, net.RGB_MEAN)
    b1 = m.read_blob("inception_4c/output", 3)
    m.close()
    monkeypatch.setenv("VQ_TSN_GROUP", "1")
    m = net.TsnNet(g, w, max_crops=3)
    items, n_items = m.launch_items()
    names = [o.name for o in m.plan.ops]
    for blk in ("3a", "3b", "4a", "4b", "4c", "4d", "5a", "5b"):
        i3, id1, id2 = (names.index("inception_%s/%s" % (blk, x)) for x in ("3x3", "double_3x3_1", "double_3x3_2"))
        assert items[i3] == items[id1] and items[id2] > items[id1]
        assert items[names.index("inception_%s/pool" % blk)] == items[i3]              # the pooling arm rides along
    for blk in ("3c", "4e"):                                                           # reduction modules: double_3x3_1 + max pool
        assert items[names.index("inception_%s/pool" % blk)] == items[names.index("inception_%s/double_3x3_1" % blk)]
    assert n_items <= len(m.plan.ops) - 18
    order = np.argsort(items, kind="stable")                       # a valid order: every producer's launch precedes its consumers'
    assert items[names.index("inception_3a/pool")] < items[names.index("inception_3b/3x3")]
    f2, p2 = m.forward(crops, 3, net.RGB_MEAN)
    b2 = m.read_blob("inception_4c/output", 3)
    m.close()
    assert (p1 == p2).all() and (f1 == f2).all() and (b1 == b2).all() and len(order) == len(names)


@pytest.mark.parametrize("channels,split", [(3, "1"), (10, "2")])
def test_every_winograd_variant_writes_all_it_owns(tsn, monkeypatch, channels, split):
    """VQ_TSN_POISON=1 fills every activation slot with NaN patterns before a forward, so an output element a kernel
    variant fails to write (and that would otherwise silently keep the value of an earlier forward or of an autotune
    launch -- which is how a wrong variant hides from a plain A/B comparison) turns into NaN.  Both Winograd workgroup
    shapes, forced on every 3x3 layer at a batch so small that every layer has ragged tile blocks and some have a
    single 64-channel block: no NaN anywhere, identical bits."""
    bi, net = tsn
    monkeypatch.setenv("VQ_TSN_POISON", "1")
    monkeypatch.setenv("VQ_TSN_SPLIT", split)
    g = bi.bn_inception(channels)
    w = net.synthetic_weights(g, seed=5)
    mean = net.RGB_MEAN if channels == 3 else net.FLOW_MEAN
    crops = np.random.default_rng(1).integers(0, 256, (6, 224, 224, channels), dtype=np.uint8)
    m = net.TsnNet(g, w, max_crops=6)
    m.forward(crops, 3, mean)
    got = {}
    from video_query_algorithms_amd import _lib as vlib
    has16 = np.array([m.layer_op(i) == vlib.VQ_OP_CONV_WINOGRAD16 for i in range(len(m.plan.ops))])
    assert has16.sum() == 19                                          # the 14 x 14 and 7 x 7 Winograd layers carry both layouts
    for variant in (0, 1, 2, 3):                                      # 2, 3: units of 16 tiles where the layer has the layout for them
        for n, paired, _ in m.tile_tables():
            t = m.layer_tiles(n, paired=paired)
            wino = t[:, 3] == 2
            t[wino, 1] = 32 * ((variant & 1) + 1)
            t[wino, 0], t[wino, 2] = 128, 8
            if variant >= 2:
                t[wino & has16, 0], t[wino & has16, 2] = 64, 16
            m.set_layer_tiles(n, t, paired=paired)
        f, p = m.forward(crops, 3, mean)
        assert np.isfinite(p).all() and np.isfinite(f).all()
        for name in m.plan.blob_loc:
            assert np.isfinite(m.read_blob(name, 6)).all(), (variant, name)
        got[variant] = p
    m.close()
    assert all((got[v] == got[0]).all() for v in (1, 2, 3))


@pytest.mark.parametrize("split", ["1", "2", "3", "2,1"])
def test_batch_split_streams_do_not_change_a_bit(tsn, monkeypatch, split):
    """VQ_TSN_SPLIT: sub-batches of one forward on separate HIP streams (default 2).  Every crop is independent and
    every tiling sums in the same order, so the features must be bit-identical to the single-stream run."""
    bi, net = tsn
    g = bi.bn_inception(3)
    w = net.synthetic_weights(g, seed=2)
    crops = np.random.default_rng(9).integers(0, 256, (6, 224, 224, 3), dtype=np.uint8)
    monkeypatch.setenv("VQ_TSN_SPLIT", "1")
    m = net.TsnNet(g, w, max_crops=6)
    f1, p1 = m.forward(crops, 3, net.RGB_MEAN)
    b1 = m.read_blob("inception_4c/output", 6)
    m.close()
    monkeypatch.setenv("VQ_TSN_SPLIT", split)
    m = net.TsnNet(g, w, max_crops=6)
    f2, p2 = m.forward(crops, 3, net.RGB_MEAN)
    b2 = m.read_blob("inception_4c/output", 6)
    f3, p3 = m.forward(crops[:3], 3, net.RGB_MEAN)          # a size the split does not divide falls back to one stream
    m.close()
    assert (p1 == p2).all() and (f1 == f2).all() and (b1 == b2).all()
    assert (p3 == p1[:3]).all() and (f3 == f1[:1]).all()


def test_cfg5_pipeline_small(tsn, tmp_path, capsys):
    """BASELINE configs[4] at toy size: features of both streams x 3 weight seeds written into the resident DB on the
    device, the CSV tree round trip, then query rounds through the Ticket / Hyperparameter drop-ins."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import e2e_cfg5
    rc = e2e_cfg5.main(["--clips", "24", "--segments", "3", "--batch-clips", "8", "--rounds", "3", "--labels", "8", "--csv-clips", "5",
                        "--out", str(tmp_path)])
    assert rc == 0
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["clips"] == 24 and out["rounds"] == 3 and out["csv_clips"] == 5
    assert 0 < out["final_matches"] <= 21 and 0.5 <= out["final_weights"]["warped_optical_flow"] <= 2.5
    assert os.path.exists(os.path.join(str(tmp_path), "synthetic_video", "UCF101_split2", "warped_optical_flow_global_pool_features.csv"))


def test_handles_are_thread_safe_and_do_not_leak(tsn):
    """The broker re-arms itself from timer threads (broker.py:91-92): two threads hammer one handle (internally
    locked) and a second handle at the same time; results stay bit-identical.  Creating and destroying handles in a loop
    must not eat device memory (streams, events and slots are released)."""
    import threading
    import torch
    bi, net = tsn
    g = _mini(bi, 32, 28, 28, 64, 3, 1, 1, pool=("MAX", 3, 2, 0))
    w = net.synthetic_weights(g, seed=4)
    crops = np.random.default_rng(2).integers(0, 256, (6, 28, 28, 32), dtype=np.uint8)
    mean = np.full(32, 120.0, dtype=np.float32)
    m1 = net.TsnNet(g, w, max_crops=6, feature_blob="gp")
    m2 = net.TsnNet(g, w, max_crops=6, feature_blob="gp")
    want, _ = m1.forward(crops, 3, mean)
    errors = []

    def worker(m, n):
        try:
            for _ in range(n):
                f, _ = m.forward(crops, 3, mean)
                if not (f == want).all():
                    errors.append("mismatch")
        except Exception as e:           # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(m1, 20)), threading.Thread(target=worker, args=(m1, 20)),
               threading.Thread(target=worker, args=(m2, 20))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    m1.close()
    m2.close()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(15):
        m = net.TsnNet(g, w, max_crops=6, feature_blob="gp")
        m.forward(crops, 3, mean)
        m.close()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, (free0, free1)


@pytest.mark.parametrize("h,w", [(360, 480), (256, 340), (240, 320), (480, 854)])
def test_device_resize_crop_against_the_pixel_loop_oracle(tsn, h, w):
    """vq_resize_crop (frame ingest on the GPU) against oracle/frames_oracle.py -- an independent
    scalar-loop restatement of the resize + crop-0 step behind calcSig_wOF.py:94,111 -- byte for byte: RGB frames and
    the 10 interleaved grey planes of a flow stack; frames already at 340x256 pass through untouched.  The host path of
    the product (tsn/frames.py, used by --host_resize) is held to the same oracle in tests/test_feature_files.py.
    The default rule is cv2's own fixed-point uint8 path (oracle: fixed_point_resize_pixel) -- integer arithmetic, so the
    device must equal it bit for bit; the exact-weight rule stays available (resize_rule="exact").  PARITY UNPINNED
    against cv2 itself (no cv2, no reference frames; frames_oracle docstring)."""
    import frames_oracle as fo
    from video_query_algorithms_amd.tsn import caffe_net
    bi, net = tsn
    rng = np.random.default_rng(h + w)
    g3 = bi.bn_inception(3)
    cn = caffe_net.CaffeNet(g3, net.synthetic_weights(g3, seed=2), max_crops=4)
    rgb = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    got = cn.crops_from_frames(rgb).cpu().numpy()
    want = np.stack([fo.crop0(f) for f in rgb])              # = fixed_point_resize_pixel on every surviving pixel
    assert got.shape == (3, 224, 224, 3) and (got == want).all()
    cn.close()
    ce = caffe_net.CaffeNet(g3, net.synthetic_weights(g3, seed=2), max_crops=4, resize_rule="exact")
    got = ce.crops_from_frames(rgb[:1]).cpu().numpy()
    assert (got[0] == fo.crop0(rgb[0], rule="exact")).all()
    ce.close()
    g10 = bi.bn_inception(10)
    cf = caffe_net.CaffeNet(g10, net.synthetic_weights(g10, seed=2), max_crops=4)
    planes = rng.integers(0, 256, (2, 10, h, w), dtype=np.uint8)
    got = cf.crops_from_frames(planes).cpu().numpy()
    want = np.stack([fo.flow_stack_crop0(snip) for snip in planes])
    assert got.shape == (2, 224, 224, 10) and (got == want).all()
    feats = cf.extract_clips_from_frames(planes, 2)
    assert (feats == cf.extract_clips(want, 2)).all()
    dev_feats = cf.extract_clips_from_frames(planes, 2, on_device=True)           # the block a rank all-gathers
    assert dev_feats.is_cuda and (dev_feats.cpu().numpy() == feats).all()
    cf.close()
    # the ten planes in ONE launch (vq_resize_crop_planes: what the device JPEG path calls on the decoder's plane-major buffer), both rules
    import ctypes as C
    import torch
    from video_query_algorithms_amd import _lib
    from video_query_algorithms_amd.tsn import frames as fr
    major = torch.from_numpy(np.ascontiguousarray(planes.transpose(1, 0, 2, 3))).cuda()          # [10][n][h][w]
    for rule in ("cv2", "exact"):
        out = torch.full((2, 224, 224, 10), 7, dtype=torch.uint8, device="cuda")
        _lib.call("vq_resize_crop_planes", C.c_void_p(major.data_ptr()), 2, h, w, 10, 2 * h * w, 340, 256, 224, fr.RESIZE_RULES[rule],
                  C.c_void_p(out.data_ptr()), 0, None)
        torch.cuda.synchronize()
        want = np.stack([fo.flow_stack_crop0(snip, rule=rule) for snip in planes])
        assert (out.cpu().numpy() == want).all(), rule
    with pytest.raises(_lib.VqError):
        _lib.call("vq_resize_crop_planes", C.c_void_p(major.data_ptr()), 2, h, w, 3, 2 * h * w, 340, 256, 224, 0, C.c_void_p(out.data_ptr()), 0, None)


def test_calcsig_as_a_fresh_process_runs_without_torch(tsn, tmp_path):
    """``python calcSig_wOF.py ...`` on one GPU never imports torch (tsn/devmem.py: the crops' device buffers, the preparation lanes' streams
    and the waits come from the library; VQ_NO_TORCH is set by main() before the library is loaded) -- 0.8 s of a 2.3 s process -- and writes
    the bytes of the in-process run of this (torch-holding) test process, host decode + device resize and --device_jpeg alike."""
    import subprocess
    import sys
    from video_query_algorithms_amd import calcSig_wOF
    bi, net = tsn
    pytest.importorskip("PIL")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_jpeg_oracle import encode, picture
    root = tmp_path / "frames"
    rng = np.random.default_rng(3)
    for clip, n in (("clip_0001", 7), ("clip_0002", 9), ("clip_0003", 6)):
        d = root / "vid" / clip
        d.mkdir(parents=True)
        for i in range(1, n + 1):
            (d / ("img_%05d.jpg" % i)).write_bytes(encode(picture(256, 340, int(rng.integers(1 << 30))), quality=90, subsampling=2))
            for p in ("flow_x", "flow_y"):
                (d / ("%s_%05d.jpg" % (p, i))).write_bytes(encode(picture(256, 340, int(rng.integers(1 << 30)))[:, :, 0], quality=90))
    protos = _write_protos(bi, tmp_path)
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-query-algorithms_amd", "calcSig_wOF.py")
    files = ("rgb_global_pool_features.csv", "warped_optical_flow_global_pool_features.csv")
    for tag, extra in (("host", []), ("device", ["--device_jpeg"])):
        args = [str(root), protos["rgb"], "synthetic:2", protos["flow"], "synthetic:5", "--num_frame_per_video", "3", "--modelname", "UCF101_split1",
                "--batch_clips", "2"] + extra
        assert calcSig_wOF.main(args + ["--outFeatures_dir", str(tmp_path / ("in_" + tag))]) == 0
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VQ_NO_TORCH")}
        env["VQ_CLI_TRACE"] = "1"
        p = subprocess.run([sys.executable, cli] + args + ["--outFeatures_dir", str(tmp_path / ("out_" + tag))], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        assert "torch imported: False" in p.stderr, p.stderr[-1000:]
        for f in files:
            assert (tmp_path / ("out_" + tag) / "vid" / "UCF101_split1" / f).read_bytes() == (tmp_path / ("in_" + tag) / "vid" / "UCF101_split1" / f).read_bytes(), (tag, f)
    # VQ_NO_TORCH=0 keeps torch in the process
    p = subprocess.run([sys.executable, cli] + args + ["--outFeatures_dir", str(tmp_path / "out_torch")], env=dict(env, VQ_NO_TORCH="0"), capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and "torch imported: True" in p.stderr, p.stderr[-1000:]


def test_calcsig_two_ranks_write_the_same_bytes_as_one(tsn, tmp_path):
    """The N > 1 path of the drop-in command line with the REAL extractor: two ranks under torch.distributed.run, both
    on this one GPU with the gloo backend (VQ_DIST_BACKEND=gloo; RCCL refuses two ranks on one device), clips sharded
    5 = 3 + 2, feature blocks kept on the device until the gather, rank 0 writes.  The CSV tree must equal the one-rank
    run byte for byte (calcSig_wOF.py:204-210: the result must not depend on how many workers shared the clips)."""
    import subprocess
    import sys
    bi, net = tsn
    from video_query_algorithms_amd.tsn import frames
    rng = np.random.default_rng(17)
    root = tmp_path / "frames"
    for clip, n in (("clip_0001", 7), ("clip_0002", 9), ("clip_0003", 7), ("clip_0004", 8), ("clip_0007", 7)):
        d = root / "vid" / clip
        d.mkdir(parents=True)
        for i in range(1, n + 1):
            frames.write_pnm(str(d / ("img_%05d.ppm" % i)), rng.integers(0, 256, (120, 160, 3), dtype=np.uint8))
            frames.write_pnm(str(d / ("flow_x_%05d.ppm" % i)), rng.integers(0, 256, (120, 160), dtype=np.uint8))
            frames.write_pnm(str(d / ("flow_y_%05d.ppm" % i)), rng.integers(0, 256, (120, 160), dtype=np.uint8))
    protos = _write_protos(bi, tmp_path)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(repo, "video-query-algorithms_amd", "calcSig_wOF.py")
    outs = {}
    for world in (1, 2):
        out_dir = tmp_path / ("features_%d" % world)
        argv = [str(root), protos["rgb"], "synthetic:2", protos["flow"], "synthetic:5", "--num_frame_per_video", "3",
                "--outFeatures_dir", str(out_dir), "--modelname", "UCF101_split1", "--frame_ext", ".ppm", "--batch_clips", "2"]
        if world == 1:
            cmd = [sys.executable, cli] + argv
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", "29541", cli] + argv + ["--gpus", "0"]
        p = subprocess.run(cmd, env=dict(os.environ, VQ_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[world] = {f: open(os.path.join(str(out_dir), "vid", "UCF101_split1", f), "rb").read()
                       for f in ("rgb_global_pool_features.csv", "warped_optical_flow_global_pool_features.csv")}
    assert outs[1] == outs[2]
    rows = outs[2]["rgb_global_pool_features.csv"].decode().split("\n")
    assert [r.split(",")[0] for r in rows[1:-1]] == ["1", "2", "3", "4", "7"] and len(rows[1].split(",")) == 1025


def test_bench_two_rank_control_flow_rehearsal(tsn):
    """bench.py under torch.distributed.run with 2 and 4 ranks, all on this one GPU through gloo (VQ_BENCH_REHEARSE=1): the
    weak-scaling control flow the driver runs on 2/4/8 GPUs -- per-rank batches, feature all-gather, barriers,
    max-over-ranks timing, one JSON line from rank 0 that says what torch.distributed saw -- must hold together.  The numbers
    of such a run mean nothing; the line's arithmetic must still be SURVEY 8(d)'s (frac x peak x ms_per_step = 390.06 GFLOP)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VQ_BENCH_REHEARSE="1")
    for world, port in ((2, "29533"), (4, "29535")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--skip-sim",
               "--skip-cpu"]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        out = json.loads(lines[0])
        assert out["n_gpus"] == world and out["config"]["global_batch"] == 32 * world and out["scaling"] == "weak" and out["value"] > 0
        dist_cfg = out["config"]["distributed"]
        assert dist_cfg["world_size"] == world and dist_cfg["backend"] == "gloo" and dist_cfg["rehearsal_on_one_gpu"] is True and dist_cfg["rccl_version"]
        roof = out["roofline"]
        assert roof["launches_per_step"] == 36
        assert len(roof["rank_ms_per_step"]["all"]) == world and roof["rank_ms_per_step"]["min"] <= roof["rank_ms_per_step"]["max"]
        assert roof["all_gather_ms_per_step"] > 0
        assert 0 < roof["matrix_pipe_frac"] < 1 and roof["matrix_pipe_frac"] < roof["kernel_frac"] < 2
        assert abs(roof["frac"] * roof["peak"] * out["ms_per_step"] - 390.06) < 0.5          # TFLOP/s x ms = GFLOP per step
        # the N > 1 line carries the consistency fields of the N = 1 line (the SCALE N = 1 line and BENCH are comparable
        # by construction): the one-stream region with its own sampled steps, the per-kernel fields taken from it
        single = out["single_stream"]
        assert single["ms_per_step"] > 0 and single["profiled_steps"] == roof["profiled_steps"] == 2       # --steps 2: every step sampled
        assert single["conv_ms_per_step"] == roof["conv_ms_per_step"] and single["kernel_frac"] == roof["kernel_frac"]
        assert "single_stream region" in out["config"]["timed_mode"] and roof["traffic_per_step"] == pytest.approx(roof["traffic"] * 36, rel=1e-4)
        assert len(lines[0]) < 4096                                 # the driver's record keeps the whole line
    # ``python bench.py --gpus 2`` with NO launcher starts its own ranks (before any GPU call) and prints the same one line
    plain = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--skip-cpu"]
    p = subprocess.run(cmd, env=plain, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["similarity"]["config"]["rows_per_gpu"] == 500_000 and out["similarity"]["value"] > 0
    assert out["similarity"]["roofline"]["kernel"].startswith("scan_tiled_kernel") and out["similarity"]["batched"]["value"] > 0
    # a rank that dies where the communicator would be built: the parent ends its peers (they would wait in the rendezvous for
    # ever) and the command exits non-zero, well inside the timeout -- children are started fresh, nothing is re-executed
    import time
    t0 = time.perf_counter()
    p = subprocess.run(cmd + ["--skip-sim"], env=dict(plain, VQ_BENCH_FAIL_INIT_RANK="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "told to fail" in p.stderr and time.perf_counter() - t0 < 120
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_tiling_tables_survive_the_process(tsn, monkeypatch, tmp_path):
    """The first forward of a batch size autotunes (~2 s of launches); the table is kept per layer graph
    (VQ_TUNE_CACHE, default next to the library) and installed when the next handle of the same graph is created --
    the command line must not re-tune in every process.  Same bits with and without the cache."""
    import glob
    import json
    bi, net = tsn
    monkeypatch.setenv("VQ_TUNE_CACHE", str(tmp_path))
    monkeypatch.setenv("VQ_TSN_SPLIT", "2")
    g = bi.bn_inception(3)
    w = net.synthetic_weights(g, seed=2)
    crops = np.random.default_rng(3).integers(0, 256, (6, 224, 224, 3), dtype=np.uint8)
    m = net.TsnNet(g, w, max_crops=6)
    assert m.tuned_sizes() == []
    f1, p1 = m.forward(crops, 3, net.RGB_MEAN)
    assert m.tuned_sizes() == [3]                                  # two sub-batches of 3 crops
    tiles = m.layer_tiles(3)
    m.close()
    files = glob.glob(str(tmp_path / "*.json"))
    assert len(files) == 1 and list(json.load(open(files[0]))) == ["3p"]          # timed side by side on the two sub-batch streams
    m = net.TsnNet(g, w, max_crops=6)
    assert m.tuned_sizes() == [3] and (m.layer_tiles(3) == tiles).all()      # installed before any forward
    f2, p2 = m.forward(crops, 3, net.RGB_MEAN)
    m.close()
    assert (p1 == p2).all() and (f1 == f2).all()
    assert len(glob.glob(str(tmp_path / "*.json"))) == 1
    m = net.TsnNet(bi.bn_inception(10), net.synthetic_weights(bi.bn_inception(10), seed=2), max_crops=6)   # another graph: another file
    assert m.tuned_sizes() == []
    m.close()
    monkeypatch.setenv("VQ_TUNE_CACHE", "0")
    m = net.TsnNet(g, w, max_crops=6)
    assert m.tuned_sizes() == []
    m.close()
    # a handle whose table was set by hand keeps its choices to itself: neither that table nor a size tuned (or borrowed) after it is saved
    monkeypatch.setenv("VQ_TUNE_CACHE", str(tmp_path))
    before = open(files[0]).read()
    m = net.TsnNet(g, w, max_crops=12)
    forced = tiles.copy()
    conv = forced[:, 0] > 0
    forced[conv & (forced[:, 3] != 2), :] = (64, 64, 32, 0)
    m.set_layer_tiles(3, forced, paired=True)
    f3, _ = m.forward(crops, 3, net.RGB_MEAN)
    m.forward(np.concatenate([crops, crops]), 3, net.RGB_MEAN)                # 12 crops: sub-batches of 6, a size without a table
    assert sorted(m.tuned_sizes()) == [3, 6]
    m.close()
    assert (f3 == f1).all() and open(files[0]).read() == before
    m = net.TsnNet(g, w, max_crops=6, tune_cache="0")                        # the keyword beats the environment
    assert m.tuned_sizes() == []
    m.close()


def test_features_from_cached_packed_weights_equal_features_from_the_weights(tmp_path, monkeypatch):
    """CaffeNet on a weights FILE: the first handle packs and stores, the second takes the packed blob from the cache (its loader is never
    called), a third with the cache switched off packs again -- the same features, bit for bit."""
    from video_query_algorithms_amd.tsn import bn_inception, caffe_net
    from video_query_algorithms_amd.tsn.net import synthetic_weights
    g = bn_inception.bn_inception(3)
    path = str(tmp_path / "w.npz")
    caffe_net.save_weights(path, synthetic_weights(g, 4))
    crops = np.random.default_rng(3).integers(0, 256, (6, 224, 224, 3), dtype=np.uint8)
    monkeypatch.setenv("VQ_WEIGHT_CACHE", str(tmp_path / "wc"))
    loads = []
    real = caffe_net.load_weights
    monkeypatch.setattr(caffe_net, "load_weights", lambda graph, spec: (loads.append(spec), real(graph, spec))[1])
    feats = []
    for _ in range(2):
        net = caffe_net.CaffeNet(g, path, 0, max_crops=6)
        feats.append(net.extract_clips(crops, 3))
        net.close()
    assert loads == [path]
    monkeypatch.setenv("VQ_WEIGHT_CACHE", "0")
    net = caffe_net.CaffeNet(g, path, 0, max_crops=6)
    feats.append(net.extract_clips(crops, 3))
    net.close()
    assert loads == [path, path]
    assert (feats[0] == feats[1]).all() and (feats[0] == feats[2]).all() and np.isfinite(feats[0]).all()


def test_shipped_tiling_tables_make_a_cold_process_time_nothing(tsn, monkeypatch, tmp_path):
    """tsn/default_tiles.json (tools/make_default_tiles.py, measured on an MI355X) holds the tables of both
    BN-Inception streams at the BASELINE batch sizes; a handle with no tuning cache installs them at creation, so the first forward of
    a cold process sweeps nothing.  Sizes in between borrow (flagged, never persisted); VQ_TSN_AUTOTUNE=1 leaves the shipped tables
    out and sweeps; the bits never depend on any of it."""
    import glob
    import json
    bi, net = tsn
    monkeypatch.delenv("VQ_TSN_AUTOTUNE", raising=False)
    monkeypatch.delenv("VQ_TSN_SPLIT", raising=False)
    monkeypatch.setenv("VQ_TUNE_CACHE", str(tmp_path))
    shipped = json.load(open(net.DEFAULT_TILES_PATH))["tables"]
    feats = {}
    for ch in (3, 10):
        g = bi.bn_inception(ch)
        w = net.synthetic_weights(g, seed=2)
        mean = net.RGB_MEAN if ch == 3 else net.FLOW_MEAN
        crops = np.random.default_rng(ch).integers(0, 256, (60, 224, 224, ch), dtype=np.uint8)
        m = net.TsnNet(g, w, max_crops=96)
        assert m.graph_key in shipped and m.default_tables == 4
        assert sorted(m.tile_tables()) == [(24, True, False), (48, False, False), (48, True, False), (96, False, False)]
        before = {(n, p): m.layer_tiles(n, paired=p).copy() for n, p, _ in m.tile_tables()}
        f48, _ = m.forward(crops[:48], 3, mean)                        # sub-batches of 24: the shipped paired table
        assert sorted(m.tile_tables()) == sorted((n, p, False) for n, p in before)
        f60, _ = m.forward(crops, 3, mean)                             # sub-batches of 30: borrowed from 24p, nothing timed
        assert (30, True, True) in m.tile_tables() and (m.layer_tiles(30, paired=True) == before[(24, True)]).all()
        for key, t in before.items():
            assert (m.layer_tiles(key[0], paired=key[1]) == t).all()
        m.close()
        assert glob.glob(str(tmp_path / "*.json")) == []               # nothing measured: nothing to keep
        feats[ch] = (crops, mean, g, w, f48, f60)
    crops, mean, g, w, f48, f60 = feats[3]
    monkeypatch.setenv("VQ_TSN_AUTOTUNE", "1")                         # the refinement: shipped tables out, every size swept and kept
    m = net.TsnNet(g, w, max_crops=96)
    assert m.default_tables == 0 and m.tile_tables() == []
    g48, _ = m.forward(crops[:48], 3, mean)
    assert m.tile_tables() == [(24, True, False)]
    m.close()
    files = glob.glob(str(tmp_path / "*.json"))
    assert len(files) == 1 and list(json.load(open(files[0]))) == ["24p"]
    assert (g48 == f48).all()
    monkeypatch.setenv("VQ_TSN_AUTOTUNE", "0")                         # never time anything: shipped tables, heuristics elsewhere
    monkeypatch.setenv("VQ_TUNE_CACHE", "0")
    m = net.TsnNet(g, w, max_crops=96)
    assert m.default_tables == 4
    h60, _ = m.forward(crops, 3, mean)
    m.close()
    assert (h60 == f60).all()


def test_a_narrow_winograd_layer_at_a_large_batch_is_cut_into_crop_ranges(tsn, monkeypatch):
    """the Winograd kernel multiplies a slot's pixel indices on 24 bits (crops x H x W < 2^23 per launch).  With 64 or more
    channels per pixel the 2^31-byte rule for slots implies that; an 8-channel slot of 128 x 128 pixels reaches 2^23 pixels at 512 crops
    with 270 MB.  The executor caps the crops of such a launch (item_limit) and covers the batch in several crop ranges: 520 crops run,
    and the first and last crops carry the bits they carry in a batch of four."""
    bi, net = tsn
    monkeypatch.delenv("VQ_TSN_TILE", raising=False)
    h, cin, cout, n = 128, 8, 32, 520
    g = _mini(bi, cin, h, h, cout, 3, 1, 1)
    w = net.synthetic_weights(g, seed=11)
    rng = np.random.default_rng(12)
    few = rng.integers(0, 256, (4, h, h, cin), dtype=np.uint8)
    crops = np.empty((n, h, h, cin), dtype=np.uint8)
    crops[:] = few[0]
    crops[:2], crops[-2:] = few[:2], few[2:]
    mean = np.linspace(100.0, 130.0, cin).astype(np.float32)
    small = net.TsnNet(g, w, max_crops=4, feature_blob="gp", winograd=True)
    assert small.layer_tiles(4)[0, 3] == 2                                  # the layer is in Winograd form
    _, ps_small = small.forward(few, 1, mean)
    small.close()
    want = to.forward(g.layers, "data", w, to.preprocess(few[:1], mean), keep=("gp",))["gp"].reshape(1, -1)
    assert np.abs(ps_small[:1] - want).max() <= 2e-5 * np.abs(want).max()
    big = net.TsnNet(g, w, max_crops=n, feature_blob="gp", winograd=True)
    _, ps = big.forward(crops, 1, mean)
    big.close()
    assert (ps[:2] == ps_small[:2]).all() and (ps[-2:] == ps_small[2:]).all() and (ps[2:-2] == ps_small[0]).all()
