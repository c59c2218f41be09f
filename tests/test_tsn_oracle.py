"""CPU: pins the pieces of the hot-path-A oracle that CAN be pinned here -- the torch calls against loop-level
restatements of the Caffe layer rules, the driver logic against calcSig_wOF.py's documented cases, and the
built-in graph against the reference's own prototxt (when the reference checkout is present)."""
import os

import numpy as np
import pytest

import tsn_oracle as to

REF_PROTO = "/root/reference/src/features_GPU_compute/models/ucf101/tsn_bn_inception_%s_deploy.prototxt"


def test_frame_ticks_follow_calcsig():
    # cnt=150: T=25 -> 1,7,...,145; T=7 -> 1,25,...,145; T=3 -> 1,75,149 (flow depth 5: 1,73,145)
    assert to.frame_ticks(150, 25, 1) == list(range(1, 146, 6))
    assert to.frame_ticks(150, 7, 1) == [1, 25, 49, 73, 97, 121, 145]
    assert to.frame_ticks(150, 3, 1) == [1, 75, 149]
    assert to.frame_ticks(150, 3, 5) == [1, 73, 145]
    assert to.frame_ticks(150, 25, 5) == list(range(1, 146, 6))
    assert to.frame_ticks(10, 25, 1) == [1] * 25                      # step == 0
    assert to.frame_ticks(5, 3, 5) == [1, 1, 1]
    with pytest.raises(ZeroDivisionError):
        to.frame_ticks(150, 1, 1)                                      # the reference divides by zero too
    assert to.flow_stack_indices(148, 150, 5) == [148, 149, 150, 150, 150]


def test_pool_output_sizes_caffe_ceil_rule():
    sizes = [112]
    for _ in range(4):
        sizes.append(to.pool_out(sizes[-1], 3, 2, 0))
    assert sizes == [112, 56, 28, 14, 7]
    assert to.pool_out(28, 3, 1, 1) == 28 and to.pool_out(7, 3, 1, 1) == 7 and to.pool_out(7, 7, 1, 0) == 1
    assert to.conv_out(224, 7, 2, 3) == 112 and to.conv_out(28, 3, 2, 1) == 14 and to.conv_out(14, 3, 2, 1) == 7


def test_torch_ops_match_loop_level_caffe_rules():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 5, 9, 11))
    for (k, s, p) in [(3, 1, 1), (3, 2, 1), (7, 2, 3), (1, 1, 0)]:
        w = rng.standard_normal((4, 5, k, k))
        b = rng.standard_normal(4)
        got = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=s, padding=p).numpy()
        assert np.abs(got - to.conv_direct(x, w, b, s, p)).max() < 1e-12
    for (k, s, p, mode) in [(3, 2, 0, "MAX"), (3, 1, 1, "AVE"), (3, 1, 1, "MAX"), (9, 1, 0, "AVE")]:
        xx = x if k != 9 else x[:, :, :, :9]
        t = torch.from_numpy(xx)
        got = (F.max_pool2d(t, k, s, p, ceil_mode=True) if mode == "MAX"
               else F.avg_pool2d(t, k, s, p, ceil_mode=True, count_include_pad=True)).numpy()
        want = to.pool_direct(xx, k, s, p, mode)
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-12
    # odd size with stride 2: ceil mode adds the partial window (8 -> 4 with k=3? (8-3)/2 = 2.5 -> 3+1 = 4)
    t = torch.from_numpy(x[:, :, :8, :8])
    assert F.max_pool2d(t, 3, 2, 0, ceil_mode=True).shape[-1] == to.pool_out(8, 3, 2, 0) == 4


def test_consensus_is_fp64_mean_of_fp32_snippets():
    rng = np.random.default_rng(1)
    ps = rng.random((6, 16)).astype(np.float32)
    c = to.consensus(ps, 3)
    assert c.dtype == np.float64 and c.shape == (2, 16)
    want = (ps[0].astype(np.float64) + ps[1] + ps[2]) / 3
    assert (c[0] == want).all()


def test_builtin_graph_equals_reference_prototxt():
    if not os.path.exists(REF_PROTO % "rgb"):
        pytest.skip("reference checkout not present (GPU box)")
    from video_query_algorithms_amd.tsn import bn_inception as bi
    for name, c in (("rgb", 3), ("flow", 10)):
        ref = bi.load_prototxt(REF_PROTO % name)
        assert ref.layers == bi.bn_inception(c).layers and ref.input_shape == (c, 224, 224)


def test_plan_matches_survey_appendix_a():
    from video_query_algorithms_amd.tsn import bn_inception as bi
    p = bi.bn_inception(3).plan(fuse=False)
    assert p.macs_per_crop() == 2_031_576_064                 # SURVEY Appendix A (fc-action excluded)
    assert bi.bn_inception(10).plan(fuse=False).macs_per_crop() == 2_306_941_952
    kinds = [o.kind for o in p.ops]
    assert kinds.count("conv") == 69 and kinds.count("avgpool") == 7 and kinds.count("maxpool") == 5
    assert kinds.count("gavgpool") == 1 and p.feature_dim == 1024
    # concat is zero-copy: 3a's four branches land at offsets 0, 64, 128, 224 of one 256-channel slot
    loc = p.blob_loc
    s = loc["inception_3a/output"][0]
    assert [loc[b] for b in ("inception_3a/1x1_bn", "inception_3a/3x3_bn", "inception_3a/double_3x3_2_bn",
                             "inception_3a/pool_proj_bn")] == [(s, 0, 64), (s, 64, 64), (s, 128, 96), (s, 224, 32)]
    assert p.tensors[s].c == 256 and (p.tensors[s].h, p.tensors[s].w) == (28, 28)
    # reduction block 3c: max-pool passes through un-projected at offset 160+96
    s3c = loc["inception_3c/output"][0]
    assert loc["inception_3c/pool"] == (s3c, 256, 320) and p.tensors[s3c].c == 576


def test_fused_plan_keeps_the_work_and_the_destinations():
    """Graph-level rewrites (sibling 1x1 merge, avg-pool / projection commute) change no MAC and no destination."""
    from video_query_algorithms_amd.tsn import bn_inception as bi
    for c, macs in ((3, 2_031_576_064), (10, 2_306_941_952)):
        g = bi.bn_inception(c)
        p0, p = g.plan(fuse=False), g.plan()
        assert p.macs_per_crop() == macs == p0.macs_per_crop()
        kinds = [o.kind for o in p.ops]
        both = bi.FOLD_POOL_MAX_COUT >= 224
        assert kinds.count("conv") == 44 and kinds.count("avgpool") == 7 and kinds.count("maxpool") == (3 if both else 4)
        # the stem max pools live in the loaders of the 1x1 GEMMs behind them (pool1: the 64-column conv2/3x3_reduce; pool2: the
        # 224-column sibling group of inception_3a, ONE 256-column tile); the pooled tensors are stubs
        folded = {o.pre_pool[2]: o for o in p.ops if o.pre_pool}
        assert set(folded) == ({"pool1/3x3_s2", "pool2/3x3_s2"} if both else {"pool1/3x3_s2"}) and folded["pool1/3x3_s2"].name == "conv2/3x3_reduce"
        for name, o in folded.items():
            pool = next(q for q in p0.ops if q.name == name)
            assert (o.src, o.src_coff, o.pre_pool[:2]) == (pool.src, pool.src_coff, (3, 2)) and name not in p.blob_loc
            assert (p.tensors[pool.dst].h, p.tensors[pool.dst].w, p.tensors[pool.dst].c) == (1, 1, 4)
        plain = {o.name: o for o in p0.ops if o.kind == "conv"}
        seen = set()
        for o in p.ops:
            if o.kind != "conv":
                continue
            for sg in (o.segments or [bi.Segment(o.name, o.bn, o.cout, o.dst, o.dst_coff, o.relu, o.bias)]):
                ref = plain[sg.name]
                seen.add(sg.name)
                assert sg.cout == ref.cout and o.cin == ref.cin and o.k == ref.k
                if sg.bias:                       # un-commuted: same destination, same activation
                    assert (sg.dst, sg.dst_coff, sg.relu) == (ref.dst, ref.dst_coff, ref.relu)
                else:                             # commuted projection: raw output to a temp, finished by the pool
                    fin = [q for q in p.ops if q.kind == "avgpool" and q.bias_from and q.bias_from[0] == sg.name]
                    assert len(fin) == 1 and (fin[0].dst, fin[0].dst_coff, fin[0].relu) == (ref.dst, ref.dst_coff, ref.relu)
                    assert fin[0].src == sg.dst and not sg.relu
        assert seen == set(plain)
        # 5b keeps its max pool in front of the projection (max does not commute with a linear map)
        assert any(o.kind == "maxpool" and o.name == "inception_5b/pool" for o in p.ops)


def test_prototxt_parser_handles_comments_and_nesting():
    from video_query_algorithms_amd.tsn import bn_inception as bi
    text = '''name: "tiny"  # a comment
    input: "data" input_dim: 1 input_dim: 3 input_dim: 8 input_dim: 8
    layer { name: "c" type: "Convolution" bottom: "data" top: "c"
      param { lr_mult: 1 } convolution_param { num_output: 32 pad: 1 kernel_size: 3 weight_filler { type: "xavier" } } }
    layer { name: "c_bn" type: "BN" bottom: "c" top: "c_bn" bn_param { frozen: true } }
    layer { name: "r" type: "ReLU" bottom: "c_bn" top: "c_bn" }
    layer { name: "gp" type: "Pooling" bottom: "c_bn" top: "gp" pooling_param { pool: AVE kernel_size: 8 stride: 1 } }
    '''
    g = bi.parse_prototxt(text)
    assert g.input_shape == (3, 8, 8) and [l.type for l in g.layers] == ["Convolution", "BN", "ReLU", "Pooling"]
    p = g.plan("gp")
    assert [o.kind for o in p.ops] == ["conv", "gavgpool"] and p.ops[0].relu and p.ops[0].bn == "c_bn"
    assert p.feature_dim == 32


def test_winograd_filter_transform_and_layout():
    """net.winograd_filters: U = G g G^T in the device layout [Cin/8][16][Cout][8].  Evaluating F(2x2,3x3) with it
    (numpy, fp64) reproduces the direct 3x3 / pad 1 correlation of the oracle."""
    import video_query_algorithms_amd  # noqa: F401
    from video_query_algorithms_amd.tsn import net
    rng = np.random.default_rng(5)
    cin, cout, h = 16, 32, 7
    W = rng.standard_normal((cout, cin, 3, 3))
    x = rng.standard_normal((2, cin, h, h))
    U = net.winograd_filters(W)
    assert U.shape == (cin // 8, 16, cout, 8) and U.dtype == np.float32
    Uf = U.astype(np.float64).transpose(2, 0, 3, 1).reshape(cout, cin, 4, 4)          # [o][c][i][j]
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
    th = (h + 1) // 2
    xp = np.zeros((2, cin, 2 * th + 2, 2 * th + 2))
    xp[:, :, 1:h + 1, 1:h + 1] = x
    y = np.zeros((2, cout, 2 * th, 2 * th))
    for ty in range(th):
        for tx in range(th):
            d = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
            V = np.einsum("ir,ncrs,js->ncij", Bt, d, Bt)
            M = np.einsum("ncij,ocij->noij", V, Uf)
            y[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ai,noij,bj->noab", At, M, At)
    want = to.conv_direct(x, W, np.zeros(cout), 1, 1)
    assert np.abs(y[:, :, :h, :h] - want).max() <= 1e-6 * np.abs(want).max()     # U is rounded to fp32


def test_space_to_depth_stem_rewrite_is_exact():
    """net.s2d_stem_weights + the space-to-depth input layout (include/vq_amd.h: vq_input_desc.s2d_pad): a k x k /
    stride-2 / pad-p convolution equals the ceil(k/2)^2 / stride-1 / pad-0 convolution over the decimated input --
    same products, so the fp64 results agree to rounding-order level."""
    import video_query_algorithms_amd  # noqa: F401
    from video_query_algorithms_amd.tsn import net
    rng = np.random.default_rng(8)
    for (c, h, k, p, cout) in [(3, 32, 7, 3, 8), (3, 17, 7, 3, 4), (2, 12, 3, 1, 4), (3, 20, 5, 2, 4)]:
        x = rng.standard_normal((2, c, h, h))
        W = rng.standard_normal((cout, c, k, k))
        want = to.conv_direct(x, W, np.zeros(cout), 2, p)
        out = want.shape[2]
        k2 = (k + 1) // 2
        hs = out + k2 - 1
        S = np.zeros((2, hs, hs, 4 * c))
        for Y in range(hs):
            for X in range(hs):
                for pp in range(2):
                    for q in range(2):
                        y, xx = 2 * Y + pp - p, 2 * X + q - p
                        if 0 <= y < h and 0 <= xx < h:
                            S[:, Y, X, (pp * 2 + q) * c:(pp * 2 + q + 1) * c] = x[:, :, y, xx]
        W2 = net.s2d_stem_weights(W)                                   # [cout][k2][k2][4c]
        got = to.conv_direct(S.transpose(0, 3, 1, 2), W2.transpose(0, 3, 1, 2), np.zeros(cout), 1, 0)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


def test_to_prototxt_round_trips_through_the_reader():
    from video_query_algorithms_amd.tsn import bn_inception as bi
    for c in (3, 10):
        g = bi.bn_inception(c)
        back = bi.parse_prototxt(bi.to_prototxt(g))
        assert back.layers == g.layers and back.input_shape == g.input_shape


def test_the_shipped_tiling_tables_are_well_formed():
    """video-query-algorithms_amd/tsn/default_tiles.json: one table set per BN-Inception stream, tables named "<crops>" (one stream) or
    "<crops>p" (timed side by side on the two sub-batch streams), [BM, BN, BK, pipelined] per layer with the layer counts of the lowered
    graphs; paired sizes are the halves of the one-stream sizes."""
    import json
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-query-algorithms_amd", "tsn", "default_tiles.json")
    doc = json.load(open(path))
    assert len(doc["tables"]) == 2 and doc["sizes"] == [48, 96, 224, 400, 448, 800]
    for key, tab in doc["tables"].items():
        assert re.fullmatch(r"[0-9a-f]{20}", key)
        alone = sorted(int(n) for n in tab if re.fullmatch(r"\d+", n))
        paired = sorted(int(n[:-1]) for n in tab if re.fullmatch(r"\d+p", n))
        assert alone == doc["sizes"] and paired == [n // 2 for n in doc["sizes"]]
        lengths = {len(t) for n, t in tab.items() if n != "one_stream"}
        assert len(lengths) == 1
        for n, t in tab.items():
            if n == "one_stream":
                assert all(int(v) in doc["sizes"] for v in t)
                continue
            rows = np.array(t)
            assert rows.shape[1] == 4 and (rows >= 0).all()
            conv = rows[rows[:, 0] > 0]
            assert len(conv) >= 40 and set(conv[:, 3].tolist()) <= {0, 1, 2, 3}
            wino = conv[conv[:, 3] == 2]
            assert len(wino) == 27 and set(wino[:, 1].tolist()) <= {32, 64}
            # units of 32 tiles (128 pixels, 8 channels per step) or -- 14 x 14 and 7 x 7 layers only -- of 16 tiles (64 pixels, 16 per step)
            assert set(map(tuple, wino[:, [0, 2]].tolist())) <= {(128, 8), (64, 16)}
