"""GPU: the two BASELINE.json configs that are about SIZE, at their full size.

configs[2]  "TSN two-stream RGB+flow (5-frame stack), 7 segments, batch=64": 448 crops per stream through the HIP
            BN-Inception, against the fp64 CPU evaluation of the layer list on an 8-crop subset per stream, plus the
            size-independent properties at the full batch (bit-determinism, batch-composition invariance, the fp64
            consensus recomputed from the per-snippet features, the data/features CSV layout read back with load_db's
            rules).  The oracle is parity-unpinned for the network arithmetic: nothing in the reference
            holds frames, weights or forward-pass outputs.
configs[4]  "End-to-end: 10k synthetic clips -> load_db layout -> 100 compute_matches weight updates": the tool
            tools/e2e_cfg5.py at 2 000 clips x 10 rounds with EVERY round checked against oracle/sim_oracle.py (the
            faithful restatements of hyperparameter.py:29-114 and ticket.py:165-180,311-356, pinned by reference-run
            goldens), and at the full 10 000 clips x 100 rounds through properties.
"""
import json
import os
import random
import sys

import numpy as np
import pytest

import sim_oracle as so
import tsn_oracle as to
from _helpers import STREAMS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T = 64, 7


@pytest.fixture(scope="module")
def tsn(gpu):
    from video_query_algorithms_amd.tsn import bn_inception, net
    return bn_inception, net


@pytest.mark.parametrize("channels", [3, 10])
def test_cfg3_two_stream_batch_64_by_7(tsn, channels, tmp_path):
    bi, net = tsn
    from video_query_algorithms_amd.tsn import feature_csv
    g = bi.bn_inception(channels)
    w = net.synthetic_weights(g, seed=2 if channels == 3 else 5)
    mean = net.RGB_MEAN if channels == 3 else net.FLOW_MEAN
    crops = np.random.default_rng(30 + channels).integers(0, 256, (B * T, 224, 224, channels), dtype=np.uint8)
    m = net.TsnNet(g, w, max_crops=B * T)
    feat, ps = m.forward(crops, T, mean)
    assert feat.shape == (B, 1024) and ps.shape == (B * T, 1024)
    assert np.isfinite(ps).all() and (ps >= 0).all() and ps.max() > 0            # post-ReLU averages
    # the consensus is the fp64 mean of the fp32 per-snippet blobs (calcSig_wOF.py:82), bit for bit
    assert (feat == to.consensus(ps, T)).all()
    # same bits on a second pass, and for any batch the crops travel in (every tiling sums in the same order)
    feat2, ps2 = m.forward(crops, T, mean)
    assert (ps2 == ps).all() and (feat2 == feat).all()
    f_small, p_small = m.forward(crops[:2 * T], T, mean)
    assert (p_small == ps[:2 * T]).all() and (f_small == feat[:2]).all()
    f_tail, p_tail = m.forward(crops[-3 * T:], T, mean)
    assert (p_tail == ps[-3 * T:]).all() and (f_tail == feat[-3:]).all()
    m.close()
    # oracle parity: 8 crops spread over the batch, fp64 evaluation of the prototxt layer list on the CPU
    pick = np.array([0, 1, 7, 100, 223, 224, 300, B * T - 1])
    ref = to.forward(g.layers, "data", w, to.preprocess(crops[pick], mean), keep=("global_pool",))["global_pool"].reshape(len(pick), -1)
    err = np.abs(ps[pick] - ref).max() / np.abs(ref).max()
    assert err <= 2e-4, err                                                       # stated whole-network tolerance
    # data/features layout: written like writeFeatures, read back like load_db (api_load_records.py:41-58)
    mode = "rgb" if channels == 3 else "warped_optical_flow"
    names = ["clip_%04d" % (i + 1) for i in range(B)]
    files = feature_csv.write_features(str(tmp_path), "synthetic_video", "/synthetic/", "UCF101_split3", "global_pool", names,
                                       {mode: feat}, {mode: "synthetic.caffemodel"})
    nsplit, per_stream = feature_csv.read_split_dir(os.path.dirname(files[0]))
    clips, back, meta = per_stream[mode]
    assert nsplit == 3 and clips.tolist() == list(range(1, B + 1)) and (back == feat).all() and back.shape == (B, 1024)
    assert meta["video"] == "synthetic_video" and meta["feature_name"] == "global_pool"


class _RoundChecker:
    """Every transition of a cfg-5 round against the oracle.  Inputs of a transition are the PRODUCT's state before
    it (teacher forcing), so a rounding-level difference in one round cannot cascade into the next."""

    def __init__(self, labels_per_round):
        self.labels_per_round = labels_per_round
        self.rounds = 0
        self.worst_w = self.worst_th = 0.0

    def start(self, feats, clip_ids, tk, hp):
        self.ids = np.asarray(clip_ids)
        x = feats.cpu().numpy()
        row = int(np.flatnonzero(self.ids == tk.ref_clip_id)[0])
        t = np.stack([[so.scale_feature(x[row, s, e].astype(np.float64)) for e in range(x.shape[2])] for s in range(x.shape[1])])
        self.o_avg = so.dense_similarities(x, t)[1]
        self.ref_row = row

    def round(self, r, tk, hp, prev_w, prev_th, labels, rng_state):
        avg = tk._avg
        assert np.abs(avg - self.o_avg).max() <= 1e-12                           # ticket.py:120-163
        w_prev = [prev_w[st] for st in STREAMS]
        s_prev = so.dense_scores(avg, w_prev)
        # the user's review set: top scores under the previous weights (ranks identical to a stable sort)
        want_rows, _ = so.dense_topk(s_prev, self.labels_per_round)
        assert [m["video_clip"] for m in labels] == self.ids[want_rows].tolist()
        assert all(m["is_match"] == bool(s_prev[row] >= prev_th) for m, row in zip(labels, want_rows))
        # hyperparameter.py:29-114 on the labelled clips (the loss reads nothing else)
        sims = {int(self.ids[row]): {st: [avg[row, si], 3] for si, st in enumerate(STREAMS)} for row in want_rows}
        o_w, o_th, _, _ = so.faithful_optimize_weights(sims, labels, STREAMS, hp.ballast, float(os.environ["COMPUTE_EPS"]))
        self.worst_w = max(self.worst_w, abs(o_w[STREAMS[1]] - hp.weights[STREAMS[1]]))
        self.worst_th = max(self.worst_th, abs(o_th - hp.threshold))
        assert abs(o_w[STREAMS[1]] - hp.weights[STREAMS[1]]) <= 1e-9 and o_w[STREAMS[0]] == hp.weights[STREAMS[0]] == 1.0
        assert abs(o_th - hp.threshold) <= 1e-9
        # ticket.py:165-180 under the new weights: bit-exact given the averages
        s_new = so.dense_scores(avg, [hp.weights[st] for st in STREAMS])
        assert (tk._score_values == s_new).all()
        # ticket.py:311-356 from the same generator state
        after = random.getstate()
        random.setstate(rng_state)
        o_matches = so.faithful_select(dict(zip(self.ids.tolist(), s_new)), tk.ref_clip_id, tk.user_matches, hp.threshold, 20,
                                       hp.near_miss_default)
        assert random.getstate() == after                                        # the product drew exactly as often
        assert list(tk.matches.items()) == list(o_matches.items())
        self.rounds += 1


def _run_cfg5(argv, capsys, observer=None):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_cfg5
    assert e2e_cfg5.main(argv, observer=observer) == 0
    return json.loads(capsys.readouterr().out.strip().splitlines()[-1])


def test_cfg5_every_round_against_the_oracle(tsn, tmp_path, capsys):
    chk = _RoundChecker(20)
    out = _run_cfg5(["--clips", "2000", "--segments", "7", "--batch-clips", "64", "--rounds", "10", "--labels", "20", "--csv-clips", "64",
                     "--out", str(tmp_path)], capsys, chk)
    assert chk.rounds == 10 and out["clips"] == 2000 and out["csv_clips"] == 64
    print("cfg5 oracle rounds: max |dw| %.2e, max |dth| %.2e" % (chk.worst_w, chk.worst_th))


class _Properties:
    def __init__(self):
        self.rounds = 0
        self.weights = []

    def start(self, feats, clip_ids, tk, hp):
        self.grid = hp.weight_grid, hp.threshold_grid

    def round(self, r, tk, hp, prev_w, prev_th, labels, rng_state):
        wg, tg = self.grid
        w = hp.weights[STREAMS[1]]
        assert hp.weights[STREAMS[0]] == 1.0 and wg[0] <= w <= wg[-1]
        assert tg[0] - 1e-5 <= hp.threshold <= tg[-1]
        sc = tk._score_values
        assert np.isfinite(sc).all() and sc.max() <= 1.0 + 1e-12
        assert abs(tk.scores[tk.ref_clip_id] - 1.0) <= 1e-12 and tk.ref_clip_id in tk.matches
        assert 1 <= len(tk.matches) <= 21                                        # 20 + the forced reference clip
        lower = hp.threshold - hp.near_miss_default * (1 - hp.threshold)
        assert all(v >= lower or c == tk.ref_clip_id for c, v in tk.matches.items())
        assert len(labels) == 20 and len({m["video_clip"] for m in labels}) == 20
        self.weights.append(w)
        self.rounds += 1


def test_cfg5_full_size_ten_thousand_clips_hundred_rounds(tsn, tmp_path, capsys):
    """BASELINE configs[4] at full size on one GPU (the 8-GPU run shards the clips; tests/test_cli_gloo.py and
    test_shard_gloo.py cover that control flow): 10 000 clips x 7 segments x 2 streams x 3 weight seeds through the
    TSN kernels into the resident DB, 200 clips through the CSV tree and back, 100 weight-update rounds."""
    prop = _Properties()
    out = _run_cfg5(["--clips", "10000", "--segments", "7", "--batch-clips", "64", "--rounds", "100", "--labels", "20",
                     "--csv-clips", "200", "--out", str(tmp_path)], capsys, prop)
    assert prop.rounds == 100 and out["clips"] == 10000 and out["csv_clips"] == 200
    assert os.path.exists(os.path.join(str(tmp_path), "synthetic_video", "UCF101_split1", "rgb_global_pool_features.csv"))
    assert 0.5 <= out["final_weights"]["warped_optical_flow"] <= 2.45 + 1e-9
    print("cfg5 full size: %.1f s extraction, %.3f s for 100 rounds" % (out["extract_total_s"], out["rounds_total_s"]))


def test_cfg5_sharded_rounds_equal_one_gpu(tsn, tmp_path):
    """BASELINE configs[4] in its N > 1 form, rehearsed with 2 and 3 ranks on this one card (gloo): clips sharded for the extraction, the
    feature blocks STAY where they were produced as the rows of a ShardedFeatureDB (no feature gather), the weight-update rounds run on
    all ranks through the same Ticket / Hyperparameter code.  A clip's features do not depend on the batch it travelled in and its score
    not on the shard it sits in, so fitted weights, threshold and review-set size equal the one-GPU run's exactly."""
    import subprocess
    argv = ["--clips", "150", "--segments", "3", "--batch-clips", "16", "--rounds", "6", "--labels", "20", "--csv-clips", "16"]
    tool = os.path.join(ROOT, "tools", "e2e_cfg5.py")

    def run(world):
        env = dict(os.environ, VQ_DIST_BACKEND="gloo", VQ_CFG5_ONE_CARD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, tool] if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                                                         "--master-addr", "127.0.0.1", "--master-port", str(29600 + world), tool]
        p = subprocess.run(cmd + argv + ["--out", str(tmp_path / ("w%d" % world))], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    one = run(1)
    for world in (2, 3):
        got = run(world)
        assert got["world"] == world and got["features_gathered"] is False
        assert got["final_weights"] == one["final_weights"] and got["final_threshold"] == one["final_threshold"]
        assert got["final_matches"] == one["final_matches"] and got["rounds"] == 6


def test_a_1600_crop_batch_crosses_the_32_bit_offset_limit_correctly(tsn):
    """`--batch_clips 64 --num_frame_per_video 25` = 1 600 crops per forward: conv1's output slot is 5.1 GB, the
    kernels address a slot with signed 32-bit byte offsets, so the executor covers such a slot with several launches over
    crop ranges.  The features of crops on both sides of every cut must equal a small-batch run bit for bit."""
    bi, net = tsn
    g = bi.bn_inception(3)
    w = net.synthetic_weights(g, seed=2)
    n, T = 1600, 25
    rng = np.random.default_rng(77)
    crops = rng.integers(0, 256, (n, 224, 224, 3), dtype=np.uint8)
    m = net.TsnNet(g, w, max_crops=n)
    feat, ps = m.forward(crops, T, net.RGB_MEAN)
    assert np.isfinite(ps).all() and (feat == to.consensus(ps, T)).all()          # T = 25 > 16: the unfused consensus path
    for lo in (0, 650, 1325, 1575):                                               # spans the cuts at 668 and 1336 crops (conv1)
        f_small, p_small = m.forward(crops[lo:lo + 25], T, net.RGB_MEAN)
        assert (p_small == ps[lo:lo + 25]).all(), lo
    m.close()
