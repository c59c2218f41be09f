"""CPU, world_size 2 over gloo: the sharded path of the drop-in ``calcSig_wOF`` command line itself -- clip shards per
rank (``shard_range``), the all-gather of the per-rank feature blocks, rank 0 writing the CSV tree -- must give the
SAME BYTES as the one-rank run (calcSig_wOF.py:195-221 + the clip-level data parallelism of :204-210).

There is no GPU here and the product has no CPU fallback, so the extractor is a stand-in defined in this file (a
deterministic function of the decoded crops); the arithmetic of the real one is checked on the GPU
(tests/test_tsn_gpu.py).  What this test pins is the control flow around it: ragged shards (7 clips over 2 ranks),
batches that do not divide a shard, several videos, both streams, the ``--gpus`` -> device mapping."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEN_DEVICES = "devices_%d.txt"


from _cli_standin import StandInNet  # noqa: E402


def _make_tree(root):
    from video_query_algorithms_amd.tsn import frames
    rng = np.random.default_rng(11)
    for video, clips in (("videoA", {"clip_0001": 6, "clip_0002": 7, "clip_0003": 6, "clip_0005": 9, "clip_0008": 6,
                                     "clip_0013": 6, "clip_0021": 8}),
                         ("videoB", {"clip_0002": 6, "clip_0004": 6, "clip_0006": 6}),
                         ("videoC", {"clip_0001": 7})):                 # fewer clips than ranks: a rank owns nothing
        for clip, n in clips.items():
            d = os.path.join(root, video, clip)
            os.makedirs(d)
            for i in range(1, n + 1):
                frames.write_pnm(os.path.join(d, "img_%05d.ppm" % i), rng.integers(0, 256, (24, 32, 3), dtype=np.uint8))
                frames.write_pnm(os.path.join(d, "flow_x_%05d.ppm" % i), rng.integers(0, 256, (24, 32), dtype=np.uint8))
                frames.write_pnm(os.path.join(d, "flow_y_%05d.ppm" % i), rng.integers(0, 256, (24, 32), dtype=np.uint8))


def _argv(frames_root, out_dir, gpus=("4", "6"), workers="2"):
    return [frames_root, "rgb.prototxt", "ucf101_split1_tsn_rgb_bn_inception_wOF.caffemodel", "flow.prototxt",
            "ucf101_split1_tsn_flow_bn_inception_wOF.caffemodel", "--num_frame_per_video", "3", "--outFeatures_dir", out_dir,
            "--modelname", "UCF101_split1", "--frame_ext", ".ppm", "--batch_clips", "2", "--num_worker", workers, "--host_resize",
            "--gpus"] + list(gpus)


def _rank(rank, world, port, frames_root, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), VQ_DIST_BACKEND="gloo")
    from video_query_algorithms_amd import calcSig_wOF
    devices = []

    def factory(*a, **kw):
        net = StandInNet(*a, **kw)
        devices.append(net.device)
        return net
    assert calcSig_wOF.main(_argv(frames_root, out_dir), net_factory=factory) == 0
    with open(os.path.join(out_dir, SEEN_DEVICES % rank), "w") as f:
        f.write(",".join(map(str, devices)))
    import torch.distributed as dist
    dist.destroy_process_group()


def _tree_bytes(out_dir):
    found = {}
    for dirpath, _, files in os.walk(out_dir):
        for fn in files:
            if fn.endswith(".csv"):
                found[os.path.relpath(os.path.join(dirpath, fn), out_dir)] = open(os.path.join(dirpath, fn), "rb").read()
    return found


def test_two_rank_cli_writes_the_same_bytes_as_one_rank(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    from video_query_algorithms_amd import calcSig_wOF
    frames_root = str(tmp_path / "frames")
    _make_tree(frames_root)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    one = str(tmp_path / "one")
    assert calcSig_wOF.main(_argv(frames_root, one), net_factory=StandInNet) == 0      # a caller's own extractor: no fan-out
    want = _tree_bytes(one)
    assert sorted(want) == ["video%s/UCF101_split1/%s_global_pool_features.csv" % (v, m) for v in "ABC"
                            for m in ("rgb", "warped_optical_flow")]
    rows = want["videoA/UCF101_split1/rgb_global_pool_features.csv"].decode().split("\n")
    assert [r.split(",")[0] for r in rows[1:-1]] == ["1", "2", "3", "5", "8", "13", "21"]     # calcSig_wOF.py:200
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    two = str(tmp_path / "two")
    os.makedirs(two)
    mp.spawn(_rank, args=(2, port, frames_root, two), nprocs=2, join=True)
    assert _tree_bytes(two) == want
    # worker -> GPU map of calcSig_wOF.py:50-55: rank g of the node takes gpu_list[g % len]
    assert open(os.path.join(two, SEEN_DEVICES % 0)).read() == "4,4"          # one net per stream
    assert open(os.path.join(two, SEEN_DEVICES % 1)).read() == "6,6"


def _run_cli(argv, env_extra, timeout=300):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VQ_FANOUT_CHILD")}
    env.update(VQ_DIST_BACKEND="gloo", **env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_cli_standin.py")] + argv, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_the_ensemble_scripts_own_command_line_fans_out_over_the_gpus_it_names(tmp_path):
    """calcSig_wOF_ensemble.sh:13-19 runs ``python calcSig_wOF.py ... --num_worker 24 ... --gpus 0 1 ...`` with NO launcher;
    the reference maps worker i to gpu_list[(i-1) % len] itself (calcSig_wOF.py:44-56, 204-210).  Spelled that way the
    drop-in must start one process per GPU and write the byte-identical CSV tree of the one-GPU run -- here with 2 and with
    4 GPUs named (videoC has ONE clip: with 4 ranks three of them own nothing of it, with 2 ranks one), and with a GPU
    named twice (two workers share it: still one process on it)."""
    frames_root = str(tmp_path / "frames")
    _make_tree(frames_root)
    one = str(tmp_path / "one")
    r = _run_cli(_argv(frames_root, one, gpus=("0",), workers="24"), {})
    assert r.returncode == 0, r.stdout
    want = _tree_bytes(one)
    assert len(want) == 6 and "one process per GPU" not in r.stdout
    for gpus in (("0", "1"), ("3", "2", "1", "0"), ("5", "5", "7")):
        out = str(tmp_path / ("fan%d" % len(gpus)))
        log = str(tmp_path / ("devices%d" % len(gpus)))
        r = _run_cli(_argv(frames_root, out, gpus=gpus, workers="24"), {"STANDIN_DEVICE_LOG": log})
        assert r.returncode == 0, r.stdout
        assert "one process per GPU" in r.stdout
        assert _tree_bytes(out) == want
        for rank, g in enumerate(dict.fromkeys(gpus)):       # rank i of the fan-out sits on the i-th DISTINCT GPU the workers reach
            assert open("%s.%d" % (log, rank)).read().split() == [g, g]          # one net per stream, both on that GPU


def test_workers_reach_only_the_gpus_the_reference_would_give_them(tmp_path):
    """--num_worker 1 --gpus 0 1: the reference's only worker takes gpu_list[0] (calcSig_wOF.py:47-55): no fan-out.
    --num_worker 3 --gpus 5 7 9 11: workers 1..3 -> GPUs 5, 7, 9."""
    from video_query_algorithms_amd import fanout
    assert fanout.worker_devices([0, 1], 1) == [0]
    assert fanout.worker_devices([5, 7, 9, 11], 3) == [5, 7, 9]
    assert fanout.worker_devices([0, 1, 2, 3, 4, 5, 6, 7], 24) == list(range(8))
    assert fanout.worker_devices([2, 2, 3], 5) == [2, 3]
    assert fanout.worker_devices(None, 4, visible=8) == [0, 1, 2, 3] and fanout.worker_devices(None, 24, visible=8) == list(range(8))
    assert fanout.worker_devices(None, 1, visible=8) == [0]
    frames_root = str(tmp_path / "frames")
    _make_tree(frames_root)
    r = _run_cli(_argv(frames_root, str(tmp_path / "o"), gpus=("0", "1"), workers="1"), {})
    assert r.returncode == 0 and "one process per GPU" not in r.stdout


def test_a_failing_rank_fails_the_command(tmp_path):
    """The reference aborts on any exception in a worker (calcSig_wOF.py:51,72,220); a child of the fan-out that dies must
    end its peers (they would wait in the all-gather for ever) and give the command a non-zero exit code."""
    frames_root = str(tmp_path / "frames")
    _make_tree(frames_root)
    r = _run_cli(_argv(frames_root, str(tmp_path / "o"), gpus=("0", "1"), workers="2"), {"STANDIN_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode != 0
    assert "told to fail" in r.stdout


def test_one_pass_ensemble_writes_the_bytes_of_three_runs(tmp_path):
    """calcSig_wOF_ensemble.sh:13-37 runs the command line three times over the SAME frame tree (split1 / split2 / split3 weights).
    With two ``--ensemble`` options ONE run reads, decodes and resizes every frame once and feeds all three weight sets; the three
    <modelname> trees must be byte-identical to three separate runs -- on one GPU and fanned out over two (global clip shards:
    11 clips of 3 videos over 2 ranks, batches that straddle videos)."""
    frames_root = str(tmp_path / "frames")
    _make_tree(frames_root)
    names = ["UCF101_split%d" % k for k in (1, 2, 3)]
    weights = [("ucf101_split%d_tsn_rgb_bn_inception_wOF.caffemodel" % k, "ucf101_split%d_tsn_flow_bn_inception_wOF.caffemodel" % k) for k in (1, 2, 3)]
    want = {}
    for name, (w_rgb, w_flow) in zip(names, weights):
        out = str(tmp_path / ("sep_" + name))
        argv = _argv(frames_root, out, gpus=("0",))
        argv[2], argv[4] = w_rgb, w_flow
        argv[argv.index("--modelname") + 1] = name
        r = _run_cli(argv, {})
        assert r.returncode == 0, r.stdout
        want.update(_tree_bytes(out))
    assert len(want) == 18                                              # 3 videos x 3 members x 2 streams
    assert len({v for v in want.values()}) == 18                        # the members' features differ
    for gpus in (("0",), ("0", "1")):
        out = str(tmp_path / ("ens%d" % len(gpus)))
        argv = _argv(frames_root, out, gpus=gpus)
        for name, (w_rgb, w_flow) in list(zip(names, weights))[1:]:
            argv += ["--ensemble", name, w_rgb, w_flow]
        r = _run_cli(argv, {})
        assert r.returncode == 0, r.stdout
        assert _tree_bytes(out) == want
        # every clip was announced once per stream, not once per member: the frames were read once
        assert r.stdout.count("for rgb modality done") == 11
    bad = _run_cli(_argv(frames_root, str(tmp_path / "bad"), gpus=("0",)) + ["--ensemble", "UCF101_split1", "a", "b"], {})
    assert bad.returncode != 0 and "modelname" in bad.stdout


def test_more_ranks_than_clips(tmp_path):
    """Two clips, three GPUs: the third rank owns no clip, builds no extractor, and still takes part in the gathers."""
    from video_query_algorithms_amd.tsn import frames
    rng = np.random.default_rng(5)
    root = str(tmp_path / "frames")
    for clip in ("clip_0001", "clip_0002"):
        d = os.path.join(root, "v", clip)
        os.makedirs(d)
        for i in range(1, 7):
            for pre, shape in (("img", (24, 32, 3)), ("flow_x", (24, 32)), ("flow_y", (24, 32))):
                frames.write_pnm(os.path.join(d, "%s_%05d.ppm" % (pre, i)), rng.integers(0, 256, shape, dtype=np.uint8))
    r1 = _run_cli(_argv(root, str(tmp_path / "one"), gpus=("0",)), {})
    r3 = _run_cli(_argv(root, str(tmp_path / "three"), gpus=("0", "1", "2"), workers="3"), {})
    assert r1.returncode == 0 and r3.returncode == 0, r1.stdout + r3.stdout
    assert _tree_bytes(str(tmp_path / "three")) == _tree_bytes(str(tmp_path / "one")) and len(_tree_bytes(str(tmp_path / "one"))) == 2


def test_groups_of_videos_are_flushed_as_they_finish(tmp_path):
    """the reference writes a video's files inside its per-video loop (calcSig_wOF.py:195-222).  The drop-in shares the
    clips out over the ranks in GROUPS of whole videos (VQ_CLI_GROUP_CLIPS; default 16 batches per rank) and gathers and queues
    a group's files before the next group starts: same bytes whatever the grouping, one rank or two -- and a failure in a later
    group leaves the files of the finished groups behind (here: videoA = group 1 is on disk, videoB + videoC = group 2 failed)."""
    frames_root = str(tmp_path / "frames")
    _make_tree(frames_root)
    one = str(tmp_path / "one")
    r = _run_cli(_argv(frames_root, one, gpus=("0",)), {})
    assert r.returncode == 0, r.stdout
    want = _tree_bytes(one)
    for gpus in (("0",), ("0", "1")):
        out = str(tmp_path / ("grouped%d" % len(gpus)))
        r = _run_cli(_argv(frames_root, out, gpus=gpus), {"VQ_CLI_GROUP_CLIPS": "4", "VQ_CLI_TRACE": "1"})
        assert r.returncode == 0, r.stdout
        assert "2 group(s)" in r.stdout
        assert _tree_bytes(out) == want
    out = str(tmp_path / "failed")
    # one rank, batches of 2 clips: group 1 (videoA, 7 clips) is 4 rgb + 4 flow calls; the 10th call belongs to group 2
    r = _run_cli(_argv(frames_root, out, gpus=("0",)), {"VQ_CLI_GROUP_CLIPS": "4", "STANDIN_FAIL_AT_CALL": "10"}, timeout=120)
    assert r.returncode != 0 and "told to fail at call 10" in r.stdout
    left = _tree_bytes(out)
    assert sorted(left) == sorted(k for k in want if k.startswith("videoA/")) and all(left[k] == want[k] for k in left)
