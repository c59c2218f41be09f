"""The drop-in for the reference's frame preparation command line (src/features_GPU_compute/build_wof_clips.py): clip
regrouping and grey conversion on the CPU, the whole command on the GPU (flow parity is unpinned: third-party binary)."""
import os

import numpy as np
import pytest


def _touch_frames(d, n, ext=".jpg"):
    os.makedirs(d)
    for i in range(1, n + 1):
        for kind in ("img", "flow_x", "flow_y"):
            with open(os.path.join(d, "%s_%05d%s" % (kind, i, ext)), "wb") as f:
                f.write(b"%s %d" % (kind.encode(), i))


def test_create_clip_follows_the_reference_rules(tmp_path):
    """build_wof_clips.py:78-128: int(n / frames_per_clip) full clips, frames renumbered from 1 inside each clip; the rest becomes
    one more clip when it lasts >= 2 s, else it is deleted."""
    import video_query_algorithms_amd  # noqa: F401
    from video_query_algorithms_amd import build_wof_clips as bw
    out = tmp_path / "out"
    _touch_frames(str(out / "vidA"), 3 * 150 + 40)          # 40 >= 2 * 15: a fourth, short clip
    _touch_frames(str(out / "vidB"), 150 + 20)              # 20 < 30: the tail is dropped
    _touch_frames(str(out / "vidC"), 12, ext=".ppm")        # shorter than 2 s: nothing survives
    assert bw.create_clip("/videos/vidA.mp4", str(out)) == 4
    assert bw.create_clip("/videos/vidB.mp4", str(out)) == 1
    assert bw.create_clip(str(tmp_path / "src" / "vidC"), str(out)) == 0
    a = out / "vidA"
    assert sorted(os.listdir(a)) == ["clip_0001", "clip_0002", "clip_0003", "clip_0004"]
    assert len(os.listdir(a / "clip_0002")) == 450 and len(os.listdir(a / "clip_0004")) == 120
    assert (a / "clip_0002" / "flow_y_00001.jpg").read_bytes() == b"flow_y 151"          # frame 151 is frame 1 of clip 2
    assert (a / "clip_0004" / "img_00040.jpg").read_bytes() == b"img 490"
    assert sorted(os.listdir(out / "vidB")) == ["clip_0001"] and len(os.listdir(out / "vidB" / "clip_0001")) == 450
    assert os.listdir(out / "vidC") == []
    # what calcSig_wOF.py's parse_directory then sees
    from video_query_algorithms_amd.tsn import frames
    d, rgb, flow = frames.parse_directory(str(a))
    assert rgb == {"clip_0001": 150, "clip_0002": 150, "clip_0003": 150, "clip_0004": 40} and flow == rgb


def test_grey_conversion_is_cv2s_fixed_point_rule():
    import video_query_algorithms_amd  # noqa: F401
    from video_query_algorithms_amd.build_wof_clips import bgr_to_grey
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 90]]], np.uint8)      # B, G, R
    assert bgr_to_grey(px).tolist() == [[255, 0, 29, 150, 76, int((10 * 1868 + 200 * 9617 + 90 * 4899 + 8192) >> 14)]]
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    ref = np.rint(0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2])
    assert np.abs(bgr_to_grey(img).astype(int) - ref).max() <= 1


@pytest.mark.gpu
def test_command_line_on_a_frame_directory(gpu, tmp_path):
    """A 'video' given as a directory of its frames (no cv2 here): a panning texture, 41 frames -> 40 img / flow_x / flow_y
    triples -> with --fps 5 --clip_time 3: two clips of 15 and a 10-frame (= 2 s) third one; the warped flow of a pure pan is
    mid-grey; the tree is what calcSig_wOF.py reads."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_warp_oracle import analytic_pair
    from video_query_algorithms_amd import build_wof_clips as bw
    from video_query_algorithms_amd.tsn import frames
    src = tmp_path / "src" / "pan"
    src.mkdir(parents=True)
    for t in range(41):
        H = np.array([[1, 0, 1.5 * t], [0, 1, -0.5 * t], [0, 0, 1.0]])
        g = analytic_pair(96, 128, H, seed=3)[1]
        frames.write_pnm(str(src / ("frame_%05d.ppm" % t)), np.repeat(g[:, :, None], 3, 2))
    out = tmp_path / "out"
    assert bw.main([str(tmp_path / "src"), str(out), "--fps", "5", "--clip_time", "3", "--max_pairs", "16"]) == 0
    d, rgb, flow = frames.parse_directory(str(out / "pan"))
    assert rgb == {"clip_0001": 15, "clip_0002": 15, "clip_0003": 10} and flow == rgb
    ext = os.path.splitext(os.listdir(out / "pan" / "clip_0001")[0])[1]
    fx = frames.imread(str(out / "pan" / "clip_0002" / ("flow_x_00003" + ext)), False)
    fy = frames.imread(str(out / "pan" / "clip_0002" / ("flow_y_00003" + ext)), False)
    assert fx.shape == (96, 128) and abs(int(np.median(fx[16:-16, 16:-16])) - 128) <= 2 and abs(int(np.median(fy[16:-16, 16:-16])) - 128) <= 2
    img = frames.imread(str(out / "pan" / "clip_0001" / ("img_00001" + ext)), True)
    want = frames.imread(str(src / "frame_00001.ppm"), True)                   # the initial frame is skipped: img 1 = frame 1
    assert img.shape == (96, 128, 3) and np.abs(img.astype(int) - want.astype(int)).max() <= (12 if ext == ".jpg" else 0)
    # the same command as a fresh process: it never imports torch (nothing here holds a tensor; VQ_NO_TORCH is set before the library is
    # loaded) and writes the same files
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-query-algorithms_amd", "build_wof_clips.py")
    env = {k: v for k, v in os.environ.items() if not k.startswith("VQ_FANOUT") and k != "VQ_NO_TORCH"}
    p = subprocess.run([sys.executable, cli, str(tmp_path / "src"), str(tmp_path / "out2"), "--fps", "5", "--clip_time", "3", "--max_pairs", "16"],
                       env=dict(env, VQ_CLI_TRACE="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "torch imported: False" in p.stderr, p.stderr[-1500:]
    assert _tree(str(out)) == _tree(str(tmp_path / "out2"))


def _run_wof(argv, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("VQ_FANOUT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "tests", "_wof_standin.py")] + argv, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def _tree(out):
    found = {}
    for dirpath, _, files in os.walk(out):
        for fn in files:
            found[os.path.relpath(os.path.join(dirpath, fn), out)] = open(os.path.join(dirpath, fn), "rb").read()
    return found


def test_num_gpu_fans_out_one_process_per_gpu_and_the_files_do_not_change(tmp_path):
    """build_wof_clips.py:66: worker i runs on GPU (i - 1) % NUM_GPU + START_GPU.  ``--num_gpu 3 --starting_gpu 2`` must put
    one process on each of GPUs 2, 3, 4 (videos dealt round-robin) and write the same files as one GPU; a window size that
    does not divide a video must not change them either; --new_width / --new_height resize the img_ frames only (the
    reference hands the VIDEO to extract_warp_gpu, :70-73)."""
    import video_query_algorithms_amd  # noqa: F401
    from video_query_algorithms_amd.tsn import frames
    rng = np.random.default_rng(4)
    src = tmp_path / "src"
    for v, (n, h, w) in enumerate([(13, 24, 32), (7, 24, 32), (12, 20, 28), (2, 24, 32), (9, 24, 32)]):
        d = src / ("vid%d" % v)
        d.mkdir(parents=True)
        for t in range(n):
            frames.write_pnm(str(d / ("frame_%05d.ppm" % t)), rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    common = ["--fps", "2", "--clip_time", "2", "--new_width", "16", "--new_height", "12"]
    one = tmp_path / "one"
    r = _run_wof([str(src), str(one)] + common + ["--max_pairs", "64"], {"STANDIN_DEVICE_LOG": str(tmp_path / "dev1")})
    assert r.returncode == 0, r.stdout
    want = _tree(str(one))
    assert "vid0/clip_0001/flow_x_00001.ppm" in want and "vid0/clip_0003/img_00004.ppm" in want       # 12 frames -> 3 clips of 4
    assert frames.imread(str(one / "vid0" / "clip_0001" / "img_00001.ppm"), True).shape == (12, 16, 3)   # resized dump
    assert frames.imread(str(one / "vid0" / "clip_0001" / "flow_x_00001.ppm"), False).shape == (24, 32)  # flow at video size
    assert open(str(tmp_path / "dev1.0")).read().split("\n")[0] == "0 24 32"
    fan = tmp_path / "fan"
    r = _run_wof([str(src), str(fan)] + common + ["--max_pairs", "4", "--num_gpu", "3", "--starting_gpu", "2"],
                 {"STANDIN_DEVICE_LOG": str(tmp_path / "dev3")})
    assert r.returncode == 0, r.stdout
    assert _tree(str(fan)) == want
    for rank in range(3):
        devs = {line.split()[0] for line in open(str(tmp_path / ("dev3.%d" % rank))).read().split("\n") if line}
        assert devs == {str(2 + rank)}
    assert r.stdout.count("number of videos found") == 1


def test_frames_read_ahead_arrive_in_order_with_their_grey_and_stop_cleanly(tmp_path):
    """iter_video_ahead: the reader threads run in front of the consumer; what arrives is iter_video's sequence with bgr_to_grey of every
    frame; a consumer that stops early leaves no thread waiting; a frame that cannot be read raises at its place in the sequence."""
    import threading

    import video_query_algorithms_amd  # noqa: F401
    from video_query_algorithms_amd import build_wof_clips as bw
    rng = np.random.default_rng(5)
    d = tmp_path / "vid"
    d.mkdir()
    frames = [rng.integers(0, 256, (24, 32, 3), dtype=np.uint8) for _ in range(41)]
    for i, f in enumerate(frames):
        np.save(str(d / ("frame_%05d.npy" % i)), f)
    got = list(bw.iter_video_ahead(str(d), readers=3, ahead=7))
    assert len(got) == 41
    for (f, g), want in zip(got, frames):
        assert (f == want).all() and (g == bw.bgr_to_grey(want)).all()
    before = threading.active_count()
    it = bw.iter_video_ahead(str(d), readers=3, ahead=7)
    for k, _ in enumerate(it):
        if k == 4:
            break
    it.close()
    assert threading.active_count() <= before                               # the pool's threads are gone with the generator
    (d / "frame_00020.npy").write_bytes(b"not an array")
    it = bw.iter_video_ahead(str(d), readers=3, ahead=7)
    seen = 0
    with pytest.raises(Exception):
        for _ in it:
            seen += 1
    assert seen == 20
    with pytest.raises(IOError):
        list(bw.iter_video_ahead(str(tmp_path)))                            # a directory without frames
