"""CPU, world_size 2 over gloo: the sharded path of the drop-in ``calcSig_wOF`` command line itself -- clip shards per
rank (``shard_range``), the all-gather of the per-rank feature blocks, rank 0 writing the CSV tree -- must give the
SAME BYTES as the one-rank run (calcSig_wOF.py:195-221 + the clip-level data parallelism of :204-210).

There is no GPU here and the product has no CPU fallback, so the extractor is a stand-in defined in this file (a
deterministic function of the decoded crops); the arithmetic of the real one is checked on the GPU
(tests/test_tsn_gpu.py).  What this test pins is the control flow around it: ragged shards (7 clips over 2 ranks),
batches that do not divide a shard, several videos, both streams, the ``--gpus`` -> device mapping."""
import hashlib
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEN_DEVICES = "devices_%d.txt"


class StandInNet:
    """Looks like tsn.caffe_net.CaffeNet to the command line; the 'features' of a clip are a hash of its crops."""
    feature_dim = 1024

    def __init__(self, net_proto, net_weights, device_id=0, max_crops=96, feature_blob="global_pool"):
        self.device, self.max_crops = device_id, max_crops

    @staticmethod
    def _clip_feature(crops):
        seed = int.from_bytes(hashlib.sha256(np.ascontiguousarray(crops).tobytes()).digest()[:8], "little")
        return np.random.default_rng(seed).random(1024) * 10.0

    def extract_clips(self, crops, T, on_device=False):
        assert crops.shape[0] % T == 0 and crops.shape[0] <= self.max_crops
        return np.stack([self._clip_feature(crops[i:i + T]) for i in range(0, crops.shape[0], T)])

    def close(self):
        pass


def _make_tree(root):
    from video_query_algorithms_amd.tsn import frames
    rng = np.random.default_rng(11)
    for video, clips in (("videoA", {"clip_0001": 6, "clip_0002": 7, "clip_0003": 6, "clip_0005": 9, "clip_0008": 6,
                                     "clip_0013": 6, "clip_0021": 8}),
                         ("videoB", {"clip_0002": 6, "clip_0004": 6, "clip_0006": 6})):
        for clip, n in clips.items():
            d = os.path.join(root, video, clip)
            os.makedirs(d)
            for i in range(1, n + 1):
                frames.write_pnm(os.path.join(d, "img_%05d.ppm" % i), rng.integers(0, 256, (24, 32, 3), dtype=np.uint8))
                frames.write_pnm(os.path.join(d, "flow_x_%05d.ppm" % i), rng.integers(0, 256, (24, 32), dtype=np.uint8))
                frames.write_pnm(os.path.join(d, "flow_y_%05d.ppm" % i), rng.integers(0, 256, (24, 32), dtype=np.uint8))


def _argv(frames_root, out_dir):
    return [frames_root, "rgb.prototxt", "ucf101_split1_tsn_rgb_bn_inception_wOF.caffemodel", "flow.prototxt",
            "ucf101_split1_tsn_flow_bn_inception_wOF.caffemodel", "--num_frame_per_video", "3", "--outFeatures_dir", out_dir,
            "--modelname", "UCF101_split1", "--frame_ext", ".ppm", "--batch_clips", "2", "--num_worker", "2", "--host_resize",
            "--gpus", "4", "6"]


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
    assert calcSig_wOF.main(_argv(frames_root, one), net_factory=StandInNet) == 0
    want = _tree_bytes(one)
    assert sorted(want) == ["videoA/UCF101_split1/rgb_global_pool_features.csv",
                            "videoA/UCF101_split1/warped_optical_flow_global_pool_features.csv",
                            "videoB/UCF101_split1/rgb_global_pool_features.csv",
                            "videoB/UCF101_split1/warped_optical_flow_global_pool_features.csv"]
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
