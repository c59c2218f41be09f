"""Where the time of building an extractor goes (the second build of a process: packed weights in the cache, block pool warm).
Run on the GPU box: python tools/profile_build.py"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from video_query_algorithms_amd.tsn import bn_inception, caffe_net, net  # noqa: E402


def build(path, ch, max_crops):
    proto = os.path.join(os.path.dirname(path), "deploy_%d.prototxt" % ch)
    if not os.path.exists(proto):
        with open(proto, "w") as f:
            f.write(bn_inception.to_prototxt(bn_inception.bn_inception(ch)))
    return caffe_net.CaffeNet(proto, path, device_id=0, max_crops=max_crops)


def main():
    tmp = os.environ.get("TMPDIR", "/tmp")
    paths = {}
    for ch, seed in ((3, 2), (10, 5)):
        p = os.path.join(tmp, "w_%d.npz" % ch)
        if not os.path.exists(p):
            caffe_net.save_weights(p, net.synthetic_weights(bn_inception.bn_inception(ch), seed))
        paths[ch] = p
    for rep in range(3):
        t0 = time.perf_counter()
        nets = [build(paths[ch], ch, 800) for ch in (3, 10)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for n in nets:
            n.close()
        print("build of both extractors, run %d: %.1f ms (close %.1f ms)" % (rep, (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    nets = [build(paths[ch], ch, 800) for ch in (3, 10)]
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
