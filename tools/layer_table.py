"""Per-layer device times of the TSN forward (HIP events), as a table: where the conv time goes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import video_query_algorithms_amd as vqa  # noqa: F401
from video_query_algorithms_amd.tsn import bn_inception, net


def main(channels=3, n_crops=96, T=3, reps=10):
    g = bn_inception.bn_inception(channels)
    m = net.TsnNet(g, net.synthetic_weights(g, seed=2), max_crops=n_crops)
    crops = torch.randint(0, 256, (n_crops, 224, 224, channels), dtype=torch.uint8, device="cuda")
    mean = net.RGB_MEAN if channels == 3 else net.FLOW_MEAN
    for _ in range(3):
        m.forward_device(crops.data_ptr(), n_crops, T, mean)
    m.set_profile(reps)
    for _ in range(reps):
        m.forward_device(crops.data_ptr(), n_crops, T, mean)
    names, kinds, ms, fl = m.layer_times()
    tiles = m.layer_tiles(n_crops)
    plan = m.plan
    tot = ms.sum()
    print("%-34s %-8s %9s %5s %5s %8s %7s %6s %s" % ("layer", "kind", "M", "N", "K", "ms", "TF/s", "%time", "tile"))
    for op, k, t, f, tl in zip(plan.ops, kinds, ms, fl, tiles):
        td = plan.tensors[op.segments[0].dst if getattr(op, "segments", None) else op.dst]
        M = n_crops * td.h * td.w
        print("%-34s %-8s %9d %5d %5d %8.4f %7.1f %6.2f %s" % (op.name[:34], k, M, op.cout, op.cin * op.k * op.k, t,
                                                             f / t / 1e9 if t > 0 else 0, 100 * t / tot,
                                                             "%dx%dx%d%s" % (tl[0], tl[1], tl[2], "p" if tl[3] else "") if tl[0] else ""))
    conv = np.array([k == "conv" for k in kinds])
    print("total %.3f ms; conv %.3f ms (%.1f TF/s = %.1f%% of 157.3); other %.3f ms" % (
        tot, ms[conv].sum(), fl[conv].sum() / ms[conv].sum() / 1e9, fl[conv].sum() / ms[conv].sum() / 1e9 / 1.573,
        ms[~conv].sum()))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 96,
         int(sys.argv[3]) if len(sys.argv) > 3 else 3)
