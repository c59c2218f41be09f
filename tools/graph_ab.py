"""A/B of the captured forward (VQ_TSN_GRAPH=1, opt-in) at cfg 2 (96 crops) and cfg 3 (448): wall clock of K back-to-back forwards on a stream, with
and without the graph, on one stream and split over two; the features must have the same bits."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from video_query_algorithms_amd.tsn import bn_inception, net


def run(n_crops, T, graph, split, steps=60):
    os.environ["VQ_TSN_GRAPH"], os.environ["VQ_TSN_SPLIT"] = graph, split
    g = bn_inception.bn_inception(3)
    m = net.TsnNet(g, net.synthetic_weights(g, seed=2), max_crops=n_crops)
    st = torch.cuda.Stream()
    m.set_stream(st.cuda_stream)
    crops = torch.randint(0, 256, (n_crops, 224, 224, 3), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for _ in range(4):
            m.forward_device(crops.data_ptr(), n_crops, T, net.RGB_MEAN)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                m.forward_device(crops.data_ptr(), n_crops, T, net.RGB_MEAN)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    feat = m.features_tensor(n_crops // T).clone().cpu().numpy()
    m.close()
    return best, feat


if __name__ == "__main__":
    for n_crops, T in ((96, 3), (448, 7)):
        ref = None
        for split in ("1", "2"):
            for graph in ("0", "1"):
                ms, feat = run(n_crops, T, graph, split)
                if ref is None:
                    ref = feat
                print("crops=%d split=%s graph=%s: %.3f ms/step = %.0f clips/s  same_bits=%s" % (n_crops, split, graph, ms, n_crops / T / ms * 1e3, (feat == ref).all()),
                      flush=True)
