"""Quick device-side timing of the TSN forward (crops resident in HBM, HIP events through the C ABI)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import video_query_algorithms_amd as vqa
from video_query_algorithms_amd._lib import call
from video_query_algorithms_amd.tsn import bn_inception, net


def bench(channels, n_crops, T, reps=5):
    g = bn_inception.bn_inception(channels)
    w = net.synthetic_weights(g, seed=2)
    m = net.TsnNet(g, w, max_crops=n_crops)
    crops = torch.randint(0, 256, (n_crops, 224, 224, channels), dtype=torch.uint8, device="cuda")
    mean = net.RGB_MEAN if channels == 3 else net.FLOW_MEAN
    torch.cuda.synchronize()
    m.forward_device(crops.data_ptr(), n_crops, T, mean)
    if os.environ.get("VQ_SHOW_LAUNCHES"):
        for o, l in zip(m.plan.ops, m.launch_items()[0]):
            print("  launch %2d  %s" % (l, o.name))
    tm = C.c_void_p()
    call("vq_timer_create", C.byref(tm))
    times = []
    for _ in range(reps):
        call("vq_timer_start", tm, None)
        m.forward_device(crops.data_ptr(), n_crops, T, mean)
        call("vq_timer_stop", tm, None)
        ms = C.c_float()
        call("vq_timer_elapsed_ms", tm, C.byref(ms))
        times.append(ms.value)
    med = sorted(times)[len(times) // 2]
    fl = m.flops_per_crop() * n_crops
    print("C=%d crops=%d T=%d: %.2f ms median (%.2f best) -> %.1f clips/s, %.1f TFLOP/s (%.1f%% of 157.3)"
          % (channels, n_crops, T, med, min(times), n_crops / T / med * 1e3, fl / med / 1e9, fl / med / 1e9 / 157.3 * 100), flush=True)
    m.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:      # quick_tsn_bench.py C:crops:T ...
        for spec in sys.argv[1:]:
            c, n, t = (int(v) for v in spec.split(":"))
            bench(c, n, t, reps=15)
    else:
        bench(3, 96, 3)
        bench(3, 448, 7)
        bench(10, 448, 7)
