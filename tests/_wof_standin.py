"""The drop-in ``build_wof_clips`` command line with a stand-in flow workspace, for the CPU tests of its process logic
(GPU fan-out over --num_gpu, streaming windows, what the flow sees under --new_width / --new_height).  The flow arithmetic
itself is checked on the GPU (tests/test_flow_gpu.py, tests/test_warp_gpu.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class StandInFlow:
    """Looks like tsn.flow.Tvl1Flow to the command line: the 'flow' of a pair is a function of the two frames and the seed of
    the window position, so any change in windowing, seeding or frame size shows in the files."""

    def __init__(self, max_pairs, h, w, device=0, **params):
        self.max_pairs, self.h, self.w, self.device = max_pairs, h, w, device
        log = os.environ.get("STANDIN_DEVICE_LOG")
        if log:
            with open(log + ".%s" % os.environ.get("VQ_FANOUT_RANK", "0"), "a") as f:
                f.write("%d %d %d\n" % (device, h, w))

    def warped_consecutive(self, frames, seed=0):
        assert frames.shape[1:] == (self.h, self.w) and 2 <= frames.shape[0] <= self.max_pairs + 1
        fx, fy = [], []
        for i in range(frames.shape[0] - 1):                  # seed + i = seed of the video + global pair index
            a, b = frames[i].astype(np.int32), frames[i + 1].astype(np.int32)
            fx.append(((a - b) * 3 + 128 + (seed + i) % 7).clip(0, 255).astype(np.uint8))
            fy.append(((a + b) // 2 + (seed + i) % 5).clip(0, 255).astype(np.uint8))
        return np.stack(fx), np.stack(fy)

    def close(self):
        pass


if __name__ == "__main__":
    from video_query_algorithms_amd import build_wof_clips
    build_wof_clips.Tvl1Flow = StandInFlow
    build_wof_clips._writer = lambda: (".ppm", lambda p, img: build_wof_clips.frames_mod.write_pnm(p + ".ppm", img))
    sys.exit(build_wof_clips.main(program=os.path.abspath(__file__)))
