"""Two builds of the library on the same frame pairs: are the TV-L1 fields and iteration counts the same BITS?
    python tools/flow_bits_ab.py <lib A> <lib B>
Each library runs in a child process (VQ_AMD_LIB) and leaves its fields in an .npz; the parent compares u1 / u2 as 32-bit patterns."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from test_flow_oracle import _shifted_pair
    from video_query_algorithms_amd.tsn import flow
    res = {}
    motions = [(3.0, -1.5), (-6.5, 2.0), (0.4, 0.3), (4.0, -3.0), (0.0, 0.0), (11.0, 7.5)]
    pairs = [_shifted_pair(256, 340, dx, dy, seed=10 + k, margin=40) for k, (dx, dy) in enumerate(motions)]
    f0, f1 = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    m = flow.Tvl1Flow(8, 256, 340)
    r = m.flow(f0, f1, iterations=True)
    res.update(d_u1=r["u1"], d_u2=r["u2"], d_it=r["iters"])
    m.close()
    for h, w in ((61, 83), (120, 97), (37, 200)):                        # ragged levels, tiles cut differently
        pairs = [_shifted_pair(h, w, 1.5, -0.75, seed=h + k, margin=8) for k in range(3)]
        f0, f1 = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
        m = flow.Tvl1Flow(4, h, w, epsilon=0.0, iterations=25, warps=2, nscales=3)
        r = m.flow(f0, f1, iterations=True)
        res.update({"f%d_u1" % h: r["u1"], "f%d_u2" % h: r["u2"], "f%d_it" % h: r["iters"]})
        m.close()
    np.savez(out, **res)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    libs = [os.path.abspath(p) for p in sys.argv[1:3]]
    outs = []
    with tempfile.TemporaryDirectory() as tmp:
        for k, lib in enumerate(libs):
            out = os.path.join(tmp, "r%d.npz" % k)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out], env=dict(os.environ, VQ_AMD_LIB=lib), check=True)
            outs.append(dict(np.load(out)))
    bad = 0
    for key in sorted(outs[0]):
        a, b = outs[0][key], outs[1][key]
        same = a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all() if a.dtype == np.float32 else (a == b).all()
        vals = (a == b).all()
        print("%-8s %-18s %s%s" % (key, a.shape, "same bits" if same else "DIFFERENT BITS", "" if same or not vals else " (equal values: zero signs)"))
        if not same:
            bad += 1
            if not vals:
                print("   max |a - b| = %g on %d elements" % (np.abs(a.astype(np.float64) - b).max(), int((a != b).sum())))
    print("ok" if not bad else "%d arrays differ" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
