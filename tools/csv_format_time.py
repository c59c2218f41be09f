"""Time of tsn/feature_csv.format_rows on a 256 x 1024 block (the command line's feature file): host only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from video_query_algorithms_amd.tsn import feature_csv as fc

x = np.random.default_rng(1).random((256, 1024))
nos = np.arange(256)
fc.format_rows(x, nos)
for _ in range(4):
    t = time.perf_counter()
    b = fc.format_rows(x, nos)
    print("%.2f ms for %d bytes" % ((time.perf_counter() - t) * 1e3, len(b)))
