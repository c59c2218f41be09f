"""Analyse a rocprofv3 --kernel-trace CSV of tools/quick_tsn_bench.py: per forward, time covered by at least one
kernel, idle gaps between kernels, and overlap between kernels of different queues."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows))
# forwards are delimited by preprocess_kernel launches
# (a forward on two sub-batch streams begins with one preprocess launch per stream: launches that follow each other within
# 100 us are one beginning)
pre = [i for i, k in enumerate(ks) if "preprocess" in k[2]]
starts = [i for n, i in enumerate(pre) if n == 0 or ks[i][0] - ks[pre[n - 1]][0] > 100_000]
print("%d kernels, %d forwards" % (len(ks), len(starts)))
for a, b in list(zip(starts, starts[1:] + [len(ks)]))[-4:]:
    seg = ks[a:b]
    t0, t1 = seg[0][0], max(k[1] for k in seg)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in sorted(seg):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(e - s for s, e, _, _ in seg)
    queues = sorted(set(k[3] for k in seg))
    print("forward: %d kernels on queues %s: span %.3f ms, busy (union) %.3f ms, idle %.3f ms, sum of durations %.3f ms (overlap %.3f ms)"
          % (len(seg), queues, (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, total / 1e6, (total - busy) / 1e6))
