#!/bin/bash
# Three driver-style runs of bench.py on ONE box (fresh process each; the tilings come with the library, nothing is swept):
# how far apart are they?  -> gpurun_out/<tag>_bench_repeat.txt    Usage: bash tools/bench_repeat.sh [tag]
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_bench_repeat.txt
mkdir -p gpurun_out
echo "# python3 bench.py --steps 20 --warmup 5 --skip-cpu --skip-flow, three fresh processes on one box: cfg 2 value (product mode), single_stream, cfg 3, scan, rounds" > $OUT
for i in 1 2 3; do
python3 bench.py --steps 20 --warmup 5 --skip-cpu --skip-flow 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
print('run $i: value %.1f clips/s (%.4f ms, frac %.4f)  single_stream %.1f (%.4f ms)  cfg3 %.1f (single %.1f)  scan %.2f q/s  query %.4f ms  weight_updates %.4f ms' % (
 l['value'], l['ms_per_step'], l['roofline']['frac'], l['single_stream']['value'], l['single_stream']['ms_per_step'], l['two_stream']['value'], l['two_stream']['single_stream'],
 l['similarity']['value'], l['rounds']['query']['ms_per_query'], l['rounds']['weight_updates']['ms_per_round']))" >> $OUT
done
python3 - <<PY >> $OUT
import re
rows=[l for l in open("$OUT") if l.startswith("run")]
v=[float(re.search(r"value ([\d.]+)", r).group(1)) for r in rows]
c=[float(re.search(r"cfg3 ([\d.]+)", r).group(1)) for r in rows]
print("spread: cfg 2 %.2f %%, cfg 3 %.2f %% (max - min over mean)" % ((max(v)-min(v))/(sum(v)/3)*100, (max(c)-min(c))/(sum(c)/3)*100))
PY
cat $OUT
