#!/bin/bash
# A/B of the timed mode of bench.py on ONE box: env toggles given as arguments, e.g.  tools/bench_ab.sh VQ_TSN_GRAPH=0 VQ_TSN_GRAPH=1
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg python bench.py --steps 20 --warmup 5 --skip-flow --skip-cpu --skip-sim --skip-two-stream 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
r=d['roofline']
print('$cfg', 'value %.0f  ms %.3f  frac %.4f  single-stream ms %.3f  (sampled: conv_ms %.3f other %.3f)' % (d['value'], d['ms_per_step'], r['frac'], d['single_stream']['ms_per_step'], r['conv_ms_per_step'], r['other_kernels_ms_per_step']))"
done
done
