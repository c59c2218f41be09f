#!/bin/bash
# A/B of the end-to-end command line on ONE box: env toggles given as arguments, e.g.  tools/e2e_ab.sh VQ_INGEST_PRIORITY=0 VQ_INGEST_PRIORITY=1
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg VQ_DEVICE_POOL_GB=120 python tools/e2e_once.py 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$cfg', 'value %.0f clips/s  %.3f s  steady %.0f' % (d['value'], d['seconds'], d['steady_state']['value']))"
done
done
