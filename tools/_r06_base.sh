set -e
python3 bench.py --details gpurun_out/r06b_bench_full.json > gpurun_out/r06b_bench.json 2> gpurun_out/r06b_bench.err
python3 - <<'PY'
import json
l=json.load(open('gpurun_out/r06b_bench.json'))
print(len(json.dumps(l,separators=(",",":"))))
print({k:l[k] for k in ('value','ms_per_step','steps')}, l['roofline'], l['single_stream'], l['config'])
print(l['two_stream']); print(l['rounds']); print(l['e2e_cli'])
PY
