"""The ONE line bench.py prints: its contract with the driver (keys, size), checked on the recorded full object of the round's profile run
(profiles/r06_bench_full.json: data written by bench.py --details on the GPU box) -- no GPU needed."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.path.join(ROOT, "profiles", "r06_bench_full.json")


@pytest.fixture(scope="module")
def bench_mod():
    sys.path.insert(0, ROOT)
    import bench
    return bench


@pytest.mark.skipif(not os.path.exists(FULL), reason="no recorded bench object")
def test_the_line_keeps_the_contract_and_its_size(bench_mod):
    full = json.load(open(FULL))
    line = bench_mod.compact(copy.deepcopy(full))
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= bench_mod.LINE_BYTES
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert key in line
    assert line["vs_baseline"] is None and line["config"]["workload"].startswith("configs[1]")
    roof = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof)
    # frac = algorithmic FLOPs of the step / the timed step / the peak
    gflop = roof["frac"] * roof["peak"] * line["ms_per_step"]            # TFLOP/s x ms = GFLOP
    assert abs(gflop - 390.06) < 0.4
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert {"query", "weight_updates"} <= set(line["rounds"])           # configs[0] and configs[4]


@pytest.mark.skipif(not os.path.exists(FULL), reason="no recorded bench object")
def test_an_eight_rank_line_sheds_numbers_instead_of_growing(bench_mod):
    full = json.load(open(FULL))
    full["roofline"]["rank_ms_per_step"] = {"min": 2.601234, "max": 2.712345, "all": [2.612345 + 0.01 * r for r in range(8)]}
    full["roofline"]["all_gather_ms_per_step"] = 0.012345
    full["config"]["distributed"] = {"backend": "nccl", "world_size": 8, "rccl_version": "2.22.3", "rehearsal_on_one_gpu": False}
    full["e2e_cli"]["fresh_process"] = {"error": "x" * 500}
    line = bench_mod.compact(full)
    assert len(json.dumps(line, separators=(",", ":"))) <= bench_mod.LINE_BYTES
    assert len(line["roofline"]["rank_ms_per_step"]["all"]) == 8 and line["config"]["distributed"]["world_size"] == 8
    assert "roofline" in line and "cpu_baseline" in line and "similarity" in line and "rounds" in line


@pytest.mark.skipif(not os.path.exists(FULL), reason="no recorded bench object")
def test_the_per_kernel_fields_of_the_line_agree_with_their_neighbours(bench_mod):
    """the line's kernel_frac / conv_ms_per_step / families come from the ONE-STREAM region (where a launch's
    duration is the kernel alone), from at least three sampled steps at the driver's --steps 20, and cannot contradict the wall
    clock of that region; roofline.traffic is per launch and says so through traffic_per_step."""
    full = json.load(open(FULL))
    line = bench_mod.compact(copy.deepcopy(full))
    roof, single = line["roofline"], line["single_stream"]
    assert bench_mod.sample_every(20) * 3 <= 20 and -(-20 // bench_mod.sample_every(20)) >= 3          # >= 3 samples at K = 20
    assert roof["profiled_steps"] >= 3 and single["profiled_steps"] == roof["profiled_steps"]
    assert "single_stream region" in line["config"]["timed_mode"]
    # the sum of the convolution launches' own durations fits inside the one-stream step they were sampled in
    assert single["conv_ms_per_step"] <= single["ms_per_step"]
    assert abs(roof["kernel_frac"] - single["kernel_frac"]) <= 0.02 and roof["conv_ms_per_step"] == single["conv_ms_per_step"]
    # kernel_frac prices the step's FLOPs against the kernels' time, single_stream.frac against the wall clock: the first cannot be lower
    assert single["kernel_frac"] >= single["frac"] - 1e-6
    gflop = single["frac"] * roof["peak"] * single["ms_per_step"]
    assert abs(gflop - 390.06) < 0.4
    # two sub-batch streams overlap: the product-mode step may be SHORTER than the sum of its kernels' one-stream durations
    assert line["ms_per_step"] < single["ms_per_step"]
    assert abs(roof["traffic_per_step"] - roof["traffic"] * roof["launches_per_step"]) <= 1e-4 * roof["traffic_per_step"]      # both rounded to six digits
    assert "single_stream" in json.dumps(line) and set(single) >= {"value", "ms_per_step", "frac", "conv_ms_per_step", "kernel_frac"}


def test_the_two_queue_trace_that_backs_the_headline_is_tracked():
    """profiles/r06_two_queue_trace.txt (tools/trace_prod.sh): per forward of the product's mode the span, the union-busy time, the sum
    of kernel durations and the queues -- the evidence that wall < sum of kernel time is overlap."""
    import re
    text = open(os.path.join(ROOT, "profiles", "r06_two_queue_trace.txt")).read()
    rows = re.findall(r"forward: (\d+) kernels on queues \[(.*?)\]: span ([\d.]+) ms, busy \(union\) ([\d.]+) ms, idle ([\d.]+) ms, sum of durations ([\d.]+) ms", text)
    assert rows
    steady = [r for r in rows if float(r[4]) < 0.1]                       # forwards that did not wait for the host
    assert steady
    for n, queues, span, busy, idle, total in steady:
        assert len(queues.split(",")) == 2 and int(n) >= 74              # 2 x (36 conv launches + preprocess + global pool ...)
        assert float(total) > 1.8 * float(span) and float(busy) <= float(span) + 1e-6
