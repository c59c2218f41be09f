"""The ONE line bench.py prints: its contract with the driver (keys, size), checked on the recorded full object of the round's profile run
(profiles/r05_bench_full.json: data written by bench.py --details on the GPU box) -- no GPU needed."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.path.join(ROOT, "profiles", "r05_bench_full.json")


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
