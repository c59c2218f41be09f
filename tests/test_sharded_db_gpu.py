"""GPU: the served, row-sharded database of sharded_db.py with the real kernels.  The box has ONE card, so N > 1 is rehearsed
with every rank on it over gloo (what moves between the ranks is the same; RCCL itself is exercised with one rank)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (3, "gloo"), (1, "nccl")])
def test_broker_round_on_served_shards_equals_one_gpu(gpu, world, backend):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_sharded_broker.py"), str(world), backend], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-4000:]


def test_host_reads_are_ordered_behind_the_scan_on_a_large_database(gpu):
    """under RCCL the database's stream is a non-blocking side stream and a scan is only queued on it.  With a
    database whose scan takes milliseconds every host read must still see THIS query's results: one rank over RCCL, 300 000 rows,
    both layouts, against a plain FeatureDB bit for bit (tests/_sharded_large.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_sharded_large.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-4000:]
