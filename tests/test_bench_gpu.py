"""bench.py contract checks on a GPU box: the single-rank line, and the multi-rank code path (histogram exchange,
device merge of the gathered runs, replicated scoring, record gather) run functionally with two ranks sharing
the one GPU through gloo (PMX_BENCH_TEST_BACKEND; RCCL itself needs one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _line(out):
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_bench_single_rank_line():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--reads-per-gpu", "200000", "--cpu-sample", "20000"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "reads/s" and d["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert d["checks"]["placed_node"] == "node_7618" and d["checks"]["mapped_fraction"] > 0.99 and d["checks"]["records_flagged"] == 0


def test_bench_two_ranks_functional():
    env = dict(os.environ, PMX_BENCH_TEST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads-per-gpu", "100000",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["checks"]["placed_node"] == "node_7618" and d["checks"]["mapped_fraction"] > 0.99 and d["checks"]["records_flagged"] == 0
