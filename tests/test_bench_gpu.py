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
    r = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--total-reads", "400000", "--cpu-sample", "20000"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "roofline_valu", "cpu_baseline", "value_device_resident", "value_host_to_host", "pcie", "dp", "real_reads"):
        assert k in d, k
    # `value`: inputs resident in HBM, two or three batches in flight; every batch equals the one-batch-at-a-time run bit for bit
    assert d["config"]["batches_in_flight"] == 3 and d["equals_device_resident_run"] is True   # (three for a batch this small, two above 3M reads)
    assert d["value_device_resident"] > 0 and d["pcie"]["h2d_GBps"] > 0 and d["pcie"]["d2h_GBps"] > 0 and d["pcie"]["bound_reads_per_s"] > 0
    assert d["scaling"] == "strong" and d["config"]["total_reads"] == 400000 and "0.4M×150bp" in d["metric"]
    assert set(("pair_share", "cells_per_step", "gcups_align_stage")) <= set(d["dp"])
    assert d["real_reads"]["value"] > 0 and d["real_reads"]["placed_node"] == "node_7618" and d["real_reads"]["records_flagged"] == 0
    assert 1 <= d["cpu_baseline"]["cores"] <= os.cpu_count() and set(d["cpu_baseline"]["kind_by_leg"]) == {"align", "place"}
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["unit"] == "reads/s" and d["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert d["checks"]["placed_node"] == d["config"]["source_node"] and d["checks"]["mapped_fraction"] > 0.99 and d["checks"]["records_flagged"] == 0
    # SURVEY 8d "parity checks in the same run": the GPU's results on the CPU sample against the CPU path's own
    o = d["checks"]["oracle"]
    assert o["reads"] == 20000 and o["histogram_equal"] and o["node_scores_bit_equal"] and o["tsv_equal"]
    assert o["records_equal"] == "20000/20000" and o["cigars_equal"] == "20000/20000" and o["cigar_ops_compared"] >= 20000
    assert d["cpu_baseline"]["threads_launched"] >= d["cpu_baseline"]["cores"]
    # per-stage rooflines: every stage with its kernel time, algorithmic bytes and fraction of its bound
    for st in d["roofline_by_stage"].values():
        assert set(("kernel_ms", "algorithmic_bytes", "achieved", "peak", "frac", "bound")) <= set(st), st


def test_bench_two_ranks_functional():
    env = dict(os.environ, PMX_BENCH_TEST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--scaling", "weak", "--reads-per-gpu", "100000",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["batches_in_flight"] == 3
    assert d["equals_device_resident_run"] is True      # three pipelines per rank, collectives in ticket order
    assert d["checks"]["placed_node"] == d["config"]["source_node"] and d["checks"]["mapped_fraction"] > 0.99 and d["checks"]["records_flagged"] == 0
    assert d["checks"]["rank0_gather_has_every_cigar"] is True


def test_bench_exchange_path_on_rccl_one_rank():
    """the N>1 code path -- RCCL process group, histogram all-gather + device merge, record / CIGAR gather, barriers, max over
    ranks -- in a one-rank group on the real backend (PMX_BENCH_FORCE_DIST; a box of the pool has one GPU)"""
    env = dict(os.environ, PMX_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--total-reads", "100000", "--no-cpu-baseline", "--no-real-reads"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1, r.stdout[:600]   # ONE JSON line (RCCL's banner goes to stderr)
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["equals_device_resident_run"] is True
    assert d["checks"]["placed_node"] == d["config"]["source_node"] and d["checks"]["mapped_fraction"] > 0.99 and d["checks"]["records_flagged"] == 0
    assert d["checks"]["rank0_gather_has_every_cigar"] is True


def test_bench_strong_scaling_two_ranks():
    """--scaling strong: the job's reads are split over the ranks (configs[2] shape, small here); records through the gather
    to rank 0 (--gather rccl: pmx_dist_gather_alignments) -- the other two-rank tests take the default, every rank downloading
    its own part into the node's shared result set (pmx_dist_plan_alignments + pmx_dist_fetch_shard_async)"""
    env = dict(os.environ, PMX_BENCH_TEST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--scaling", "strong",
                        "--total-reads", "200000", "--no-cpu-baseline", "--gather", "rccl"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = _line(r.stdout)
    assert d["scaling"] == "strong" and d["config"]["total_reads"] == 200000 and d["config"]["reads_per_gpu"] == 100000
    assert d["equals_device_resident_run"] is True and d["checks"]["rank0_gather_has_every_cigar"] is True


def test_two_rank_gather_equals_single_rank():
    """rank 0 reconstructs every CIGAR of rank 1, and the merged records equal a single-rank run bit for bit"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29535", os.path.join("tests", "dist_gpu_worker.py")], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert d["n_records"] == d["n_expected"] == 40000
    assert d["fields_equal"] and d["cigars_equal"] and d["flagged"] == 0
    assert d["rank1_base_nonzero"] and d["rank1_has_cigars"] and d["multi_op_cigars"] > 1000
