"""The multi-GPU exchange behind the C ABI (pmx_dist_*, panmap_amd/csrc/api_dist.hip) on a one-GPU box: two ranks as two
processes on the one device over the library's host-directory test transport (RCCL refuses two ranks on one device), and a
one-rank group on RCCL itself (librccl loaded by the library, communicator, all-gather, the gather's local part)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_ranks(world, env_extra):
    procs = []
    for r in range(world):
        env = dict(os.environ, PMX_RANK=str(r), PMX_WORLD=str(world), **env_extra)
        procs.append(subprocess.Popen([sys.executable, os.path.join("tests", "dist_c_worker.py")], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs, done = [], []
    for p in procs:
        so, se = p.communicate(timeout=900)
        done.append((p.returncode, so, se))
    # (every rank's end is reported: the rank that failed first is usually not the one whose wait timed out)
    assert all(rc == 0 for rc, _, _ in done), [(rc, so[-500:], se[-1500:]) for rc, so, se in done]
    for rc, so, se in done:
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][7:]))
    return sorted(outs, key=lambda d: d["rank"])


def test_two_ranks_through_the_c_abi_equal_a_single_rank(tmp_path):
    """shards seeded on two ranks, histograms merged by pmx_dist_merge_histograms, replicated scoring, records + CIGAR arenas
    gathered by pmx_dist_gather_alignments with cigar_off rebased: everything equals a single-rank run bit for bit"""
    d0, d1 = _run_ranks(2, {"PMX_DIST_HOST_DIR": str(tmp_path)})
    assert d0["placed"] == d1["placed"] == "node_7618"
    assert d0["n_records"] == d0["n_expected"] == 40000 and d0["per_rank"] == [20000, 20000]
    assert d0["hist_equal"] and d0["scores_equal"] and d0["fields_equal"] and d0["cigars_equal"] and d0["async_same"] and d0["flagged"] == 0
    assert d0["words_per_rank"][1] > 0 and d0["multi_op_cigars"] > 1000
    assert d1["n_records"] == 0 and d1["n_words"] == 0        # only the root holds the gathered set
    # --dedup over the ranks (pmx_dist_dedup_reads): every distinct read counted once although its copies sit on two ranks
    assert d0["dedup_hist_equal"] and d0["dedup_matters"] and d0["dedup_kept"] + d1["dedup_kept"] == d0["distinct_reads"]
    # the transport removes a round's files once the next round is complete; the files of the LAST round stay (a rank cannot
    # know that every peer has read its last file; the owner of the directory removes them): one per rank, same round
    left = sorted(f for f in os.listdir(tmp_path) if f.startswith("x"))
    assert len(left) == 2 and left[0].rsplit(".", 1)[0] == left[1].rsplit(".", 1)[0], left
    # the one-node form (pmx_dist_plan_alignments + pmx_dist_fetch_shard_async): every rank downloads its own part to its
    # place in buffers both ranks map; the result is byte for byte the gathered set
    assert d0["sharded_equals_gathered"] and d0["plan"][:2] == [0, 0] and d1["plan"][0] == 20000 and d1["plan"][1] == d0["words_per_rank"][0]


@pytest.mark.parametrize("mode", ["unequal", "empty", "nocigar"])
def test_two_ranks_at_the_edges_of_the_size_exchange(tmp_path, mode):
    """pmx_dist_* with ranks of UNEQUAL shard sizes, an EMPTY shard and a rank whose reads map nowhere (records without one
    CIGAR word): the count exchange, the max-padded histogram planes, the exact-size point-to-point gather and the rebase
    of cigar_off (api_dist.hip) at their edges -- merged histogram, placement, every record and every CIGAR still equal the
    single-rank run, through the gather and through the sharded download"""
    d0, d1 = _run_ranks(2, {"PMX_DIST_HOST_DIR": str(tmp_path), "PMX_SHARD_MODE": mode})
    assert d0["placed"] == d1["placed"] == "node_7618" and d0["mode"] == mode
    assert d0["n_records"] == d0["n_expected"] and d0["hist_equal"] and d0["scores_equal"] and d0["fields_equal"] and d0["cigars_equal"] and d0["flagged"] == 0
    assert d0["sharded_equals_gathered"]
    n0, n1 = d0["per_rank"]
    if mode == "unequal":
        assert n0 == 28000 and n1 == 12000 and d0["words_per_rank"][1] > 0 and d1["plan"][0] == 28000
    elif mode == "empty":
        assert n0 == 40000 and n1 == 0 and d0["words_per_rank"][1] == 0 and d1["shard"] == [40000, 40000]
    else:
        assert n0 == 30000 and n1 == 6000 and d0["words_per_rank"][1] == 0     # rank 1: records, no CIGAR


@pytest.mark.parametrize("with_torch", [True, False])
def test_one_rank_group_on_rccl(with_torch):
    """the same calls on the real transport: the library loads librccl itself, makes the communicator and runs its
    all-gathers in a group of one (a box of the pool has one GPU; the N > 1 run is the driver's) -- inside a torch process
    (bench.py: the RCCL copy torch ships) and without torch (the panmap command line: the system's)"""
    env = {k: v for k, v in os.environ.items() if k != "PMX_DIST_HOST_DIR"}
    env.update(PMX_RANK="0", PMX_WORLD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if not with_torch:
        env["PMX_WORKER_NO_TORCH"] = "1"
    p = subprocess.run([sys.executable, os.path.join("tests", "dist_c_worker.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert d["placed"] == "node_7618" and d["n_records"] == d["n_expected"] == 40000
    assert d["hist_equal"] and d["fields_equal"] and d["cigars_equal"] and d["async_same"] in (True, None) and d["flagged"] == 0
