"""--refine (src/placement.cpp:440-698, src/mm_align.c:122-199): candidate selection on the host against a direct restatement
of the reference's text, the alignment score on the device against the reference's own aligner (golden vectors made by
tests/golden/make_refine_golden.py with oracle/_ref), and the two together."""
import collections
import importlib.util
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _restated_refine(parent, scores5, best, score_node, top_pct=0.01, max_top_n=150, radius=2, max_nb=150):
    """refineTopCandidates, statement by statement (floats for the seed scores, std::greater on (score, index) pairs, BFS
    with the size check at the pop, parent before children)"""
    n = len(parent)
    children = [[] for _ in range(n)]
    for i in range(1, n):
        children[parent[i]].append(i)
    f32 = scores5.astype(np.float32)

    def within(start):
        out, seen, q = [], {start}, collections.deque([(start, 0)])
        while q and len(out) < max_nb:
            node, d = q.popleft()
            if node != start:
                out.append(node)
            if d >= radius:
                continue
            if node != 0 and parent[node] not in seen:
                seen.add(parent[node]); q.append((parent[node], d + 1))
            for c in children[node]:
                if c not in seen:
                    seen.add(c); q.append((c, d + 1))
        return out
    expanded, allc = [], set()
    for m in range(5):
        scored = sorted(((float(f32[i, m]), i) for i in range(n) if f32[i, m] > 0), reverse=True)
        base = set()
        if scored:
            k = max(min(int(len(scored) * top_pct), max_top_n), 1)
            base = {i for _, i in scored[:k]}
        if best[m] != 0xFFFFFFFF:
            base.add(best[m])
        ex = set()
        for b in base:
            ex.add(b)
            ex.update(within(b))
        expanded.append(ex)
        allc |= ex
    sc = {c: score_node(c) for c in sorted(allc)}
    node, score = [], []
    for m in range(5):
        bi = None
        for c in sorted(expanded[m], reverse=True):     # any order: the tie rules decide
            if bi is None or sc[c] > sc[bi] or (sc[c] == sc[bi] and (f32[c, m] > f32[bi, m] or (f32[c, m] == f32[bi, m] and c < bi))):
                bi = c
        node.append(0xFFFFFFFF if bi is None else bi)
        score.append(0 if bi is None else sc[bi])
    return sorted(allc), node, score


def test_refine_selection_matches_the_restated_reference(pmx):
    rng = np.random.default_rng(9)
    for trial in range(30):
        n = int(rng.integers(2, 400))
        parent = np.zeros(n, np.uint32)
        for i in range(1, n):   # random trees with polytomies (star nodes) and long paths
            parent[i] = rng.integers(max(0, i - 3), i) if trial % 3 == 0 else (rng.integers(0, i) if trial % 3 == 1 else min(i - 1, int(rng.integers(0, 4))))
        scores5 = np.round(rng.random((n, 5)) * (rng.random((n, 5)) > 0.3), 2)     # zeros and exact ties
        best = [int(np.argmax(scores5[:, m])) if scores5[:, m].max() > 0 else 0xFFFFFFFF for m in range(5)]
        align = rng.integers(-50, 0, n)                                            # plenty of equal alignment scores
        kw = dict(top_pct=float(rng.choice([0.01, 0.1, 0.5])), max_top_n=int(rng.choice([3, 150])), radius=int(rng.integers(0, 4)),
                  max_nb=int(rng.choice([2, 7, 150])))
        want_c, want_node, want_score = _restated_refine(parent, scores5, best, lambda c: int(align[c]), **kw)
        got = pmx.refine_top_candidates(parent, scores5, best, lambda c: int(align[c]),
                                        pmx.RefineParams(kw["top_pct"], kw["max_top_n"], kw["radius"], kw["max_nb"]))
        if not want_c:
            assert not got["ran"]
            continue
        assert got["ran"] and got["candidates"].tolist() == want_c and got["candidate_scores"].tolist() == [int(align[c]) for c in want_c]
        assert got["node"] == want_node and got["score"] == want_score, (trial, kw)
    # a failing callback aborts the whole refinement with its error
    with pytest.raises(ZeroDivisionError):
        pmx.refine_top_candidates(np.zeros(3, np.uint32), np.ones((3, 5)), [0] * 5, lambda c: 1 // 0)


def _golden_inputs(pmx):
    spec = importlib.util.spec_from_file_location("make_refine_golden", os.path.join(GOLDEN, "make_refine_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pm, cases = mod.inputs(pmx)
    return pm, cases, json.load(open(os.path.join(GOLDEN, "refine_golden.json")))


def test_refine_golden_matches_the_compiled_reference(pmx, oracle):
    """(where oracle/_ref exists) the committed vectors are what the reference's score_reads_vs_reference returns"""
    pm, cases, gold = _golden_inputs(pmx)
    reads, paired = cases["synthetic_paired"]
    for nd in ("node_7618", "node_9000"):
        assert oracle.ref_score_reads(pm.genome(nd), reads, paired) == gold["synthetic_paired"][nd]


@pytest.mark.gpu
def test_score_reads_equals_the_reference_aligner(pmx):
    """pmx_align_score_reads == score_reads_vs_reference: real and synthetic reads, pairs with mate 2 as sequenced (what
    --refine feeds, so the mates lie on opposite strands) and in the aligner's orientation, single reads"""
    pm, cases, gold = _golden_inputs(pmx)
    ctx = pmx.Context(0)
    al = None
    for name, (reads, paired) in cases.items():
        rs = pmx.ReadSet(ctx, reads)
        mean_len = int(sum(len(r) for r in reads) // len(reads))
        for nd, want in gold[name].items():
            g = pm.genome(nd)
            if al is None:
                al = pmx.Aligner(ctx, g, mean_len)
            else:
                al.set_reference(g, mean_len)
            assert al.score_reads(rs, paired, False) == want, (name, nd)
        rs.close()
    # an odd number of reads cannot be scored as pairs here (the reference maps the last one alone)
    rs = pmx.ReadSet(ctx, cases["real_single"][0][:5])
    with pytest.raises(pmx.PmxError):
        al.score_reads(rs, True, False)
    # the drop-in with the reference's own signature (src/mm_align.h:13-17): same numbers from host strings, and an odd
    # paired set maps its last read alone (src/mm_align.c:178-185) -- against the compiled reference itself
    from oracle import oracle as orc
    for name in ("synthetic_paired", "real_single"):
        reads, paired = cases[name]
        nd = next(iter(gold[name]))
        assert pmx.score_reads_vs_reference(pm.genome(nd), reads, paired) == gold[name][nd], name
    reads, _ = cases["synthetic_paired"]
    odd = reads[:301]
    g = pm.genome("node_7618")
    assert pmx.score_reads_vs_reference(g, odd, True) == orc.ref_score_reads(g, odd, True)


@pytest.mark.gpu
def test_refine_on_the_device_end_to_end(pmx, sars, sars_index):
    """placement of 2,000 synthetic pairs of node_7618, then --refine with the default parameters: the candidates and the
    per-metric winners are those of the restated reference logic fed with the device's own alignment scores; the source
    node (or a node with the same genome) wins every metric"""
    ctx = pmx.Context(0)
    src = sars.genome("node_7618")
    concat, off = pmx.simulate_paired_reads(src, 2000, seed=5)
    reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    rs = pmx.ReadSet(ctx, reads)
    placer = pmx.Placer(ctx, sars_index)
    params = pmx.TraversalParams()
    placer.reset()
    placer.add_reads(rs, params)
    res = placer.score(params, len(reads))
    refined = pmx.refine_placement(ctx, placer, sars, res, rs, True, 150, pmx.RefineParams(0.01, 20, 2, 20))
    assert refined["ran"] and len(refined["candidates"]) > 5
    score_of = dict(zip(refined["candidates"].tolist(), refined["candidate_scores"].tolist()))
    parent = sars_index.arrays()["parent"]
    want_c, want_node, want_score = _restated_refine(parent, placer.node_outputs()[0], res.best_index, lambda c: score_of[c], 0.01, 20, 2, 20)
    assert refined["candidates"].tolist() == want_c and refined["node"] == want_node and refined["score"] == want_score
    best = max(score_of.values())
    for m in range(5):
        assert refined["score"][m] == best and sars.genome(refined["node"][m]) == src
    tsv = pmx.format_refined_tsv(refined, sars.node_id)
    assert tsv.count("\n") == 5 and tsv.startswith("refined_log_raw\t%d\t" % best)
