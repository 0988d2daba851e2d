#!/usr/bin/env python3
"""--refine with the reference's default parameters on the repository's example sample (2 x 51,169 reads): how many
candidates, how long.  usage: refine_timing.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import panmap_amd as pmx  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
pm = pmx.Panman(os.path.join(G, "sars_20000_twilight_dipper.panman"))
raw = pmx.extract_read_sequences(os.path.join(G, "isolate_R1.fastq.gz"), os.path.join(G, "isolate_R2.fastq.gz"))
ctx = pmx.Context(0)
rs = pmx.ReadSet(ctx, raw)
mean = int(sum(len(x) for x in raw) // len(raw))
index = pmx.Index.build(pm, k=19, s=8, t=0, l=3, open_syncmer=False, flank_mask=250)
placer = pmx.Placer(ctx, index)
params = pmx.TraversalParams()
placer.reset()
placer.add_reads(rs, params)
res = placer.score(params, len(raw))
for it in range(2):
    t0 = time.perf_counter()
    r = pmx.refine_placement(ctx, placer, pm, res, rs, True, mean)
    el = time.perf_counter() - t0
    n = len(r["candidates"])
    print("refine: %d candidates x %d reads in %.3f s = %.1f ms per candidate, %.1f M read alignments/s; refined nodes %s scores %s"
          % (n, len(raw), el, el / n * 1e3, n * len(raw) / el / 1e6, [pm.node_id(x) for x in r["node"]], r["score"]))
