#!/usr/bin/env python3
"""The repository's real example reads (tests/golden/isolate_R{1,2}.fastq.gz, x8) through place + align on one GPU: the
real-reads leg of bench.py on its own (no 10M synthetic set to generate), with the tier statistics and a digest of the
records + CIGARs so that two builds can be compared.  Usage: python tools/real_reads.py [--steps N] [--copies C]"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--copies", type=int, default=8)
    args = ap.parse_args()
    import torch  # noqa: F401  (first: its bundled HIP runtime must be the one loaded)
    import panmap_amd as pmx
    golden = os.path.join(os.path.dirname(HERE), "tests", "golden")
    pm = pmx.Panman(os.path.join(golden, "sars_20000_twilight_dipper.panman"))
    index = pmx.Index.build(pm, k=19, s=8, t=0, l=3, open_syncmer=False, flank_mask=250)
    params = pmx.TraversalParams()
    seqs, _, _ = pmx.read_fastq_paired(os.path.join(golden, "isolate_R1.fastq.gz"), os.path.join(golden, "isolate_R2.fastq.gz"))
    rr = seqs * args.copies
    ctx = pmx.Context(0)
    placer = pmx.Placer(ctx, index)
    rs = pmx.ReadSet(ctx, rr, pack=False)
    mean_len = int(sum(len(r) for r in rr[:2000]) // 2000)
    aligner = None
    al_ms, out = [], None
    for it in range(args.steps + 1):
        if it == 1 or args.steps == 0:
            ctx.synchronize()
            t0 = time.perf_counter()
        rs.pack()
        placer.reset()
        placer.add_reads(rs, params)
        res = placer.score(params, len(rr))
        ref = pm.genome(int(res.best_index[4]))
        if aligner is None:
            aligner = pmx.Aligner(ctx, ref, mean_len)
        else:
            aligner.set_reference(ref, mean_len)
        aligner.align_readset(rs, paired=True, revcomp_mate2=False)
        ctx.synchronize()
        if it:
            al_ms.append(ctx.kernel_ms("align"))
    el = time.perf_counter() - t0 if args.steps else 1.0
    recs, cig = aligner.fetch()
    # digest by content: records in pair order, each with its own CIGAR words (the arena order is not deterministic)
    h = hashlib.sha256()
    co, nc = recs["cigar_off"].astype(np.int64), recs["n_cigar"].astype(np.int64)
    flat = np.concatenate([cig[o:o + n] for o, n in zip(co.tolist(), nc.tolist())]) if len(recs) else np.zeros(0, np.uint32)
    r2 = recs.copy()
    r2["cigar_off"] = 0
    h.update(r2.tobytes())
    h.update(np.ascontiguousarray(flat).tobytes())
    print(json.dumps(dict(reads=len(rr), reads_per_s=len(rr) * args.steps / el, ms_per_step=el / args.steps * 1e3, align_ms=al_ms,
                          node=pm.node_id(int(res.best_index[4])), mapped=float(np.mean(recs["mapped"])),
                          flagged=int(np.sum((recs["flags"] & 3) != 0)), digest=h.hexdigest()[:16], tiers=aligner.stats())))


if __name__ == "__main__":
    main()
