#!/usr/bin/env python3
"""Generates tests/golden/align_golden.json.gz: expected outputs of the REFERENCE aligner (oracle/_ref, compiled
from /root/reference/src/mm_align.c + vendored minimap2 by oracle/Makefile) for inputs every box can
regenerate deterministically.  Data only: inputs are named, not stored (the isolate FASTQs and the placed
genome are fixtures already; the synthetic pairs come from panmap_amd.simulate_paired_reads with a fixed seed).

Run in the build container (needs oracle/_ref, i.e. /root/reference):  python3 tests/golden/make_align_golden.py
"""
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def inputs(pmx):
    genome = b"".join(l.strip() for l in open(os.path.join(HERE, "isolate.ref.fa"), "rb") if not l.startswith(b">"))
    seqs, _, _ = pmx.read_fastq_paired(os.path.join(HERE, "isolate_R1.fastq.gz"), os.path.join(HERE, "isolate_R2.fastq.gz"))
    real = seqs[40000:43000]                                    # 1500 real pairs (R2 already reverse-complemented)
    concat, off = pmx.simulate_paired_reads(genome, 1000, seed=77, sub_rate=0.01)
    syn = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    syn = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(syn)]
    return genome, {"real_40000_43000": real, "synthetic_seed77_sub1pct": syn}


def main():
    import panmap_amd as pmx
    from oracle import oracle as orc
    genome, sets = inputs(pmx)
    out = {}
    for name, reads in sets.items():
        res = orc.ref_align_reads_direct(genome, reads, True, 8)
        rows = []
        for w in res:
            row = [int(w["mapped"])]
            for m in ("r1", "r2"):
                r = w[m]
                row.append([r["pos"], r["rs"], r["re"], r["qs"], r["qe"], r["mapq"], r["rev"], r["proper_frag"], [int(c) for c in r["cigar"]]])
            rows.append(row)
        out[name] = rows
    path = os.path.join(HERE, "align_golden.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as fh:
        fh.write(json.dumps(out, separators=(",", ":")).encode())
    print(path, os.path.getsize(path), "bytes;", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
