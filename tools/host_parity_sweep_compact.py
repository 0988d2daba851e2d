#!/usr/bin/env python3
"""Host parity sweep of the compact align tier (both forms, two-kernel and fused, 16- and 32-bit position words) against the
reference's own aligner (oracle/_ref): read length, insert size, substitution rate, indels, several genomes of the SARS
tree.  CPU only (the kernel source built for the host, tests/hostsim); every pair a form finishes must carry the
reference's record.  python tools/host_parity_sweep_compact.py [pairs per case]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import panmap_amd as pmx          # noqa: E402
import align_checks as ac         # noqa: E402
from oracle import oracle         # noqa: E402


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    pm = pmx.Panman(os.path.join(ROOT, "tests", "golden", "sars_20000_twilight_dipper.panman"))
    rng = np.random.default_rng(2026)
    cases = []
    for node in ("node_7618", "node_1", "node_9000", "node_15000"):
        for read_len, insert, sub, indel_every in ((150, 300.0, 0.002, 0), (150, 160.0, 0.004, 0), (150, 120.0, 0.01, 0), (100, 110.0, 0.003, 0),
                                                    (75, 80.0, 0.002, 0), (125, 200.0, 0.02, 9), (160, 170.0, 0.001, 0), (150, 500.0, 0.03, 0)):
            cases.append((node, read_len, insert, sub, indel_every))
    total = bad_total = 0
    for node, read_len, insert, sub, indel_every in cases:
        g = pm.genome(node)
        concat, off = pmx.simulate_paired_reads(g, n_pairs, read_len=read_len, seed=int(rng.integers(1, 1 << 30)), sub_rate=sub,
                                                mean_insert=max(insert, float(read_len)), sd_insert=insert / 8)
        reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
        reads = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(reads)]
        if indel_every:
            for i in range(0, len(reads), indel_every):
                r = bytearray(reads[i])
                p = int(rng.integers(10, max(11, len(r) - 10)))
                if i % 2:
                    del r[p:p + int(rng.integers(1, 6))]
                else:
                    r[p:p] = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(1, 6))))
                reads[i] = bytes(r)
        want = oracle.ref_align_reads_direct(g, reads, True, 8)
        line = "%-10s len %3d insert %3.0f sub %.3f indel/%d:" % (node, read_len, insert, sub, indel_every)
        for name, env in (("first form", {}), ("second form", {"PMX_HS_COMPACT_MULTI": "1"}), ("two kernels", {"PMX_HS_COMPACT_SPLIT": "1"}),
                          ("two kernels, second form", {"PMX_HS_COMPACT_SPLIT": "1", "PMX_HS_COMPACT_MULTI": "1"}),
                          ("32-bit positions, second form", {"PMX_HS_COMPACT_POS32": "1", "PMX_HS_COMPACT_MULTI": "1"})):
            for k in ("PMX_HS_COMPACT_MULTI", "PMX_HS_COMPACT_SPLIT", "PMX_HS_COMPACT_POS32"):
                os.environ.pop(k, None)
            os.environ.update(env)
            got, done = ac.hostsim_align_compact(g, reads)
            idx = [i for i in range(len(want)) if done[i]]
            bad = ac.compare_results([got[i] for i in idx], [want[i] for i in idx])
            total += len(idx)
            bad_total += len(bad)
            line += "  %s %d/%d%s" % (name, len(idx), len(want), "" if not bad else " BAD %d" % len(bad))
            if bad:
                print(bad[:3])
        print(line, flush=True)
    print("records compared: %d, differing: %d" % (total, bad_total))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
