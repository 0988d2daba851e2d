"""Randomized parity sweep (GPU box): single-end short reads, long reads of both long-read presets (plain and with a large
rearrangement each), and --refine scoring (pairs with mate 2 as sequenced) against the compiled reference (oracle/_ref).
Not collected by pytest.
usage: python tests/parity_sweep_other.py [seed] [configs]"""
import os, sys, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import panmap_amd as pmx
import align_checks as ac
from oracle import oracle as orc
G = os.path.join(ROOT, "tests", "golden")
pm = pmx.Panman(os.path.join(G, "sars_20000_twilight_dipper.panman"))
ctx = pmx.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
n_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nflag = 0
al = None
nbad = 0
for it in range(n_cfg):
    node = int(rng.integers(1, pm.num_nodes))
    g = pm.genome(node)
    mode = it % 4
    if mode == 0:      # single-end short reads
        read_len = int(rng.choice([50, 100, 150, 250, 400]))
        concat, off = pmx.simulate_paired_reads(g, 3000, read_len=min(read_len, 300), seed=int(rng.integers(1, 1 << 30)), sub_rate=float(rng.choice([0.001, 0.01, 0.05])))
        reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(0, len(off) - 1, 2)]
        paired = False
    elif mode == 1:    # long reads, both long-read presets
        read_len = int(rng.choice([600, 2000, 4000, 6000, 12000]))
        reads = pmx.simulate_long_reads(g, 150, read_len=read_len, seed=int(rng.integers(1, 1 << 30)), sub=float(rng.choice([0.005, 0.02, 0.04])), ins=0.01, dele=0.01)
        paired = False
    elif mode == 3:    # long reads with a deletion / insertion / inversion / duplication each
        read_len = int(rng.choice([1200, 2500, 5500, 8000, 14000]))
        reads = ac.rearranged_long_reads(pmx, g, 150, read_len, int(rng.integers(1, 1 << 30)), sub=float(rng.choice([0.01, 0.03, 0.06])))
        paired = False
    else:              # refine-style scoring, mate 2 as sequenced
        concat, off = pmx.simulate_paired_reads(g, 3000, seed=int(rng.integers(1, 1 << 30)), sub_rate=0.01)
        reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
        paired = True
    mean = int(sum(len(r) for r in reads) // len(reads))
    other = pm.genome(int(rng.integers(1, pm.num_nodes))) if mode == 2 else g
    if al is None: al = pmx.Aligner(ctx, other, mean)
    else: al.set_reference(other, mean)
    if mode == 2:
        rs = pmx.ReadSet(ctx, reads)
        got = al.score_reads(rs, True, False); want = orc.ref_score_reads(other, reads, True)
        ok = got == want
        rs.close()
        print(it, "score", node, got, want, "OK" if ok else "BAD", flush=True)
    else:
        got = al.align_reads(reads, paired=False)
        want = orc.ref_align_reads_direct(g, reads, False, 16)
        bad = ac.compare_results(got, want)
        fl = sum(1 for x in got if x["flags"] & 3)
        ok = not bad
        nflag += fl
        print(it, ("single", "long", "", "rearranged")[mode], node, mean, "bad", len(bad), "flagged", fl, "mapped", sum(w["mapped"] for w in want), bad[:2], flush=True)
    nbad += 0 if ok else 1
print("mismatching configs:", nbad, "flagged records:", nflag)
