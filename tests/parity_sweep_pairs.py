"""Randomized parity sweep (GPU box): read pairs of random shape -- read length, insert, error rates, indels / N, mate
orientation, random nodes of both trees -- through the library against the compiled reference aligner (oracle/_ref).
Not collected by pytest (minutes of GPU time); tests/test_align_gpu.py::test_varied_pair_shapes_exact keeps eight shapes.
usage: python tests/parity_sweep_pairs.py [seed] [configs]"""
import os, sys, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import panmap_amd as pmx
import align_checks as ac
from oracle import oracle as orc
G = os.path.join(ROOT, "tests", "golden")
pm = pmx.Panman(os.path.join(G, "sars_20000_twilight_dipper.panman"))
rsv = pmx.Panman(os.path.join(G, "rsv_4K.panman"))
ctx = pmx.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
al = None
nbad = 0
dump = []
t0 = time.time()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    tree = pm if rng.random() < 0.8 else rsv
    node = int(rng.integers(1, tree.num_nodes))
    g = tree.genome(node)
    if len(g) < 2000: continue
    read_len = int(rng.choice([50, 75, 100, 125, 150, 151, 160]))
    mean_insert = float(rng.choice([read_len, read_len * 1.2, 2 * read_len, 300, 450, 600]))
    sub = float(rng.choice([0, 0.001, 0.002, 0.005, 0.01, 0.03, 0.08]))
    indel_every = int(rng.choice([0, 0, 0, 3, 9, 25]))
    as_seq = bool(rng.random() < 0.3)
    n = 8000
    concat, off = pmx.simulate_paired_reads(g, n, read_len=read_len, seed=int(rng.integers(1, 1 << 30)), sub_rate=sub, mean_insert=max(mean_insert, float(read_len)), sd_insert=max(mean_insert / 8, 1.0))
    reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    if not as_seq:
        reads = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(reads)]
    if indel_every:
        for i in range(0, len(reads), indel_every):
            r = bytearray(reads[i]); p = int(rng.integers(5, max(6, len(r) - 5)))
            k = int(rng.integers(0, 3))
            if k == 0: del r[p:p + int(rng.integers(1, 8))]
            elif k == 1: r[p:p] = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(1, 8))))
            else: r[p:p + int(rng.integers(1, 4))] = b"N"
            reads[i] = bytes(r)
    mean = int(sum(len(r) for r in reads) // len(reads))
    if al is None: al = pmx.Aligner(ctx, g, mean)
    else: al.set_reference(g, mean)
    got = al.align_reads(reads, paired=True)
    want = orc.ref_align_reads_direct(g, reads, True, 16)
    bad = ac.compare_results(got, want)
    fl = sum(1 for x in got if x["flags"] & 3)
    st = al.stats()
    print(it, tree is pm, node, len(g), read_len, mean_insert, sub, indel_every, as_seq, "bad", len(bad), "flagged", fl, "compact", st["compact_tier_items"], "mapped", sum(w["mapped"] for w in want), flush=True)
    if bad:
        nbad += 1
        print("   first:", bad[:2])
        seen = set()
        for b in bad:
            if b[0] in seen or len(seen) >= 4: continue
            seen.add(b[0])
            dump.append(dict(tree="sars" if tree is pm else "rsv", node=node, mean=mean, pair=[reads[2 * b[0]].decode(), reads[2 * b[0] + 1].decode()], what=str(b), it=it))
print("configs with mismatches:", nbad, "time", time.time() - t0)
import json
json.dump(dump, open(os.path.join(ROOT, "gpurun_out", "sweep_bad.json"), "w"))
