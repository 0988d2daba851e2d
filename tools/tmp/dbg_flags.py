import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import panmap_amd as pmx
G = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
pm = pmx.Panman(os.path.join(G, "sars_20000_twilight_dipper.panman"))
raw = pmx.extract_read_sequences(os.path.join(G, "isolate_R1.fastq.gz"), os.path.join(G, "isolate_R2.fastq.gz"))
ctx = pmx.Context(0)
rs = pmx.ReadSet(ctx, raw)
mean = int(sum(len(x) for x in raw) // len(raw))
index = pmx.Index.build(pm, k=19, s=8, t=0, l=3, open_syncmer=False, flank_mask=250)
placer = pmx.Placer(ctx, index)
params = pmx.TraversalParams()
placer.reset(); placer.add_reads(rs, params)
res = placer.score(params, len(raw))
print(res.best_index)
al = [None]
def score(node):
    g = pm.genome(node)
    if al[0] is None: al[0] = pmx.Aligner(ctx, g, mean)
    else: al[0].set_reference(g, mean)
    try:
        return al[0].score_reads(rs, True, False)
    except Exception as e:
        recs, _ = al[0].fetch()
        fl = recs["flags"] & 3
        idx = np.flatnonzero(fl)
        print(node, pm.node_id(node), "ERR", len(idx), idx[:10], fl[idx[:10]], al[0].stats(), len(g))
        for i in idx[:3]:
            print(raw[i - (i & 1)], raw[i - (i & 1) + 1])
        return 0
r = pmx.refine_top_candidates(index.arrays()["parent"], placer.node_outputs()[0], res.best_index, score, pmx.RefineParams(0.01, 4, 2, 3))
print(r)
