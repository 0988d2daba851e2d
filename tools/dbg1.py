import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, panmap_amd as pmx
G='/root/repo/tests/golden'
g=b"".join(l.strip() for l in open(G+"/isolate.ref.fa","rb") if not l.startswith(b">"))
concat,off=pmx.simulate_paired_reads(g,2000,seed=21)
reads=[bytes(concat[off[i]:off[i+1]]) for i in range(len(off)-1)]
reads=[r if i%2==0 else pmx.reverse_complement(r) for i,r in enumerate(reads)]
ctx=pmx.Context(0)
al=pmx.Aligner(ctx,g,150)
print("aligning", flush=True)
got=al.align_reads(reads,paired=True)
print("ok", sum(x["mapped"] for x in got), al.stats(), flush=True)
