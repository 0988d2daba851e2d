import sys, time, collections
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, panmap_amd as pmx, align_checks as ac
from oracle import oracle as orc
G='/root/repo/tests/golden'
g=b"".join(l.strip() for l in open(G+"/isolate.ref.fa","rb") if not l.startswith(b">"))
n=int(sys.argv[1]); L=int(sys.argv[2])
reads=pmx.simulate_long_reads(g,n,read_len=L,seed=43)
ctx=pmx.Context(0)
al=pmx.Aligner(ctx,g,int(np.mean([len(r) for r in reads])))
rs=pmx.ReadSet(ctx,reads)
for it in range(2):
    ctx.synchronize(); t=time.time(); al.align_readset(rs,paired=False); ctx.synchronize(); dt=time.time()-t
    print("gpu align %d reads of %d: %.1f ms -> %.0f reads/s"%(n,L,dt*1e3,n/dt), al.stats(), flush=True)
recs,cig=al.fetch()
got=pmx.records_to_results(recs,cig,False)
t=time.time(); want=orc.ref_align_reads_direct(g,reads,False,16); print("ref %.2fs"%(time.time()-t))
bad=ac.compare_results(got,want)
print("bad",len(bad),bad[:6]); print("flags",collections.Counter(int(x["flags"])&3 for x in got), "mapped", sum(w["mapped"] for w in want))
