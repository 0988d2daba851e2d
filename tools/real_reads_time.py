"""Align-stage throughput on the repository's real example reads (tests/golden/isolate_R*.fastq.gz, the list repeated 8x ~ 0.8 M
reads): a harder workload than bench.py's synthetic reads (20 % of the pairs need a DP).  Run from the repo root on a GPU box."""
import sys, time, os, numpy as np
sys.path.insert(0,'.')
import panmap_amd as pmx
G='tests/golden'
g=b"".join(l.strip() for l in open(G+"/isolate.ref.fa","rb") if not l.startswith(b">"))
seqs,_,_=pmx.read_fastq_paired(G+"/isolate_R1.fastq.gz",G+"/isolate_R2.fastq.gz")
reads=seqs*8   # ~800k reads
ctx=pmx.Context(0)
rs=pmx.ReadSet(ctx,reads)
al=pmx.Aligner(ctx,g,int(np.mean([len(r) for r in seqs])))
for it in range(3):
    ctx.synchronize(); t=time.perf_counter(); al.align_readset(rs,paired=True,revcomp_mate2=False); ctx.synchronize(); dt=time.perf_counter()-t
    print("real reads x8: %d reads align %.1f ms -> %.2f M reads/s"%(len(reads),dt*1e3,len(reads)/dt/1e6))
