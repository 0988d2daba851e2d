#!/usr/bin/env python3
"""The real example reads (tests/golden/isolate_R{1,2}.fastq.gz x8) through place + align, on their own: the
workload of bench.py's `real_reads` leg for rocprofv3 / PMX_ALIGN_VERBOSE runs.  usage: real_reads.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import panmap_amd as pmx  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
golden = os.path.join(ROOT, "tests", "golden")
dev = torch.device("cuda", 0)
pm = pmx.Panman(os.path.join(golden, "sars_20000_twilight_dipper.panman"))
index = pmx.Index.build(pm, k=19, s=8, t=0, l=3, open_syncmer=False, flank_mask=250)
ctx = pmx.Context(0)
placer = pmx.Placer(ctx, index)
params = pmx.TraversalParams()
seqs, _, _ = pmx.read_fastq_paired(os.path.join(golden, "isolate_R1.fastq.gz"), os.path.join(golden, "isolate_R2.fastq.gz"))
rr = seqs * 8
cb, off = pmx.concat_reads(rr)
concat = np.frombuffer(cb, np.uint8).copy()
mean = int(off[-1] // len(rr))
d_concat = torch.from_numpy(concat).to(dev)
d_off = torch.from_numpy(off).to(dev)
rs = pmx.ReadSet.wrap_device(ctx, d_concat.data_ptr(), d_off.data_ptr(), len(rr), int(concat.size), int(np.max(np.diff(off))), keepalive=(d_concat, d_off))
aligner = None
for it in range(steps + 1):
    if it == 1:
        ctx.synchronize()
        t0 = time.perf_counter()
    rs.pack()
    placer.reset()
    placer.add_reads(rs, params)
    res = placer.score(params, len(rr))
    ref = pm.genome(int(res.best_index[4]))
    if aligner is None:
        aligner = pmx.Aligner(ctx, ref, mean)
    else:
        aligner.set_reference(ref, mean)
    aligner.align_readset(rs, paired=True, revcomp_mate2=False)
    ctx.synchronize()
    print("step", it, "align stage ms", ctx.kernel_ms("align"), "seed", ctx.kernel_ms("seed"), flush=True)
el = time.perf_counter() - t0
print("reads/s", len(rr) * steps / el, "ms/step", el / steps * 1e3, aligner.stats())
