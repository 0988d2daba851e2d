#!/usr/bin/env python3
"""--meta on a synthetic 5-way mixture (the shape of BASELINE configs[4]; the demo's own reads are absent from the reference
checkout): 200k reads drawn 40/25/20/10/5 % from five nodes of the SARS-CoV-2 20k tree; times the stages of pmx.Meta."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    import panmap_amd as pmx
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    pm = pmx.Panman(os.path.join(ROOT, "tests", "golden", "sars_20000_twilight_dipper.panman"))
    ctx = pmx.Context(0)
    t0 = time.perf_counter()
    meta = pmx.Meta.build(ctx, pm)
    t_index = time.perf_counter() - t0
    names = ["node_7618", "node_1000", "node_12000", "node_3000", "node_17000"]
    shares = [0.40, 0.25, 0.20, 0.10, 0.05]
    parts, offs, base = [], [np.zeros(1, np.int64)], 0
    for i, (nm, sh) in enumerate(zip(names, shares)):
        c, o = pmx.simulate_paired_reads(pm.genome(nm), int(n_reads * sh) // 2, seed=10 + i)
        parts.append(c if isinstance(c, np.ndarray) else np.frombuffer(c, np.uint8))
        offs.append(np.asarray(o[1:], np.int64) + base)
        base += int(o[-1])
    concat, offsets = np.concatenate(parts), np.concatenate(offs)
    t0 = time.perf_counter()
    meta.set_reads(concat=concat, offsets=offsets)
    t_reads = time.perf_counter() - t0
    t0 = time.perf_counter()
    meta.score(1000)
    ctx.synchronize()
    t_score = time.perf_counter() - t0
    t0 = time.perf_counter()
    haps = meta.em()
    t_em = time.perf_counter() - t0
    info = meta.em_info()
    print(json.dumps(dict(reads=len(offsets) - 1, distinct_reads=meta.n_reads, candidates=int(len(meta.candidates())), index_s=t_index, read_seedmers_s=t_reads,
                          score_s=t_score, em_s=t_em, em_info=info, ms_per_iteration=t_em / max(info["iterations"], 1) * 1e3,
                          top=[(meta.index.node_id(int(n)), round(float(p), 6)) for n, p, _ in haps[:6]])))


if __name__ == "__main__":
    main()
