#!/usr/bin/env python3
"""BASELINE configs[4] on a synthetic stand-in of the README demo (its reads are absent from the reference checkout): reads
drawn from the five leaves the reference's golden abundance file names, --em-delta-threshold 1e-5 as the demo command."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import panmap_amd as pmx
    from panmap_amd import _lib
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    pm = pmx.Panman(os.path.join(ROOT, "tests", "golden", "sars_20000_twilight_dipper.panman"))
    ctx = pmx.Context(0)
    meta = pmx.Meta.build(ctx, pm)
    names = [l.split("\t")[0] for l in open(os.path.join(ROOT, "tests", "golden", "example.mgsr.abundance.out"))][:5]
    shares = [0.50, 0.20, 0.15, 0.10, 0.05]
    parts, offs, base = [], [np.zeros(1, np.int64)], 0
    for i, (nm, sh) in enumerate(zip(names, shares)):
        c, o = pmx.simulate_paired_reads(pm.genome(nm), int(n_reads * sh) // 2, seed=10 + i)
        parts.append(c)
        offs.append(np.asarray(o[1:], np.int64) + base)
        base += int(o[-1])
    concat, offsets = np.concatenate(parts), np.concatenate(offs)
    meta.set_reads(concat=concat, offsets=offsets)
    for top_oc, delta in ((1000, 1e-5), (1000, 0.0)):
        t0 = time.perf_counter()
        meta.score(top_oc)
        mp = _lib.MetaParams()
        mp.em_delta_threshold = delta
        haps = meta.em(mp)
        dt = time.perf_counter() - t0
        print(json.dumps(dict(top_oc=top_oc, delta=delta, s=dt, distinct=meta.n_reads, cands=int(len(meta.candidates())), info=meta.em_info(),
                              n_haps=len(haps), top=[(pm.node_id(int(n)), round(p, 5), len(m)) for n, p, m in haps[:12]])), flush=True)
    src = [pm.find_node(n) for n in names]
    print("sources", src)


if __name__ == "__main__":
    main()
