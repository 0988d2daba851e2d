#!/usr/bin/env python3
"""Why pairs leave the compact align tier: a histogram of the `return PMX_C_BAIL` / `bail = true` sites, per source line.

Study tool (CPU, no GPU): builds the host form of the kernel source (tests/hostsim) from a scratch copy of
panmap_amd/csrc/align in which every bail site also records its line number, and runs the real read pairs of
tests/golden (or synthetic ones) through `compact_map_pair`.  The product sources are not touched.

  python tools/compact_bail_reasons.py [--pairs 20000] [--synthetic] [--multi]
"""
import argparse
import collections
import ctypes as C
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(tmp):
    src = os.path.join(ROOT, "panmap_amd", "csrc")
    shutil.copytree(os.path.join(src, "align"), os.path.join(tmp, "align"))
    shutil.copytree(os.path.join(src, "device"), os.path.join(tmp, "device"))
    for f in os.listdir(src):
        if f.endswith((".h", ".hpp")):
            shutil.copy(os.path.join(src, f), tmp)
    p = os.path.join(tmp, "align", "aln_compact.hpp")
    text = open(p).read()
    text = text.replace("#define PMX_C_BAIL 1", "static int pmx_c_bail_line = 0;\n#define PMX_C_BAIL (pmx_c_bail_line = pmx_c_bail_line ? pmx_c_bail_line : __LINE__, 1)", 1)
    # the seeding's bail flag: note where it was raised first
    text = re.sub(r"\bbail = true;", "bail = (pmx_c_bail_line = pmx_c_bail_line ? pmx_c_bail_line : __LINE__, true);", text)
    open(p, "w").write(text)
    hs = open(os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")).read()
    hs = hs.replace("        done[it] = rc == PMX_C_DONE ? 1 : 0;",
                    "        done[it] = rc == PMX_C_DONE ? 1 : 0;\n        if (hs_bail_lines) hs_bail_lines[it] = rc == PMX_C_DONE ? 0 : pmx_c_bail_line;\n        pmx_c_bail_line = 0;", 1)
    hs = hs.replace("extern \"C\" void hs_compact_trace(", "static int* hs_bail_lines = nullptr;\nextern \"C\" void hs_bail_out(int* p) { hs_bail_lines = p; }\nextern \"C\" void hs_compact_trace(", 1)
    open(os.path.join(tmp, "hostsim.cpp"), "w").write(hs)
    so = os.path.join(tmp, "libhostsim_bail.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-unused-value",
                    "-fvisibility-inlines-hidden", "-Wl,-Bsymbolic", "-I" + tmp, os.path.join(tmp, "hostsim.cpp"), "-o", so], check=True)
    return so, text.splitlines()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=20000)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--multi", action="store_true", help="the tier's second form (several regions per mate)")
    args = ap.parse_args()
    if args.multi:
        os.environ["PMX_HS_COMPACT_MULTI"] = "1"
    import panmap_amd as pmx
    import align_checks as ac
    golden = os.path.join(ROOT, "tests", "golden")
    g = b"".join(l.strip() for l in open(os.path.join(golden, "isolate.ref.fa"), "rb") if not l.startswith(b">"))
    if args.synthetic:
        concat, off = pmx.simulate_paired_reads(g, args.pairs, seed=5, sub_rate=0.005)
        reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
        reads = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(reads)]
    else:
        seqs, _, _ = pmx.read_fastq_paired(os.path.join(golden, "isolate_R1.fastq.gz"), os.path.join(golden, "isolate_R2.fastq.gz"))
        reads = seqs[:2 * args.pairs]
    with tempfile.TemporaryDirectory() as tmp:
        so, lines = build(tmp)
        L = C.CDLL(so)
        n = len(reads)
        arr = (C.c_char_p * n)(*reads)
        lens = (C.c_int * n)(*[len(r) for r in reads])
        recs = (ac.Rec * n)()
        cig = np.zeros(n, np.uint32)
        done = np.zeros(n // 2, np.int8)
        where = np.zeros(n // 2, np.int32)
        L.hs_bail_out(where.ctypes.data_as(C.POINTER(C.c_int)))
        L.hs_align_compact.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.POINTER(ac.Rec), C.c_void_p, C.c_void_p]
        rc = L.hs_align_compact(g, len(g), n, arr, lens, 0, recs, cig.ctypes.data, done.ctypes.data)
        assert rc == 0
    np_ = n // 2
    print("%d pairs, %d finished by the compact tier (%.1f %%)" % (np_, int(done.sum()), 100.0 * done.sum() / np_))
    hist = collections.Counter(int(x) for x in where[done == 0])
    for line, c in hist.most_common():
        print("%7d  %5.1f %%  line %4d: %s" % (c, 100.0 * c / np_, line, lines[line - 1].strip()[:150] if line else "?"))


if __name__ == "__main__":
    main()
