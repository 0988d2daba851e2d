#!/usr/bin/env python3
"""Line similarity of a product file against reference files, the way the round-2 verdict measured it: comments and white
space stripped, then the share of the file's lines that (a) equal a reference line, (b) have a difflib ratio >= 0.8 with one.
usage: tools/similarity.py FILE REF [REF ...]"""
import difflib
import re
import sys


def norm_lines(path):
    txt = open(path, errors="replace").read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = []
    for ln in txt.splitlines():
        ln = re.sub(r"//.*", "", ln)
        ln = re.sub(r"\s+", "", ln)
        if len(ln) >= 8:            # braces, else, short declarations say nothing
            out.append(ln)
    return out


def main():
    mine = norm_lines(sys.argv[1])
    ref = []
    for p in sys.argv[2:]:
        ref += norm_lines(p)
    ref_set = set(ref)
    exact = sum(1 for l in mine if l in ref_set)
    near = 0
    by_len = {}
    for r in ref:
        by_len.setdefault(len(r) // 8, []).append(r)
    for l in mine:
        if l in ref_set:
            near += 1
            continue
        cands = []
        for b in range(max(0, len(l) // 8 - 3), len(l) // 8 + 4):
            cands += by_len.get(b, [])
        sm = difflib.SequenceMatcher(None, l, "", autojunk=False)
        hit = False
        for r in cands:
            sm.set_seq2(r)
            if sm.real_quick_ratio() >= 0.8 and sm.quick_ratio() >= 0.8 and sm.ratio() >= 0.8:
                hit = True
                break
        near += hit
    print("%s: %d lines, exact %.1f %%, near(>=0.8) %.1f %%" % (sys.argv[1], len(mine), 100.0 * exact / max(len(mine), 1), 100.0 * near / max(len(mine), 1)))


if __name__ == "__main__":
    main()
