"""Divergence of the compact tier's chain fill (align/aln_compact.hpp) measured on the host build of the kernel source
(tests/hostsim with its per-anchor trace hook): per pair, how many predecessors every anchor evaluates; per wave of 64 pairs in
launch order, what the kernel pays (the sum over anchor indices of the slowest lane) against what a per-lane state machine,
an evaluation budget with a second pass, or waves of pairs of equal (predicted) cost would pay.  DENSE=1 draws the fragments
from an 827-base stretch of the genome, i.e. at the depth of the 10M-read batch (~130 pairs per start position): the waves
of the real launch are then as homogeneous as the pair order by both mates' keys makes them.  profiles/r04/README.md quotes it."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import panmap_amd as pmx
from align_checks import hostsim, Rec
pm = pmx.Panman("tests/golden/sars_20000_twilight_dipper.panman")
g = pm.genome("node_7618")
npairs = 64 * 600
sub = g[8000:8000 + 227 + 600] if os.environ.get("DENSE") else g
concat, off = pmx.simulate_paired_reads(sub, npairs, seed=3)
reads = [bytes(concat[off[i]:off[i+1]]) for i in range(2 * npairs)]
L = hostsim(False)
L.hs_align_compact.restype = C.c_int
L.hs_align_compact.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.POINTER(Rec), C.c_void_p, C.c_void_p]
L.hs_compact_trace.argtypes = [C.c_void_p]
n = len(reads)
arr = (C.c_char_p * n)(*reads); lens = (C.c_int * n)(*[len(r) for r in reads])
recs = (Rec * n)(); cig = np.zeros(n, np.uint32); done = np.zeros(n // 2, np.int8)
trace = np.zeros((npairs, 129), np.uint8)
L.hs_compact_trace(trace.ctypes.data)
rc = L.hs_align_compact(g, len(g), n, arr, lens, 1, recs, cig.ctypes.data, done.ctypes.data)
L.hs_compact_trace(None)
ra = np.frombuffer(recs, dtype=np.dtype([("rs","<i4"),("re","<i4"),("qs","<i4"),("qe","<i4"),("mapq","u1"),("rev","u1"),("pf","u1"),("mapped","u1"),("nc","<u2"),("fl","<u2"),("co","<u4"),("sc","<i4")]))
k1 = ra["rs"][0::2] >> 4; k2 = ra["rs"][1::2] >> 4
order = np.lexsort((k2, k1))
tr = trace[order]; ok = done[order] == 1
na = tr[:, 0].astype(int); trips = tr[:, 1:65].astype(int); resc = tr[:, 65:129].astype(int)
print("pairs", npairs, "done", ok.mean(), "mean anchors", na.mean(), "mean evals/pair", trips.sum(1).mean(), "mean rescans/pair", resc.sum(1).mean())
A = B = A2 = B2 = 0
for w in range(0, npairs, 64):
    t = trips[w:w+64]; r = resc[w:w+64]; n_ = na[w:w+64]
    A += t.max(0).sum(); B += t.sum(1).max()          # sum over anchors of the wave's slowest lane / slowest lane's own total
    A2 += r.max(0).sum(); B2 += r.sum(1).max()
nw = npairs // 64
print("per wave: inner loop steps now %.1f, flattened %.1f ; rescan steps now %.1f, flattened %.1f ; anchors max %.1f" % (A / nw, B / nw, A2 / nw, B2 / nw, np.mean([na[w:w+64].max() for w in range(0, npairs, 64)])))
tot = trips.sum(1)
print("evals per pair percentiles", np.percentile(tot, [10, 50, 75, 90, 95, 98, 99, 99.9]))
def wave_cost(t):   # sum over anchors of the slowest lane
    return t.max(0).sum()
base = sum(wave_cost(trips[w:w+64]) for w in range(0, npairs, 64)) / nw
for B in (60, 100, 150, 200, 300):
    heavy = tot > B
    # first pass: a lane stops once it has spent B evaluations (its remaining anchors cost nothing)
    c1 = 0
    cum = np.cumsum(trips, 1)
    capped = np.where(cum <= B, trips, np.maximum(0, B - (cum - trips)))
    for w in range(0, npairs, 64):
        c1 += wave_cost(capped[w:w+64])
    hv = trips[heavy]
    hv = hv[np.argsort(tot[heavy])]     # heavy pairs grouped by their cost
    c2 = sum(wave_cost(hv[w:w+64]) for w in range(0, len(hv), 64))
    print("budget %4d: heavy %.3f, pass-1 steps/wave %.1f, pass-2 adds %.1f per original wave -> total %.1f (now %.1f)" % (B, heavy.mean(), c1 / nw, c2 / nw, (c1 + c2) / nw, base))
# ideal: every wave made of pairs of equal cost
srt = trips[np.argsort(tot)]
print("sorted by cost (ideal homogeneous waves): %.1f" % (sum(wave_cost(srt[w:w+64]) for w in range(0, npairs, 64)) / nw))
print("---- proxies")
rs1 = ra["rs"][0::2][order]; re1 = ra["re"][0::2][order]; rs2 = ra["rs"][1::2][order]; re2 = ra["re"][1::2][order]
ov = np.maximum(0, np.minimum(re1, re2) - np.maximum(rs1, rs2))
print("corr(total evals, overlap)", np.corrcoef(tot, ov)[0, 1], " corr(total, anchors)", np.corrcoef(tot, na)[0, 1])
def cost_sorted(key):
    o = np.argsort(key, kind="stable")
    t = trips[o]
    return sum(wave_cost(t[w:w+64]) for w in range(0, npairs, 64)) / nw
print("now (locality order)            %.1f" % base)
print("sorted by overlap               %.1f" % cost_sorted(ov))
print("sorted by overlap>>3            %.1f" % cost_sorted(ov >> 3))
print("sorted by (overlap>>3, anchors) %.1f" % cost_sorted((ov >> 3) * 64 + na))
print("sorted by anchors               %.1f" % cost_sorted(na))
print("sorted by true total (ideal)    %.1f" % cost_sorted(tot))
# within blocks of 4096 launch positions only (keeps most of the locality order)
def cost_sorted_blocked(key, B):
    c = 0
    for b0 in range(0, npairs, B):
        o = np.argsort(key[b0:b0+B], kind="stable")
        t = trips[b0:b0+B][o]
        c += sum(wave_cost(t[w:w+64]) for w in range(0, len(t), 64))
    return c / nw
for B in (1024, 4096, 16384):
    print("sorted by (overlap>>3, anchors) within blocks of %5d: %.1f" % (B, cost_sorted_blocked((ov >> 3) * 64 + na, B)))
