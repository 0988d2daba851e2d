"""--meta: haplotype deconvolution of a mixed sample on the device (pmx_meta_*; src/main.cpp:1192-1313 runDeconvolution).
Host-side mirror of the reference's flow: reads -> Meta.set_reads -> Meta.score (candidates + parsimony scores) -> Meta.em
-> the `<prefix>.mgsr.abundance.out` text (node[,merged nodes...]<TAB>proportion with five decimals, by proportion)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib
from ._lib import check
from .api import Context, Index, Panman, concat_reads

ORIENTED = 0x100     # PMX_INDEX_ORIENTED
ORIENT_XOR = 0x9e3779b97f4a7c15


class Meta:
    def __init__(self, ctx: Context, index: Index, index_oriented: Index):
        self.ctx, self.index, self.index_oriented = ctx, index, index_oriented
        self._h = C.c_void_p()
        check(lib.pmx_meta_create(ctx._h, index._h, index_oriented._h, C.byref(self._h)), "pmx_meta_create")

    @classmethod
    def build(cls, ctx: Context, pm: Panman, k=19, s=8, t=0, l=3, open_syncmer=False, flank_mask=0):
        """both indexes of one tree (the MGSR index of the reference masks no flanks: flank_mask 0)"""
        idx = Index.build(pm, k=k, s=s, t=t, l=l, open_syncmer=open_syncmer, flank_mask=flank_mask)
        oidx = Index.build(pm, k=k, s=s, t=t, l=l, open_syncmer=open_syncmer, flank_mask=flank_mask, mode=ORIENTED)
        return cls(ctx, idx, oidx)

    def set_reads(self, reads=None, concat=None, offsets=None):
        if reads is not None:
            cb, offsets = concat_reads(reads)
            concat = np.frombuffer(cb, np.uint8)
        concat = np.ascontiguousarray(concat, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.int64)
        check(lib.pmx_meta_set_reads(self.ctx._h, self._h, concat.ctypes.data, offsets.ctypes.data, len(offsets) - 1), "pmx_meta_set_reads")

    def set_dust(self, threshold: float):
        """--dust: the next set_reads drops reads whose DUST score is non-zero and above `threshold` (100 = off)"""
        check(lib.pmx_meta_set_dust(self._h, float(threshold)), "pmx_meta_set_dust")

    def score(self, top_oc: int = 1000, candidates=None):
        if candidates is not None:
            c = np.ascontiguousarray(candidates, np.uint32)
            check(lib.pmx_meta_score(self.ctx._h, self._h, top_oc, c.ctypes.data, len(c)), "pmx_meta_score")
        else:
            check(lib.pmx_meta_score(self.ctx._h, self._h, top_oc, None, 0), "pmx_meta_score")

    def em(self, params: "_lib.MetaParams" = None):
        mp = params if params is not None else _lib.MetaParams()
        check(lib.pmx_meta_em(self.ctx._h, self._h, C.byref(mp)), "pmx_meta_em")
        return self.haplotypes()

    # ---- accessors
    @property
    def n_reads(self) -> int:
        return int(lib.pmx_meta_num_reads(self._h))

    def candidates(self) -> np.ndarray:
        out = np.zeros(int(lib.pmx_meta_num_candidates(self._h)), np.uint32)
        check(lib.pmx_meta_candidates(self._h, out.ctypes.data, len(out)), "pmx_meta_candidates")
        return out

    def overlap_coefficients(self) -> np.ndarray:
        out = np.zeros(self.index.info.n_nodes, np.float64)
        check(lib.pmx_meta_overlap_coefficients(self._h, out.ctypes.data, len(out)), "pmx_meta_overlap_coefficients")
        return out

    def read_info(self):
        n = self.n_reads
        ns, mult = np.zeros(n, np.int64), np.zeros(n, np.int64)
        check(lib.pmx_meta_read_info(self._h, ns.ctypes.data, mult.ctypes.data, n), "pmx_meta_read_info")
        return ns, mult

    def read_seedmers(self):
        n = self.n_reads
        ns, _ = self.read_info()
        tot = int(ns.sum())
        off, h, rev = np.zeros(n + 1, np.int64), np.zeros(max(tot, 1), np.uint64), np.zeros(max(tot, 1), np.uint8)
        check(lib.pmx_meta_read_seedmers(self._h, off.ctypes.data, h.ctypes.data, rev.ctypes.data, len(h)), "pmx_meta_read_seedmers")
        return off, h[:tot], rev[:tot]

    def scores(self) -> np.ndarray:
        n, c = self.n_reads, int(lib.pmx_meta_num_candidates(self._h))
        out = np.zeros((n, c), np.uint16)
        check(lib.pmx_meta_scores(self.ctx._h, self._h, out.ctypes.data, out.size), "pmx_meta_scores")
        return out

    def haplotypes(self):
        """[(node dfs index, proportion, [merged candidates])] by proportion, descending"""
        out = []
        for i in range(int(lib.pmx_meta_num_haplotypes(self._h))):
            node, prop, nm = C.c_uint32(0), C.c_double(0), C.c_int64(0)
            check(lib.pmx_meta_haplotype(self._h, i, C.byref(node), C.byref(prop), C.byref(nm), None, 0), "pmx_meta_haplotype")
            mem = np.zeros(max(nm.value, 1), np.uint32)
            check(lib.pmx_meta_haplotype(self._h, i, None, None, None, mem.ctypes.data, len(mem)), "pmx_meta_haplotype")
            out.append((int(node.value), float(prop.value), [int(x) for x in mem[:nm.value]]))
        return out

    def em_info(self):
        r, it, llh = C.c_int32(0), C.c_int32(0), C.c_double(0)
        check(lib.pmx_meta_em_info(self._h, C.byref(r), C.byref(it), C.byref(llh)), "pmx_meta_em_info")
        return dict(rounds=int(r.value), iterations=int(it.value), log_likelihood=float(llh.value))

    def close(self):
        if self._h:
            lib.pmx_meta_free(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_dust(seq: bytes, window: int = 64) -> float:
    """mgsr::getDust (src/mgsr.cpp:1505-1568) by the library (pmx_read_dust, host)"""
    return float(lib.pmx_read_dust(seq, len(seq), window))


def format_abundance(haplotypes, node_id) -> str:
    """`<prefix>.mgsr.abundance.out` (src/main.cpp:1288-1306): id[,merged ids]<TAB>proportion, %.5f, by proportion"""
    lines = []
    for node, prop, members in haplotypes:
        lines.append(",".join([node_id(node)] + [node_id(x) for x in members]) + "\t%.5f" % prop)
    return "\n".join(lines) + ("\n" if lines else "")
