"""Host-side mirror of the reference's interfaces for the index -> place -> align path.

Names follow the reference: ``TraversalParams`` (src/placement.hpp:28-54), ``PlacementResult``
(:157-235), ``place_lite`` = placement::placeLite (:237-244), ``align_reads_direct``
(src/mm_align.h:44-53).  All compute goes through the C ABI of libpanmap_amd.so.
"""
import ctypes as C
import threading
import os
import dataclasses
import gzip
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import lib, check

METRICS = ("log_raw", "log_cosine", "containment", "weighted_containment", "log_containment")


# ------------------------------------------------------------------------------------- host
class Panman:
    """PanMAN pangenome (replaces loadPanMAN, src/main.cpp:313-325)."""

    def __init__(self, path: str):
        self._h = C.c_void_p()
        check(lib.pmx_panman_open(path.encode(), C.byref(self._h)), f"pmx_panman_open({path})")
        self.path = path

    def close(self):
        if self._h:
            lib.pmx_panman_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_nodes(self) -> int:
        return lib.pmx_panman_num_nodes(self._h)

    @property
    def num_blocks(self) -> int:
        return lib.pmx_panman_num_blocks(self._h)

    def node_id(self, i: int) -> str:
        s = lib.pmx_panman_node_id(self._h, i)
        if s is None:
            raise IndexError(i)
        return s.decode()

    def parent(self, i: int) -> int:
        return lib.pmx_panman_parent(self._h, i)

    def find_node(self, node_id: str) -> int:
        return lib.pmx_panman_find_node(self._h, node_id.encode())

    def genome(self, node) -> bytes:
        """panmapUtils::getStringFromReference (src/panmap_utils.cpp:182-190), ungapped."""
        i = node if isinstance(node, int) else self.find_node(node)
        if i < 0:
            raise KeyError(node)
        # one reconstruction per call: the (per-thread) buffer of the previous call is tried first (genomes of one tree are
        # about the same length) and grown when the answer does not fit
        tl = self.__dict__.setdefault("_tl", threading.local())
        buf = getattr(tl, "gbuf", None)
        n = lib.pmx_panman_node_genome(self._h, i, buf, len(buf) if buf is not None else 0)
        if n < 0:
            raise _lib.PmxError(-3, "pmx_panman_node_genome")
        if buf is None or n > len(buf):
            buf = tl.gbuf = C.create_string_buffer(max(n + n // 8, 1))
            lib.pmx_panman_node_genome(self._h, i, buf, len(buf))
        return buf.raw[:n]


class Index:
    """Single-sample seed index (IndexBuilder, src/index_single_mode.hpp:207-214)."""

    def __init__(self, handle):
        self._h = handle
        self.info = _lib.IndexInfo()
        check(lib.pmx_index_get_info(self._h, C.byref(self.info)), "pmx_index_get_info")

    @classmethod
    def build(cls, pm: Panman, k=19, s=8, t=0, l=3, open_syncmer=False, flank_mask=250, mode=0, max_nodes=-1) -> "Index":
        """mode 0 = automatic, 1 = from-scratch re-seeding of every node, 2 = incremental DFS (see pmx_index_build_ex)"""
        h = C.c_void_p()
        check(lib.pmx_index_build_ex(pm._h, k, s, t, l, int(open_syncmer), flank_mask, mode, max_nodes, C.byref(h)), "pmx_index_build_ex")
        return cls(h)

    @classmethod
    def from_arrays(cls, k, s, t, l, open_syncmer, parent, offsets, hashes, parent_counts, child_counts, flank_mask=0):
        parent = np.ascontiguousarray(parent, np.uint32)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        hashes = np.ascontiguousarray(hashes, np.uint64)
        pc = np.ascontiguousarray(parent_counts, np.int16)
        cc = np.ascontiguousarray(child_counts, np.int16)
        info = _lib.IndexInfo(k, s, t, l, int(open_syncmer), 0, flank_mask, 0, len(parent), len(hashes))
        h = C.c_void_p()
        check(lib.pmx_index_from_arrays(C.byref(info), parent.ctypes.data, offsets.ctypes.data, hashes.ctypes.data,
                                        pc.ctypes.data, cc.ctypes.data, C.byref(h)), "pmx_index_from_arrays")
        return cls(h)

    @classmethod
    def load(cls, path: str) -> "Index":
        """read a `.idx` file (the reference's single-sample index container, src/placement.cpp:1009-1092)"""
        h = C.c_void_p()
        check(lib.pmx_index_load(os.fsencode(path), C.byref(h)), "pmx_index_load")
        return cls(h)

    def save(self, path: str, zstd_level: int = 3, uncompressed: bool = False):
        """write the `.idx` container (src/index_single_mode.cpp:1593-1640)"""
        check(lib.pmx_index_save(self._h, os.fsencode(path), int(zstd_level), int(uncompressed)), "pmx_index_save")

    @staticmethod
    def read_header(path: str):
        """(k, s, t, l, open, hpc, uncompressed) from the 32-byte header, or None if the file has none"""
        info = _lib.IndexInfo()
        unc = C.c_int()
        if lib.pmx_index_read_header(os.fsencode(path), C.byref(info), C.byref(unc)) != 0:
            return None
        return dict(k=info.k, s=info.s, t=info.t, l=info.l, open=bool(info.open_syncmer), hpc=bool(info.hpc), uncompressed=bool(unc.value))

    def node_id(self, dfs_index: int) -> str:
        p = lib.pmx_index_node_id(self._h, int(dfs_index))
        return p.decode() if p else ""

    def close(self):
        if self._h:
            lib.pmx_index_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _arr(self, fn, dtype, n):
        p = fn(self._h)
        if not p or n == 0:
            return np.zeros(0, dtype)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,)).copy()

    def arrays(self):
        n, m = self.info.n_nodes, self.info.n_changes
        return dict(parent=self._arr(lib.pmx_index_parents, np.uint32, n),
                    offsets=self._arr(lib.pmx_index_offsets, np.uint64, n + 1),
                    hash=self._arr(lib.pmx_index_hashes, np.uint64, m),
                    parent_count=self._arr(lib.pmx_index_parent_counts, np.int16, m),
                    child_count=self._arr(lib.pmx_index_child_counts, np.int16, m))


# ------------------------------------------------------------------------------------ reads
def _open_maybe_gz(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")


def read_fastx(path: str):
    """FASTA/FASTQ(.gz) -> (names, seqs, quals); quals '' for FASTA.  kseq semantics: multi-line
    records tolerated, name = up to first whitespace (src/seeding.cpp:237-243)."""
    names, seqs, quals = [], [], []
    with _open_maybe_gz(path) as f:
        data = f.read()
    lines = data.split(b"\n")
    i, n = 0, len(lines)
    while i < n:
        ln = lines[i].rstrip(b"\r")
        if not ln:
            i += 1
            continue
        if ln[:1] == b">":
            name = ln[1:].split()[0] if len(ln) > 1 else b""
            i += 1
            parts = []
            while i < n and lines[i][:1] not in (b">", b"@"):
                parts.append(lines[i].rstrip(b"\r"))
                i += 1
            names.append(name); seqs.append(b"".join(parts)); quals.append(b"")
        elif ln[:1] == b"@":
            name = ln[1:].split()[0] if len(ln) > 1 else b""
            i += 1
            parts = []
            while i < n and lines[i][:1] != b"+":
                parts.append(lines[i].rstrip(b"\r"))
                i += 1
            seq = b"".join(parts)
            i += 1  # '+'
            q = b""
            while i < n and len(q) < len(seq):
                q += lines[i].rstrip(b"\r")
                i += 1
            names.append(name); seqs.append(seq); quals.append(q)
        else:
            i += 1
    return names, seqs, quals


_RC = bytes.maketrans(b"ACGT", b"TGCA")


def reload_options():
    """read the PMX_* switches from the environment again (the library reads them once, at its first use)"""
    lib.pmx_options_reload()


def describe_options() -> str:
    """the library's one table of switches: name, class, meaning, current value"""
    n = int(lib.pmx_options_describe(None, 0))
    buf = C.create_string_buffer(n)
    lib.pmx_options_describe(buf, n)
    return buf.value.decode()


def reverse_complement(seq: bytes) -> bytes:
    """seeding::reverseComplement (src/seeding.cpp:271-284): only upper-case ACGT are complemented."""
    return seq.translate(_RC)[::-1]


def extract_read_sequences(reads1: str, reads2: str = "") -> List[bytes]:
    """extractReadSequences (src/placement.cpp:164-197): R1 then R2, interleaved; R2 NOT rev-comped."""
    s1 = read_fastx(reads1)[1]
    if not reads2:
        return s1
    s2 = read_fastx(reads2)[1]
    if len(s1) != len(s2):
        raise ValueError(f"File {reads2} does not contain the same number of reads as {reads1}")
    out = [None] * (2 * len(s1))
    out[0::2] = s1
    out[1::2] = s2
    return out


def read_fastq_paired(reads1: str, reads2: str = ""):
    """seeding::readFastqPaired (src/seeding.cpp:231-269): R2 reverse-complemented, quals reversed,
    pairs interleaved; missing quals -> 'I'."""
    n1, s1, q1 = read_fastx(reads1)
    q1 = [q if q else b"I" * len(s) for s, q in zip(s1, q1)]
    if not reads2:
        return s1, q1, n1
    n2, s2, q2 = read_fastx(reads2)
    if len(s1) != len(s2):
        raise ValueError(f"Error: {reads2} does not contain the same number of reads as {reads1}")
    s2 = [reverse_complement(s) for s in s2]
    q2 = [(q if q else b"I" * len(s))[::-1] for s, q in zip(s2, q2)]
    n = len(s1)
    seqs, quals, names = [None] * (2 * n), [None] * (2 * n), [None] * (2 * n)
    seqs[0::2], seqs[1::2] = s1, s2
    quals[0::2], quals[1::2] = q1, q2
    names[0::2], names[1::2] = n1, n2
    return seqs, quals, names


class FastxReads:
    """Reads parsed by the native (C++) FASTA/FASTQ reader: flat numpy views of the library's buffers
    (`seq`, `qual`: uint8; `off`, `name_off`: int64 offsets) that upload to the device without a Python list in between."""

    def __init__(self, handle):
        self._h = handle
        n = lib.pmx_fastx_num_reads(self._h)
        ptrs = [C.c_void_p() for _ in range(5)]
        check(lib.pmx_fastx_views(self._h, *[C.byref(p) for p in ptrs]), "pmx_fastx_views")
        self.n = int(n)

        def view(ptr, count, ctype, dtype):
            if count == 0 or not ptr.value:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).view(dtype)

        self.off = view(ptrs[2], self.n + 1, C.c_int64, np.int64)
        self.name_off = view(ptrs[4], self.n + 1, C.c_int64, np.int64)
        total = int(self.off[-1]) if self.n >= 0 and len(self.off) else 0
        self.seq = view(ptrs[0], total, C.c_uint8, np.uint8)
        self.qual = view(ptrs[1], total, C.c_uint8, np.uint8)
        self.names_concat = view(ptrs[3], int(self.name_off[-1]) if len(self.name_off) else 0, C.c_uint8, np.uint8)

    def lists(self):
        """(seqs, quals, names) as Python lists of bytes: the shape read_fastq_paired returns"""
        sb, qb, nb = self.seq.tobytes(), self.qual.tobytes(), self.names_concat.tobytes()
        o, no = self.off, self.name_off
        return ([sb[o[i]:o[i + 1]] for i in range(self.n)], [qb[o[i]:o[i + 1]] for i in range(self.n)],
                [nb[no[i]:no[i + 1]] for i in range(self.n)])

    def close(self):
        if self._h:
            lib.pmx_fastx_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_fastq_paired_native(reads1: str, reads2: str = "") -> FastxReads:
    """seeding::readFastqPaired (src/seeding.cpp:231-269) by the library's C++ reader (pmx_fastx_read_paired)."""
    h = C.c_void_p()
    check(lib.pmx_fastx_read_paired(reads1.encode(), reads2.encode() if reads2 else None, C.byref(h)), "pmx_fastx_read_paired")
    return FastxReads(h)


def read_fastx_native(path: str) -> FastxReads:
    """one FASTA/FASTQ(.gz) file in file order (kseq conventions); FASTA records have zero bytes as qualities"""
    h = C.c_void_p()
    check(lib.pmx_fastx_read(path.encode(), C.byref(h)), "pmx_fastx_read")
    return FastxReads(h)


def concat_reads(reads: Sequence[bytes]):
    lens = np.fromiter((len(r) for r in reads), np.int64, len(reads))
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    return b"".join(reads), off


# ----------------------------------------------------------------------------------- device
class Context:
    """One GPU (one process rank)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        check(lib.pmx_ctx_create(device, C.byref(self._h)), "pmx_ctx_create")
        self.device = device

    def synchronize(self):
        check(lib.pmx_ctx_synchronize(self._h), "pmx_ctx_synchronize")

    @property
    def stream(self) -> int:
        return lib.pmx_ctx_stream(self._h) or 0

    def kernel_ms(self, name: str) -> float:
        return lib.pmx_last_kernel_ms(self._h, name.encode())

    def close(self):
        if self._h:
            lib.pmx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReadSet:
    def __init__(self, ctx: Context, reads=None, concat: Optional[bytes] = None, offsets=None, pack=True):
        self.ctx = ctx
        if reads is not None:
            concat, offsets = concat_reads(reads)
        offsets = np.ascontiguousarray(offsets, np.int64)
        self._h = C.c_void_p()
        self._keep = (concat, offsets)
        buf = (C.c_char * max(len(concat), 1)).from_buffer_copy(concat if len(concat) else b"\0") if not isinstance(concat, np.ndarray) else None
        ptr = C.addressof(buf) if buf is not None else concat.ctypes.data
        check(lib.pmx_readset_upload(ctx._h, ptr, offsets.ctypes.data, len(offsets) - 1, C.byref(self._h)), "pmx_readset_upload")
        self.n_reads = len(offsets) - 1
        self.total_bases = int(offsets[-1] - offsets[0])
        if pack:
            self.pack()

    @classmethod
    def from_fastx(cls, ctx: Context, fx: "FastxReads", with_qualities: bool = False, pack: bool = True) -> "ReadSet":
        """upload the reads of the native FASTA/FASTQ reader (flat buffers, no Python list in between)"""
        self = cls(ctx, concat=fx.seq, offsets=fx.off, pack=pack)
        self._keep = (fx,)
        if with_qualities:
            if len(fx.qual) != self.total_bases:
                raise ValueError("quality strings must have the lengths of the reads")
            q = np.ascontiguousarray(fx.qual) if len(fx.qual) else np.zeros(1, np.uint8)
            check(lib.pmx_readset_set_qualities(ctx._h, self._h, q.ctypes.data), "pmx_readset_set_qualities")
        return self

    @classmethod
    def wrap_device(cls, ctx: Context, d_concat_ptr: int, d_offsets_ptr: int, n_reads: int, total_bytes: int, max_len: int, keepalive=None):
        self = cls.__new__(cls)
        self.ctx = ctx
        self._h = C.c_void_p()
        self._keep = keepalive
        check(lib.pmx_readset_wrap_device(ctx._h, d_concat_ptr, d_offsets_ptr, n_reads, total_bytes, max_len, C.byref(self._h)),
              "pmx_readset_wrap_device")
        self.n_reads = n_reads
        self.total_bases = total_bytes
        return self

    def rewrap_device(self, d_concat_ptr: int, d_offsets_ptr: int, n_reads: int, total_bytes: int, max_len: int = 0, keepalive=None):
        """point this read set at another batch in device memory (buffers reused, offsets handled on the device)"""
        check(lib.pmx_readset_rewrap_device(self.ctx._h, self._h, d_concat_ptr, d_offsets_ptr, n_reads, total_bytes, max_len),
              "pmx_readset_rewrap_device")
        self._keep = keepalive
        self.n_reads = n_reads
        self.total_bases = total_bytes

    def pack(self):
        check(lib.pmx_readset_pack(self.ctx._h, self._h), "pmx_readset_pack")

    def order_pairs(self):
        """enqueue the align stage's pair order now, on a side stream (pmx_readset_order_pairs): optional, same results"""
        check(lib.pmx_readset_order_pairs(self.ctx._h, self._h), "pmx_readset_order_pairs")

    def pack_range(self, r0: int, r1: int):
        """streaming: pack the reads [r0, r1) alone (their bases have landed in the wrapped buffer)"""
        check(lib.pmx_readset_pack_range(self.ctx._h, self._h, int(r0), int(r1)), "pmx_readset_pack_range")

    def set_qualities(self, quals):
        """attach the FASTQ quality strings (list of bytes, same lengths as the reads) for --min-seed-quality"""
        qc = b"".join(quals)
        if len(qc) != self.total_bases:
            raise ValueError("quality strings must have the lengths of the reads")
        buf = (C.c_char * max(len(qc), 1)).from_buffer_copy(qc if qc else b"\0")
        check(lib.pmx_readset_set_qualities(self.ctx._h, self._h, C.addressof(buf)), "pmx_readset_set_qualities")

    def close(self):
        if self._h:
            lib.pmx_readset_free(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclasses.dataclass
class TraversalParams:
    """Result-affecting subset of placement::TraversalParams (src/placement.hpp:28-54); k/s/t/l/open
    always come from the index (src/placement.cpp:1094-1101).  seedMaskFraction defaults to the CLI
    value 0 (src/main.cpp:1967), not the struct's 0.001."""
    seedMaskFraction: float = 0.0
    minSeedQuality: int = 0
    dedupReads: bool = False
    trimStart: int = 0
    trimEnd: int = 0
    minReadSupport: int = -1
    forceLeaf: bool = False

    def to_c(self) -> _lib.PlaceParams:
        p = _lib.PlaceParams()
        p.min_seed_quality = int(self.minSeedQuality)
        p.seed_mask_fraction = self.seedMaskFraction
        p.min_read_support = self.minReadSupport
        p.trim_start, p.trim_end = self.trimStart, self.trimEnd
        p.dedup_reads = int(self.dedupReads)
        p.force_leaf = int(self.forceLeaf)
        return p


@dataclasses.dataclass
class PlacementResult:
    """placement::PlacementResult (src/placement.hpp:157-235), per metric in TSV order."""
    best_score: List[float]
    best_index: List[int]
    tied_indices: List[np.ndarray]
    n_reads: int = 0
    n_unique_seeds: int = 0
    readUniqueSeedCount: int = 0
    totalReadSeedFrequency: int = 0
    min_support: int = 0
    readMagnitude: float = 0.0
    logContainmentDenominator: float = 0.0
    weightedContainmentDenominator: float = 0.0

    @property
    def bestLogContainmentNodeIndex(self):
        return self.best_index[4]


class Placer:
    """Device-resident index + seed histogram of one sample (the place stage)."""

    def __init__(self, ctx: Context, index: Index):
        self.ctx, self.index = ctx, index
        self._h = C.c_void_p()
        check(lib.pmx_place_create(ctx._h, index._h, C.byref(self._h)), "pmx_place_create")
        self.n_nodes = index.info.n_nodes

    def reset(self):
        check(lib.pmx_place_reset(self.ctx._h, self._h), "pmx_place_reset")

    def add_reads(self, rs: ReadSet, params: TraversalParams = TraversalParams()):
        cp = params.to_c()
        check(lib.pmx_place_add_reads(self.ctx._h, self._h, rs._h, C.byref(cp)), "pmx_place_add_reads")

    def add_reads_range(self, rs: ReadSet, r0: int, r1: int, params: TraversalParams = TraversalParams()):
        """seed the reads [r0, r1) of a read set (packed so far) into the histogram"""
        cp = params.to_c()
        check(lib.pmx_place_add_reads_range(self.ctx._h, self._h, rs._h, int(r0), int(r1), C.byref(cp)), "pmx_place_add_reads_range")

    def histogram(self):
        n = lib.pmx_place_histogram_size(self.ctx._h, self._h)
        if n < 0:
            raise _lib.PmxError(n, "pmx_place_histogram_size")
        h, c = np.zeros(n, np.uint64), np.zeros(n, np.int64)
        check(lib.pmx_place_histogram_export(self.ctx._h, self._h, h.ctypes.data, c.ctypes.data, n), "pmx_place_histogram_export")
        return h, c

    def merge(self, hashes, counts):
        hashes = np.ascontiguousarray(hashes, np.uint64)
        counts = np.ascontiguousarray(counts, np.int64)
        check(lib.pmx_place_histogram_merge(self.ctx._h, self._h, hashes.ctypes.data, counts.ctypes.data, len(hashes)),
              "pmx_place_histogram_merge")

    def histogram_size(self) -> int:
        n = lib.pmx_place_histogram_size(self.ctx._h, self._h)
        if n < 0:
            raise _lib.PmxError(n, "pmx_place_histogram_size")
        return n

    def export_device(self, d_hash_ptr: int, d_count_ptr: int, cap: int):
        check(lib.pmx_place_histogram_export_device(self.ctx._h, self._h, d_hash_ptr, d_count_ptr, cap), "pmx_place_histogram_export_device")

    def histogram_entries(self) -> int:
        """distinct seeds so far, without sorting them (the multi-GPU exchange sizes its buffers with it)"""
        n = lib.pmx_place_histogram_entries(self.ctx._h, self._h)
        if n < 0:
            raise _lib.PmxError(n, "pmx_place_histogram_entries")
        return n

    def export_device_unsorted(self, d_hash_ptr: int, d_count_ptr: int, cap: int):
        check(lib.pmx_place_histogram_export_device_unsorted(self.ctx._h, self._h, d_hash_ptr, d_count_ptr, cap),
              "pmx_place_histogram_export_device_unsorted")

    def merge_device_parts(self, d_hash_ptr: int, d_count_ptr: int, part_stride: int, sizes, skip_part: int):
        sz = np.ascontiguousarray(sizes, np.int64)
        check(lib.pmx_place_histogram_merge_device_parts(self.ctx._h, self._h, d_hash_ptr, d_count_ptr, part_stride, sz.ctypes.data, len(sz), skip_part),
              "pmx_place_histogram_merge_device_parts")

    def merge_device(self, d_hash_ptr: int, d_count_ptr: int, n: int):
        check(lib.pmx_place_histogram_merge_device(self.ctx._h, self._h, d_hash_ptr, d_count_ptr, n), "pmx_place_histogram_merge_device")

    def score(self, params: TraversalParams = TraversalParams(), n_reads: int = 0) -> PlacementResult:
        cp, res = params.to_c(), _lib.PlaceResult()
        check(lib.pmx_place_score(self.ctx._h, self._h, C.byref(cp), n_reads, C.byref(res)), "pmx_place_score")
        tied = []
        for m in range(5):
            t = np.zeros(res.n_tied[m], np.uint32)
            check(lib.pmx_place_tied(self._h, m, t.ctypes.data, len(t)), "pmx_place_tied")
            tied.append(t)
        return PlacementResult(list(res.best_score), list(res.best_index), tied, res.n_reads, res.n_unique_seeds,
                               res.n_kept_seeds, res.total_seed_freq, res.min_support, res.log_read_magnitude,
                               res.log_containment_den, res.weighted_containment_den)

    def node_outputs(self):
        n = self.n_nodes
        sc, me, ct = np.zeros((n, 5)), np.zeros((n, 5)), np.zeros((n, 2), np.int64)
        check(lib.pmx_place_node_outputs(self.ctx._h, self._h, sc.ctypes.data, me.ctypes.data, ct.ctypes.data), "pmx_place_node_outputs")
        return sc, me, ct

    def kept_seeds(self):
        n = lib.pmx_place_kept_seeds(self.ctx._h, self._h, None, None, 0)
        h, l = np.zeros(max(n, 0), np.uint64), np.zeros(max(n, 0), np.float64)
        if n > 0:
            lib.pmx_place_kept_seeds(self.ctx._h, self._h, h.ctypes.data, l.ctypes.data, n)
        return h, l

    def close(self):
        if self._h:
            lib.pmx_place_free(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def format_placement_tsv(result: PlacementResult, node_id) -> str:
    """The `<prefix>.placement.tsv` text (src/placement.cpp:1952-2003): score %.6f, tied ids comma-joined."""
    lines = ["metric\tscore\tnodes"]
    for m, name in enumerate(METRICS):
        tied = result.tied_indices[m]
        if len(tied):
            ids = ",".join(node_id(int(t)) for t in tied)
        else:
            ids = node_id(result.best_index[m]) if result.best_index[m] != 0xFFFFFFFF else ""
        lines.append("%s\t%.6f\t%s" % (name, result.best_score[m], ids))
    return "\n".join(lines) + "\n"


def place_lite(ctx: Context, placer: Placer, reads1: str, reads2: str, output_path: str,
               params: TraversalParams = TraversalParams(), node_id=None) -> PlacementResult:
    """placement::placeLite (src/placement.hpp:237-244): reads -> seed histogram -> node scores ->
    best/tied nodes, and the TSV at output_path."""
    seqs = extract_read_sequences(reads1, reads2) if reads1 else []
    placer.reset()
    if seqs:
        rs = ReadSet(ctx, seqs)
        placer.add_reads(rs, params)
        rs.close()
    res = placer.score(params, len(seqs))
    if output_path and node_id is not None:
        with open(output_path, "w") as f:
            f.write(format_placement_tsv(res, node_id))
    return res


def refine_top_candidates(parent, scores5, best_index, score_node, params=None):
    """refineTopCandidates (src/placement.cpp:516-698) over plain arrays: parent[n] (DFS indices), scores5[n][5] (metric
    order of the TSV), best_index[5]; score_node(dfs_index) -> int (minus the reads' total edit distance against that
    node's genome).  -> dict(ran, score[5], node[5], candidates, candidate_scores)"""
    parent = np.ascontiguousarray(parent, np.uint32)
    scores5 = np.ascontiguousarray(scores5, np.float64)
    n = len(parent)
    assert scores5.shape == (n, 5)
    best = (C.c_uint32 * 5)(*[int(b) & 0xFFFFFFFF for b in best_index])
    rp = params if params is not None else _lib.RefineParams()
    err = []

    def _cb(_user, node, out):
        try:
            out[0] = int(score_node(int(node)))
            return 0
        except Exception as e:     # noqa: BLE001  (reported after the C call returns)
            err.append(e)
            return -3
    cb = _lib.REFINE_SCORE_FN(_cb)
    res = _lib.RefineResult()
    cand = np.zeros(n, np.uint32)
    cs = np.zeros(n, np.int64)
    rc = lib.pmx_refine_top_candidates(parent.ctypes.data, n, scores5.ctypes.data, best, C.byref(rp), cb, None, C.byref(res), cand.ctypes.data,
                                       cs.ctypes.data, n)
    if err:
        raise err[0]
    check(rc, "pmx_refine_top_candidates")
    k = res.n_candidates
    return dict(ran=bool(res.ran), score=list(res.score), node=list(res.node), candidates=cand[:k].copy(), candidate_scores=cs[:k].copy())


def refine_candidates(parent, scores5, best_index, params=None) -> np.ndarray:
    """the nodes --refine aligns against (steps 1-2 of refineTopCandidates), ascending"""
    parent = np.ascontiguousarray(parent, np.uint32)
    scores5 = np.ascontiguousarray(scores5, np.float64)
    n = len(parent)
    best = (C.c_uint32 * 5)(*[int(b) & 0xFFFFFFFF for b in best_index])
    rp = params if params is not None else _lib.RefineParams()
    cand = np.zeros(n, np.uint32)
    k = lib.pmx_refine_candidates(parent.ctypes.data, n, scores5.ctypes.data, best, C.byref(rp), cand.ctypes.data, n)
    if k < 0:
        check(int(k), "pmx_refine_candidates")
    return cand[:k].copy()


def refine_placement(ctx: Context, placer: "Placer", pm: Panman, result: "PlacementResult", rs: "ReadSet", paired: bool, mean_read_len: int,
                     params=None, aligner=None, streams: int = 4):
    """--refine on the device: every candidate's genome is indexed (pmx_aligner_set_reference) and the reads are aligned
    against it (pmx_align_score_reads); reads as extractReadSequences leaves them (mate 2 as sequenced,
    src/placement.cpp:164-197, 1910-1914).  The candidates are independent: `streams` aligners, each with its own context
    (stream) and host thread, score them concurrently -- one sample's reads do not fill the GPU, and most of a small
    batch's time is the latency of its few hard pairs."""
    import concurrent.futures
    import threading
    parent = placer.index.arrays()["parent"]
    scores5 = placer.node_outputs()[0]
    cands = refine_candidates(parent, scores5, result.best_index, params)
    scores = {}
    if len(cands):
        first = int(cands[0])      # serially: also computes the read set's locality order once
        al0 = aligner if aligner is not None else Aligner(ctx, pm.genome(first), mean_read_len)
        if aligner is not None:
            al0.set_reference(pm.genome(first), mean_read_len)
        scores[first] = al0.score_reads(rs, paired, False)
        ctx.synchronize()
        n_workers = max(1, min(int(streams), len(cands) - 1))
        local = threading.local()
        pool0 = {"taken": False}
        lock = threading.Lock()

        def work(node):
            if not hasattr(local, "al"):
                with lock:
                    mine = not pool0["taken"]
                    pool0["taken"] = True
                if mine:
                    local.ctx, local.al = ctx, al0
                else:
                    local.ctx = Context(ctx.device)
                    local.al = Aligner(local.ctx, pm.genome(node), mean_read_len)
            local.al.set_reference(pm.genome(node), mean_read_len)
            return node, local.al.score_reads(rs, paired, False)
        with concurrent.futures.ThreadPoolExecutor(n_workers) as ex:
            for node, sc in ex.map(work, [int(c) for c in cands[1:]]):
                scores[node] = sc
    return refine_top_candidates(parent, scores5, result.best_index, lambda node: scores[node], params)


def format_refined_tsv(refined, node_id) -> str:
    """the refined_<metric> lines of <prefix>.placement.tsv (src/placement.cpp:1987-2000)"""
    if not refined["ran"]:
        return ""
    lines = []
    for m, name in enumerate(METRICS):
        if refined["node"][m] != 0xFFFFFFFF:
            lines.append("refined_%s\t%.0f\t%s" % (name, float(refined["score"][m]), node_id(refined["node"][m])))
    return "\n".join(lines) + ("\n" if lines else "")


# ------------------------------------------------------------------------------------ align
def last_error() -> bytes:
    """pmx_last_error() of the calling thread (the drop-in boundary has no return code: it reports here)"""
    return lib.pmx_last_error()


DP_RESULT_DTYPE = np.dtype([("served", "<i4"), ("max", "<u4"), ("zdropped", "<i4"), ("max_q", "<i4"), ("max_t", "<i4"), ("mqe", "<i4"), ("mqe_t", "<i4"),
                            ("mte", "<i4"), ("mte_q", "<i4"), ("score", "<i4"), ("n_cigar", "<i4"), ("reach_end", "<i4"), ("cigar", "<u4", (20,))])
REC_DTYPE = np.dtype([("rs", "<i4"), ("re", "<i4"), ("qs", "<i4"), ("qe", "<i4"), ("mapq", "u1"), ("rev", "u1"),
                      ("proper_frag", "u1"), ("mapped", "u1"), ("n_cigar", "<u2"), ("flags", "<u2"), ("cigar_off", "<u4"),
                      ("score", "<i4")])
ALN_OVERFLOW, ALN_UNSUPPORTED, ALN_HAS_ALN = 1, 2, 4
INT_MAX = 2147483647


class Aligner:
    """Device-resident minimizer index of one reference genome + the map/align kernel
    (setup_minimap2 + align workers, src/mm_align.c:118-188, 304-354)."""

    def __init__(self, ctx: Context, reference: bytes, mean_read_len: int):
        self.ctx = ctx
        self._h = C.c_void_p()
        self._ref = bytes(reference)
        check(lib.pmx_aligner_create(ctx._h, self._ref, len(self._ref), int(mean_read_len), C.byref(self._h)), "pmx_aligner_create")

    def set_reference(self, reference: bytes, mean_read_len: int):
        self._ref = bytes(reference)
        check(lib.pmx_aligner_set_reference(self.ctx._h, self._h, self._ref, len(self._ref), int(mean_read_len)), "pmx_aligner_set_reference")

    def score_reads(self, rs: ReadSet, paired: bool, revcomp_mate2: bool = False) -> int:
        """score_reads_vs_reference (src/mm_align.c:144-199): minus the total edit distance of the reads against the
        aligner's reference"""
        out = C.c_int64(0)
        check(lib.pmx_align_score_reads(self.ctx._h, self._h, rs._h, int(paired), int(revcomp_mate2), C.byref(out)), "pmx_align_score_reads")
        return int(out.value)

    def index_digest(self):
        """(occurrences, distinct minimizers, mid_occ, digest, built_on_device) of the current reference index"""
        out = (C.c_uint64 * 5)()
        check(lib.pmx_aligner_index_digest(self.ctx._h, self._h, out), "pmx_aligner_index_digest")
        return tuple(int(x) for x in out)

    def align_readset(self, rs: ReadSet, paired: bool, revcomp_mate2: bool = False):
        check(lib.pmx_align_readset(self.ctx._h, self._h, rs._h, int(paired), int(revcomp_mate2)), "pmx_align_readset")

    def fetch(self):
        n = lib.pmx_align_num_records(self._h)
        words = lib.pmx_align_cigar_words(self.ctx._h, self._h)
        if words < 0:
            raise _lib.PmxError(int(words), "pmx_align_cigar_words")
        recs = np.zeros(max(n, 1), REC_DTYPE)
        cig = np.zeros(max(words, 1), np.uint32)
        check(lib.pmx_align_fetch(self.ctx._h, self._h, recs.ctypes.data, len(recs), cig.ctypes.data, len(cig)), "pmx_align_fetch")
        return recs[:n], cig[:words]

    def fetch_async(self, host_records_ptr: int, n_records: int, host_cigars_ptr: int, cigar_cap: int, stream: int = 0):
        """enqueue the download of the last results into (pinned) host buffers on `stream` (a raw hipStream_t; 0 = the
        context's) without waiting; the next align_readset waits for it before it overwrites the device buffers"""
        check(lib.pmx_align_fetch_async(self.ctx._h, self._h, host_records_ptr, n_records, host_cigars_ptr, cigar_cap, stream or None), "pmx_align_fetch_async")

    def copy_records_device(self, d_ptr: int, n_records: int):
        check(lib.pmx_align_copy_records_device(self.ctx._h, self._h, d_ptr, n_records), "pmx_align_copy_records_device")

    def copy_cigars_device(self, d_ptr: int, n_words: int):
        check(lib.pmx_align_copy_cigars_device(self.ctx._h, self._h, d_ptr, n_words), "pmx_align_copy_cigars_device")

    def cigar_words(self) -> int:
        return int(lib.pmx_align_cigar_words(self.ctx._h, self._h))

    def scoring(self) -> dict:
        """the DP scoring of the preset in force: a, b, q, e, q2, e2, sc_ambi, zdrop, end_bonus"""
        v = (C.c_int32 * 9)()
        check(lib.pmx_align_scoring(self._h, v), "pmx_align_scoring")
        return dict(zip(("a", "b", "q", "e", "q2", "e2", "sc_ambi", "zdrop", "end_bonus"), (int(x) for x in v)))

    def dp_batch(self, queries, targets, w, zdrop, end_bonus, flag, reps: int = 1):
        """ksw_extd2 on (query, target) pairs of nt4 codes through the grouped DP service (pmx_align_dp_batch).
        -> (structured array of DP_RESULT_DTYPE, milliseconds of one service launch)"""
        n = len(queries)
        qa = [np.asarray(a, np.uint8) for a in queries]
        ta = [np.asarray(b, np.uint8) for b in targets]
        qs, ts = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
        if n:
            qs[1:] = np.cumsum([len(a) for a in qa])
            ts[1:] = np.cumsum([len(b) for b in ta])
        ts += qs[-1]                                       # all queries first, then all targets
        seqs = np.ascontiguousarray(np.concatenate(qa + ta)) if n else np.zeros(1, np.uint8)
        out = np.zeros(n, DP_RESULT_DTYPE)
        ms = C.c_double(0)

        def arr(x):
            return np.ascontiguousarray(np.broadcast_to(np.asarray(x, np.int32), (n,)))
        w_, z_, e_, f_ = arr(w), arr(zdrop), arr(end_bonus), arr(flag)
        check(lib.pmx_align_dp_batch(self.ctx._h, self._h, seqs.ctypes.data, qs.ctypes.data, ts.ctypes.data, n, w_.ctypes.data, z_.ctypes.data,
                                     e_.ctypes.data, f_.ctypes.data, out.ctypes.data, int(reps), C.byref(ms)), "pmx_align_dp_batch")
        return out, float(ms.value)

    def stats(self) -> dict:
        """work statistics of the last align_readset call (DP cells = q * min(t, 2w+1) per ksw2 call)"""
        st = _lib.AlignStats()
        check(lib.pmx_align_get_stats(self.ctx._h, self._h, C.byref(st)), "pmx_align_get_stats")
        return {k: int(getattr(st, k)) for k, _ in st._fields_ if k != "reserved"}

    @property
    def n_records(self) -> int:
        return lib.pmx_align_num_records(self._h)

    def align_reads(self, reads, paired: bool, revcomp_mate2: bool = False):
        """-> list of per-pair (or per-read) dicts shaped like align_pair_result_t."""
        rs = ReadSet(self.ctx, reads)
        self.align_readset(rs, paired, revcomp_mate2)
        recs, cig = self.fetch()
        rs.close()
        return records_to_results(recs, cig, paired)

    def close(self):
        if self._h:
            lib.pmx_aligner_free(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dist:
    """One rank of the multi-GPU exchange (pmx_dist_*: RCCL over xGMI behind the C ABI).
    `uid` = the 128 bytes rank 0 made with Dist.unique_id() and shipped to every rank."""
    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Dist.ID_BYTES)
        check(lib.pmx_dist_unique_id(buf), "pmx_dist_unique_id")
        return buf.raw

    def __init__(self, ctx: Context, uid: bytes, rank: int, world: int):
        assert len(uid) == Dist.ID_BYTES
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self._h = C.c_void_p()
        self._uid = C.create_string_buffer(bytes(uid), Dist.ID_BYTES)
        check(lib.pmx_dist_init(ctx._h, self._uid, self.rank, self.world, C.byref(self._h)), "pmx_dist_init")

    def barrier(self):
        check(lib.pmx_dist_barrier(self._h), "pmx_dist_barrier")

    def merge_histograms(self, placer: Placer):
        """all-gather + merge: every rank's placer then holds the histogram of the whole sample"""
        check(lib.pmx_dist_merge_histograms(self._h, placer._h), "pmx_dist_merge_histograms")

    def dedup_reads(self, placer: Placer, rs: ReadSet) -> int:
        """--dedup over the whole sample: prepares the placer's keep mask for `rs` (follow with add_reads(rs, dedupReads=True));
        returns the reads this rank still keeps"""
        kept = C.c_int64(0)
        check(lib.pmx_dist_dedup_reads(self._h, placer._h, rs._h, C.byref(kept)), "pmx_dist_dedup_reads")
        return int(kept.value)

    def gather_alignments(self, aligner: "Aligner", root: int = 0):
        """records + CIGAR arena of the aligner's last call to `root` -> (n_records, n_words) there, (0, 0) elsewhere"""
        nr, nw = C.c_int64(0), C.c_int64(0)
        check(lib.pmx_dist_gather_alignments(self._h, aligner._h, int(root), C.byref(nr), C.byref(nw)), "pmx_dist_gather_alignments")
        return int(nr.value), int(nw.value)

    def plan_alignments(self, aligner: "Aligner"):
        """one-node form of the gather: -> (record_base, word_base, total_records, total_words) of this rank's part in the one
        result set of all ranks (pmx_dist_plan_alignments); follow with fetch_shard_async"""
        v = [C.c_int64(0) for _ in range(4)]
        check(lib.pmx_dist_plan_alignments(self._h, aligner._h, *[C.byref(x) for x in v]), "pmx_dist_plan_alignments")
        return tuple(int(x.value) for x in v)

    def fetch_shard_async(self, aligner: "Aligner", host_records_all_ptr: int, host_cigars_all_ptr: int, stream: int = 0):
        """download this rank's records (cigar_off rebased) and CIGAR words to their place in the whole result set's host buffers"""
        check(lib.pmx_dist_fetch_shard_async(self._h, aligner._h, host_records_all_ptr, host_cigars_all_ptr, stream or None), "pmx_dist_fetch_shard_async")

    def gathered_device_pointers(self):
        return int(lib.pmx_dist_gathered_records(self._h) or 0), int(lib.pmx_dist_gathered_cigars(self._h) or 0)

    def rank_counts(self):
        r, w = np.zeros(self.world, np.int64), np.zeros(self.world, np.int64)
        check(lib.pmx_dist_rank_counts(self._h, r.ctypes.data, w.ctypes.data), "pmx_dist_rank_counts")
        return r, w

    def fetch_gathered(self, n_records: int, n_words: int):
        recs = np.zeros(max(n_records, 1), REC_DTYPE)
        cig = np.zeros(max(n_words, 1), np.uint32)
        check(lib.pmx_dist_fetch_gathered(self._h, recs.ctypes.data, len(recs), cig.ctypes.data, len(cig)), "pmx_dist_fetch_gathered")
        return recs[:n_records], cig[:n_words]

    def fetch_gathered_async(self, host_records_ptr: int, n_records: int, host_cigars_ptr: int, cigar_cap: int, stream: int = 0):
        check(lib.pmx_dist_fetch_gathered_async(self._h, host_records_ptr, n_records, host_cigars_ptr, cigar_cap, stream or None), "pmx_dist_fetch_gathered_async")

    def close(self):
        if self._h:
            lib.pmx_dist_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _one_result(r, cig):
    if not r["mapped"] or not (r["flags"] & ALN_HAS_ALN):
        return dict(pos=INT_MAX, rs=0, re=0, qs=0, qe=0, mapq=0, rev=0, proper_frag=0, cigar=[])
    return dict(pos=int(r["rs"]) + 1, rs=int(r["rs"]), re=int(r["re"]), qs=int(r["qs"]), qe=int(r["qe"]), mapq=int(r["mapq"]),
                rev=int(r["rev"]), proper_frag=int(r["proper_frag"]),
                cigar=[int(x) for x in cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]])


_UNMAPPED = dict(pos=INT_MAX, rs=0, re=0, qs=0, qe=0, mapq=0, rev=0, proper_frag=0, cigar=[])


def records_to_results(recs, cig, paired: bool):
    """Records flagged ALN_OVERFLOW / ALN_UNSUPPORTED are invalid: like pmx_align_reads_direct, they are reported
    unmapped (never as an alignment); `flags` still says why."""
    out = []
    if paired:
        for i in range(len(recs) // 2):
            a, b = recs[2 * i], recs[2 * i + 1]
            fl = int(a["flags"] | b["flags"])
            if fl & (ALN_OVERFLOW | ALN_UNSUPPORTED):
                out.append(dict(mapped=0, r1=dict(_UNMAPPED), r2=dict(_UNMAPPED), flags=fl))
            else:
                out.append(dict(mapped=int(a["mapped"]), r1=_one_result(a, cig), r2=_one_result(b, cig), flags=fl))
    else:
        for r in recs:
            fl = int(r["flags"])
            if fl & (ALN_OVERFLOW | ALN_UNSUPPORTED):
                out.append(dict(mapped=0, r1=dict(_UNMAPPED), r2=None, flags=fl))
            else:
                out.append(dict(mapped=int(r["mapped"]), r1=_one_result(r, cig), r2=None, flags=fl))
    return out


def align_reads_direct(reference: bytes, reads, paired: bool, n_threads: int = 1):
    """The reference's C-ABI boundary (src/mm_align.h:44-53) through libpanmap_amd.so."""
    n = len(reads)
    arr = (C.c_char_p * n)(*reads)
    quals = (C.c_char_p * n)(*[b"I" * len(r) for r in reads])
    names = (C.c_char_p * n)(*[b"r%d" % i for i in range(n)])
    lens = (C.c_int * n)(*[len(r) for r in reads])
    n_res = n // 2 if paired else n
    res = (_lib.AlignPairResult * max(n_res, 1))()
    lib.pmx_align_reads_direct(reference, b"ref", n, arr, quals, names, lens, res, paired, n_threads)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]

    def unpack(ra):
        cg = [ra.cigar[i] for i in range(ra.n_cigar)] if ra.cigar else []
        if ra.cigar:
            libc.free(C.cast(ra.cigar, C.c_void_p))
        return dict(pos=ra.pos, rs=ra.rs, re=ra.re, qs=ra.qs, qe=ra.qe, mapq=ra.mapq, rev=ra.rev, proper_frag=ra.proper_frag, cigar=cg)
    return [dict(mapped=res[i].mapped, r1=unpack(res[i].r1), r2=unpack(res[i].r2) if paired else None) for i in range(n_res)]


def score_reads_vs_reference(reference: bytes, reads, paired: bool, kmer_size: int = 0) -> int:
    """the reference's scorer boundary (src/mm_align.h:13-17) through libpanmap_amd.so"""
    n = len(reads)
    arr = (C.c_char_p * n)(*reads)
    lens = (C.c_int * n)(*[len(r) for r in reads])
    return int(lib.pmx_score_reads_vs_reference(reference, n, arr, lens, kmer_size, paired))


def write_bam(bam_path: str, ref_name: str, ref_len: int, seqs, quals, names, results, paired: bool):
    """BAM + .bai egress of alignment results (alignAndWriteBam, src/conversion.cpp:426-538).  seqs / quals / names are
    what the aligner was given (R2 reverse-complemented, its qualities reversed); results as returned by
    align_reads_direct / Aligner.align_reads."""
    n = len(seqs)
    arr = (C.c_char_p * n)(*seqs)
    qarr = (C.c_char_p * n)(*quals)
    narr = (C.c_char_p * n)(*names)
    lens = (C.c_int * n)(*[len(r) for r in seqs])
    n_res = n // 2 if paired else n
    res = (_lib.AlignPairResult * max(n_res, 1))()
    keep = []

    def fill(dst, d):
        if d is None:
            return
        for f in ("pos", "rs", "re", "qs", "qe", "mapq", "rev", "proper_frag"):
            setattr(dst, f, int(d[f]))
        cg = (C.c_uint32 * max(len(d["cigar"]), 1))(*[int(x) for x in d["cigar"]])
        keep.append(cg)
        dst.n_cigar = len(d["cigar"])
        dst.cigar = C.cast(cg, C.POINTER(C.c_uint32))
    for i in range(n_res):
        res[i].mapped = int(results[i]["mapped"])
        fill(res[i].r1, results[i]["r1"])
        fill(res[i].r2, results[i].get("r2"))
    check(lib.pmx_write_bam(bam_path.encode(), ref_name.encode(), ref_len, n, arr, qarr, narr, lens, res, paired), "pmx_write_bam")
