"""Deterministic synthetic read generators for the benchmark configs.

Short reads: fragment start uniform, insert ~ round(N(300,30)) clamped to [150,600], R1 = first 150 bp
forward, R2 = reverse complement of the last 150 bp (FR); i.i.d. substitutions at `sub_rate`, no indels,
no N.  Reads are returned in FASTQ orientation (R2 as sequenced), interleaved R1,R2,...
RNG: numpy PCG64 seeded with `seed` (documented draw order: inserts, starts, error mask, error bases).
Deviation from SURVEY.md 8d, on purpose: the survey sketches a splitmix64 stream and a source leaf chosen as the
splitmix64(42) mod 20000-th leaf; this generator uses numpy's PCG64 and callers pass the genome (bench.py: node_7618,
the node the repository's example sample places on).  Nothing downstream depends on which deterministic stream it is.
"""
import numpy as np

_COMP = np.arange(256, dtype=np.uint8)   # anything but ACGT (N runs of a genome) complements to itself
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


def simulate_paired_reads(genome: bytes, n_pairs: int, read_len: int = 150, seed: int = 42, sub_rate: float = 0.002,
                          mean_insert: float = 300.0, sd_insert: float = 30.0, threads: int = 1):
    """-> (concat uint8 array, offsets int64[n_reads+1]) with n_reads = 2*n_pairs, fixed read_len.
    More than a million pairs are drawn in blocks of a million (block c with seed + 1000003 * c), which bounds the
    generator's memory (the error mask costs 8 bytes per base while it exists); up to a million pairs the stream is the
    single-block one the committed fixtures were made with.  The blocks are independent streams: `threads` > 1 draws several
    at a time (same result; it only pays on hosts with memory bandwidth to spare -- measured slower on an 8-core box)."""
    block = 1000000
    if n_pairs > block:
        from concurrent.futures import ThreadPoolExecutor
        out = np.empty(2 * n_pairs * read_len, np.uint8)

        def draw(c_lo):
            c, lo = c_lo
            part, _ = simulate_paired_reads(genome, min(block, n_pairs - lo), read_len, seed + 1000003 * c, sub_rate, mean_insert, sd_insert)
            out[2 * lo * read_len:2 * lo * read_len + part.size] = part
        with ThreadPoolExecutor(max(1, int(threads))) as ex:
            list(ex.map(draw, enumerate(range(0, n_pairs, block))))
        return out, np.arange(2 * n_pairs + 1, dtype=np.int64) * read_len
    g = np.frombuffer(genome, np.uint8)
    G = len(g)
    rng = np.random.Generator(np.random.PCG64(seed))
    ins = np.clip(np.rint(rng.normal(mean_insert, sd_insert, n_pairs)), read_len, min(600, G)).astype(np.int64)
    start = (rng.random(n_pairs) * (G - ins + 1)).astype(np.int64)
    ar = np.arange(read_len, dtype=np.int64)
    r1 = g[start[:, None] + ar[None, :]]
    # R2 = reverse complement of the last read_len bases of the fragment
    r2 = _COMP[g[(start + ins - 1)[:, None] - ar[None, :]]]
    reads = np.empty((2 * n_pairs, read_len), np.uint8)
    reads[0::2] = r1
    reads[1::2] = r2
    err = rng.random(reads.shape) < sub_rate
    n_err = int(err.sum())
    if n_err:
        shift = rng.integers(1, 4, n_err)
        code = np.zeros(256, np.uint8)
        for i, b in enumerate(b"ACGT"):
            code[b] = i
        bases = np.frombuffer(b"ACGT", np.uint8)
        reads[err] = bases[(code[reads[err]] + shift) & 3]
    off = np.arange(2 * n_pairs + 1, dtype=np.int64) * read_len
    return reads.reshape(-1), off


def simulate_long_reads(genome: bytes, n_reads: int, read_len: int = 10000, seed: int = 43, sub=0.02, ins=0.015, dele=0.015):
    """Single-end long reads with substitutions / insertions / deletions (config 4).  -> list of bytes."""
    g = np.frombuffer(genome, np.uint8)
    G = len(g)
    rng = np.random.Generator(np.random.PCG64(seed))
    bases = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for _ in range(n_reads):
        st = int(rng.integers(0, max(G - read_len, 0) + 1))
        frag = g[st:st + read_len]
        if rng.random() < 0.5:
            frag = _COMP[frag[::-1]]
        u = rng.random(len(frag))
        keep = u >= dele
        issub = (u >= dele) & (u < dele + sub)
        frag = frag.copy()
        ns = int(issub.sum())
        if ns:
            frag[issub] = bases[rng.integers(0, 4, ns)]
        isins = rng.random(len(frag)) < ins
        pieces = np.where(keep, 1, 0) + np.where(isins, 1, 0)
        res = np.empty(int(pieces.sum()), np.uint8)
        pos = np.cumsum(pieces) - pieces
        res[pos[keep] + isins[keep]] = frag[keep]
        ni = int(isins.sum())
        if ni:
            res[pos[isins]] = bases[rng.integers(0, 4, ni)]
        out.append(res.tobytes())
    return out
