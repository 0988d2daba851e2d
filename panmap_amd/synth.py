"""Deterministic synthetic read generators for the benchmark configs.

Short reads: fragment start uniform, insert ~ round(N(300,30)) clamped to [150,600], R1 = first 150 bp
forward, R2 = reverse complement of the last 150 bp (FR); i.i.d. substitutions at `sub_rate`, no indels,
no N.  Reads are returned in FASTQ orientation (R2 as sequenced), interleaved R1,R2,...
RNG: numpy PCG64 seeded with `seed` (documented draw order: inserts, starts, error mask, error bases).
`simulate_paired_reads` (numpy PCG64; callers pass the genome) is what the committed fixtures and the tests were made
with.  `simulate_paired_reads_8d` + `source_leaf_8d` are SURVEY.md 8d to the letter -- the (splitmix64(42) mod 20000)-th leaf
of the tree, ONE splitmix64 stream with a fixed draw order -- and what bench.py runs (round 4).
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


# ------------------------------------------------------------------------------------------------ SURVEY.md 8d, to the letter
_SM_GAMMA = np.uint64(0x9E3779B97F4A7C15)
DRAWS_PER_PAIR = 35          # 1 start + 2 insert (Box-Muller) + 16 error gaps + 16 replacement bases
_MAX_ERR = 16


def splitmix64(seed: int, first: int, count: int) -> np.ndarray:
    """outputs number first .. first + count - 1 (0-based) of the splitmix64 stream seeded with `seed`: the stream is
    counter-based (state_i = seed + (i + 1) * gamma), so any stretch of it is computed directly"""
    with np.errstate(over="ignore"):
        i = np.arange(first + 1, first + 1 + count, dtype=np.uint64)
        z = np.uint64(seed) + i * _SM_GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def source_leaf_8d(parent: np.ndarray, seed: int = 42) -> int:
    """DFS index of the (splitmix64(seed) mod n_leaves)-th leaf of the tree, leaves counted in DFS order (`parent`: the
    DFS-ordered parent array, parent[0] ignored)"""
    parent = np.asarray(parent)
    has_child = np.zeros(len(parent), bool)
    has_child[parent[1:]] = True
    leaves = np.nonzero(~has_child)[0]
    return int(leaves[int(splitmix64(seed, 0, 1)[0] % np.uint64(len(leaves)))])


def simulate_paired_reads_8d(genome: bytes, n_pairs: int, first_pair: int = 0, read_len: int = 150, seed: int = 42, sub_rate: float = 0.002,
                             mean_insert: float = 300.0, sd_insert: float = 30.0):
    """pairs first_pair .. first_pair + n_pairs - 1 of the job's ONE read stream (a rank of a sharded job draws
    its own stretch; the reads of the job are the same whatever the number of ranks).
    Draw order: pair p owns outputs [p * 35, (p + 1) * 35) of the splitmix64(seed) stream, uniform u = (z >> 11) * 2^-53:
      0      fragment start = floor(u * (G - insert + 1))           (insert first, from draws 1-2)
      1, 2   insert = clamp(round(mean + sd * sqrt(-2 ln u1) * cos(2 pi u2)), read_len, min(600, G))      (u1 = 0 -> 2^-53)
      3..18  substitution errors, i.i.d. `sub_rate` per base over the 2 * read_len bases of the pair (R1 then R2 as
             sequenced) as geometric gaps: gap_j = floor(ln(u) / ln(1 - sub_rate)) error-free bases before the j-th error
      19..34 the j-th error's base: the (1 + floor(3 u))-th next base in A C G T order, cyclically
    (more than 16 errors in a pair -- probability below 1e-19 at 0.2 % -- are not drawn.)
    -> (concat uint8, offsets int64[2 n_pairs + 1]); reads in FASTQ orientation, interleaved R1, R2."""
    g = np.frombuffer(genome, np.uint8)
    G = len(g)
    out = np.empty(2 * n_pairs * read_len, np.uint8)
    code = np.zeros(256, np.uint8)
    for i, b in enumerate(b"ACGT"):
        code[b] = i
    bases = np.frombuffer(b"ACGT", np.uint8)
    ar = np.arange(read_len, dtype=np.int64)
    block = 1000000
    for lo in range(0, n_pairs, block):
        n = min(block, n_pairs - lo)
        z = splitmix64(seed, (first_pair + lo) * DRAWS_PER_PAIR, n * DRAWS_PER_PAIR).reshape(n, DRAWS_PER_PAIR)
        u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        u1 = np.maximum(u[:, 1], 1.0 / 9007199254740992.0)
        nrm = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u[:, 2])
        ins = np.clip(np.rint(mean_insert + sd_insert * nrm), read_len, min(600, G)).astype(np.int64)
        start = np.floor(u[:, 0] * (G - ins + 1)).astype(np.int64)
        reads = np.empty((2 * n, read_len), np.uint8)
        reads[0::2] = g[start[:, None] + ar[None, :]]
        reads[1::2] = _COMP[g[(start + ins - 1)[:, None] - ar[None, :]]]
        if sub_rate > 0:
            gaps = np.floor(np.log(np.maximum(u[:, 3:3 + _MAX_ERR], 1.0 / 9007199254740992.0)) / np.log1p(-sub_rate)).astype(np.int64)
            pos = np.cumsum(gaps + 1, axis=1) - 1                                # position of the j-th error among the pair's bases
            hit = pos < 2 * read_len
            shift = 1 + np.floor(3.0 * u[:, 3 + _MAX_ERR:3 + 2 * _MAX_ERR]).astype(np.int64)
            pr, pj = np.nonzero(hit)
            flat = reads.reshape(n, 2 * read_len)                                # (R1 then R2 of a pair are adjacent rows)
            at = pos[pr, pj]
            flat[pr, at] = bases[(code[flat[pr, at]] + shift[pr, pj]) & 3]
        out[2 * lo * read_len:2 * (lo + n) * read_len] = reads.reshape(-1)
    return out, np.arange(2 * n_pairs + 1, dtype=np.int64) * read_len
