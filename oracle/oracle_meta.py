"""ORACLE (test infrastructure only -- never imported by the product path): a direct restatement of the --meta scoring and
EM of the reference (src/mgsr.cpp), written for clarity, not speed.

  * node_seed_counts: the (forward, reverse) occurrence counts of every seedmer hash in a node's genome, by walking the
    oriented index root -> node (what kminmerOnRefCount holds when scoreReadsHelper visits the node, src/mgsr.cpp:7246-7275);
  * read_scores: per read, f = its seedmers the genome holds in the read's orientation, r = in the other one; score
    max(f, r) (updateReadsForSeed, :7237-7259: a seedmer counts once per orientation whatever its number of occurrences);
  * overlap_coefficient: distinct read seedmer hashes the genome holds / the genome's distinct hashes (:5685-5745);
  * square_em: updateProps / normalizeProps / getExp / runSquareEM / removeLowPropNodes (:4341-4490), numpy float64.

parity: UNPINNED against the reference itself -- the MGSR index and the EM need panman v0.1.4, TBB, Eigen and abseil, none of
which can be built here; the reference holds no golden vector for this mode except the e2e expectation of
src/test/e2e/run_e2e.sh:182-204 (a 70/30 mixture on rsv_4K recovered within stated ranges), which tests/test_meta_gpu.py
reproduces.  The demo's reads (examples/data/reads/sars20000_5hap_*) are absent from the checkout (.MISSING_LARGE_BLOBS)."""
import numpy as np

ORIENT_XOR = 0x9e3779b97f4a7c15


def node_seed_counts(arrays, node, read_hashes=None):
    """{hash: [forward count, reverse count]} of `node` from the oriented index arrays (Index.arrays() of a
    PMX_INDEX_ORIENTED build); an oriented key is told from a forward one through `read_hashes` when given (a key whose
    XOR-ed form is a read hash is a reverse occurrence of that hash), else every key is kept as its own entry"""
    parent, off = arrays["parent"], arrays["offsets"]
    path, v = [], int(node)
    while True:
        path.append(v)
        if v == 0:
            break
        v = int(parent[v])
    state = {}
    for v in reversed(path):
        for h, cc in zip(arrays["hash"][off[v]:off[v + 1]].tolist(), arrays["child_count"][off[v]:off[v + 1]].tolist()):
            if cc == 0:
                state.pop(h, None)
            else:
                state[h] = cc
    if read_hashes is None:
        return state
    out = {}
    for key, c in state.items():
        if key in read_hashes:
            out.setdefault(key, [0, 0])[0] += c
        elif (key ^ ORIENT_XOR) in read_hashes:
            out.setdefault(key ^ ORIENT_XOR, [0, 0])[1] += c
    return out


def _rol(x, r):
    r &= 63
    return ((x << r) | (x >> (64 - r))) & 0xFFFFFFFFFFFFFFFF if r else x


def seedmers(seq: bytes, k, s, l, open_syncmer=False, t=0):
    """the seedmers of a sequence as the reference's query side lists them (src/mgsr.cpp:1855-1950): over every window of l
    consecutive syncmers F = rol-xor of the hashes left to right, R right to left; a window with F != R gives
    (min(F, R), R < F).  The syncmers come from the oracle's rollingSyncmers restatement (oracle/oracle_place.c, pinned by the
    compiled reference's known answers) -- nothing of the product is involved."""
    from . import oracle as orc
    assert l >= 2
    sy = [h for h, _, is_s, _ in orc.rolling_syncmers(seq, k, s, open_syncmer, t, return_all=True) if is_s]
    out = []
    for i in range(len(sy) - l + 1):
        f = r = 0
        for q in range(l):
            f = _rol(f, k) ^ sy[i + q]
            r = _rol(r, k) ^ sy[i + l - 1 - q]
        if f != r:
            out.append((min(f, r), r < f))
    return out


def genome_seed_counts(genome: bytes, k, s, l, open_syncmer=False, t=0):
    """{hash: [forward, reverse]} occurrence counts of a genome's seedmers, from the genome string alone"""
    out = {}
    for h, rev in seedmers(genome, k, s, l, open_syncmer, t):
        out.setdefault(h, [0, 0])[1 if rev else 0] += 1
    return out


def read_scores(counts, read_off, seed_hash, seed_rev):
    """scores of every read at one node given node_seed_counts(..., read_hashes)"""
    n = len(read_off) - 1
    out = np.zeros(n, np.int64)
    for r in range(n):
        f = b = 0
        for i in range(read_off[r], read_off[r + 1]):
            c = counts.get(int(seed_hash[i]))
            if c is None:
                continue
            rev = int(seed_rev[i])
            f += c[rev] > 0          # the genome holds it the way the read does
            b += c[1 - rev] > 0
        out[r] = max(f, b)
    return out


def overlap_coefficient(arrays_unoriented, node, read_hashes):
    st = node_seed_counts(arrays_unoriented, node)
    return (sum(1 for h in st if h in read_hashes) / len(st)) if st else 0.0


def _normalize(p):
    p = np.where(p <= 0, 1e-12, p)
    return p / p.sum()


def square_em(scores, n_seedmers, weight, error_rate=0.005, eta=1e-5, delta_threshold=0.0, max_iterations=1000, max_rounds=5, prop_threshold=0.005):
    """scores [reads][columns] (reads without any score excluded by the caller) -> (kept column indices, proportions)"""
    scores = np.asarray(scores, np.int64)
    n = np.asarray(n_seedmers, np.int64)[:, None]
    probs = np.power(error_rate, (n - scores).astype(np.float64)) * np.power(1.0 - error_rate, scores.astype(np.float64))
    w = np.asarray(weight, np.float64)
    cols = np.arange(scores.shape[1])
    inv_total = 1.0 / w.sum()
    props = None
    for _ in range(max(1, max_rounds)):
        P = probs[:, cols]
        k = len(cols)
        props = np.full(k, 1.0 / k)

        def step(p):
            inv = 1.0 / (P @ p)
            return np.array([(w * (P[:, i] * p[i] * inv)).sum() * inv_total for i in range(k)])

        def llh_of(p):
            return float((w * np.log(P @ p)).sum())
        llh = 0.0
        for _it in range(max_iterations):
            p0 = props
            p1 = _normalize(step(p0))
            p2 = _normalize(step(p1))
            r = p1 - p0
            v = (p2 - p1) - r
            with np.errstate(divide="ignore", invalid="ignore"):
                alpha = -np.linalg.norm(r) / np.linalg.norm(v)
                sq = _normalize(p0 - 2.0 * alpha * r + alpha * alpha * v)
                l2, lsq = llh_of(p2), llh_of(sq)
            if lsq > l2 - eta:
                props, diff, llh = sq, lsq - llh, lsq
            else:
                props, diff, llh = p2, l2 - llh, l2
            if delta_threshold == 0:
                if abs(diff) < eta:
                    break
            elif np.abs(props - p0).max() < delta_threshold:
                break
        # removeLowPropNodes (src/mgsr.cpp:4445-4490) runs after every round, the last allowed one included
        # (src/main.cpp:1263-1271); a removal resets the survivors' proportions to uniform
        keep = props >= prop_threshold
        if keep.all():
            break
        cols = cols[keep]
        props = np.full(len(cols), 1.0 / len(cols)) if len(cols) else np.zeros(0)
        if not len(cols):
            break
    return cols, props


def discard_rows(max_score, n_seedmers, discard):
    """reads that enter the EM (src/main.cpp:1229-1240): a positive best score that is not below int(n * discard)"""
    mx = np.asarray(max_score, np.int64)
    thr = (np.asarray(n_seedmers, np.float64) * float(discard)).astype(np.int64)    # static_cast<int>: truncation
    return (mx > 0) & ~(mx < thr)


def read_scores_np(counts, read_off, seed_hash, seed_rev):
    """read_scores with the per-seedmer look-ups done by numpy (same numbers; for the 10^5-read cases)"""
    read_off = np.asarray(read_off, np.int64)
    seed_hash = np.asarray(seed_hash, np.uint64)
    seed_rev = np.asarray(seed_rev).astype(bool)
    if not counts:
        return np.zeros(len(read_off) - 1, np.int64)
    keys = np.fromiter(counts.keys(), np.uint64, len(counts))
    fw = np.fromiter((c[0] > 0 for c in counts.values()), bool, len(counts))
    rv = np.fromiter((c[1] > 0 for c in counts.values()), bool, len(counts))
    order = np.argsort(keys)
    keys, fw, rv = keys[order], fw[order], rv[order]
    pos = np.minimum(np.searchsorted(keys, seed_hash), len(keys) - 1)
    hit = keys[pos] == seed_hash
    same = hit & np.where(seed_rev, rv[pos], fw[pos])       # the genome holds it the way the read does
    other = hit & np.where(seed_rev, fw[pos], rv[pos])
    cs = np.concatenate([[0], np.cumsum(same)])
    co = np.concatenate([[0], np.cumsum(other)])
    return np.maximum(cs[read_off[1:]] - cs[read_off[:-1]], co[read_off[1:]] - co[read_off[:-1]]).astype(np.int64)


def get_dust(seq: bytes, window_size: int = 64) -> float:
    """mgsr::getDust (src/mgsr.cpp:1505-1568): Prinseq-scaled DUST score over triplets.  Bases other than ACGT (either
    case) are skipped altogether (the triplet register runs over them); while fewer than `window_size` triplets were seen
    the score only accumulates; from then on the triplet that leaves the window is taken out first (its count decremented
    when positive, the new count subtracted), the entering one added, and the maximum of the running score is kept.  Result:
    200 max / (W (W - 1)) once a full window was seen, else 200 score / (v (v + 1)) with v = triplets - 1 when there are at
    least two triplets, else 0.  `--dust T` (< 100) drops a read whose score is non-zero and > T (:1593-1594, :1833-1834)."""
    assert window_size >= 3
    code = {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3}
    counts = [0] * 64
    window = [0] * window_size
    cur = best = 0
    kmer = 0
    valid = -3
    for ch in seq:
        b = code.get(ch)
        if b is None:
            continue
        kmer = ((kmer << 2) | b) & 63
        valid += 1
        if valid < 0:
            continue
        slot = valid % window_size
        if valid >= window_size:
            out = window[slot]
            if counts[out] > 0:
                counts[out] -= 1
                cur -= counts[out]
            cur += counts[kmer]
            counts[kmer] += 1
            best = max(best, cur)
        else:
            cur += counts[kmer]
            counts[kmer] += 1
        window[slot] = kmer
    n_kmers = valid + 1
    if valid >= window_size:
        return (200.0 * best) / (window_size * (window_size - 1))
    if n_kmers > 1:
        return (200.0 * cur) / (valid * (valid + 1))
    return 0.0
