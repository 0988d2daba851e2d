"""A9 on its own: the grouped DP service (panmap_amd/csrc/align_kernel_dpg.hip, through pmx_align_dp_batch) against
ksw_extd2_sse of the compiled reference (src/3rdparty/minimap2/ksw2_extd2_sse.c:28-400) on the same inputs, field by field of
ksw_extz_t and CIGAR by CIGAR: the three call shapes of mm_align1 (gap fill with the approximate maximum, align.c:744; right
extension, :790; left extension with the gaps right-aligned and the CIGAR reversed, :699), sequences from near-identical to
unrelated, ambiguous bases, one-base sides, Z-drops that fire, end bonuses on both sides of the decision."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

APPROX_MAX, EXTZ_ONLY, RIGHT, REV_CIGAR = 0x08, 0x40, 0x02, 0x80
KINDS = (APPROX_MAX, EXTZ_ONLY, EXTZ_ONLY | RIGHT | REV_CIGAR)
FIELDS = ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "n_cigar", "reach_end")


@pytest.fixture(scope="module")
def aligner(pmx, ctx):
    rng = np.random.default_rng(1)
    ref = bytes(rng.choice(list(b"ACGT"), 3000).astype(np.uint8))
    return pmx.Aligner(ctx, ref, 150)


def _mutate(rng, seq, sub, indel):
    out = []
    for c in seq:
        r = rng.random()
        if r < indel / 2:
            continue
        if r < indel:
            out += [c, int(rng.integers(0, 4))]
        elif r < indel + sub:
            out.append(int((c + 1 + rng.integers(0, 3)) % 4))
        else:
            out.append(int(c))
    return out or [0]


def _requests(rng, n):
    qs, ts, ws, zs, es, fs = [], [], [], [], [], []
    for i in range(n):
        kind = KINDS[i % 3]
        tl = int(rng.integers(1, 129))
        t = [int(x) for x in rng.integers(0, 4, tl)]
        mode = i % 7
        if mode == 0:
            q = [int(x) for x in rng.integers(0, 4, int(rng.integers(1, 129)))]                 # unrelated
        elif mode == 1:
            q = _mutate(rng, t, 0.25, 0.1)                                                     # diverged: Z-drops fire
        else:
            q = _mutate(rng, t, 0.03 * (mode - 1), 0.02 * (mode - 2))
        if kind != APPROX_MAX and rng.random() < 0.5:
            q = q[:max(1, len(q) // 2)]                                                        # an extension's target overhangs
        q = q[:128]
        if rng.random() < 0.15:
            for _ in range(int(rng.integers(1, 4))):
                (q if rng.random() < 0.5 else t)[int(rng.integers(0, min(len(q), len(t))))] = 4   # N
        qs.append(q); ts.append(t)
        ws.append(-1 if rng.random() < 0.3 else max(len(q), len(t)) - 1 + int(rng.integers(0, 200)))
        zs.append(int(rng.choice([-1, 20, 50, 100, 400])))
        es.append(int(rng.choice([-1, 0, 2, 5, 10, 40])))
        fs.append(kind)
    return qs, ts, ws, zs, es, fs


def _compare(oracle, aligner, qs, ts, ws, zs, es, fs):
    sc = aligner.scoring()
    mat = oracle.simple_mat(sc["a"], sc["b"], sc["sc_ambi"])
    got, _ = aligner.dp_batch(qs, ts, ws, zs, es, fs)
    bad, served = [], 0
    for i in range(len(qs)):
        want = oracle.ref_ksw_extd2(qs[i], ts[i], mat, sc["q"], sc["e"], sc["q2"], sc["e2"], ws[i], zs[i], es[i], fs[i])
        g = got[i]
        if not g["served"]:
            assert want["n_cigar"] > 20, (i, "an eligible request was not served", len(qs[i]), len(ts[i]), ws[i], fs[i], want)
            continue
        served += 1
        if any(int(g[f]) != want[f] for f in FIELDS) or [int(x) for x in g["cigar"][:want["n_cigar"]]] != want["cigar"]:
            bad.append((i, {f: (int(g[f]), want[f]) for f in FIELDS if int(g[f]) != want[f]}, len(qs[i]), len(ts[i]), ws[i], zs[i], es[i], hex(fs[i])))
    assert not bad, bad[:5]
    return served


def test_random_requests_equal_ksw_extd2_sse(oracle, aligner):
    rng = np.random.default_rng(2024)
    qs, ts, ws, zs, es, fs = _requests(rng, 6000)
    assert _compare(oracle, aligner, qs, ts, ws, zs, es, fs) > 5500


def test_edges(oracle, aligner):
    """one-base sides, full 128 x 128, all-N, identical sequences, a query longer than the target, ties on every diagonal"""
    rng = np.random.default_rng(7)
    full = [int(x) for x in rng.integers(0, 4, 128)]
    cases = [([0], [0]), ([1], [2]), ([0], full), (full, [3]), (full, full), (full, full[::-1]), ([4] * 40, [4] * 50), ([0] * 100, [0] * 128),
             ([0, 1] * 60, [1, 0] * 64), (full[:100], full[28:]), (full[5:], full[:100])]
    qs, ts, ws, zs, es, fs = [], [], [], [], [], []
    for q, t in cases:
        for kind in KINDS:
            for z, eb in ((-1, -1), (30, 0), (100, 10), (400, 50)):
                qs.append(q); ts.append(t); ws.append(-1); zs.append(z); es.append(eb); fs.append(kind)
    assert _compare(oracle, aligner, qs, ts, ws, zs, es, fs) >= len(qs) - 12


def test_requests_outside_the_service_are_left_alone(aligner):
    """a side beyond 128 bases, a band that cuts the matrix, another flag combination: not served (the align tiers run those on the
    wave-per-request kernels), and nothing is written for them"""
    q, t = [0, 1, 2, 3] * 20, [0, 1, 2, 3] * 20
    got, _ = aligner.dp_batch([q, q + q, q, q, []], [t, t, t + t + t, t, t], [-1, -1, -1, 10, -1], [100] * 5, [0] * 5, [EXTZ_ONLY, EXTZ_ONLY, EXTZ_ONLY, EXTZ_ONLY, EXTZ_ONLY])
    assert [int(x) for x in got["served"]] == [1, 0, 0, 0, 0]
    got, _ = aligner.dp_batch([q], [t], [-1], [100], [0], [0])
    assert int(got["served"][0]) == 0
