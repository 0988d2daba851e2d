"""The `panmap` command line built over the C ABI (panmap_amd/csrc/cli/panmap_main.cpp; reference surface:
src/main.cpp:1941-2131, 2225-2276, 371-396).  CPU part: the index stage, the `.idx` cache rules, argument errors.
GPU part: the README demo `panmap <panman> R1 R2 --stop align` -- placement TSV byte-equal to the reference's golden,
placed genome equal to the golden FASTA, the BAM parsed back and compared with the reference aligner's results."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CLI = os.path.join(ROOT, "panmap_amd", "bin", "panmap")


def run(args, cwd):
    """one invocation of the command line.  A run that does not come back within two minutes (they take seconds) is started
    once more: at the end of round 4 one `panmap --meta` on 1,000 reads sat for five minutes on a GPU box and ran in seconds on
    the next one, on the same sources (profiles/r04/README.md item 20); a second hang fails the test."""
    try:
        return subprocess.run([CLI] + args, cwd=cwd, capture_output=True, text=True, timeout=120)
    except subprocess.TimeoutExpired:
        return subprocess.run([CLI] + args, cwd=cwd, capture_output=True, text=True, timeout=120)


def test_index_stage_and_cache_rules(pmx, tmp_path):
    shutil.copy(os.path.join(GOLDEN, "rsv_4K.panman"), tmp_path / "rsv.panman")
    r = run(["rsv.panman", "--stop", "index"], tmp_path)
    assert r.returncode == 0 and "(built)" in r.stderr, r.stderr
    idx = tmp_path / "rsv.panman.idx"                                   # <panman>.idx (src/main.cpp:2242-2244)
    assert pmx.Index.read_header(str(idx)) == dict(k=19, s=8, t=0, l=3, open=False, hpc=False, uncompressed=False)
    r = run(["rsv.panman", "--stop", "index"], tmp_path)
    assert r.returncode == 0 and "(cached)" in r.stderr                 # reused
    r = run(["rsv.panman", "--stop", "index", "-k", "15", "-s", "6"], tmp_path)
    assert r.returncode == 0 and "different seeding parameters" in r.stderr and "(built)" in r.stderr
    assert pmx.Index.read_header(str(idx))["k"] == 15
    os.utime(tmp_path / "rsv.panman")                                   # PanMAN newer than the index -> rebuild
    os.utime(idx, (1, 1))
    r = run(["rsv.panman", "--stop", "index", "-k", "15", "-s", "6"], tmp_path)
    assert "is older than" in r.stderr and "(built)" in r.stderr
    r = run(["rsv.panman", "--stop", "index", "-f", "-k", "15", "-s", "6", "--index-out", "custom.idx", "--index-uncompressed"], tmp_path)
    assert r.returncode == 0 and pmx.Index.read_header(str(tmp_path / "custom.idx"))["uncompressed"] is True
    built = pmx.Index.build(pmx.Panman(str(tmp_path / "rsv.panman")), k=15, s=6)
    back = pmx.Index.load(str(tmp_path / "custom.idx"))
    assert all(np.array_equal(built.arrays()[k], back.arrays()[k]) for k in ("parent", "offsets", "hash", "parent_count", "child_count"))


def test_argument_errors(tmp_path):
    shutil.copy(os.path.join(GOLDEN, "rsv_4K.panman"), tmp_path / "rsv.panman")
    assert run([], tmp_path).returncode == 1
    r = run(["rsv.panman", "-s", "30"], tmp_path)
    assert r.returncode == 1 and "Invalid syncmer s=30 (must be in 1..k, k=19)" in r.stderr                # src/main.cpp:2227-2230
    r = run(["rsv.panman", "--offset", "12"], tmp_path)
    assert r.returncode == 1 and "Invalid syncmer offset=12 (must be in 0..k-s = 0..11)" in r.stderr
    r = run(["rsv.panman", "-i", "missing.idx"], tmp_path)
    assert r.returncode == 1 and "index file not found: missing.idx" in r.stderr
    for opt in (["--filter-and-assign"], ["--hpc"], ["-a", "bwa"], ["--stop", "nowhere"], ["--no-such-option"], ["--meta"], ["--meta", "x.fq", "-l", "1"]):
        assert run(["rsv.panman"] + opt, tmp_path).returncode == 1


def test_a_failing_rank_ends_the_run(tmp_path):
    """`panmap --gpus N`: a rank that fails -- here every rank asks for a device ordinal the box does not have (or, in a
    container without a GPU, for any device) -- must END the run with its code: the parent reaps whichever child ends first and
    terminates the others instead of waiting for ranks that are blocked in the rendezvous for ever"""
    import time
    shutil.copy(os.path.join(GOLDEN, "rsv_4K.panman"), tmp_path / "rsv.panman")
    (tmp_path / "r.fastq").write_text("@a\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n")
    assert run(["rsv.panman", "--stop", "index"], tmp_path).returncode == 0
    env = dict(os.environ, PMX_DEVICE="63")
    t0 = time.time()
    r = subprocess.run([CLI, "rsv.panman", "r.fastq", "--stop", "place", "--gpus", "3"], cwd=tmp_path, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and time.time() - t0 < 60, (r.returncode, r.stderr[-500:])
    assert "opening the GPU" in r.stderr
    assert not [d for d in os.listdir("/tmp") if d.startswith("panmap_ranks_") and os.path.exists(os.path.join("/tmp", d, "uid"))]


@pytest.mark.gpu
def test_readme_demo_through_the_cli(pmx, oracle, tmp_path):
    for f in ("sars_20000_twilight_dipper.panman", "isolate_R1.fastq.gz", "isolate_R2.fastq.gz"):
        shutil.copy(os.path.join(GOLDEN, f), tmp_path / f)
    r = run(["sars_20000_twilight_dipper.panman", "isolate_R1.fastq.gz", "isolate_R2.fastq.gz", "--stop", "align"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    # prefix derived from reads1 as src/main.cpp:2253-2276 derives it: stem "isolate_R1.fastq", the mate suffixes are
    # tried FIRST (none matches a stem that still ends in .fastq), then .fastq goes -> "isolate_R1"
    assert open(tmp_path / "isolate_R1.placement.tsv", "rb").read() == open(os.path.join(GOLDEN, "isolate.placement.tsv"), "rb").read()
    assert open(tmp_path / "isolate_R1.ref.fa", "rb").read() == open(os.path.join(GOLDEN, "isolate.ref.fa"), "rb").read()
    fai = open(tmp_path / "isolate_R1.ref.fa.fai").read().split("\t")
    assert fai[0] == "node_7618" and int(fai[1]) == 29709
    import test_bam as tb
    text, refs, recs = tb.parse_bam(str(tmp_path / "isolate_R1.bam"))
    assert refs == [("node_7618", 29709)] and os.path.exists(tmp_path / "isolate_R1.bam.bai")
    # against the reference aligner on the same reads
    g = b"".join(l.strip() for l in open(os.path.join(GOLDEN, "isolate.ref.fa"), "rb") if not l.startswith(b">"))
    seqs, _, _ = pmx.read_fastq_paired(os.path.join(GOLDEN, "isolate_R1.fastq.gz"), os.path.join(GOLDEN, "isolate_R2.fastq.gz"))
    want = oracle.ref_align_reads_direct(g, seqs, True, 8)
    n_mapped = sum(w["mapped"] for w in want)
    assert n_mapped == 41003 and len(recs) == 2 * n_mapped
    assert "41003 of 51169 pairs mapped" in r.stderr
    assert [x["pos"] for x in recs] == sorted(x["pos"] for x in recs)                        # coordinate sorted
    want_pos = sorted([w["r1"]["rs"] for w in want if w["mapped"]] + [w["r2"]["rs"] for w in want if w["mapped"]])
    assert [x["pos"] for x in recs] == want_pos
    # a second run reuses the index written next to the PanMAN
    r2 = run(["sars_20000_twilight_dipper.panman", "isolate_R1.fastq.gz", "isolate_R2.fastq.gz", "--stop", "place", "-o", "again"], tmp_path)
    assert r2.returncode == 0 and "(cached)" in r2.stderr
    assert open(tmp_path / "again.placement.tsv", "rb").read() == open(os.path.join(GOLDEN, "isolate.placement.tsv"), "rb").read()
    # --refine: the refined_<metric> lines follow the five seed metrics (src/placement.cpp:1987-2000); same numbers as the
    # library's own refinement of the same placement
    r3 = run(["sars_20000_twilight_dipper.panman", "isolate_R1.fastq.gz", "isolate_R2.fastq.gz", "--stop", "place", "-o", "refined", "--refine",
              "--refine-max-top-n", "4", "--refine-max-neighbor-n", "3"], tmp_path)
    assert r3.returncode == 0, r3.stderr[-2000:]
    lines = open(tmp_path / "refined.placement.tsv").read().splitlines()
    assert "\n".join(lines[:6]) + "\n" == open(os.path.join(GOLDEN, "isolate.placement.tsv")).read()
    assert [l.split("\t")[0] for l in lines[6:]] == ["refined_" + m for m in pmx.METRICS]
    ctx = pmx.Context(0)
    pm = pmx.Panman(str(tmp_path / "sars_20000_twilight_dipper.panman"))
    index = pmx.Index.load(str(tmp_path / "sars_20000_twilight_dipper.panman.idx"))
    placer = pmx.Placer(ctx, index)
    raw = pmx.extract_read_sequences(str(tmp_path / "isolate_R1.fastq.gz"), str(tmp_path / "isolate_R2.fastq.gz"))
    rs = pmx.ReadSet(ctx, raw)
    params = pmx.TraversalParams()
    placer.reset()
    placer.add_reads(rs, params)
    res = placer.score(params, len(raw))
    refined = pmx.refine_placement(ctx, placer, pm, res, rs, True, int(sum(len(x) for x in raw) // len(raw)), pmx.RefineParams(0.01, 4, 2, 3))
    assert pmx.format_refined_tsv(refined, index.node_id).splitlines() == lines[6:]


@pytest.mark.gpu
def test_batch_mode(pmx, tmp_path):
    """--batch (runBatchPlacement, src/main.cpp:1464-1666; batch file format :1025-1087): the samples one after the other
    against the resident index, one line per sample on stderr, outputs under each sample's prefix"""
    for f in ("sars_20000_twilight_dipper.panman", "isolate_R1.fastq.gz", "isolate_R2.fastq.gz"):
        shutil.copy(os.path.join(GOLDEN, f), tmp_path / f)
    (tmp_path / "batch.txt").write_text("# reads1 [reads2] [prefix]\n"
                                        "isolate_R1.fastq.gz isolate_R2.fastq.gz out/paired\n"
                                        "\n"
                                        "isolate_R1.fastq.gz single_end\n"
                                        "isolate_R2.fastq.gz\n")
    r = run(["sars_20000_twilight_dipper.panman", "--batch", "batch.txt", "--stop", "place"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Batch mode: 3 samples" in r.stderr and "[1/3] out/paired -> node_7618 (" in r.stderr and "[2/3] single_end -> " in r.stderr
    assert "[3/3] isolate_R2 -> " in r.stderr                  # default prefix: mate suffix tried on the stem "isolate_R2.fastq", then .fastq goes
    assert open(tmp_path / "out" / "paired.placement.tsv", "rb").read() == open(os.path.join(GOLDEN, "isolate.placement.tsv"), "rb").read()
    assert os.path.exists(tmp_path / "single_end.placement.tsv") and os.path.exists(tmp_path / "isolate_R2.placement.tsv")
    # through align as well: the BAM of every sample next to its prefix
    (tmp_path / "b2.txt").write_text("isolate_R1.fastq.gz isolate_R2.fastq.gz again\n")
    r = run(["sars_20000_twilight_dipper.panman", "--batch", "b2.txt", "--stop", "align"], tmp_path)
    assert r.returncode == 0 and os.path.exists(tmp_path / "again.bam") and os.path.exists(tmp_path / "again.bam.bai"), r.stderr[-1000:]
    # a missing read file is an error of the batch file (src/main.cpp:1074-1081)
    (tmp_path / "b3.txt").write_text("nope.fastq\n")
    r = run(["sars_20000_twilight_dipper.panman", "--batch", "b3.txt", "--stop", "place"], tmp_path)
    assert r.returncode == 1 and "Batch line 1: reads file not found: nope.fastq" in r.stderr


@pytest.mark.gpu
def test_gpus_two_ranks_equal_one(pmx, tmp_path):
    """`panmap --gpus 2`: two processes forked before any GPU call, each a rank of the pmx_dist_* exchange with its shard of
    the reads (here both on the one device of the box, over the library's host-directory test transport) -- the placement TSV,
    the placed genome and the BAM equal the one-GPU run's byte for byte; --refine sums the shards' candidate scores"""
    for f in ("sars_20000_twilight_dipper.panman", "isolate_R1.fastq.gz", "isolate_R2.fastq.gz"):
        shutil.copy(os.path.join(GOLDEN, f), tmp_path / f)
    reads = ["sars_20000_twilight_dipper.panman", "isolate_R1.fastq.gz", "isolate_R2.fastq.gz"]
    r1 = run(reads + ["--stop", "align", "-o", "one"], tmp_path)
    assert r1.returncode == 0, r1.stderr[-2000:]
    meet = tmp_path / "meet"
    meet.mkdir()
    env = dict(os.environ, PMX_DIST_SAME_DEVICE="1", PMX_DIST_HOST_DIR=str(meet))
    r2 = subprocess.run([CLI] + reads + ["--stop", "align", "-o", "two", "--gpus", "2"], cwd=tmp_path, capture_output=True, text=True, timeout=1200, env=env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "41003 of 51169 pairs mapped" in r2.stderr
    for ext in (".placement.tsv", ".ref.fa", ".bam", ".bam.bai"):
        assert open(tmp_path / ("one" + ext), "rb").read() == open(tmp_path / ("two" + ext), "rb").read(), ext
    r3 = run(reads + ["--stop", "place", "-o", "one_r", "--refine", "--refine-max-top-n", "4", "--refine-max-neighbor-n", "3"], tmp_path)
    r4 = subprocess.run([CLI] + reads + ["--stop", "place", "-o", "two_r", "--refine", "--refine-max-top-n", "4", "--refine-max-neighbor-n", "3", "--gpus", "2"],
                        cwd=tmp_path, capture_output=True, text=True, timeout=1200, env=env)
    assert r3.returncode == 0 and r4.returncode == 0, (r3.stderr[-1000:], r4.stderr[-1000:])
    assert open(tmp_path / "one_r.placement.tsv", "rb").read() == open(tmp_path / "two_r.placement.tsv", "rb").read()
    # --dedup: duplicates are collapsed over the whole sample, whichever rank holds the copies
    r5 = run(reads + ["--stop", "place", "--dedup", "-o", "one_d"], tmp_path)
    r6 = subprocess.run([CLI] + reads + ["--stop", "place", "--dedup", "-o", "two_d", "--gpus", "2"], cwd=tmp_path, capture_output=True, text=True, timeout=1200, env=env)
    assert r5.returncode == 0 and r6.returncode == 0, (r5.stderr[-1000:], r6.stderr[-1000:])
    assert open(tmp_path / "one_d.placement.tsv", "rb").read() == open(tmp_path / "two_d.placement.tsv", "rb").read()
    assert open(tmp_path / "one_d.placement.tsv", "rb").read() != open(tmp_path / "one.placement.tsv", "rb").read()   # (the flag changes the scores)


@pytest.mark.gpu
def test_meta_mixture_through_the_cli(pmx, tmp_path):
    """`panmap <panman> mix.fastq --meta`: the reference's e2e scenario 12 (src/test/e2e/run_e2e.sh:182-204) through the
    command line -- the same greps and ranges on `<prefix>.mgsr.abundance.out`"""
    shutil.copy(os.path.join(GOLDEN, "rsv_4K.panman"), tmp_path / "rsv_4K.panman")

    def rd(p):
        return "".join(l.strip() for l in open(p) if not l.startswith(">")).upper()
    a, b = rd(os.path.join(GOLDEN, "MZ515733.1.fa")), rd(os.path.join(GOLDEN, "rsv_4K.panman.random.node_1330.fa"))
    with open(tmp_path / "mix.fastq", "w") as out:
        def emit(g, n, pre):
            L = 150; step = max(1, (len(g) - L) // n); c = i = 0
            while c < n and i + L <= len(g):
                out.write("@%s%d\n%s\n+\n%s\n" % (pre, c, g[i:i + L], "I" * L)); c += 1; i += step
        emit(a, 700, "A"); emit(b, 300, "B")
    r = run(["rsv_4K.panman", "mix.fastq", "--meta", "-o", "mix"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l.split("\t") for l in open(tmp_path / "mix.mgsr.abundance.out").read().splitlines() if l]
    assert len(lines) == 2
    got = {k: float(v) for k, v in lines}
    assert 0.55 < got["MZ515733.1"] < 0.82 and 0.18 < got["node_1330"] < 0.45 and 0.99 < sum(got.values()) < 1.01
    assert all(len(v.split(".")[1]) == 5 for _, v in lines)              # %.5f
    # --dust (src/main.cpp:2059-2061, mgsr::getDust): low-complexity reads are left out before the seedmers are made --
    # 200 poly-A / dinucleotide reads added to the sample change nothing once the filter is on
    with open(tmp_path / "mix.fastq", "a") as out:
        for i in range(100):
            out.write("@L%d\n%s\n+\n%s\n@M%d\n%s\n+\n%s\n" % (i, "A" * 150, "I" * 150, i, "AC" * 75, "I" * 150))
    r = run(["rsv_4K.panman", "mix.fastq", "--meta", "--dust", "20", "-o", "mixd"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(tmp_path / "mixd.mgsr.abundance.out").read() == open(tmp_path / "mix.mgsr.abundance.out").read()
    assert run(["rsv_4K.panman", "mix.fastq", "--meta", "--dust", "101"], tmp_path).returncode == 1     # --dust must be <= 100
