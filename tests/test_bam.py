"""BAM / BAI egress (panmap_amd/csrc/host/bam_writer.cpp) against an independent restatement of the reference's
record rules (src/conversion.cpp:257-388, 426-538) and of the container formats (SAM spec v1.6 sections 4, 5.2),
parsed back with nothing but zlib.  Alignment results come from the reference aligner itself (oracle/_ref), so
the test needs no GPU."""
import os
import struct
import zlib

import pytest

import align_checks as ac
from conftest import GOLDEN

NT16 = "=ACMGRSVTWYHKDBN"


def bgzf_blocks(data):
    """-> list of (file offset, uncompressed bytes)"""
    out, off = [], 0
    while off < len(data):
        assert data[off:off + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", data, off + 10)[0]
        assert data[off + 12:off + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", data, off + 16)[0] + 1
        raw = zlib.decompress(data[off + 12 + xlen:off + bsize - 8], -15)
        crc, isize = struct.unpack_from("<II", data, off + bsize - 8)
        assert isize == len(raw) and crc == zlib.crc32(raw)
        out.append((off, raw))
        off += bsize
    return out


def parse_bam(path):
    data = open(path, "rb").read()
    blocks = bgzf_blocks(data)
    assert blocks[-1][1] == b"", "BGZF EOF marker missing"
    stream = b"".join(b for _, b in blocks)
    # virtual offset of every stream position's block start
    starts, pos = {}, 0
    for off, b in blocks:
        starts[pos] = off
        pos += len(b)
    assert stream[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", stream, 4)[0]
    text = stream[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", stream, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", stream, p)[0]
        name = stream[p + 4:p + 4 + l_name - 1].decode()
        l_ref = struct.unpack_from("<i", stream, p + 4 + l_name)[0]
        refs.append((name, l_ref))
        p += 8 + l_name
    recs = []
    block_pos = sorted(starts)
    while p < len(stream):
        bs = struct.unpack_from("<i", stream, p)[0]
        ref_id, pos_, l_rn, mapq, bin_, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", stream, p + 4)
        q = p + 36
        name = stream[q:q + l_rn - 1].decode()
        q += l_rn
        cigar = list(struct.unpack_from("<%dI" % n_cig, stream, q))
        q += 4 * n_cig
        packed = stream[q:q + (l_seq + 1) // 2]
        seq = "".join(NT16[b >> 4] + NT16[b & 15] for b in packed)[:l_seq]
        q += (l_seq + 1) // 2
        qual = stream[q:q + l_seq]
        q += l_seq
        assert q == p + 4 + bs
        import bisect
        bstart = block_pos[bisect.bisect_right(block_pos, p) - 1]
        recs.append(dict(name=name, ref_id=ref_id, pos=pos_, mapq=mapq, bin=bin_, flag=flag, cigar=cigar, seq=seq, qual=qual, mtid=mtid,
                         mpos=mpos, tlen=tlen, voff=starts[bstart] << 16 | (p - bstart), vend_stream=q))
        p = q
    return text, refs, recs


def reg2bin(beg, end):
    end -= 1
    for sh, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return base + (beg >> sh)
    return 0


def expected_records(seqs, quals, names, results, paired):
    """src/conversion.cpp:288-388 + :426-499 restated"""
    comp = {"A": "T", "a": "T", "T": "A", "t": "A", "C": "G", "c": "G", "G": "C", "g": "C"}
    out = []

    def one(i, aln, is_read1, mate):
        L = len(seqs[i])
        name = names[i].decode()
        if len(name) >= 2 and name[-2] == "/" and name[-1] in "12":
            name = name[:-2]
        eff_rev = (0 if aln["rev"] else 1) if (paired and not is_read1) else aln["rev"]
        flag = 0
        if paired:
            mate_rev = (0 if mate["rev"] else 1) if is_read1 else mate["rev"]
            flag = 1 | (2 if aln["proper_frag"] else 0) | (16 if eff_rev else 0) | (32 if mate_rev else 0) | (64 if is_read1 else 128)
        elif aln["rev"]:
            flag = 16
        clip5 = L - aln["qe"] if aln["rev"] else aln["qs"]
        clip3 = aln["qs"] if aln["rev"] else L - aln["qe"]
        cig = ([clip5 << 4 | 4] if clip5 else []) + list(aln["cigar"]) + ([clip3 << 4 | 4] if clip3 else [])
        s = seqs[i].decode()
        ql = quals[i]
        if aln["rev"]:
            s = "".join(comp.get(c, "N") for c in reversed(s))
            ql = ql[::-1]
        else:
            s = "".join(c.upper() if c.upper() in NT16 else "N" for c in s)
        tlen, mtid, mpos = 0, -1, -1
        if paired:
            t5 = aln["re"] - 1 if eff_rev else aln["rs"]
            m5 = mate["re"] - 1 if mate_rev else mate["rs"]
            tlen = m5 - t5
            tlen += 1 if tlen > 0 else (-1 if tlen < 0 else 0)
            mtid, mpos = 0, mate["rs"]
        rlen = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8)) or 1
        return dict(sort=aln["pos"], name=name, pos=aln["rs"], mapq=aln["mapq"], bin=reg2bin(aln["rs"], aln["rs"] + rlen), flag=flag, cigar=cig,
                    seq=s, qual=bytes(b - 33 for b in ql), mtid=mtid, mpos=mpos, tlen=tlen, end=aln["rs"] + rlen)
    for k, res in enumerate(results):
        if not res["mapped"]:
            continue
        if paired:
            out.append(one(2 * k, res["r1"], True, res["r2"]))
            out.append(one(2 * k + 1, res["r2"], False, res["r1"]))
        else:
            out.append(one(k, res["r1"], False, None))
    return out


@pytest.mark.parametrize("paired", [True, False])
def test_bam_and_bai(pmx, oracle, tmp_path, paired):
    g = b"".join(l.strip() for l in open(os.path.join(GOLDEN, "isolate.ref.fa"), "rb") if not l.startswith(b">"))
    seqs, quals, names = pmx.read_fastq_paired(os.path.join(GOLDEN, "isolate_R1.fastq.gz"), os.path.join(GOLDEN, "isolate_R2.fastq.gz"))
    seqs, quals, names = seqs[:3000], quals[:3000], names[:3000]
    names[0] = names[0] + b"/1"
    results = oracle.ref_align_reads_direct(g, seqs, paired, 4)
    path = str(tmp_path / "out.bam")
    pmx.write_bam(path, "node_7618", len(g), seqs, quals, names, results, paired)
    text, refs, recs = parse_bam(path)
    assert text == "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:node_7618\tLN:%d\n" % len(g)
    assert refs == [("node_7618", len(g))]
    want = expected_records(seqs, quals, names, results, paired)
    assert len(recs) == len(want) > 1000
    # sorted by the 1-based pos key; records with equal keys may come in either order
    assert [r["pos"] for r in recs] == sorted(r["pos"] for r in recs)
    key = lambda r: (r["pos"], r["name"], r["flag"], r["seq"], r["tlen"], r["qual"])
    for got, exp in zip(sorted(recs, key=key), sorted(want, key=key)):
        for f in ("name", "pos", "mapq", "bin", "flag", "cigar", "seq", "qual", "mtid", "mpos", "tlen"):
            assert got[f] == exp[f], (f, got["name"], got[f], exp[f])
        assert got["ref_id"] == 0

    # ---- BAI: every record must be reachable through its bin's chunks and the linear index
    bai = open(path + ".bai", "rb").read()
    assert bai[:4] == b"BAI\x01" and struct.unpack_from("<i", bai, 4)[0] == 1
    p = 8
    n_bin = struct.unpack_from("<i", bai, p)[0]
    p += 4
    bins = {}
    for _ in range(n_bin):
        b, n_chunk = struct.unpack_from("<Ii", bai, p)
        p += 8
        bins[b] = [struct.unpack_from("<QQ", bai, p + 16 * i) for i in range(n_chunk)]
        p += 16 * n_chunk
    n_intv = struct.unpack_from("<i", bai, p)[0]
    p += 4
    linear = list(struct.unpack_from("<%dQ" % n_intv, bai, p))
    p += 8 * n_intv
    assert struct.unpack_from("<Q", bai, p)[0] == 0 and p + 8 == len(bai)
    meta = bins.pop(37450)
    assert meta[1] == (len(recs), 0) and meta[0][0] == recs[0]["voff"]
    for r, exp in zip(sorted(recs, key=key), sorted(want, key=key)):
        chunks = bins[r["bin"]]
        assert any(c[0] <= r["voff"] < c[1] for c in chunks), r["name"]
        for w in range(exp["pos"] >> 14, ((exp["end"] - 1) >> 14) + 1):
            assert linear[w] <= r["voff"]
    assert n_intv == (max(e["end"] - 1 for e in want) >> 14) + 1
