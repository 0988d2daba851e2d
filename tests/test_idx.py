"""`.idx` container (src/index_single_mode.cpp:1561-1640 writer, src/placement.cpp:1009-1092 +
src/zstd_compression.cpp:100-260 reader, schema src/index_lite.capnp:36-70): round trips of the host index through both
payload forms, the byte layout of the header and of the Cap'n Proto message checked with an independent reader written
here from the encoding spec, and the loader's validation errors."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def rsv_index(pmx):
    pm = pmx.Panman(os.path.join(GOLDEN, "rsv_4K.panman"))
    return pm, pmx.Index.build(pm, k=19, s=8, t=0, l=3, open_syncmer=False, flank_mask=250)


def _same(a, b):
    aa, bb = a.arrays(), b.arrays()
    return all(np.array_equal(aa[k], bb[k]) for k in ("parent", "offsets", "hash", "parent_count", "child_count"))


@pytest.mark.parametrize("uncompressed", [False, True])
def test_round_trip(pmx, rsv_index, tmp_path, uncompressed):
    pm, ix = rsv_index
    path = str(tmp_path / "rsv.idx")
    ix.save(path, zstd_level=3, uncompressed=uncompressed)
    assert pmx.Index.read_header(path) == dict(k=19, s=8, t=0, l=3, open=False, hpc=False, uncompressed=uncompressed)
    back = pmx.Index.load(path)
    assert _same(ix, back)
    assert (back.info.k, back.info.s, back.info.t, back.info.l, back.info.open_syncmer) == (19, 8, 0, 3, 0)
    n = back.info.n_nodes
    assert [back.node_id(i) for i in (0, 1, n - 1)] == [pm.node_id(i) for i in (0, 1, n - 1)]


def _capnp_root(msg):
    """independent mini reader (single segment): returns a getter for the root struct's data bytes and pointers"""
    nseg, w0 = struct.unpack_from("<II", msg, 0)
    assert nseg == 0
    seg = msg[8:8 + 8 * w0]
    root_ptr, = struct.unpack_from("<Q", seg, 0)
    assert root_ptr & 3 == 0
    off = ((root_ptr & 0xffffffff) >> 2)
    dwords, pwords = (root_ptr >> 32) & 0xffff, root_ptr >> 48
    data_at = 8 * (1 + off)
    return seg, data_at, dwords, pwords


def _list(seg, ptr_at):
    p, = struct.unpack_from("<Q", seg, ptr_at)
    assert p & 3 == 1
    off = (p & 0xffffffff) >> 2
    if off >= 1 << 29:
        off -= 1 << 30
    return ptr_at + 8 + 8 * off, (p >> 32) & 7, p >> 35


def test_wire_layout_follows_the_schema(pmx, rsv_index, tmp_path):
    """header bytes as encodeIndexHeader writes them; LiteIndex laid out as capnp lays out the schema: k s t l at bytes
    0/2/4/6, open = bit 64, hpc = bit 65, formatVersion (ordinal 17) in the 16-bit hole at byte 10, 11 pointers"""
    _, ix = rsv_index
    path = str(tmp_path / "raw.idx")
    ix.save(path, uncompressed=True)
    raw = open(path, "rb").read()
    assert raw[:4] == b"PMI1" and struct.unpack_from("<IIIII", raw, 4) == (1, 19, 8, 0, 3)
    assert raw[24:27] == bytes([0, 0, 1]) and raw[27:32] == bytes(5)
    seg, at, dwords, pwords = _capnp_root(raw[32:])
    assert (dwords, pwords) == (2, 11)
    assert struct.unpack_from("<HHHH", seg, at) == (19, 8, 0, 3)
    assert seg[at + 8] & 3 == 0 and struct.unpack_from("<H", seg, at + 10)[0] == 4
    arr = ix.arrays()
    ptrs = at + 16
    # nodeChangeOffsets @9 = pointer 4: List(UInt64) of n + 1
    loc, esize, cnt = _list(seg, ptrs + 8 * 4)
    assert esize == 5 and cnt == len(arr["offsets"])
    assert np.array_equal(np.frombuffer(seg, "<u8", cnt, loc), arr["offsets"])
    # seedChangeChildCounts @8 = pointer 3: List(List(Int16)), one segment here
    loc, esize, cnt = _list(seg, ptrs + 8 * 3)
    assert esize == 6 and cnt == 1
    iloc, iesize, icnt = _list(seg, loc)
    assert iesize == 3 and icnt == len(arr["child_count"])
    assert np.array_equal(np.frombuffer(seg, "<i2", icnt, iloc), arr["child_count"])
    # liteTree @5 = pointer 0 -> struct {liteNodes, blockRanges}; liteNodes: composite of (1 data word, 1 pointer)
    p, = struct.unpack_from("<Q", seg, ptrs)
    assert p & 3 == 0 and (p >> 32) & 0xffff == 0 and p >> 48 == 2
    lt = ptrs + 8 + 8 * ((p & 0xffffffff) >> 2)
    loc, esize, words = _list(seg, lt)
    tag, = struct.unpack_from("<Q", seg, loc)
    assert esize == 7 and (tag & 0xffffffff) >> 2 == len(arr["parent"]) and (tag >> 32) & 0xffff == 1 and tag >> 48 == 1
    assert words == 2 * len(arr["parent"])
    par = np.frombuffer(seg, "<u4", 4 * len(arr["parent"]), loc + 8)[0::4]
    assert np.array_equal(par, arr["parent"])


def test_compressed_payload_is_independent_zstd_frames_with_checksums(pmx, rsv_index, tmp_path):
    _, ix = rsv_index
    path = str(tmp_path / "z.idx")
    ix.save(path, zstd_level=1, uncompressed=False)
    raw = open(path, "rb").read()
    assert raw[26] == 0 and raw[32:36] == bytes([0x28, 0xb5, 0x2f, 0xfd])      # zstd frame magic
    fhd = raw[36]
    assert fhd & 0x04                                                          # Content_Checksum_flag (ZSTD_c_checksumFlag = 1)
    assert (fhd >> 6) != 0 or (fhd & 0x20)                                     # Frame_Content_Size is recorded (the reader needs it)


def test_loader_validation_errors(pmx, rsv_index, tmp_path):
    """src/placement.cpp:1013-1047: stale format version, missing struct-of-arrays fields; plus truncation and garbage"""
    _, ix = rsv_index
    path = str(tmp_path / "raw.idx")
    ix.save(path, uncompressed=True)
    raw = bytearray(open(path, "rb").read())
    seg, at, _, _ = _capnp_root(bytes(raw[32:]))

    def load_mutated(mut, name):
        p = str(tmp_path / name)
        open(p, "wb").write(bytes(mut))
        with pytest.raises(pmx.PmxError) as e:
            pmx.Index.load(p)
        return str(e.value)

    stale = bytearray(raw)
    struct.pack_into("<H", stale, 32 + 8 + at + 10, 3)
    assert "Index format version 3 is incompatible with this panmap (expects 4)" in load_mutated(stale, "stale.idx")
    nosoa = bytearray(raw)
    struct.pack_into("<Q", nosoa, 32 + 8 + at + 16 + 8 * 1, 0)                 # seedChangeHashes pointer nulled
    assert "Index missing required V3 fields" in load_mutated(nosoa, "nosoa.idx")
    assert "not a panmap index" in load_mutated(b"not an index at all, just forty bytes of text....", "junk.idx")
    assert load_mutated(raw[:len(raw) // 2], "cut.idx")                        # truncated: a format error, not a crash
    wild = bytearray(raw)
    struct.pack_into("<Q", wild, 32 + 8 + at + 16 + 8 * 4, (0x1fffffff << 2) | 1 | (5 << 32) | (1000 << 35))   # offsets list far outside
    assert "outside its segment" in load_mutated(wild, "wild.idx")
    # a list pointer that declares narrower elements than the field's type passes the bounds check of ITS size and
    # would then be read 8x (2x) past it -- the element-size code is checked against the type that is read
    ptrs = 32 + 8 + at + 16
    narrow = bytearray(raw)
    w, = struct.unpack_from("<Q", narrow, ptrs + 8 * 4)                        # nodeChangeOffsets: List(UInt64) -> declared as bytes
    struct.pack_into("<Q", narrow, ptrs + 8 * 4, (w & ~(7 << 32)) | (2 << 32))
    assert "element width" in load_mutated(narrow, "narrow_offsets.idx")
    for fld, code in ((1, 2), (2, 2), (3, 1)):                                 # inner lists of hashes / counts declared as bytes / bits
        m = bytearray(raw)
        loc, esize, cnt = _list(bytes(m[32 + 8:]), at + 16 + 8 * fld)
        assert esize == 6 and cnt == 1
        w, = struct.unpack_from("<Q", m, 32 + 8 + loc)
        struct.pack_into("<Q", m, 32 + 8 + loc, (w & ~(7 << 32)) | (code << 32))
        assert "element width" in load_mutated(m, "narrow_%d.idx" % fld)
    assert pmx.Index.read_header(str(tmp_path / "junk.idx")) is None


def test_loaded_index_places_like_the_built_one(pmx, tmp_path):
    """the example sample's SARS index through the file: same arrays, so the place stage sees the same index"""
    pm = pmx.Panman(os.path.join(GOLDEN, "sars_20000_twilight_dipper.panman"))
    ix = pmx.Index.build(pm)
    path = str(tmp_path / "sars.idx")
    ix.save(path, zstd_level=1)
    assert os.path.getsize(path) < 0.6 * (12 * ix.info.n_changes)            # compressed
    back = pmx.Index.load(path)
    assert _same(ix, back) and back.info.n_changes == 2422076
