// See idx_file.hpp.
#include "idx_file.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <stdexcept>
#include <vector>

#include "capnp_lite.hpp"

namespace pmx {
namespace {

constexpr uint32_t kMagic = 0x31494D50u;   // "PMI1" little-endian
constexpr uint32_t kHeaderVersion = 1;
constexpr size_t kHeaderSize = 32;
constexpr size_t kFrameSize = (size_t)64 << 20;
constexpr uint64_t kSegmentElems = 500000000ULL;   // LiteTree::SEED_CHANGE_SEGMENT (src/panmap_utils.hpp:77)

// ---- zstd through dlopen: the handful of prototypes of zstd.h this file needs (stable ABI since 1.4)
struct Zstd {
    void* h = nullptr;
    size_t (*compressBound)(size_t) = nullptr;
    void* (*createCCtx)() = nullptr;
    size_t (*freeCCtx)(void*) = nullptr;
    size_t (*CCtx_setParameter)(void*, int, int) = nullptr;
    size_t (*compress2)(void*, void*, size_t, const void*, size_t) = nullptr;
    size_t (*decompress)(void*, size_t, const void*, size_t) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    const char* (*getErrorName)(size_t) = nullptr;
    size_t (*findFrameCompressedSize)(const void*, size_t) = nullptr;
    unsigned long long (*getFrameContentSize)(const void*, size_t) = nullptr;
};
constexpr int kZstdCompressionLevel = 100, kZstdChecksumFlag = 201;   // ZSTD_c_compressionLevel, ZSTD_c_checksumFlag
constexpr unsigned long long kContentSizeUnknown = 0ULL - 1, kContentSizeError = 0ULL - 2;

const Zstd& zstd() {
    static Zstd z;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"libzstd.so.1", "libzstd.so"}) {
            z.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (z.h) break;
        }
        if (!z.h) return;
        auto sym = [&](const char* n) { return dlsym(z.h, n); };
        z.compressBound = (size_t(*)(size_t))sym("ZSTD_compressBound");
        z.createCCtx = (void* (*)())sym("ZSTD_createCCtx");
        z.freeCCtx = (size_t(*)(void*))sym("ZSTD_freeCCtx");
        z.CCtx_setParameter = (size_t(*)(void*, int, int))sym("ZSTD_CCtx_setParameter");
        z.compress2 = (size_t(*)(void*, void*, size_t, const void*, size_t))sym("ZSTD_compress2");
        z.decompress = (size_t(*)(void*, size_t, const void*, size_t))sym("ZSTD_decompress");
        z.isError = (unsigned (*)(size_t))sym("ZSTD_isError");
        z.getErrorName = (const char* (*)(size_t))sym("ZSTD_getErrorName");
        z.findFrameCompressedSize = (size_t(*)(const void*, size_t))sym("ZSTD_findFrameCompressedSize");
        z.getFrameContentSize = (unsigned long long (*)(const void*, size_t))sym("ZSTD_getFrameContentSize");
        if (!z.compressBound || !z.createCCtx || !z.freeCCtx || !z.CCtx_setParameter || !z.compress2 || !z.decompress || !z.isError ||
            !z.getErrorName || !z.findFrameCompressedSize || !z.getFrameContentSize) {
            dlclose(z.h);
            z.h = nullptr;
        }
    });
    if (!z.h) throw std::runtime_error("libzstd.so.1 not available: compressed .idx files cannot be read or written (use the uncompressed form)");
    return z;
}

std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> buf((size_t)n);
    if (n > 0 && !f.read(reinterpret_cast<char*>(buf.data()), n)) throw std::runtime_error("cannot read " + path);
    return buf;
}

bool decode_header(const uint8_t* h, size_t n, IdxHeader& out) {
    if (n < kHeaderSize) return false;
    auto get32 = [&](size_t off) { uint32_t v; std::memcpy(&v, h + off, 4); return v; };
    if (get32(0) != kMagic || get32(4) != kHeaderVersion) return false;
    out.k = (int32_t)get32(8); out.s = (int32_t)get32(12); out.t = (int32_t)get32(16); out.l = (int32_t)get32(20);
    out.hpc = h[24] != 0; out.open = h[25] != 0; out.uncompressed = h[26] != 0;
    return true;
}

// ---- a single-segment Cap'n Proto builder: just enough of the encoding (https://capnproto.org/encoding.html) to lay out
// the LiteIndex message; every allocation is zero-filled, pointers are written once their target exists
struct Builder {
    std::vector<uint64_t> w;
    size_t alloc(size_t words) {
        const size_t at = w.size();
        w.resize(at + words, 0);
        return at;
    }
    static uint64_t off30(size_t ptr_at, size_t target) {
        const int64_t o = (int64_t)target - (int64_t)ptr_at - 1;
        if (o < -(1LL << 29) || o >= (1LL << 29)) throw std::runtime_error("index too large for the single-segment .idx writer");
        return (uint64_t)((uint32_t)(int32_t)o << 2);
    }
    void set_struct_ptr(size_t ptr_at, size_t target, unsigned dwords, unsigned pwords) {
        w[ptr_at] = (off30(ptr_at, target) & 0xfffffffcULL) | (uint64_t)dwords << 32 | (uint64_t)pwords << 48;
    }
    void set_list_ptr(size_t ptr_at, size_t target, unsigned esize, uint64_t count) {
        if (count >= (1ULL << 29)) throw std::runtime_error("list too long for a Cap'n Proto list pointer");
        w[ptr_at] = (off30(ptr_at, target) & 0xfffffffcULL) | 1ULL | (uint64_t)esize << 32 | count << 35;
    }
    // list of fixed-width primitives copied from memory
    void prim_list(size_t ptr_at, const void* src, size_t n, unsigned elem_bytes) {
        const unsigned code = elem_bytes == 1 ? 2 : elem_bytes == 2 ? 3 : elem_bytes == 4 ? 4 : 5;
        const size_t at = alloc((n * elem_bytes + 7) / 8);
        if (n) std::memcpy(reinterpret_cast<uint8_t*>(w.data() + at), src, n * elem_bytes);
        set_list_ptr(ptr_at, at, code, n);
    }
    void text(size_t ptr_at, const std::string& s) {
        const size_t at = alloc((s.size() + 1 + 7) / 8);
        std::memcpy(reinterpret_cast<uint8_t*>(w.data() + at), s.data(), s.size());
        set_list_ptr(ptr_at, at, 2, s.size() + 1);
    }
};

}  // namespace

bool read_idx_header(const std::string& path, IdxHeader& out) {
    std::ifstream f(path, std::ios::binary);
    uint8_t h[kHeaderSize];
    if (!f.read(reinterpret_cast<char*>(h), kHeaderSize)) return false;
    return decode_header(h, kHeaderSize, out);
}

void save_idx(const LiteIndex& ix, const std::string& path, int zstd_level, bool uncompressed) {
    const size_t n = ix.n_nodes();
    if (ix.offsets.size() != n + 1 || ix.node_id.size() != n) throw std::runtime_error("index arrays are inconsistent");
    const uint64_t total = ix.offsets[n];
    if (ix.hash.size() != total || ix.parent_count.size() != total || ix.child_count.size() != total)
        throw std::runtime_error("index arrays are inconsistent");
    Builder b;
    b.alloc(1);                                  // root pointer
    const size_t root = b.alloc(2 + 11);         // LiteIndex: 2 data words, 11 pointers
    b.set_struct_ptr(0, root, 2, 11);
    uint8_t* d = reinterpret_cast<uint8_t*>(b.w.data() + root);
    auto put16 = [&](size_t off, uint16_t v) { std::memcpy(reinterpret_cast<uint8_t*>(b.w.data() + root) + off, &v, 2); };
    put16(0, (uint16_t)ix.params.k); put16(2, (uint16_t)ix.params.s); put16(4, (uint16_t)ix.params.t); put16(6, (uint16_t)ix.params.l);
    d[8] = (uint8_t)((ix.params.open ? 1 : 0) | (ix.hpc ? 2 : 0));   // open @4 = bit 64, hpc @10 = bit 65
    put16(10, kIdxFormatVersion);                                     // formatVersion @17: the 16-bit hole at bytes 10..11
    const size_t P = root + 2;                   // pointer section
    // liteTree @5 (pointer 0): { liteNodes, blockRanges }
    {
        const size_t lt = b.alloc(2);
        b.set_struct_ptr(P + 0, lt, 0, 2);
        // liteNodes: composite list of { data: parentIndex u32, identicalToParent bit 32 | ptr: id }
        const size_t tag = b.alloc(1 + 2 * n);
        b.w[tag] = ((uint64_t)n << 2) | (uint64_t)1 << 32 | (uint64_t)1 << 48;
        b.set_list_ptr(lt + 0, tag, 7, 2 * n);
        for (size_t i = 0; i < n; ++i) {
            const bool identical = i > 0 && ix.offsets[i + 1] == ix.offsets[i];
            b.w[tag + 1 + 2 * i] = (uint64_t)(i == 0 ? 0u : ix.parent[i]) | (uint64_t)(identical ? 1 : 0) << 32;
        }
        for (size_t i = 0; i < n; ++i) b.text(tag + 1 + 2 * i + 1, ix.node_id[i]);
        // blockRanges (pointer 1): the place stage does not read them; left empty
    }
    // seedChangeHashes / ParentCounts / ChildCounts @6..8: List(List(T)), outer list = segments of <= 5e8 elements
    const size_t n_seg = (size_t)std::max<uint64_t>(1, (total + kSegmentElems - 1) / kSegmentElems);
    auto soa = [&](size_t ptr_index, const void* src, unsigned elem_bytes) {
        const size_t outer = b.alloc(n_seg);
        b.set_list_ptr(P + ptr_index, outer, 6, n_seg);
        for (size_t sg = 0; sg < n_seg; ++sg) {
            const uint64_t lo = sg * kSegmentElems, hi = std::min<uint64_t>(total, lo + kSegmentElems);
            b.prim_list(outer + sg, static_cast<const uint8_t*>(src) + lo * elem_bytes, (size_t)(hi - lo), elem_bytes);
        }
    };
    soa(1, ix.hash.data(), 8);
    soa(2, ix.parent_count.data(), 2);
    soa(3, ix.child_count.data(), 2);
    b.prim_list(P + 4, ix.offsets.data(), n + 1, 8);   // nodeChangeOffsets @9
    // (pointers 5..10: the mgsr fields and the substitution matrix stay null)

    // flat array: u32 segment count - 1, u32 words of segment 0, the segment
    std::vector<uint8_t> flat(8 + 8 * b.w.size());
    const uint32_t zero = 0, words = (uint32_t)b.w.size();
    if (b.w.size() > 0xffffffffULL) throw std::runtime_error("index too large for one Cap'n Proto segment");
    std::memcpy(flat.data(), &zero, 4);
    std::memcpy(flat.data() + 4, &words, 4);
    std::memcpy(flat.data() + 8, b.w.data(), 8 * b.w.size());

    uint8_t header[kHeaderSize] = {0};
    auto put32 = [&](size_t off, uint32_t v) { std::memcpy(header + off, &v, 4); };
    put32(0, kMagic); put32(4, kHeaderVersion);
    put32(8, (uint32_t)ix.params.k); put32(12, (uint32_t)ix.params.s); put32(16, (uint32_t)ix.params.t); put32(20, (uint32_t)ix.params.l);
    header[24] = ix.hpc ? 1 : 0; header[25] = ix.params.open ? 1 : 0; header[26] = uncompressed ? 1 : 0;

    std::ofstream out(path, std::ios::binary | std::ios::trunc);
    if (!out) throw std::runtime_error("cannot write " + path);
    out.write(reinterpret_cast<const char*>(header), kHeaderSize);
    if (uncompressed) {
        out.write(reinterpret_cast<const char*>(flat.data()), (std::streamsize)flat.size());
    } else {
        const Zstd& z = zstd();
        for (size_t at = 0; at < flat.size() || at == 0; at += kFrameSize) {
            const size_t len = std::min(kFrameSize, flat.size() - at);
            std::vector<uint8_t> frame(z.compressBound(len));
            void* cctx = z.createCCtx();
            if (!cctx) throw std::runtime_error("ZSTD_createCCtx failed");
            z.CCtx_setParameter(cctx, kZstdCompressionLevel, zstd_level);
            z.CCtx_setParameter(cctx, kZstdChecksumFlag, 1);
            const size_t r = z.compress2(cctx, frame.data(), frame.size(), flat.data() + at, len);
            z.freeCCtx(cctx);
            if (z.isError(r)) throw std::runtime_error(std::string("ZSTD compression failed: ") + z.getErrorName(r));
            out.write(reinterpret_cast<const char*>(frame.data()), (std::streamsize)r);
            if (flat.size() == 0) break;
        }
    }
    if (!out) throw std::runtime_error("failed to write " + path);
}

void load_idx(const std::string& path, LiteIndex& out) {
    const std::vector<uint8_t> file = read_file(path);
    IdxHeader h;
    if (!decode_header(file.data(), file.size(), h)) throw std::runtime_error(path + " is not a panmap index (no PMI1 header)");
    std::vector<uint8_t> inflated;
    const uint8_t* payload = file.data() + kHeaderSize;
    size_t payload_len = file.size() - kHeaderSize;
    if (!h.uncompressed) {   // concatenated zstd frames (src/zstd_compression.cpp:141-205)
        const Zstd& z = zstd();
        std::vector<std::pair<size_t, size_t>> frames;   // (offset, compressed size)
        std::vector<unsigned long long> sizes;
        unsigned long long total = 0;
        for (size_t pos = 0; pos < payload_len;) {
            const size_t fc = z.findFrameCompressedSize(payload + pos, payload_len - pos);
            if (z.isError(fc)) break;
            const unsigned long long fs = z.getFrameContentSize(payload + pos, fc);
            if (fs == kContentSizeError) throw std::runtime_error("Not a valid ZSTD frame in: " + path);
            if (fs == kContentSizeUnknown) throw std::runtime_error("Cannot determine uncompressed size for a frame in: " + path);
            frames.emplace_back(pos, fc);
            sizes.push_back(fs);
            total += fs;
            pos += fc;
        }
        if (frames.empty()) throw std::runtime_error("No valid ZSTD frames in: " + path);
        inflated.resize((size_t)total);
        size_t at = 0;
        for (size_t i = 0; i < frames.size(); ++i) {
            const size_t r = z.decompress(inflated.data() + at, (size_t)sizes[i], payload + frames[i].first, frames[i].second);
            if (z.isError(r) || r != sizes[i]) throw std::runtime_error("ZSTD decompression failed for: " + path);
            at += (size_t)sizes[i];
        }
        payload = inflated.data();
        payload_len = inflated.size();
    }
    if (payload_len % 8 != 0 && h.uncompressed) payload_len -= payload_len % 8;
    // capnp words must be 8-byte aligned: the mapped payload starts 32 bytes into the file, the inflated one at a vector
    std::vector<uint64_t> aligned((payload_len + 7) / 8);
    std::memcpy(aligned.data(), payload, payload_len);
    capnp::Message msg;
    msg.parse(reinterpret_cast<const uint8_t*>(aligned.data()), payload_len);
    const capnp::StructR root = capnp::root(msg);
    if (!root.valid()) throw std::runtime_error("index message has no root");
    const uint16_t version = root.get<uint16_t>(10);
    if (version != kIdxFormatVersion)
        throw std::runtime_error("Index format version " + std::to_string(version) + " is incompatible with this panmap (expects " +
                                 std::to_string(kIdxFormatVersion) + "). Rebuild the index (delete the .idx and rerun).");
    LiteIndex ix;
    ix.params.k = root.get<uint16_t>(0); ix.params.s = root.get<uint16_t>(2); ix.params.t = root.get<uint16_t>(4); ix.params.l = root.get<uint16_t>(6);
    ix.params.open = root.bit(64);
    ix.hpc = root.bit(65);
    if (ix.params.k != h.k || ix.params.s != h.s || ix.params.t != h.t || ix.params.l != h.l || ix.params.open != h.open || ix.hpc != h.hpc)
        throw std::runtime_error("index header and payload disagree on the seeding parameters");
    const capnp::StructR tree = capnp::as_struct(root.ptr(0));
    const capnp::ListR nodes = tree.valid() ? capnp::as_list(tree.ptr(0)) : capnp::ListR();
    const size_t n = nodes.size();
    ix.node_id.resize(n);
    ix.parent.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const capnp::StructR nd = nodes.struct_at((uint32_t)i);
        ix.node_id[i] = capnp::as_text(nd.ptr(0));
        ix.parent[i] = i == 0 ? 0u : nd.get<uint32_t>(0);
        if (i > 0 && ix.parent[i] >= i) throw std::runtime_error("index tree is not in DFS pre-order (parent index >= node index)");
    }
    if (root.ptr(1).null() || root.ptr(2).null() || root.ptr(3).null() || root.ptr(4).null())
        throw std::runtime_error("Index missing required V3 fields (seedChangeHashes, etc). V2 is no longer supported.");
    const capnp::ListR offs = capnp::as_list(root.ptr(4));
    if (offs.size() < n + 1)
        throw std::runtime_error("Struct-of-arrays format offsets size mismatch: " + std::to_string(offs.size()) + " vs " + std::to_string(n + 1));
    ix.offsets.resize(n + 1);
    for (size_t i = 0; i <= n; ++i) ix.offsets[i] = offs.prim<uint64_t>((uint32_t)i);
    for (size_t i = 0; i < n; ++i)
        if (ix.offsets[i + 1] < ix.offsets[i]) throw std::runtime_error("index node offsets are not monotone");
    const uint64_t total = ix.offsets[n];
    auto gather = [&](unsigned ptr_index, unsigned elem_bytes, void* dst) {
        const capnp::ListR outer = capnp::as_list(root.ptr(ptr_index));
        uint64_t at = 0;
        for (uint32_t sg = 0; sg < outer.size(); ++sg) {
            const capnp::ListR inner = capnp::as_list(outer.ptr_at(sg));
            const uint64_t cnt = inner.size();
            if (at + cnt > total) throw std::runtime_error("index seed-change arrays are longer than the node offsets say");
            if (cnt) std::memcpy(static_cast<uint8_t*>(dst) + at * elem_bytes, inner.bytes((uint32_t)elem_bytes), (size_t)cnt * elem_bytes);
            at += cnt;
        }
        if (at != total) throw std::runtime_error("index seed-change arrays are shorter than the node offsets say");
    };
    ix.hash.resize((size_t)total);
    ix.parent_count.resize((size_t)total);
    ix.child_count.resize((size_t)total);
    gather(1, 8, ix.hash.data());
    gather(2, 2, ix.parent_count.data());
    gather(3, 2, ix.child_count.data());
    out = std::move(ix);
}

}  // namespace pmx
