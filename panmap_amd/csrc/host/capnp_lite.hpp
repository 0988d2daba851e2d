// Minimal read-only Cap'n Proto message walker (struct / list / far pointers, text, data).
// Only what the PanMAN (`.panman`) and panmap index (`.idx`) payloads need; no schema compiler.
// Wire format: https://capnproto.org/encoding.html (stream framing + pointer encoding).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pmx {
namespace capnp {

struct Message {
    std::vector<const uint64_t*> seg;
    std::vector<uint32_t> seg_words;

    // every object a pointer leads to must lie inside its segment (a truncated or crafted file must end in a format
    // error, never in an out-of-bounds read: the capnp library bounds-checks every pointer as well)
    void check(uint32_t sid, const uint64_t* at, uint64_t words) const {
        if (sid >= seg.size()) throw std::runtime_error("capnp: bad segment id");
        const uint64_t* b = seg[sid];
        if (at < b || (uint64_t)(at - b) > seg_words[sid] || words > seg_words[sid] - (uint64_t)(at - b))
            throw std::runtime_error("capnp: pointer target outside its segment");
    }

    // stream-framed message: u32 nseg-1, u32 sizes[nseg] (words), pad to 8 B, segments
    void parse(const uint8_t* buf, size_t len) {
        if (len < 8) throw std::runtime_error("capnp: truncated header");
        uint32_t nseg;
        std::memcpy(&nseg, buf, 4);
        nseg += 1;
        size_t hdr = 4 + 4 * (size_t)nseg;
        hdr = (hdr + 7) & ~(size_t)7;
        if (len < hdr) throw std::runtime_error("capnp: truncated segment table");
        size_t off = hdr;
        for (uint32_t i = 0; i < nseg; ++i) {
            uint32_t w;
            std::memcpy(&w, buf + 4 + 4 * (size_t)i, 4);
            if (off + 8 * (size_t)w > len) throw std::runtime_error("capnp: truncated segment");
            seg.push_back(reinterpret_cast<const uint64_t*>(buf + off));
            seg_words.push_back(w);
            off += 8 * (size_t)w;
        }
    }
};

struct Ptr {
    const Message* m = nullptr;
    uint32_t seg = 0;
    const uint64_t* p = nullptr;  // location of the pointer word
    bool null() const { return !p || *p == 0; }
};

struct StructR;
struct ListR;

// Resolved object location after following far pointers.
struct Resolved {
    uint64_t tag = 0;              // struct/list pointer word describing the object
    const uint64_t* target = nullptr;  // first word of the object content
    uint32_t seg = 0;
    bool ok = false;
};

inline Resolved resolve(const Ptr& ptr) {
    Resolved r;
    if (ptr.null()) return r;
    uint64_t w = *ptr.p;
    const Message* m = ptr.m;
    uint32_t kind = (uint32_t)(w & 3);
    if (kind == 2) {  // far pointer
        bool dbl = (w >> 2) & 1;
        uint32_t off = (uint32_t)((w >> 3) & 0x1fffffff);
        uint32_t sid = (uint32_t)(w >> 32);
        if (sid >= m->seg.size()) throw std::runtime_error("capnp: bad far segment");
        const uint64_t* pad = m->seg[sid] + off;
        m->check(sid, pad, dbl ? 2 : 1);
        if (!dbl) {
            uint64_t pw = *pad;
            int32_t o = (int32_t)((int32_t)(uint32_t)(pw & 0xffffffffu) >> 2);
            if ((pw & 3) == 2) throw std::runtime_error("capnp: far pointer lands on a far pointer");
            r.tag = pw;
            r.target = pad + 1 + o;
            r.seg = sid;
            r.ok = true;
            return r;
        }
        uint64_t far2 = pad[0];
        uint32_t off2 = (uint32_t)((far2 >> 3) & 0x1fffffff);
        uint32_t sid2 = (uint32_t)(far2 >> 32);
        if (sid2 >= m->seg.size() || (far2 & 3) != 2 || ((far2 >> 2) & 1)) throw std::runtime_error("capnp: bad double-far landing pad");
        r.tag = pad[1];
        r.target = m->seg[sid2] + off2;
        r.seg = sid2;
        r.ok = true;
        return r;
    }
    int32_t o = (int32_t)((int32_t)(uint32_t)(w & 0xffffffffu) >> 2);
    r.tag = w;
    r.target = ptr.p + 1 + o;
    r.seg = ptr.seg;
    r.ok = true;
    return r;
}

struct StructR {
    const Message* m = nullptr;
    uint32_t seg = 0;
    const uint64_t* data = nullptr;
    uint16_t dwords = 0, pwords = 0;

    bool valid() const { return data != nullptr; }
    uint64_t word(unsigned i) const { return i < dwords ? data[i] : 0; }
    template <class T> T get(unsigned byte_off) const {
        T v{};
        if (byte_off + sizeof(T) <= 8u * dwords) std::memcpy(&v, reinterpret_cast<const uint8_t*>(data) + byte_off, sizeof(T));
        return v;
    }
    bool bit(unsigned bit_off) const { return (word(bit_off >> 6) >> (bit_off & 63)) & 1; }
    Ptr ptr(unsigned i) const {
        Ptr p;
        p.m = m;
        p.seg = seg;
        p.p = i < pwords ? data + dwords + i : nullptr;
        return p;
    }
};

inline StructR as_struct(const Ptr& ptr) {
    StructR s;
    Resolved r = resolve(ptr);
    if (!r.ok) return s;
    if ((r.tag & 3) != 0) throw std::runtime_error("capnp: expected struct pointer");
    s.m = ptr.m;
    s.seg = r.seg;
    s.data = r.target;
    s.dwords = (uint16_t)(r.tag >> 32);
    s.pwords = (uint16_t)(r.tag >> 48);
    ptr.m->check(r.seg, r.target, (uint64_t)s.dwords + s.pwords);
    return s;
}

struct ListR {
    const Message* m = nullptr;
    uint32_t seg = 0;
    const uint64_t* base = nullptr;
    uint32_t n = 0;
    uint32_t esize = 0;  // capnp element size code
    uint16_t dwords = 0, pwords = 0;  // composite only

    uint32_t size() const { return n; }
    StructR struct_at(uint32_t i) const {
        StructR s;
        if (esize != 7) throw std::runtime_error("capnp: list is not composite");
        s.m = m;
        s.seg = seg;
        s.data = base + (size_t)i * (dwords + pwords);
        s.dwords = dwords;
        s.pwords = pwords;
        return s;
    }
    Ptr ptr_at(uint32_t i) const {
        if (esize != 6) throw std::runtime_error("capnp: list is not a pointer list");
        Ptr p;
        p.m = m;
        p.seg = seg;
        p.p = base + i;
        return p;
    }
    // bytes per element of a primitive list (codes 2..5); 0 for void / bit / pointer / composite lists
    uint32_t elem_bytes() const { return esize >= 2 && esize <= 5 ? 1u << (esize - 2) : 0u; }
    // as_list() validated n elements of the width the POINTER declares: an accessor of another width would read past that
    // range (a crafted list that declares bytes and is read as 64-bit words), so the width and the index are checked here
    template <class T> T prim(uint32_t i) const {
        if (elem_bytes() != sizeof(T)) throw std::runtime_error("capnp: list element width differs from the field's type");
        if (i >= n) throw std::runtime_error("capnp: list index out of range");
        T v;
        std::memcpy(&v, reinterpret_cast<const uint8_t*>(base) + (size_t)i * sizeof(T), sizeof(T));
        return v;
    }
    bool bit_at(uint32_t i) const {
        if (esize != 1) throw std::runtime_error("capnp: list is not a bit list");
        if (i >= n) throw std::runtime_error("capnp: list index out of range");
        return (reinterpret_cast<const uint8_t*>(base)[i >> 3] >> (i & 7)) & 1;
    }
    // the raw content of a primitive list whose elements are `width` bytes wide (n * width bytes are inside the segment)
    const uint8_t* bytes(uint32_t width) const {
        if (n != 0 && elem_bytes() != width) throw std::runtime_error("capnp: list element width differs from the field's type");
        return reinterpret_cast<const uint8_t*>(base);
    }
};

inline ListR as_list(const Ptr& ptr) {
    ListR l;
    Resolved r = resolve(ptr);
    if (!r.ok) return l;
    if ((r.tag & 3) != 1) throw std::runtime_error("capnp: expected list pointer");
    l.m = ptr.m;
    l.seg = r.seg;
    l.esize = (uint32_t)((r.tag >> 32) & 7);
    uint32_t cnt = (uint32_t)(r.tag >> 35);
    if (l.esize == 7) {
        ptr.m->check(r.seg, r.target, 1 + (uint64_t)cnt);   // tag word + content words
        uint64_t tag = *r.target;
        l.n = (uint32_t)((tag & 0xffffffffu) >> 2);
        l.dwords = (uint16_t)(tag >> 32);
        l.pwords = (uint16_t)(tag >> 48);
        l.base = r.target + 1;
        if ((uint64_t)l.n * ((uint64_t)l.dwords + l.pwords) > cnt) throw std::runtime_error("capnp: composite list overruns its word count");
    } else {
        static const uint32_t bits[7] = {0, 1, 8, 16, 32, 64, 64};
        l.n = cnt;
        l.base = r.target;
        ptr.m->check(r.seg, r.target, ((uint64_t)cnt * bits[l.esize] + 63) / 64);
    }
    return l;
}

inline std::string as_text(const Ptr& ptr) {
    ListR l = as_list(ptr);
    if (!l.base || l.n == 0) return std::string();
    return std::string(reinterpret_cast<const char*>(l.base), l.n - 1);  // drop NUL
}

inline StructR root(const Message& m) {
    Ptr p;
    p.m = &m;
    p.seg = 0;
    p.p = m.seg.at(0);
    return as_struct(p);
}

}  // namespace capnp
}  // namespace pmx
