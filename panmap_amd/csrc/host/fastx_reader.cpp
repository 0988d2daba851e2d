#include "fastx_reader.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace pmx {
namespace {

// whole inflated file (gzread also passes plain files through)
std::vector<char> slurp(const std::string& path) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    gzbuffer(f, 1 << 20);
    std::vector<char> buf;
    size_t used = 0;
    for (;;) {
        if (buf.size() - used < (1u << 22)) buf.resize(std::max<size_t>(buf.size() * 2, (size_t)1 << 24));
        const int got = gzread(f, buf.data() + used, (unsigned)std::min<size_t>(buf.size() - used, (size_t)1 << 30));
        if (got < 0) { gzclose(f); throw std::runtime_error("read error in " + path); }
        if (got == 0) break;
        used += (size_t)got;
    }
    gzclose(f);
    buf.resize(used);
    return buf;
}

struct Line { const char* p; size_t n; };

// next line without its terminator (and without a trailing CR); false at end of data
bool next_line(const char*& cur, const char* end, Line& ln) {
    if (cur >= end) return false;
    const char* e = (const char*)memchr(cur, '\n', (size_t)(end - cur));
    const char* stop = e ? e : end;
    ln.p = cur;
    ln.n = (size_t)(stop - cur);
    if (ln.n > 0 && ln.p[ln.n - 1] == '\r') --ln.n;
    cur = e ? e + 1 : end;
    return true;
}

void push_name(FastxReads& out, const Line& hdr) {   // header line without its '>' / '@': up to the first white space
    size_t k = 1;
    while (k < hdr.n && hdr.p[k] != ' ' && hdr.p[k] != '\t' && hdr.p[k] != '\v' && hdr.p[k] != '\f' && hdr.p[k] != '\r') ++k;
    if (hdr.n > 1) out.names.insert(out.names.end(), hdr.p + 1, hdr.p + k);
    out.name_off.push_back((int64_t)out.names.size());
}

}  // namespace

void read_fastx(const std::string& path, FastxReads& out) {
    out = FastxReads();
    out.off.push_back(0);
    out.name_off.push_back(0);
    const std::vector<char> data = slurp(path);
    const char* cur = data.data();
    const char* end = cur + data.size();
    Line ln;
    bool have = next_line(cur, end, ln);
    while (have) {
        if (ln.n == 0 || (ln.p[0] != '>' && ln.p[0] != '@')) { have = next_line(cur, end, ln); continue; }   // skip to a header
        const bool fastq = ln.p[0] == '@';
        const Line hdr = ln;
        const size_t seq0 = out.seq.size();
        // sequence lines until a line that starts with a marker ('+' closes a FASTQ sequence; '>' / '@' start the next record)
        have = next_line(cur, end, ln);
        while (have && !(ln.n > 0 && (ln.p[0] == '>' || ln.p[0] == '@' || ln.p[0] == '+'))) {
            out.seq.insert(out.seq.end(), ln.p, ln.p + ln.n);
            have = next_line(cur, end, ln);
        }
        const size_t slen = out.seq.size() - seq0;
        if (have && ln.p[0] == '+' && fastq) {
            // quality lines until at least as many characters as bases
            const size_t q0 = out.qual.size();
            have = next_line(cur, end, ln);
            while (have && out.qual.size() - q0 < slen) {
                out.qual.insert(out.qual.end(), ln.p, ln.p + ln.n);
                have = next_line(cur, end, ln);
            }
            if (out.qual.size() - q0 != slen) {   // kseq_read returns -2: the reading loop of the reference ends here
                out.seq.resize(seq0);
                out.qual.resize(q0);
                break;
            }
        } else if (have && ln.p[0] == '+') {
            have = next_line(cur, end, ln);       // (a '+' line after a FASTA record: kseq would read a quality; not produced by any writer)
        }
        if (!fastq || out.qual.size() < out.seq.size()) out.qual.resize(out.seq.size(), 0);   // FASTA: no qualities (marked 0)
        push_name(out, hdr);
        out.off.push_back((int64_t)out.seq.size());
    }
}

void read_fastq_paired(const std::string& path1, const std::string& path2, FastxReads& out) {
    FastxReads a;
    read_fastx(path1, a);
    auto fill_missing_qual = [](FastxReads& r) {
        for (int64_t i = 0; i < r.n(); ++i)
            if (r.off[i + 1] > r.off[i] && r.qual[(size_t)r.off[i]] == 0) std::fill(r.qual.begin() + r.off[i], r.qual.begin() + r.off[i + 1], 'I');
    };
    fill_missing_qual(a);
    if (path2.empty()) { out = std::move(a); return; }
    FastxReads b;
    read_fastx(path2, b);
    if (a.n() != b.n()) throw std::runtime_error("Error: " + path2 + " does not contain the same number of reads as " + path1);
    fill_missing_qual(b);
    out = FastxReads();
    out.seq.reserve(a.seq.size() + b.seq.size());
    out.qual.reserve(a.qual.size() + b.qual.size());
    out.names.reserve(a.names.size() + b.names.size());
    out.off.push_back(0);
    out.name_off.push_back(0);
    for (int64_t i = 0; i < a.n(); ++i) {
        out.seq.insert(out.seq.end(), a.seq.begin() + a.off[i], a.seq.begin() + a.off[i + 1]);
        out.qual.insert(out.qual.end(), a.qual.begin() + a.off[i], a.qual.begin() + a.off[i + 1]);
        out.names.insert(out.names.end(), a.names.begin() + a.name_off[i], a.names.begin() + a.name_off[i + 1]);
        out.off.push_back((int64_t)out.seq.size());
        out.name_off.push_back((int64_t)out.names.size());
        // mate 2: reverse complement (upper-case ACGT only, src/seeding.cpp:271-284), qualities reversed
        for (int64_t j = b.off[i + 1] - 1; j >= b.off[i]; --j) {
            const char c = b.seq[(size_t)j];
            out.seq.push_back(c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c);
            out.qual.push_back(b.qual[(size_t)j]);
        }
        out.names.insert(out.names.end(), b.names.begin() + b.name_off[i], b.names.begin() + b.name_off[i + 1]);
        out.off.push_back((int64_t)out.seq.size());
        out.name_off.push_back((int64_t)out.names.size());
    }
}

}  // namespace pmx
