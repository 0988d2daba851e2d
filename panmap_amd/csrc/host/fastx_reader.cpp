#include "fastx_reader.hpp"
#include "device/pmx_options.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <stdexcept>
#include <thread>

namespace pmx {
namespace {

// whole inflated file (gzread also passes plain files through).  The path is opened ONCE: a FIFO / process substitution /
// /dev/stdin cannot be rewound or reopened, so only a regular file takes the one-read fast path (magic read with pread,
// which does not move the offset), and everything else hands the very same descriptor to zlib.
std::vector<char> slurp(const std::string& path) {
    const int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) throw std::runtime_error("cannot open " + path);
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
        unsigned char magic[2] = {0, 0};
        const ssize_t got = pread(fd, magic, 2, 0);
        if (!(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {
            // a plain regular file: one read of its size (gzread would pass it through in small buffers)
            std::vector<char> whole((size_t)st.st_size);
            size_t rd = 0;
            while (rd < whole.size()) {
                const ssize_t r = pread(fd, whole.data() + rd, whole.size() - rd, (off_t)rd);
                if (r < 0) { close(fd); throw std::runtime_error("read error in " + path); }
                if (r == 0) break;      // the file shrank under us: keep what is there
                rd += (size_t)r;
            }
            close(fd);
            whole.resize(rd);
            return whole;
        }
    }
    gzFile f = gzdopen(fd, "rb");       // owns fd from here on (gzclose closes it)
    if (!f) { close(fd); throw std::runtime_error("cannot open " + path); }
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

namespace {
// the general (kseq-compatible) parser over [cur, end): records appended to `out` (which holds its leading offsets)
void parse_general(const char* cur, const char* end, FastxReads& out) {
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

// Strict four-line FASTQ over [cur, end), `cur` at a record start, in two passes so that the pieces of a file can be written
// straight into the final arrays: measure() counts records / bases / name bytes and returns false as soon as a record is not
// "@name / bases / + / as many qualities" with non-empty bases (the caller then parses the whole file with the general
// parser); fill() writes the piece at its place.
struct PieceSize { size_t n_rec = 0, n_seq = 0, n_name = 0; };
size_t name_len(const Line& hdr) {
    size_t k = 1;
    while (k < hdr.n && hdr.p[k] != ' ' && hdr.p[k] != '\t' && hdr.p[k] != '\v' && hdr.p[k] != '\f' && hdr.p[k] != '\r') ++k;
    return hdr.n > 1 ? k - 1 : 0;
}
bool strict4_measure(const char* cur, const char* end, PieceSize& sz) {
    Line h, sq, pl, ql;
    while (cur < end) {
        if (!next_line(cur, end, h)) break;
        if (h.n == 0 && cur >= end) break;                     // trailing newline
        if (h.n == 0 || h.p[0] != '@') return false;
        if (!next_line(cur, end, sq) || !next_line(cur, end, pl) || !next_line(cur, end, ql)) return false;
        if (sq.n == 0 || sq.p[0] == '@' || sq.p[0] == '+' || sq.p[0] == '>' || pl.n == 0 || pl.p[0] != '+' || ql.n != sq.n) return false;
        ++sz.n_rec;
        sz.n_seq += sq.n;
        sz.n_name += name_len(h);
    }
    return true;
}
void strict4_fill(const char* cur, const char* end, FastxReads& out, size_t rec0, size_t seq0, size_t name0) {
    Line h, sq, pl, ql;
    size_t r = rec0, sp = seq0, np = name0;
    while (cur < end) {
        if (!next_line(cur, end, h)) break;
        if (h.n == 0 && cur >= end) break;
        next_line(cur, end, sq); next_line(cur, end, pl); next_line(cur, end, ql);
        memcpy(&out.seq[sp], sq.p, sq.n);
        memcpy(&out.qual[sp], ql.p, sq.n);
        const size_t nl = name_len(h);
        if (nl) memcpy(&out.names[np], h.p + 1, nl);
        sp += sq.n; np += nl; ++r;
        out.off[r] = (int64_t)sp;
        out.name_off[r] = (int64_t)np;
    }
}

// first record start at or after `from` (a line "@..." followed by a line, a "+..." line and a line as long as the second one;
// what the reference's fqNextRecord accepts, src/placement.cpp:97-119), or `end`
const char* next_record_start(const char* begin, const char* end, const char* from) {
    const char* o = from;
    if (o > begin)
        while (o < end && o[-1] != '\n') ++o;
    while (o < end) {
        if (*o == '@') {
            const char* c = o;
            Line h, sq, pl, ql;
            if (next_line(c, end, h) && next_line(c, end, sq) && next_line(c, end, pl) && pl.n > 0 && pl.p[0] == '+' && next_line(c, end, ql) && ql.n == sq.n && sq.n > 0)
                return o;
        }
        const char* nl = (const char*)memchr(o, '\n', (size_t)(end - o));
        if (!nl) return end;
        o = nl + 1;
    }
    return end;
}

}  // namespace

void read_fastx(const std::string& path, FastxReads& out) {
    out = FastxReads();
    out.off.push_back(0);
    out.name_off.push_back(0);
    const std::vector<char> data = slurp(path);
    const char* begin = data.data();
    const char* end = begin + data.size();
    // Large four-line FASTQ (what sequencers and the benchmark write): the file is cut at record starts and the pieces are
    // parsed side by side, in file order (the reference does the same for uncompressed input, src/placement.cpp:120-161;
    // inflating a .gz stays serial there and here).  Anything else -- FASTA, wrapped records, a piece that does not parse as
    // strict four-line records -- goes through the general parser as before: same result either way.
    unsigned n_thr = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = pmx::opt_str(pmx::O_FASTX_THREADS)) n_thr = (unsigned)std::max(1, atoi(e));
    if (n_thr > 1 && data.size() >= ((size_t)8 << 20) && begin[0] == '@') {
        std::vector<const char*> cut(n_thr + 1, end);
        cut[0] = begin;
        for (unsigned i = 1; i < n_thr; ++i) cut[i] = next_record_start(begin, end, begin + data.size() / n_thr * i);
        for (unsigned i = 1; i < n_thr; ++i) cut[i] = std::max(cut[i], cut[i - 1]);
        std::vector<PieceSize> size(n_thr);
        std::vector<int> ok(n_thr, 1);
        auto on_pieces = [&](const std::function<void(unsigned)>& body) {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < n_thr; ++t) pool.emplace_back([&, t]() { try { body(t); } catch (...) { ok[t] = 0; } });
            for (auto& th : pool) th.join();
        };
        on_pieces([&](unsigned t) { ok[t] = strict4_measure(cut[t], cut[t + 1], size[t]) ? 1 : 0; });
        if (std::all_of(ok.begin(), ok.end(), [](int v) { return v == 1; })) {
            std::vector<size_t> rec0(n_thr + 1, 0), seq0(n_thr + 1, 0), name0(n_thr + 1, 0);
            for (unsigned t = 0; t < n_thr; ++t) { rec0[t + 1] = rec0[t] + size[t].n_rec; seq0[t + 1] = seq0[t] + size[t].n_seq; name0[t + 1] = name0[t] + size[t].n_name; }
            out.seq.resize(seq0[n_thr]); out.qual.resize(seq0[n_thr]); out.names.resize(name0[n_thr]);
            out.off.assign(rec0[n_thr] + 1, 0); out.name_off.assign(rec0[n_thr] + 1, 0);
            on_pieces([&](unsigned t) { strict4_fill(cut[t], cut[t + 1], out, rec0[t], seq0[t], name0[t]); });
            if (std::all_of(ok.begin(), ok.end(), [](int v) { return v == 1; })) return;
            out = FastxReads();
            out.off.push_back(0);
            out.name_off.push_back(0);
        }
    }
    parse_general(begin, end, out);
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
