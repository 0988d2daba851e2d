// BAM egress of the align stage (host): restates what the reference does after align_reads_direct returns --
// build_bam_from_result (src/conversion.cpp:288-388), compute_sam_flags (:257-274), compute_tlen (:276-286)
// and the sort / write / index tail of alignAndWriteBam (:426-538) -- on top of the BAM, BGZF and BAI formats
// (SAM spec v1.6 sections 4.1, 4.2, 5.2).  The reference goes through htslib (bam_set1, sam_hdr_write,
// bam_write1, sam_index_build); htslib is not linked here, the container formats are written directly with zlib.
// Byte parity of the compressed file is not a goal (it depends on the deflate implementation htslib was built
// with); the decompressed BAM stream and the index contents are what tests/test_bam.py checks.
#include "bam_writer.hpp"
#include "device/pmx_options.hpp"

#include <zlib.h>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <thread>

namespace pmx {
namespace {

enum { BAM_FPAIRED = 1, BAM_FPROPER_PAIR = 2, BAM_FUNMAP = 4, BAM_FMUNMAP = 8, BAM_FREVERSE = 16, BAM_FMREVERSE = 32, BAM_FREAD1 = 64,
       BAM_FREAD2 = 128 };
enum { BAM_CSOFT_CLIP = 4 };

// src/conversion.cpp:257-274
uint16_t compute_sam_flags(bool is_paired, bool is_read1, uint8_t rev, uint8_t mate_rev, uint8_t proper_frag, bool mate_unmapped) {
    uint16_t flag = 0;
    if (is_paired) {
        flag |= BAM_FPAIRED;
        if (proper_frag) flag |= BAM_FPROPER_PAIR;
        if (rev) flag |= BAM_FREVERSE;
        if (mate_rev) flag |= BAM_FMREVERSE;
        if (mate_unmapped) flag |= BAM_FMUNMAP;
        flag |= is_read1 ? BAM_FREAD1 : BAM_FREAD2;
    } else if (rev) flag |= BAM_FREVERSE;
    return flag;
}

// src/conversion.cpp:276-286
int32_t compute_tlen(int32_t this_rs, int32_t this_re, uint8_t this_rev, int32_t mate_rs, int32_t mate_re, uint8_t mate_rev) {
    const int this_pos5 = this_rev ? this_re - 1 : this_rs;
    const int mate_pos5 = mate_rev ? mate_re - 1 : mate_rs;
    int tlen = mate_pos5 - this_pos5;
    if (tlen > 0) tlen++;
    else if (tlen < 0) tlen--;
    return tlen;
}

// htslib seq_nt16_table: "=ACMGRSVTWYHKDBN", case-insensitive, everything else 15 (N)
uint8_t nt16(char c) {
    static const char* codes = "=ACMGRSVTWYHKDBN";
    if (c >= 'a' && c <= 'z') c = (char)(c - 32);
    if (c == '=') return 0;
    for (int i = 1; i < 16; ++i)
        if (codes[i] == c) return (uint8_t)i;
    return 15;
}

// hts_reg2bin(beg, end, 14, 5) (SAM spec 5.3)
int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

void put32(std::string& s, uint32_t v) { s.append(reinterpret_cast<const char*>(&v), 4); }
void put16(std::string& s, uint16_t v) { s.append(reinterpret_cast<const char*>(&v), 2); }
void put64(std::string& s, uint64_t v) { s.append(reinterpret_cast<const char*>(&v), 8); }

struct Rec {
    int32_t sort_pos;     // read_align_t::pos (1-based) -- the reference's sort key
    int32_t pos, end;     // 0-based start, end (pos + reference length)
    std::string bytes;    // block_size + record
};

// build_bam_from_result (src/conversion.cpp:288-388) + bam_set1's encoding
Rec build_record(const std::string& qname_full, const std::string& seq, const std::string& qual, const read_align_t& aln, int read_len,
                 bool is_paired, bool is_read1, uint8_t mate_rev, int32_t mate_pos, int32_t this_rs, int32_t this_re, int32_t mate_rs,
                 int32_t mate_re, uint8_t proper_frag, bool mate_unmapped) {
    std::string qname = qname_full;
    if (qname.size() >= 2 && qname[qname.size() - 2] == '/' && (qname.back() == '1' || qname.back() == '2')) qname.resize(qname.size() - 2);
    const uint8_t effective_rev = (is_paired && !is_read1) ? (uint8_t)!aln.rev : aln.rev;
    const uint16_t flag = compute_sam_flags(is_paired, is_read1, effective_rev, mate_rev, proper_frag, mate_unmapped);
    const uint32_t clip5 = aln.rev ? (uint32_t)(read_len - aln.qe) : (uint32_t)aln.qs;
    const uint32_t clip3 = aln.rev ? (uint32_t)aln.qs : (uint32_t)(read_len - aln.qe);
    std::vector<uint32_t> cigar;
    if (clip5 > 0) cigar.push_back(clip5 << 4 | BAM_CSOFT_CLIP);
    for (int j = 0; j < aln.n_cigar; ++j) cigar.push_back(aln.cigar[j]);
    if (clip3 > 0) cigar.push_back(clip3 << 4 | BAM_CSOFT_CLIP);

    std::string bam_seq((size_t)read_len, 'N'), bam_qual((size_t)read_len, '\0');
    if (aln.rev) {
        for (int i = 0; i < read_len; ++i) {
            const char c = seq[(size_t)(read_len - 1 - i)];
            switch (c) {
                case 'A': case 'a': bam_seq[(size_t)i] = 'T'; break;
                case 'T': case 't': bam_seq[(size_t)i] = 'A'; break;
                case 'C': case 'c': bam_seq[(size_t)i] = 'G'; break;
                case 'G': case 'g': bam_seq[(size_t)i] = 'C'; break;
                default: bam_seq[(size_t)i] = 'N'; break;
            }
        }
        for (int i = 0; i < read_len; ++i) bam_qual[(size_t)i] = (char)(qual[(size_t)(read_len - 1 - i)] - 33);
    } else {
        bam_seq = seq.substr(0, (size_t)read_len);
        for (int i = 0; i < read_len; ++i) bam_qual[(size_t)i] = (char)(qual[(size_t)i] - 33);
    }
    int32_t tlen = 0, mtid = -1, mpos = -1;
    if (is_paired) {
        tlen = compute_tlen(this_rs, this_re, effective_rev, mate_rs, mate_re, mate_rev);
        mtid = 0;
        mpos = mate_pos;
    }
    // reference length of the CIGAR (bam_cigar2rqlens): M, D, N, =, X consume the reference
    int64_t rlen = 0;
    if (!(flag & BAM_FUNMAP))
        for (uint32_t c : cigar) {
            const uint32_t op = c & 0xf;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += c >> 4;
        }
    if (rlen == 0) rlen = 1;
    const int32_t pos = aln.rs;

    Rec r;
    r.sort_pos = aln.pos;
    r.pos = pos;
    r.end = (int32_t)(pos + rlen);
    std::string& b = r.bytes;
    const uint32_t l_read_name = (uint32_t)qname.size() + 1;
    const uint32_t block_size = 32 + l_read_name + 4 * (uint32_t)cigar.size() + (uint32_t)((read_len + 1) / 2) + (uint32_t)read_len;
    put32(b, block_size);
    put32(b, 0);                                          // refID
    put32(b, (uint32_t)pos);
    b.push_back((char)l_read_name);
    b.push_back((char)aln.mapq);
    put16(b, (uint16_t)reg2bin(pos, pos + rlen));
    put16(b, (uint16_t)cigar.size());
    put16(b, flag);
    put32(b, (uint32_t)read_len);
    put32(b, (uint32_t)mtid);
    put32(b, (uint32_t)mpos);
    put32(b, (uint32_t)tlen);
    b.append(qname);
    b.push_back('\0');
    for (uint32_t c : cigar) put32(b, c);
    for (int i = 0; i < read_len; i += 2) {
        const uint8_t hi = nt16(bam_seq[(size_t)i]), lo = i + 1 < read_len ? nt16(bam_seq[(size_t)i + 1]) : 0;
        b.push_back((char)(hi << 4 | lo));
    }
    b.append(bam_qual);
    return r;
}

// ----------------------------------------------------------------------------------------------- BGZF
// A BGZF file is a sequence of independent gzip members of at most 64 KiB of payload (SAM spec 4.1): the blocks are laid
// out first (which bytes go into which block: a cheap serial pass that also fixes every record's virtual offsets up to
// the blocks' file positions), compressed by a pool of threads, and written in order.  htslib compresses with a pool of
// workers as well (bgzf_mt, which samtools drives through the reference's `-@ threads`).
constexpr size_t kBgzfBlock = 0xff00;   // htslib's BGZF_BLOCK_SIZE

void deflate_block(const std::string& in, std::string& out) {
    out.resize(in.size() + 1024);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2");
    zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(in.data()));
    zs.avail_in = (uInt)in.size();
    zs.next_out = reinterpret_cast<Bytef*>(&out[18]);
    zs.avail_out = (uInt)(out.size() - 18 - 8);
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); throw std::runtime_error("deflate"); }
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    const size_t bsize = clen + 18 + 8;
    const unsigned char hdr[18] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, (unsigned char)((bsize - 1) & 0xff),
                                   (unsigned char)((bsize - 1) >> 8)};
    memcpy(&out[0], hdr, 18);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef*>(in.data()), (uInt)in.size());
    const uint32_t isize = (uint32_t)in.size();
    memcpy(&out[18 + clen], &crc, 4);
    memcpy(&out[18 + clen + 4], &isize, 4);
    out.resize(bsize);
}

// the payload of the blocks, in file order, and where the next byte would go
struct BlockLayout {
    std::vector<std::string> blocks;   // closed blocks
    std::string open;                  // the block being filled
    struct Pos { size_t block; uint32_t off; };
    Pos tell() const { return Pos{blocks.size(), (uint32_t)open.size()}; }
    void close_block() {
        if (open.empty()) return;
        blocks.emplace_back();
        blocks.back().swap(open);
    }
    void write(const char* p, size_t n) {
        while (n > 0) {
            const size_t room = kBgzfBlock - open.size(), k = std::min(room, n);
            open.append(p, k);
            p += k;
            n -= k;
            if (open.size() == kBgzfBlock) close_block();
        }
    }
    // a record stays inside one block when it fits (as htslib's bgzf_write does for BAM records), so that the virtual
    // offset of a record start never points at a block boundary mid-record; returns where the record starts
    Pos begin_record(size_t rec_size) {
        if (open.size() + rec_size > kBgzfBlock && !open.empty() && rec_size <= kBgzfBlock) close_block();
        return tell();
    }
};

unsigned worker_count(size_t n_items, size_t per_thread) {
    unsigned n = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = pmx::opt_str(pmx::O_BAM_THREADS)) n = (unsigned)std::max(1, atoi(e));
    return (unsigned)std::max<size_t>(1, std::min<size_t>(n, n_items / std::max<size_t>(per_thread, 1) + 1));
}

template <class F>
void parallel_for(size_t n, size_t grain, F&& body) {   // body(begin, end) over [0, n) in pieces of `grain`, dynamic
    const unsigned n_thr = worker_count(n, grain);
    if (n_thr <= 1) { body((size_t)0, n); return; }
    std::atomic<size_t> next{0};
    std::vector<std::string> errs(n_thr);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_thr; ++t)
        pool.emplace_back([&, t]() {
            try {
                for (;;) {
                    const size_t b = next.fetch_add(grain);
                    if (b >= n) break;
                    body(b, std::min(n, b + grain));
                }
            } catch (const std::exception& e) { errs[t] = e.what(); if (errs[t].empty()) errs[t] = "worker failed"; }
        });
    for (auto& th : pool) th.join();
    for (const std::string& e : errs)
        if (!e.empty()) throw std::runtime_error(e);
}

}  // namespace

int write_bam(const std::string& bam_path, const std::string& ref_name, int64_t ref_len, const std::vector<std::string>& seqs,
              const std::vector<std::string>& quals, const std::vector<std::string>& names, const align_pair_result_t* results, int64_t n_results,
              bool paired, bool write_index) {
    // ---- records of the mapped results, in input order (built by the worker pool: two slots per pair)
    const size_t per = paired ? 2 : 1;
    std::vector<Rec> slots((size_t)n_results * per);
    std::vector<uint8_t> used((size_t)n_results, 0);
    parallel_for((size_t)n_results, 4096, [&](size_t b, size_t e) {
        for (size_t k = b; k < e; ++k) {
            const align_pair_result_t& res = results[k];
            if (!res.mapped) continue;
            used[k] = 1;
            if (paired) {
                const size_t i1 = 2 * k, i2 = i1 + 1;
                slots[i1] = build_record(names[i1], seqs[i1], quals[i1], res.r1, (int)seqs[i1].size(), true, true, (uint8_t)!res.r2.rev, res.r2.rs,
                                         res.r1.rs, res.r1.re, res.r2.rs, res.r2.re, res.r1.proper_frag, false);
                slots[i2] = build_record(names[i2], seqs[i2], quals[i2], res.r2, (int)seqs[i2].size(), true, false, res.r1.rev, res.r1.rs, res.r2.rs,
                                         res.r2.re, res.r1.rs, res.r1.re, res.r2.proper_frag, false);
            } else {
                slots[k] = build_record(names[k], seqs[k], quals[k], res.r1, (int)seqs[k].size(), false, false, 0, -1, 0, 0, 0, 0, 0, false);
            }
        }
    });
    std::vector<Rec> recs;
    recs.reserve(slots.size());
    for (size_t k = 0; k < (size_t)n_results; ++k)
        if (used[k])
            for (size_t q = 0; q < per; ++q) recs.push_back(std::move(slots[k * per + q]));
    slots.clear();
    // the reference sorts (pos, record) pairs with std::sort on pos only (src/conversion.cpp:499)
    std::vector<std::pair<int32_t, size_t>> order(recs.size());
    for (size_t i = 0; i < recs.size(); ++i) order[i] = {recs[i].sort_pos, i};
    std::sort(order.begin(), order.end(), [](const std::pair<int32_t, size_t>& a, const std::pair<int32_t, size_t>& b) { return a.first < b.first; });

    const std::string text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:" + ref_name + "\tLN:" + std::to_string(ref_len) + "\n";
    std::string hdr("BAM\1", 4);
    put32(hdr, (uint32_t)text.size());
    hdr += text;
    put32(hdr, 1);
    put32(hdr, (uint32_t)ref_name.size() + 1);
    hdr += ref_name;
    hdr.push_back('\0');
    put32(hdr, (uint32_t)ref_len);

    // ---- block layout + where every record begins and ends
    BlockLayout lay;
    lay.write(hdr.data(), hdr.size());
    lay.close_block();   // records start in their own block, as sam_hdr_write leaves them
    std::vector<BlockLayout::Pos> rec_beg(order.size()), rec_end(order.size());
    for (size_t i = 0; i < order.size(); ++i) {
        const Rec& r = recs[order[i].second];
        rec_beg[i] = lay.begin_record(r.bytes.size());
        lay.write(r.bytes.data(), r.bytes.size());
        rec_end[i] = lay.tell();
    }
    lay.close_block();
    // ---- compress (worker pool), place, write
    std::vector<std::string> packed(lay.blocks.size());
    parallel_for(lay.blocks.size(), 8, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) { deflate_block(lay.blocks[i], packed[i]); std::string().swap(lay.blocks[i]); }
    });
    std::vector<uint64_t> file_off(packed.size() + 1, 0);
    for (size_t i = 0; i < packed.size(); ++i) file_off[i + 1] = file_off[i] + packed[i].size();
    auto voff = [&](const BlockLayout::Pos& p) { return file_off[p.block] << 16 | (uint64_t)p.off; };
    {
        FILE* f = fopen(bam_path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot open " + bam_path);
        bool ok = true;
        for (const std::string& b : packed) ok = ok && fwrite(b.data(), 1, b.size(), f) == b.size();
        static const unsigned char eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        ok = ok && fwrite(eof, 1, 28, f) == 28;
        if (fclose(f) != 0 || !ok) throw std::runtime_error("BAM write failed");
    }

    // BAI (SAM spec 5.2): bins -> chunks of virtual offsets, 16 kb linear index, htslib's metadata pseudo-bin
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> linear;
    uint64_t n_mapped = 0, off_beg = 0, off_end = 0;
    int64_t last_bin = -1;
    for (size_t i = 0; i < order.size(); ++i) {
        const Rec& r = recs[order[i].second];
        const uint64_t vbeg = voff(rec_beg[i]), vend = voff(rec_end[i]);
        if (n_mapped == 0) off_beg = vbeg;
        off_end = vend;
        ++n_mapped;
        const uint32_t bin = (uint32_t)reg2bin(r.pos, r.end);
        auto& chunks = bins[bin];
        if ((int64_t)bin == last_bin && !chunks.empty()) chunks.back().second = vend;   // consecutive records of one bin: one chunk
        else chunks.emplace_back(vbeg, vend);
        last_bin = bin;
        const size_t w0 = (size_t)(r.pos >> 14), w1 = (size_t)((r.end - 1) >> 14);
        if (linear.size() <= w1) linear.resize(w1 + 1, UINT64_MAX);
        for (size_t wi = w0; wi <= w1; ++wi)
            if (linear[wi] == UINT64_MAX || vbeg < linear[wi]) linear[wi] = vbeg;
    }
    if (!write_index) return 0;
    for (size_t i = linear.size(); i-- > 1;)
        if (linear[i - 1] == UINT64_MAX) linear[i - 1] = linear[i];   // windows without a start inherit the next one (hts_idx_finish)
    std::string bai("BAI\1", 4);
    put32(bai, 1);
    put32(bai, (uint32_t)(bins.size() + (n_mapped ? 1 : 0)));
    for (const auto& kv : bins) {
        put32(bai, kv.first);
        put32(bai, (uint32_t)kv.second.size());
        for (const auto& c : kv.second) { put64(bai, c.first); put64(bai, c.second); }
    }
    if (n_mapped) {   // htslib's metadata pseudo-bin
        put32(bai, 37450);
        put32(bai, 2);
        put64(bai, off_beg); put64(bai, off_end);
        put64(bai, n_mapped); put64(bai, 0);
    }
    put32(bai, (uint32_t)linear.size());
    for (uint64_t v : linear) put64(bai, v);
    put64(bai, 0);   // n_no_coor
    FILE* f = fopen((bam_path + ".bai").c_str(), "wb");
    if (!f) return 1;
    const bool ok = fwrite(bai.data(), 1, bai.size(), f) == bai.size();
    return (fclose(f) == 0 && ok) ? 0 : 1;
}

}  // namespace pmx
