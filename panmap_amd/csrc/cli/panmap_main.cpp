// `panmap <panman> [reads1] [reads2] [options]` -- the reference's command line (src/main.cpp:1941-2131, 2225-2276) for
// the stages this library implements, written against the C ABI only (include/panmap_amd.h):
//   index   PanMAN -> seed index, cached as `<panman>.idx` (reused when newer than the PanMAN and built with the same
//           seeding parameters, src/main.cpp:371-396; -f rebuilds; -i loads a given file; --index-out names the output)
//   place   reads -> `<prefix>.placement.tsv` (src/placement.cpp:1952-2003)
//   align   placed genome -> `<prefix>.ref.fa` (+ .fai), reads aligned to it -> `<prefix>.bam` (+ .bai)
// `--stop index|place|align` ends after that stage.  The later stages of the reference (genotype, consensus), --meta,
// the bwa backend and HPC seeds are outside this library: asking for them is an error, not a silent
// no-op (a `--stop` beyond align stops after align with a note).  Output prefix: -o, else derived from reads1 as the
// reference derives it.  Exit code 130 on SIGINT.
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <thread>
#include <algorithm>
#include <chrono>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "panmap_amd.h"

namespace {

struct Config {
    std::string panman, reads1, reads2, output, index, index_out, stop = "consensus", aligner = "minimap2", batch;
    int threads = 1, k = 19, s = 8, t = 0, l = 3, flank_mask = 250, zstd_level = 7;
    bool open_syncmer = false, hpc = false, force_reindex = false, index_uncompressed = false, quiet = false;
    double seed_mask_fraction = 0.0;
    bool dedup = false, force_leaf = false;
    int trim_start = 0, trim_end = 0, min_seed_quality = 0, min_read_support = -1;
    bool meta = false;     // --meta: haplotype deconvolution (src/main.cpp:1192-1313)
    int64_t top_oc = 1000;
    double em_convergence = 0.00001, em_delta = 0.0, discard = 0.0, dust = 100.0;
    int em_max_iterations = 1000, em_max_rounds = 5;
    int gpus = 1;          // --gpus N: one process per GPU, reads sharded, RCCL exchange (pmx_dist_*)
    bool refine = false;   // src/main.cpp:186-190, 2002-2011
    double refine_top_pct = 0.01;
    int refine_max_top_n = 150, refine_neighbor_radius = 2, refine_max_neighbor_n = 150;
};

void on_sigint(int) { _exit(130); }

struct Fatal { std::string msg; int code; };
[[noreturn]] void die(const std::string& msg, int code = 1) { throw Fatal{msg, code}; }   // main (or the --batch loop) reports it
void check(int rc, const char* what) {
    if (rc != PMX_OK) die(std::string(what) + ": " + pmx_last_error());
}
void say(const Config& c, const char* stage, const std::string& what) {
    if (!c.quiet) fprintf(stderr, "[%s] %s\n", stage, what.c_str());
}
bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
time_t mtime(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 ? st.st_mtime : 0; }

void usage() {
    fputs("Usage: panmap <panman> [reads1.fq[.gz]] [reads2.fq[.gz]] [options]\n"
          "  -o, --output PREFIX        output prefix (default: derived from reads1)\n"
          "  -t, --threads N            accepted (the GPU owns the parallelism)\n"
          "      --stop STAGE           index|place|align (later stages are not part of this build)\n"
          "      --batch FILE           one sample per line: reads1 [reads2] [prefix]; the index stays resident\n"
          "      --meta                 estimate haplotype abundances of a mixed sample -> <prefix>.mgsr.abundance.out\n"
          "      --top-oc N --em-convergence-threshold F --em-delta-threshold F --em-maximum-iterations N --em-maximum-rounds N --discard F --dust F\n"
          "      --gpus N               N processes, one per GPU: reads sharded, seed index replicated, RCCL exchange\n"
          "      --refine               re-rank the top candidates by aligning the reads against them\n"
          "      --refine-top-pct F / --refine-max-top-n N / --refine-neighbor-radius N / --refine-max-neighbor-n N\n"
          "  -i, --index PATH           load a pre-built index       --index-out PATH   write the built index here\n"
          "  -f, --reindex              force rebuild                --zstd-level N     (default 7)   --index-uncompressed\n"
          "  -k N  -s N  -l N  --offset N  --open-syncmer  --flank-mask N  --seed-mask-fraction F\n"
          "      --dedup --trim-start N --trim-end N --min-seed-quality N --min-read-support N --force-leaf\n"
          "  -a, --aligner minimap2     -q, --quiet   -h, --help   -V, --version\n", stderr);
}

Config parse(int argc, char** argv) {
    Config c;
    std::vector<std::string> pos;
    auto need = [&](int& i) -> const char* { if (i + 1 >= argc) die(std::string("option ") + argv[i] + " needs a value"); return argv[++i]; };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        std::string val;
        const size_t eq = a.rfind("--", 0) == 0 ? a.find('=') : std::string::npos;   // --opt=value
        bool has_val = false;
        if (eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); has_val = true; }
        auto v = [&]() -> std::string { return has_val ? val : std::string(need(i)); };
        if (a == "-h" || a == "--help" || a == "--help-all") { usage(); exit(0); }
        else if (a == "-V" || a == "--version") { puts(pmx_version()); exit(0); }
        else if (a == "-o" || a == "--output") c.output = v();
        else if (a == "-t" || a == "--threads") c.threads = atoi(v().c_str());
        else if (a == "--stop") c.stop = v();
        else if (a == "-i" || a == "--index") c.index = v();
        else if (a == "--index-out") c.index_out = v();
        else if (a == "-f" || a == "--reindex") c.force_reindex = true;
        else if (a == "-k" || a == "--kmer") c.k = atoi(v().c_str());
        else if (a == "-s" || a == "--syncmer") c.s = atoi(v().c_str());
        else if (a == "--offset") c.t = atoi(v().c_str());
        else if (a == "-l" || a == "--lmer") c.l = atoi(v().c_str());
        else if (a == "--open-syncmer") c.open_syncmer = true;
        else if (a == "--hpc") c.hpc = true;
        else if (a == "--flank-mask") c.flank_mask = atoi(v().c_str());
        else if (a == "--seed-mask-fraction") c.seed_mask_fraction = atof(v().c_str());
        else if (a == "--zstd-level") c.zstd_level = atoi(v().c_str());
        else if (a == "--index-uncompressed") c.index_uncompressed = true;
        else if (a == "-a" || a == "--aligner") c.aligner = v();
        else if (a == "--dedup") c.dedup = true;
        else if (a == "--trim-start") c.trim_start = atoi(v().c_str());
        else if (a == "--trim-end") c.trim_end = atoi(v().c_str());
        else if (a == "--min-seed-quality") c.min_seed_quality = atoi(v().c_str());
        else if (a == "--min-read-support") c.min_read_support = atoi(v().c_str());
        else if (a == "--force-leaf") c.force_leaf = true;
        else if (a == "-q" || a == "--quiet") c.quiet = true;
        else if (a == "-v" || a == "--verbose" || a == "--no-color" || a == "--no-progress") {}
        else if (a == "--batch") c.batch = v();
        else if (a == "--gpus") c.gpus = atoi(v().c_str());
        else if (a == "--refine") c.refine = true;
        else if (a == "--refine-top-pct") c.refine_top_pct = atof(v().c_str());
        else if (a == "--refine-max-top-n") c.refine_max_top_n = atoi(v().c_str());
        else if (a == "--refine-neighbor-radius") c.refine_neighbor_radius = atoi(v().c_str());
        else if (a == "--refine-max-neighbor-n") c.refine_max_neighbor_n = atoi(v().c_str());
        else if (a == "--meta") c.meta = true;
        else if (a == "--top-oc") c.top_oc = atoll(v().c_str());
        else if (a == "--em-convergence-threshold") c.em_convergence = atof(v().c_str());
        else if (a == "--em-delta-threshold") c.em_delta = atof(v().c_str());
        else if (a == "--em-maximum-iterations") c.em_max_iterations = atoi(v().c_str());
        else if (a == "--em-maximum-rounds") c.em_max_rounds = atoi(v().c_str());
        else if (a == "--discard") c.discard = atof(v().c_str());
        else if (a == "--dust") c.dust = atof(v().c_str());
        else if (a == "--filter-and-assign" || a == "--impute" || a == "--extent-guard" || a == "--reference-node" ||
                 a == "--dump-sequence" || a == "--dump-all-scores")
            die("option " + a + " belongs to a part of panmap this build does not implement (index / place / align only)");
        else if (a.size() > 1 && a[0] == '-') die("unknown option " + a + " (see --help)");
        else pos.push_back(a);
    }
    if (pos.empty()) { usage(); exit(1); }
    c.panman = pos[0];
    if (pos.size() > 1) c.reads1 = pos[1];
    if (pos.size() > 2) c.reads2 = pos[2];
    if (pos.size() > 3) die("at most two read files");
    return c;
}

// src/main.cpp:2253-2276
std::string derive_prefix(const Config& c) {
    if (c.reads1.empty()) return c.panman;
    std::string stem = c.reads1;
    const size_t slash = stem.find_last_of('/');
    if (slash != std::string::npos) stem = stem.substr(slash + 1);
    const size_t dot = stem.find_last_of('.');
    if (dot != std::string::npos && dot > 0) stem = stem.substr(0, dot);            // fs::path::stem()
    auto strip = [&](std::initializer_list<const char*> sfx) {
        for (const char* s : sfx) {
            const size_t n = strlen(s);
            if (stem.size() > n && stem.compare(stem.size() - n, n, s) == 0) { stem.erase(stem.size() - n); return; }
        }
    };
    strip({"_R1", "_R2", "_1", "_2", ".R1", ".R2", ".1", ".2"});
    strip({".fastq", ".fq"});
    return stem;
}

// readBatchFiles (src/main.cpp:1025-1087): one sample per line, `reads1 [reads2] [prefix]`; '#' starts a comment line; a
// single optional field is reads2 when it looks like a FASTQ, else the output prefix; default prefix = reads1 without its
// directory-independent mate suffix and .fastq / .fq
struct BatchEntry { std::string reads1, reads2, prefix; };
std::vector<BatchEntry> read_batch_file(const std::string& path) {
    std::ifstream in(path);
    if (!in.is_open()) die("Cannot open batch file: " + path);
    std::vector<BatchEntry> out;
    std::string line;
    size_t line_no = 0;
    while (std::getline(in, line)) {
        ++line_no;
        const size_t b = line.find_first_not_of(" \t\r\n"), e2 = line.find_last_not_of(" \t\r\n");
        if (b == std::string::npos) continue;
        line = line.substr(b, e2 - b + 1);
        if (line[0] == '#') continue;
        std::istringstream iss(line);
        BatchEntry e;
        std::string f2, f3;
        iss >> e.reads1 >> f2 >> f3;
        if (e.reads1.empty()) continue;
        if (!f2.empty()) {
            if (!f3.empty()) { e.reads2 = f2; e.prefix = f3; }
            else {
                std::string lower = f2;
                for (char& ch : lower) ch = (char)tolower((unsigned char)ch);
                if (lower.find(".fastq") != std::string::npos || lower.find(".fq") != std::string::npos) e.reads2 = f2;
                else e.prefix = f2;
            }
        }
        if (e.prefix.empty()) {
            const size_t slash = e.reads1.find_last_of('/');
            const std::string dir = slash == std::string::npos ? "" : e.reads1.substr(0, slash);
            std::string stem = slash == std::string::npos ? e.reads1 : e.reads1.substr(slash + 1);
            const size_t dot = stem.find_last_of('.');
            if (dot != std::string::npos && dot > 0) stem = stem.substr(0, dot);
            auto strip = [&](std::initializer_list<const char*> sfx) {
                for (const char* sx : sfx) {
                    const size_t n = strlen(sx);
                    if (stem.size() > n && stem.compare(stem.size() - n, n, sx) == 0) { stem.erase(stem.size() - n); return; }
                }
            };
            strip({"_R1", "_R2", "_1", "_2", ".R1", ".R2"});
            strip({".fastq", ".fq"});
            e.prefix = dir.empty() ? stem : dir + "/" + stem;
        }
        if (!exists(e.reads1)) die("Batch line " + std::to_string(line_no) + ": reads file not found: " + e.reads1);
        if (!e.reads2.empty() && !exists(e.reads2)) die("Batch line " + std::to_string(line_no) + ": reads file not found: " + e.reads2);
        out.push_back(e);
    }
    return out;
}
void mkdirs(const std::string& dir) {   // fs::create_directories
    std::string cur;
    for (size_t i = 0; i <= dir.size(); ++i) {
        if (i == dir.size() || dir[i] == '/') {
            if (!cur.empty() && !exists(cur)) mkdir(cur.c_str(), 0777);
        }
        if (i < dir.size()) cur += dir[i];
    }
}

// cachedIndexUsable (src/main.cpp:371-396)
bool cached_index_usable(const Config& c) {
    if (exists(c.panman) && mtime(c.index) < mtime(c.panman)) {
        fprintf(stderr, "panmap: cached index %s is older than %s; rebuilding.\n", c.index.c_str(), c.panman.c_str());
        return false;
    }
    pmx_index_info h;
    int unc = 0;
    if (pmx_index_read_header(c.index.c_str(), &h, &unc) != PMX_OK) {
        fprintf(stderr, "panmap: cached index %s has no readable param header (old format/corrupt); rebuilding.\n", c.index.c_str());
        return false;
    }
    if (h.k != c.k || h.s != c.s || h.t != c.t || h.l != c.l || (h.hpc != 0) != c.hpc || (h.open_syncmer != 0) != c.open_syncmer) {
        fprintf(stderr, "panmap: cached index %s was built with different seeding parameters (k/s/t/l/hpc/open-syncmer); rebuilding.\n", c.index.c_str());
        return false;
    }
    return true;
}

char comp(char b) {   // seeding::reverseComplement (src/seeding.cpp:271-284): A/C/G/T only, everything else kept
    switch (b) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return b; }
}

}  // namespace

// per-sample resources (freed when the sample is done, whatever way it ends)
struct SampleGuard {
    pmx_ctx* ctx;
    pmx_fastx *f1 = nullptr, *f2 = nullptr;
    pmx_readset* rs = nullptr;
    pmx_aligner* al = nullptr;
    ~SampleGuard() {
        if (al) pmx_aligner_free(ctx, al);
        if (rs) pmx_readset_free(ctx, rs);
        if (f1) pmx_fastx_free(f1);
        if (f2) pmx_fastx_free(f2);
    }
};

// One sample through place (+ --refine) (+ align) against the resident index: c.reads1 / c.reads2 / c.output name it.
// Returns the placed node; throws Fatal.
// With `dist` (--gpus N) this process is one rank: it seeds and aligns ITS contiguous, pair-aligned shard of the reads; the
// histograms are merged over the ranks before the (replicated) scoring, candidate scores of --refine are summed, and the
// records + CIGAR arenas are gathered to rank 0, which writes every output file.
std::string run_sample(const Config& c, int stop, pmx_panman*& pm, pmx_index* idx, pmx_ctx* ctx, pmx_place* pl, int dev, pmx_dist* dist = nullptr) {
    const int rank = dist ? pmx_dist_rank(dist) : 0, world = dist ? pmx_dist_world(dist) : 1;
    const bool writer = rank == 0;
    // ------------------------------------------------------------------------------------------------ reads
    SampleGuard g{ctx};
    pmx_fastx *&f1 = g.f1, *&f2 = g.f2;
    check(pmx_fastx_read(c.reads1.c_str(), &f1), "reading reads1");
    const bool paired = !c.reads2.empty();
    if (paired) check(pmx_fastx_read(c.reads2.c_str(), &f2), "reading reads2");
    const int64_t n1 = pmx_fastx_num_reads(f1), n2 = paired ? pmx_fastx_num_reads(f2) : 0;
    if (paired && n1 != n2) die("File " + c.reads2 + " does not contain the same number of reads as " + c.reads1);   // src/placement.cpp:189-192
    const char *s1, *q1, *nm1, *s2 = nullptr, *q2 = nullptr, *nm2 = nullptr;
    const int64_t *o1, *no1, *o2 = nullptr, *no2 = nullptr;
    check(pmx_fastx_views(f1, &s1, &q1, &o1, &nm1, &no1), "reads1 views");
    if (paired) check(pmx_fastx_views(f2, &s2, &q2, &o2, &nm2, &no2), "reads2 views");
    // interleaved R1, R2 in FASTQ orientation (extractReadSequences: R2 is NOT reverse-complemented for placement)
    const int64_t n_reads = n1 + n2;
    std::string concat, quals;
    std::vector<int64_t> off((size_t)n_reads + 1, 0);
    concat.reserve((size_t)(o1[n1] + (paired ? o2[n2] : 0)));
    quals.reserve(concat.capacity());
    for (int64_t i = 0; i < n1; ++i) {
        off[(size_t)(paired ? 2 * i : i)] = (int64_t)concat.size();
        concat.append(s1 + o1[i], (size_t)(o1[i + 1] - o1[i]));
        quals.append(q1 + o1[i], (size_t)(o1[i + 1] - o1[i]));
        if (paired) {
            off[(size_t)(2 * i + 1)] = (int64_t)concat.size();
            concat.append(s2 + o2[i], (size_t)(o2[i + 1] - o2[i]));
            quals.append(q2 + o2[i], (size_t)(o2[i + 1] - o2[i]));
        }
    }
    off[(size_t)n_reads] = (int64_t)concat.size();
    // this rank's shard: reads [lo, hi), mates together
    const int64_t unit = paired ? 2 : 1, n_units = n_reads / unit;
    const int64_t lo = n_units * rank / world * unit, hi = rank == world - 1 ? n_reads : n_units * (rank + 1) / world * unit;

    // ------------------------------------------------------------------------------------------------ place
    pmx_readset*& rs = g.rs;
    check(pmx_readset_upload(ctx, concat.data(), off.data() + lo, hi - lo, &rs), "uploading the reads");
    check(pmx_readset_pack(ctx, rs), "packing the reads");
    pmx_place_params pp;
    memset(&pp, 0, sizeof(pp));
    pp.seed_mask_fraction = c.seed_mask_fraction; pp.min_read_support = c.min_read_support; pp.trim_start = c.trim_start; pp.trim_end = c.trim_end;
    pp.dedup_reads = c.dedup; pp.force_leaf = c.force_leaf; pp.min_seed_quality = c.min_seed_quality;
    if (c.min_seed_quality > 0) check(pmx_readset_set_qualities(ctx, rs, quals.data() + off[(size_t)lo]), "attaching the qualities");
    pmx_place_result res;
    check(pmx_place_reset(ctx, pl), "place reset");
    if (dist && c.dedup && c.min_seed_quality <= 0) check(pmx_dist_dedup_reads(dist, pl, rs, nullptr), "collapsing duplicate reads over the ranks");
    check(pmx_place_add_reads(ctx, pl, rs, &pp), "seeding the reads");
    if (dist) check(pmx_dist_merge_histograms(dist, pl), "merging the ranks' seed histograms");
    check(pmx_place_score(ctx, pl, &pp, n_reads, &res), "scoring the tree");
    static const char* metric_names[5] = {"log_raw", "log_cosine", "containment", "weighted_containment", "log_containment"};
    // --refine (refineTopCandidates, src/placement.cpp:516-698, called at :1910-1914): every candidate's genome is indexed on the
    // device and the reads -- as extractReadSequences leaves them, mate 2 as sequenced -- are aligned against it
    pmx_refine_result refined;
    memset(&refined, 0, sizeof(refined));
    if (c.refine) {
        if (!pm && pmx_panman_open(c.panman.c_str(), &pm) != PMX_OK) {
            fprintf(stderr, "panmap: warning: Failed to load full tree for refinement, disabling refinement\n");
        } else {
            pmx_index_info info;
            check(pmx_index_get_info(idx, &info), "index info");
            std::vector<double> scores5((size_t)info.n_nodes * 5);
            check(pmx_place_node_outputs(ctx, pl, scores5.data(), nullptr, nullptr), "node scores");
            pmx_refine_params rp;
            memset(&rp, 0, sizeof(rp));
            rp.top_pct = c.refine_top_pct; rp.max_top_n = c.refine_max_top_n; rp.neighbor_radius = c.refine_neighbor_radius; rp.max_neighbor_n = c.refine_max_neighbor_n;
            std::vector<uint32_t> cands((size_t)info.n_nodes);
            const int64_t n_cand = pmx_refine_candidates(pmx_index_parents(idx), info.n_nodes, scores5.data(), res.best_index, &rp, cands.data(), (int64_t)cands.size());
            if (n_cand < 0) die(std::string("refining the placement: ") + pmx_last_error());
            cands.resize((size_t)n_cand);
            // The candidates are independent: a few aligners, each with its own context (stream) and host thread, score them
            // concurrently -- one sample's reads do not fill the GPU.  The first one runs alone (it also computes the read set's
            // locality order, which the others then share read-only).
            std::vector<int64_t> cand_score((size_t)n_cand, 0);
            const int mean_len = (int)(concat.size() / (size_t)std::max<int64_t>(n_reads, 1));
            std::atomic<int64_t> next{1};
            std::atomic<int> failed{0};
            std::string fail_msg;
            std::mutex fail_mu;
            auto score_range = [&](pmx_ctx* wctx, pmx_aligner*& al, int64_t i) {
                // one reconstruction per candidate: the worker's buffer keeps the size of the previous genome (genomes of one
                // tree differ by a few bases) and grows only when the returned length says so
                thread_local std::string genome;
                if (genome.size() < 1024) genome.resize(1024);
                int64_t len = pmx_panman_node_genome(pm, (int64_t)cands[(size_t)i], &genome[0], (int64_t)genome.size());
                if (len > (int64_t)genome.size()) {
                    genome.resize((size_t)len + (size_t)len / 64);
                    len = pmx_panman_node_genome(pm, (int64_t)cands[(size_t)i], &genome[0], (int64_t)genome.size());
                }
                if (len <= 0) { cand_score[(size_t)i] = 0; return PMX_OK; }   // (scoreNodeByAlignment returns 0 for an empty genome, src/placement.cpp:496-499)
                int rc2 = al ? pmx_aligner_set_reference(wctx, al, genome.data(), len, mean_len) : pmx_aligner_create(wctx, genome.data(), len, mean_len, &al);
                if (rc2 != PMX_OK) return rc2;
                return pmx_align_score_reads(wctx, al, rs, paired ? 1 : 0, 0, &cand_score[(size_t)i]);
            };
            pmx_aligner* al0 = nullptr;
            if (n_cand > 0) {
                check(score_range(ctx, al0, 0), "refining the placement");
                check(pmx_ctx_synchronize(ctx), "refining the placement");
                int n_workers = 4;
                if (const char* e = getenv("PMX_REFINE_STREAMS")) n_workers = std::max(1, atoi(e));
                n_workers = (int)std::min<int64_t>(n_workers, std::max<int64_t>(n_cand - 1, 1));
                std::vector<std::thread> pool;
                for (int wk = 0; wk < n_workers; ++wk)
                    pool.emplace_back([&, wk]() {
                        pmx_ctx* wctx = ctx;
                        pmx_aligner* al = wk == 0 ? al0 : nullptr;
                        if (wk > 0 && pmx_ctx_create(dev, &wctx) != PMX_OK) { failed = 1; return; }
                        for (;;) {
                            const int64_t i = next.fetch_add(1);
                            if (i >= n_cand || failed.load()) break;
                            if (score_range(wctx, al, i) != PMX_OK) {
                                std::lock_guard<std::mutex> g(fail_mu);
                                if (!failed.exchange(1)) fail_msg = pmx_last_error();
                                break;
                            }
                        }
                        if (wk > 0) { if (al) pmx_aligner_free(wctx, al); pmx_ctx_destroy(wctx); }
                        else al0 = al;
                    });
                for (auto& t : pool) t.join();
                if (failed.load()) die("refining the placement: " + fail_msg);
            }
            if (al0) pmx_aligner_free(ctx, al0);
            if (dist) check(pmx_dist_sum_i64(dist, cand_score.data(), (int64_t)cand_score.size()), "summing the candidate scores over the ranks");
            struct Lookup { const std::vector<uint32_t>* nodes; const std::vector<int64_t>* scores; } lk{&cands, &cand_score};
            auto lookup = [](void* user, uint32_t node, int64_t* score) -> int {
                const Lookup& l = *(const Lookup*)user;
                const auto it = std::lower_bound(l.nodes->begin(), l.nodes->end(), node);
                if (it == l.nodes->end() || *it != node) return PMX_ERR_ARG;
                *score = (*l.scores)[(size_t)(it - l.nodes->begin())];
                return PMX_OK;
            };
            check(pmx_refine_top_candidates(pmx_index_parents(idx), info.n_nodes, scores5.data(), res.best_index, &rp, lookup, &lk, &refined, nullptr, nullptr, 0),
                  "refining the placement");
            if (!refined.ran) fprintf(stderr, "panmap: warning: Refinement skipped: no nodes with positive scores\n");
            else say(c, "place", "refined against " + std::to_string(refined.n_candidates) + " candidates");
        }
    }
    if (writer) {
        const std::string path = c.output + ".placement.tsv";
        FILE* f = fopen(path.c_str(), "w");
        if (!f) die("cannot write " + path);
        fputs("metric\tscore\tnodes\n", f);
        for (int m = 0; m < 5; ++m) {
            std::string ids;
            if (res.n_tied[m] > 0) {
                std::vector<uint32_t> tied((size_t)res.n_tied[m]);
                check(pmx_place_tied(pl, m, tied.data(), (int64_t)tied.size()), "tied nodes");
                for (size_t j = 0; j < tied.size(); ++j) { if (j) ids += ","; ids += pmx_index_node_id(idx, tied[j]); }
            } else if (res.best_index[m] != UINT32_MAX) ids = pmx_index_node_id(idx, res.best_index[m]);
            fprintf(f, "%s\t%.6f\t%s\n", metric_names[m], res.best_score[m], ids.c_str());
        }
        if (refined.ran)   // src/placement.cpp:1987-2000
            for (int m = 0; m < 5; ++m)
                if (refined.node[m] != UINT32_MAX) fprintf(f, "refined_%s\t%.0f\t%s\n", metric_names[m], (double)refined.score[m], pmx_index_node_id(idx, refined.node[m]));
        fclose(f);
        say(c, "place", path);
    }
    if (res.best_index[4] == UINT32_MAX) die("No placement found");
    const std::string node_id = pmx_index_node_id(idx, res.best_index[4]);
    say(c, "place", node_id + " (log_containment " + std::to_string(res.best_score[4]) + ")");
    if (stop == 1) return node_id;

    // ------------------------------------------------------------------------------------------------ align
    if (!pm) check(pmx_panman_open(c.panman.c_str(), &pm), "opening the PanMAN");
    const int64_t node = pmx_panman_find_node(pm, node_id.c_str());
    if (node < 0) die("placed node '" + node_id + "' is not in the PanMAN");
    std::string genome((size_t)pmx_panman_node_genome(pm, node, nullptr, 0), '\0');
    pmx_panman_node_genome(pm, node, &genome[0], (int64_t)genome.size());
    if (genome.empty()) die("Empty sequence for node '" + node_id + "', cannot align");
    if (writer) {
        const std::string fa = c.output + ".ref.fa";
        FILE* f = fopen(fa.c_str(), "w");
        if (!f) die("Cannot write reference file: " + fa);
        fprintf(f, ">%s\n%s\n", node_id.c_str(), genome.c_str());
        fclose(f);
        FILE* g = fopen((fa + ".fai").c_str(), "w");   // faidx: name, length, offset of the first base, bases per line, bytes per line
        if (g) { fprintf(g, "%s\t%zu\t%zu\t%zu\t%zu\n", node_id.c_str(), genome.size(), node_id.size() + 2, genome.size(), genome.size() + 1); fclose(g); }
        say(c, "align", fa);
    }
    pmx_aligner*& al = g.al;
    check(pmx_aligner_create(ctx, genome.data(), (int64_t)genome.size(), (int)(concat.size() / (size_t)std::max<int64_t>(n_reads, 1)), &al), "indexing the placed genome");
    check(pmx_align_readset(ctx, al, rs, paired ? 1 : 0, paired ? 1 : 0), "aligning");   // mate 2 reverse-complemented on the device
    std::vector<pmx_aln_record> recs((size_t)n_reads);
    std::vector<uint32_t> arena;
    if (dist) {
        int64_t g_records = 0, g_words = 0;
        check(pmx_dist_gather_alignments(dist, al, 0, &g_records, &g_words), "gathering the ranks' alignments");
        if (!writer) return node_id;
        if (g_records != n_reads) die("the gathered alignment records do not cover the sample");
        arena.resize((size_t)std::max<int64_t>(g_words, 1));
        check(pmx_dist_fetch_gathered(dist, recs.data(), n_reads, arena.data(), (int64_t)arena.size()), "fetching the gathered alignments");
    } else {
        const int64_t words = pmx_align_cigar_words(ctx, al);
        arena.resize((size_t)std::max<int64_t>(words, 1));
        check(pmx_align_fetch(ctx, al, recs.data(), n_reads, arena.data(), (int64_t)arena.size()), "fetching the alignments");
    }
    // what alignAndWriteBam holds after the aligner call: R2 reverse-complemented, its qualities reversed (src/seeding.cpp:231-269)
    std::vector<std::string> seq_s((size_t)n_reads), qual_s((size_t)n_reads), name_s((size_t)n_reads);
    for (int64_t r = 0; r < n_reads; ++r) {
        const bool second = paired && (r & 1);
        const int64_t i = paired ? r / 2 : r;
        const char* nm = second ? nm2 + no2[i] : nm1 + no1[i];
        name_s[(size_t)r].assign(nm, (size_t)((second ? no2[i + 1] - no2[i] : no1[i + 1] - no1[i])));
        while (!name_s[(size_t)r].empty() && name_s[(size_t)r].back() == '\0') name_s[(size_t)r].pop_back();
        std::string sq(concat, (size_t)off[(size_t)r], (size_t)(off[(size_t)r + 1] - off[(size_t)r]));
        std::string ql(quals, (size_t)off[(size_t)r], sq.size());
        for (char& ch : ql) if (ch == '\0') ch = 'I';   // FASTA input: missing qualities
        if (second) {
            std::string rc(sq.rbegin(), sq.rend());
            for (char& ch : rc) ch = comp(ch);
            sq.swap(rc);
            std::string rq(ql.rbegin(), ql.rend());
            ql.swap(rq);
        }
        seq_s[(size_t)r].swap(sq);
        qual_s[(size_t)r].swap(ql);
    }
    std::vector<const char*> seq_p((size_t)n_reads), qual_p((size_t)n_reads), name_p((size_t)n_reads);
    std::vector<int> lens((size_t)n_reads);
    for (int64_t r = 0; r < n_reads; ++r) { seq_p[(size_t)r] = seq_s[(size_t)r].c_str(); qual_p[(size_t)r] = qual_s[(size_t)r].c_str(); name_p[(size_t)r] = name_s[(size_t)r].c_str(); lens[(size_t)r] = (int)seq_s[(size_t)r].size(); }
    const int64_t n_items = paired ? n_reads / 2 : n_reads;
    std::vector<align_pair_result_t> results((size_t)std::max<int64_t>(n_items, 1));
    int64_t n_mapped = 0, n_withheld = 0;
    auto fill = [&](const pmx_aln_record& r, read_align_t* o) {
        memset(o, 0, sizeof(*o));
        if (r.mapped && (r.flags & PMX_ALN_HAS_ALN)) {
            o->pos = r.rs + 1; o->rs = r.rs; o->re = r.re; o->qs = r.qs; o->qe = r.qe;
            o->mapq = r.mapq; o->rev = r.rev; o->proper_frag = r.proper_frag; o->n_cigar = r.n_cigar;
            o->cigar = arena.data() + r.cigar_off;   // (points into the arena: nothing to free)
        } else o->pos = INT_MAX;
    };
    for (int64_t k = 0; k < n_items; ++k) {
        align_pair_result_t& out = results[(size_t)k];
        memset(&out, 0, sizeof(out));
        const pmx_aln_record& a = recs[(size_t)(paired ? 2 * k : k)];
        const bool invalid = paired ? ((a.flags | recs[(size_t)(2 * k + 1)].flags) & 3) != 0 : (a.flags & 3) != 0;
        out.r1.pos = out.r2.pos = INT_MAX;
        if (invalid) { ++n_withheld; continue; }
        if (!a.mapped) continue;
        out.mapped = 1;
        ++n_mapped;
        fill(a, &out.r1);
        if (paired) fill(recs[(size_t)(2 * k + 1)], &out.r2);
    }
    const std::string bam = c.output + ".bam";
    check(pmx_write_bam(bam.c_str(), node_id.c_str(), (int64_t)genome.size(), (int)n_reads, seq_p.data(), qual_p.data(), name_p.data(), lens.data(),
                        results.data(), paired), "writing the BAM");
    say(c, "align", bam + " (" + std::to_string(n_mapped) + " of " + std::to_string(n_items) + (paired ? " pairs" : " reads") + " mapped" +
                    (n_withheld ? ", " + std::to_string(n_withheld) + " invalid records withheld" : "") + ")");
    return node_id;
}

// --meta (runDeconvolution, src/main.cpp:1192-1313): reads of a mixed sample -> `<prefix>.mgsr.abundance.out`, one line per
// estimated haplotype: node id (+ the nodes merged into it, comma-joined) <TAB> proportion with five decimals, by proportion.
// The two indexes of the tree (the place stage's and its oriented form, without flank mask) are built in memory.
int run_meta(const Config& c) {
    if (c.reads1.empty()) die("--meta needs reads");
    if (c.discard < 0.0 || c.discard > 1.0) die("--discard must be between 0 and 1");   // src/main.cpp:1358-1361
    if (c.dust > 100.0) die("--dust must be <= 100");                                    // src/main.cpp:1353-1356
    if (c.l < 2) die("--meta needs l >= 2 in this build (the orientation of a lone syncmer is not indexed)");
    if (c.gpus > 1) die("--meta runs on one GPU in this build");
    pmx_panman* pm = nullptr;
    check(pmx_panman_open(c.panman.c_str(), &pm), "opening the PanMAN");
    pmx_index *idx = nullptr, *oidx = nullptr;
    {   // the two builds are independent: side by side
        int rc1 = PMX_OK, rc2 = PMX_OK;
        std::string e2;
        std::thread th([&]() { rc2 = pmx_index_build_ex(pm, c.k, c.s, c.t, c.l, c.open_syncmer ? 1 : 0, 0, PMX_INDEX_ORIENTED, -1, &oidx); if (rc2 != PMX_OK) e2 = pmx_last_error(); });
        rc1 = pmx_index_build_ex(pm, c.k, c.s, c.t, c.l, c.open_syncmer ? 1 : 0, 0, 0, -1, &idx);
        th.join();
        check(rc1, "building the index");
        if (rc2 != PMX_OK) die("building the oriented index: " + e2);
    }
    say(c, "index", "seed index + oriented seed index (in memory)");
    pmx_fastx *f1 = nullptr, *f2 = nullptr;
    check(pmx_fastx_read(c.reads1.c_str(), &f1), "reading reads1");
    if (!c.reads2.empty()) check(pmx_fastx_read(c.reads2.c_str(), &f2), "reading reads2");
    const char *s1, *q1, *nm1;
    const int64_t *o1, *no1;
    check(pmx_fastx_views(f1, &s1, &q1, &o1, &nm1, &no1), "reads1 views");
    std::string concat(s1, (size_t)o1[pmx_fastx_num_reads(f1)]);
    std::vector<int64_t> off(o1, o1 + pmx_fastx_num_reads(f1) + 1);
    if (f2) {   // mates as sequenced, one after the other (the score of a read does not depend on its place in the list)
        const char *s2, *q2, *nm2;
        const int64_t *o2, *no2;
        check(pmx_fastx_views(f2, &s2, &q2, &o2, &nm2, &no2), "reads2 views");
        const int64_t n2 = pmx_fastx_num_reads(f2), base = (int64_t)concat.size();
        concat.append(s2, (size_t)o2[n2]);
        for (int64_t i = 1; i <= n2; ++i) off.push_back(base + o2[i]);
    }
    int dev = 0;
    if (const char* e = getenv("PMX_DEVICE")) dev = atoi(e);
    pmx_ctx* ctx = nullptr;
    check(pmx_ctx_create(dev, &ctx), "opening the GPU");
    pmx_meta* m = nullptr;
    check(pmx_meta_create(ctx, idx, oidx, &m), "uploading the indexes");
    check(pmx_meta_set_dust(m, c.dust), "--dust");
    check(pmx_meta_set_reads(ctx, m, concat.data(), off.data(), (int64_t)off.size() - 1), "seeding the reads");
    check(pmx_meta_score(ctx, m, c.top_oc, nullptr, 0), "scoring the reads against the candidate nodes");
    say(c, "meta", std::to_string(pmx_meta_num_reads(m)) + " distinct reads x " + std::to_string(pmx_meta_num_candidates(m)) + " candidate nodes");
    pmx_meta_params mp;
    memset(&mp, 0, sizeof(mp));
    mp.error_rate = 0.005; mp.em_convergence = c.em_convergence; mp.em_delta_threshold = c.em_delta; mp.prop_threshold = 0.005; mp.discard = c.discard;
    mp.em_max_iterations = c.em_max_iterations; mp.em_max_rounds = c.em_max_rounds;
    check(pmx_meta_em(ctx, m, &mp), "estimating the abundances");
    const std::string path = c.output + ".mgsr.abundance.out";
    FILE* f = fopen(path.c_str(), "w");
    if (!f) die("cannot write " + path);
    const int64_t n_h = pmx_meta_num_haplotypes(m);
    if (n_h == 0) fprintf(stderr, "No reads remain for node scoring and EM after discarding low-score reads... Exiting... \n");   // src/main.cpp:1244-1247
    for (int64_t i = 0; i < n_h; ++i) {
        uint32_t node = 0;
        double prop = 0;
        int64_t n_mem = 0;
        check(pmx_meta_haplotype(m, i, &node, &prop, &n_mem, nullptr, 0), "haplotype");
        std::vector<uint32_t> mem((size_t)std::max<int64_t>(n_mem, 1));
        check(pmx_meta_haplotype(m, i, nullptr, nullptr, nullptr, mem.data(), (int64_t)mem.size()), "haplotype members");
        std::string ids = pmx_index_node_id(idx, node);
        for (int64_t k = 0; k < n_mem; ++k) { ids += ","; ids += pmx_index_node_id(idx, mem[(size_t)k]); }
        fprintf(f, "%s\t%.5f\n", ids.c_str(), prop);
    }
    fclose(f);
    say(c, "meta", path + " (" + std::to_string(n_h) + " haplotypes)");
    pmx_meta_free(ctx, m);
    pmx_ctx_destroy(ctx);
    pmx_fastx_free(f1);
    if (f2) pmx_fastx_free(f2);
    pmx_index_close(idx); pmx_index_close(oidx);
    pmx_panman_close(pm);
    return 0;
}

int real_main(int argc, char** argv) {
    signal(SIGINT, on_sigint);
    Config c = parse(argc, argv);
    if (c.s <= 0 || c.s > c.k) die("Invalid syncmer s=" + std::to_string(c.s) + " (must be in 1..k, k=" + std::to_string(c.k) + ")");
    if (c.t < 0 || c.t > c.k - c.s) die("Invalid syncmer offset=" + std::to_string(c.t) + " (must be in 0..k-s = 0.." + std::to_string(c.k - c.s) + ")");
    if (c.hpc) die("--hpc (homopolymer-compressed seeds) is not implemented in this build");
    if (c.aligner != "minimap2") die("aligner '" + c.aligner + "' is not implemented in this build (minimap2 only)");
    int stop = c.stop == "index" ? 0 : c.stop == "place" ? 1 : c.stop == "align" ? 2 : (c.stop == "genotype" || c.stop == "consensus") ? 3 : -1;
    if (stop < 0) die("--stop expects index|place|align|genotype|consensus");
    if (c.index.empty()) c.index = c.index_out.empty() ? c.panman + ".idx" : c.index_out;
    else if (!exists(c.index)) die("index file not found: " + c.index + " (--index expects a pre-built index; use --index-out to build at a custom path)");
    if (c.output.empty()) c.output = derive_prefix(c);
    if (!c.batch.empty() && !c.reads1.empty()) die("--batch takes the read files from the batch file, not from the command line");
    if (c.gpus < 1) die("--gpus expects a positive number");

    if (c.meta) return run_meta(c);

    // ------------------------------------------------------------------------------------------------ index
    pmx_panman* pm = nullptr;
    pmx_index* idx = nullptr;
    if (exists(c.index) && !c.force_reindex && cached_index_usable(c)) {
        check(pmx_index_load(c.index.c_str(), &idx), "loading the index");
        say(c, "index", c.index + " (cached)");
    } else {
        check(pmx_panman_open(c.panman.c_str(), &pm), "opening the PanMAN");
        check(pmx_index_build(pm, c.k, c.s, c.t, c.l, c.open_syncmer ? 1 : 0, c.flank_mask, &idx), "building the index");
        check(pmx_index_save(idx, c.index.c_str(), c.zstd_level, c.index_uncompressed ? 1 : 0), "writing the index");
        say(c, "index", c.index + " (built)");
    }
    if (stop == 0 || (c.reads1.empty() && c.batch.empty())) return 0;

    // ------------------------------------------------------------------------------------------------ --gpus N
    // One process per GPU, forked HERE: the index (and the PanMAN, when it was opened) is in memory and nothing has touched
    // a GPU yet, so every rank inherits them and opens its own device.  The ranks meet through RCCL (pmx_dist_*); the
    // communicator's id travels through a file in a private directory.  The parent only waits.
    int rank = 0, world = 1;
    std::string meet_dir;
    if (c.gpus > 1) {
        if (!c.batch.empty()) die("--gpus shards ONE sample over the GPUs; run --batch per GPU instead");
        char tmpl[] = "/tmp/panmap_ranks_XXXXXX";
        if (!mkdtemp(tmpl)) die("cannot create a rendezvous directory under /tmp");
        meet_dir = tmpl;
        fflush(stdout); fflush(stderr);
        std::vector<pid_t> kids;
        bool child = false;
        for (int r = 0; r < c.gpus; ++r) {
            const pid_t pid = fork();
            if (pid < 0) die("fork failed");
            if (pid == 0) { child = true; rank = r; world = c.gpus; break; }
            kids.push_back(pid);
        }
        if (!child) {
            // Wait for whichever rank ends first.  A rank that fails (no such device, unreadable reads, any die() between two
            // collectives) leaves the others blocked in ncclCommInitRank / a collective for ever: the first failure ends the
            // run -- the remaining children (and only they) get SIGTERM, are reaped, and its code is returned.
            int worst = 0;
            size_t left = kids.size();
            while (left > 0) {
                int st = 0;
                const pid_t k = waitpid(-1, &st, 0);
                if (k < 0) {
                    if (errno == EINTR) continue;
                    worst = worst ? worst : 1;
                    break;
                }
                auto it = std::find(kids.begin(), kids.end(), k);
                if (it == kids.end()) continue;
                *it = -1;
                --left;
                const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
                if (code && !worst) {
                    worst = code;
                    for (pid_t o : kids) if (o > 0) kill(o, SIGTERM);
                }
            }
            unlink((meet_dir + "/uid.tmp").c_str());
            unlink((meet_dir + "/uid").c_str());
            rmdir(meet_dir.c_str());
            pmx_index_close(idx);
            if (pm) pmx_panman_close(pm);
            return worst;
        }
        if (rank > 0) c.quiet = true;
    }

    // ------------------------------------------------------------------------------------------------ samples
    pmx_ctx* ctx = nullptr;
    int dev = 0;
    if (const char* e = getenv("PMX_DEVICE")) dev = atoi(e);
    if (world > 1 && !getenv("PMX_DIST_SAME_DEVICE")) dev += rank;   // (PMX_DIST_SAME_DEVICE: functional test on a one-GPU box)
    check(pmx_ctx_create(dev, &ctx), "opening the GPU");
    pmx_dist* dist = nullptr;
    if (world > 1) {
        char uid[PMX_DIST_ID_BYTES];
        const std::string uid_path = meet_dir + "/uid";
        if (rank == 0) {
            check(pmx_dist_unique_id(uid), "creating the communicator id");
            FILE* f = fopen((uid_path + ".tmp").c_str(), "wb");
            if (!f || fwrite(uid, 1, sizeof(uid), f) != sizeof(uid)) die("cannot write the communicator id");
            fclose(f);
            if (rename((uid_path + ".tmp").c_str(), uid_path.c_str()) != 0) die("cannot publish the communicator id");
        } else {
            FILE* f = nullptr;
            for (int tries = 0; tries < 600000 && !(f = fopen(uid_path.c_str(), "rb")); ++tries) usleep(500);
            if (!f || fread(uid, 1, sizeof(uid), f) != sizeof(uid)) die("rank 0 never published the communicator id");
            fclose(f);
        }
        check(pmx_dist_init(ctx, uid, rank, world, &dist), "joining the ranks (RCCL)");
    }
    pmx_place* pl = nullptr;
    check(pmx_place_create(ctx, idx, &pl), "uploading the index");
    int rc = 0;
    if (c.batch.empty()) run_sample(c, stop, pm, idx, ctx, pl, dev, dist);
    else {
        // runBatchPlacement (src/main.cpp:1464-1666): the samples of the batch file one after the other against the index
        // that stays on the device; one line per sample on stderr, a failed sample does not stop the batch
        const std::vector<BatchEntry> samples = read_batch_file(c.batch);
        if (samples.empty()) die("No samples found in batch file");
        fprintf(stderr, "Batch mode: %zu samples\n", samples.size());
        int ok = 0, failed = 0;
        for (size_t i = 0; i < samples.size(); ++i) {
            Config sc = c;
            sc.reads1 = samples[i].reads1; sc.reads2 = samples[i].reads2; sc.output = samples[i].prefix; sc.quiet = true;
            const size_t slash = sc.output.find_last_of('/');
            if (slash != std::string::npos && slash > 0) mkdirs(sc.output.substr(0, slash));
            const auto t0 = std::chrono::steady_clock::now();
            std::string node, err;
            try { node = run_sample(sc, stop, pm, idx, ctx, pl, dev); }
            catch (const Fatal& f) { err = f.msg; }
            const long long ms = (long long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (!err.empty()) {
                fprintf(stderr, "[%zu/%zu] %s -> %s (%lldms)\n", i + 1, samples.size(), sc.output.c_str(), err == "No placement found" ? "NO PLACEMENT" : ("failed: " + err).c_str(), ms);
                ++failed;
            } else {
                fprintf(stderr, "[%zu/%zu] %s -> %s (%lldms)\n", i + 1, samples.size(), sc.output.c_str(), node.c_str(), ms);
                ++ok;
            }
        }
        fprintf(stderr, "Batch complete: %d placed, %d failed\n", ok, failed);
        rc = failed ? 1 : 0;
    }
    if (stop > 2 && rank == 0) fprintf(stderr, "panmap: note: stages after align (genotype, consensus) are not part of this build; stopped after align.\n");
    if (dist) { (void)pmx_dist_barrier(dist); pmx_dist_free(dist); }
    pmx_place_free(ctx, pl);
    pmx_ctx_destroy(ctx);
    pmx_index_close(idx);
    if (pm) pmx_panman_close(pm);
    return rc;
}

int main(int argc, char** argv) {
    try {
        return real_main(argc, argv);
    } catch (const Fatal& f) {
        fprintf(stderr, "panmap: error: %s\n", f.msg.c_str());
        return f.code;
    }
}
