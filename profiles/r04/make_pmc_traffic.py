#!/usr/bin/env python3
"""Turns the rocprofv3 PMC passes of tools/profile_r04_pmc.sh (FETCH_SIZE, WRITE_SIZE and the SQ counters, each collected in
its OWN run as the MI355X guide prescribes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2, they do not fit one pass) into
profiles/r04/pmc_traffic.json -- the per-launch HBM traffic and wave-instruction counts that bench.py quotes as
roofline.traffic / roofline_valu, but only while `source_digest` equals the digest of the sources it runs on.

usage: make_pmc_traffic.py <FETCH_SIZE pass dir> <WRITE_SIZE pass dir> <SQ pass dir> <reads_per_gpu> <read_len> <out.json>

Units / corrections (MI355X_MICROARCH.md, "HBM [CDNA4]"): both size counters are in units of 1024 B as rocprofv3 prints them
(FETCH_SIZE = TCC_EA0_RDREQ x 64 B / 1024); on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B, so it is DOUBLED;
WRITE_SIZE is taken as is.  The corrected figure is an upper bound for kernels whose accesses are scattered 4..16-byte
granules; the uncorrected one is kept next to it."""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def per_launch(d, counter, kernel_re):
    """d: a rocprofv3 output directory, or one of the filtered pmc_*.csv copies kept under profiles/; kernel_re: a regular
    expression on the kernel name (k_align_compact16 must not swallow k_align_compact16_multi)"""
    vals = {}
    files = [d] if os.path.isfile(d) else glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and re.search(kernel_re, r["Kernel_Name"]):
                vals[int(r["Dispatch_Id"])] = vals.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [vals[k] for k in sorted(vals)]


def main():
    fetch_dir, write_dir, sq_dir, reads, read_len, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    import bench
    res = {"reads_per_gpu": reads, "read_len": read_len, "source_digest": bench.source_digest(), "source_digest_place": bench.source_digest_place(),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (separate passes), tools/profile_r04_pmc.sh"}
    for name, sub in (("dominant_kernel", r"k_align_compact(16|32)\("), ("k_align_compact_multi", r"k_align_compact(16|32)_multi"), ("k_compact_seeds", "k_compact_seeds"), ("k_align_reads_tpp", "k_align_reads_tpp"), ("k_align_reads_t1", "k_align_reads_t1"),
                      ("k_align_dp_group", "k_align_dp_group"), ("k_seed_histogram", "k_seed_histogram"), ("k_collapse_reads", "k_collapse_reads"),
                      ("k_score_chains", "k_score_chains"), ("k_score_terms", "k_score_terms"), ("k_table_compact", "k_table_compact"), ("k_pack_reads", "k_pack_reads")):
        f = per_launch(fetch_dir, "FETCH_SIZE", sub)
        w = per_launch(write_dir, "WRITE_SIZE", sub)
        if not f or not w:
            continue
        if name in ("k_align_reads_tpp", "k_seed_histogram"):      # several launches per step: the per-step sum over the largest ones is not
            pass                                                    # meaningful per launch; the mean is what is reported
        fetch_kb, write_kb = sum(f) / len(f), sum(w) / len(w)
        ent = {"name": "k_align_compact16 (all pairs, one launch per step)" if name == "dominant_kernel" else name, "launches_seen": len(f),
               "FETCH_SIZE_raw_bytes": fetch_kb * 1024, "WRITE_SIZE_raw_bytes": write_kb * 1024,
               "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024, "hbm_bytes_per_launch_uncorrected": fetch_kb * 1024 + write_kb * 1024}
        valu, salu = per_launch(sq_dir, "SQ_INSTS_VALU", sub), per_launch(sq_dir, "SQ_INSTS_SALU", sub)
        lds = per_launch(sq_dir, "SQ_INSTS_LDS", sub)
        wc, wa = per_launch(sq_dir, "SQ_WAVE_CYCLES", sub), per_launch(sq_dir, "SQ_WAIT_ANY", sub)
        if valu:
            ent["valu_wave_insts_per_launch"] = sum(valu) / len(valu)
            ent["salu_wave_insts_per_launch"] = sum(salu) / len(salu) if salu else None
            ent["lds_wave_insts_per_launch"] = sum(lds) / len(lds) if lds else None
            ent["wave_cycles_per_launch"] = sum(wc) / len(wc) if wc else None
            ent["wait_any_share"] = (sum(wa) / sum(wc)) if wc and wa and sum(wc) > 0 else None
        res[name] = ent
    # the whole seeding stage per STEP (bench.py roofline_by_stage["seed"].traffic): every launch of its kernels in one step
    steps = max(res["k_pack_reads"]["launches_seen"], 1) if "k_pack_reads" in res else 1
    tot = 0.0
    for key in ("k_seed_histogram", "k_collapse_reads"):
        if key in res:
            tot += res[key]["hbm_bytes_per_launch"] * res[key]["launches_seen"] / steps
    res["seed_stage"] = {"hbm_bytes_per_step": tot, "kernels": "k_collapse_reads + k_seed_histogram_ks (all launches of a step)", "steps_seen": steps}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
