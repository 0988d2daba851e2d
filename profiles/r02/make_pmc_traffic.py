#!/usr/bin/env python3
"""Turns the two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as the MI355X
guide prescribes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2, they do not fit one pass) into
profiles/r02/pmc_traffic.json, the per-launch HBM traffic bench.py reports as roofline.traffic.

usage: make_pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <reads_per_gpu> <read_len> <out.json>

Units / corrections (MI355X_MICROARCH.md, "HBM [CDNA4]"): both counters are in KiB-like units of 1024 B as
rocprofv3 prints them (FETCH_SIZE = TCC_EA0_RDREQ x 64 B / 1024); on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 B, so it is DOUBLED; WRITE_SIZE is taken as is.  The access pattern of this kernel (scattered
16-byte and 4-byte granules) is not the calibrated wide streaming read, so the corrected figure is an upper
bound of the fetched bytes; the uncorrected one is kept next to it."""
import csv
import glob
import json
import sys


def per_launch(d, counter, kernel_substr):
    vals = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]:
                vals[int(r["Dispatch_Id"])] = vals.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [vals[k] for k in sorted(vals)]


def main():
    fetch_dir, write_dir, reads, read_len, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    res = {"reads_per_gpu": reads, "read_len": read_len, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)"}
    for name, sub in (("dominant_kernel", "k_align_compact"), ("k_align_reads_tpp_bails", "k_align_reads_tpp"), ("k_align_reads_t1_bails", "k_align_reads_t1"),
                      ("k_seed_histogram", "k_seed_histogram"),
                      ("k_score_chains", "k_score_chains")):
        f = per_launch(fetch_dir, "FETCH_SIZE", sub)
        w = per_launch(write_dir, "WRITE_SIZE", sub)
        if not f or not w:
            continue
        if name == "k_align_reads_tpp_bails":      # the bail launch is the largest dispatch of that kernel in each step
            f, w = [max(f)], [max(w)]
        fetch_kb, write_kb = sum(f) / len(f), sum(w) / len(w)
        res[name] = {"name": sub + ("16 (all pairs, one launch per step)" if name == "dominant_kernel" else ""), "launches_seen": len(f), "FETCH_SIZE_raw_bytes": fetch_kb * 1024, "WRITE_SIZE_raw_bytes": write_kb * 1024,
                     "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
                     "hbm_bytes_per_launch_uncorrected": fetch_kb * 1024 + write_kb * 1024}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
