#!/usr/bin/env python3
"""Condense the counter CSVs written by tools/pmc_passes.sh into one JSON summary.

usage: tools/pmc_summary.py <passes dir> <frames> <algorithmic bytes per frame> <out.json> [command text]

Counter units on gfx950, as used below: FETCH_SIZE / WRITE_SIZE are in KB, and FETCH_SIZE counts a 128-byte
read request as 64 bytes (doubled here); SQ_ACTIVE_INST_* tick once per 4 clocks per SIMD (1024 SIMDs);
SQ_LDS_IDX_ACTIVE is in clocks per CU (256 CUs); GRBM_GUI_ACTIVE is summed over the 8 XCDs.
"""
import collections
import csv
import glob
import json
import sys


def main():
    out_dir, frames, bytes_per_frame, dest = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    command = sys.argv[5] if len(sys.argv) > 5 else ""
    total = collections.defaultdict(float)
    launches = collections.Counter()
    kernel = None
    for path in glob.glob(out_dir + "/pass*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            if "planes" in name or "minsum" in name or "regular" in name or "heavy" in name or "spa" in name:
                kernel = name
                total[row["Counter_Name"]] += float(row["Counter_Value"])
                launches[row["Counter_Name"]] += 1
    n = max(launches.values())
    c = {k: v / launches[k] for k, v in sorted(total.items())}       # per launch
    rd = 2.0 * c["FETCH_SIZE"] * 1024.0
    wr = c["WRITE_SIZE"] * 1024.0
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    algo = frames * bytes_per_frame
    derived = {
        "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": (rd + wr) / algo,
        "gpu_cycles_per_xcd": cyc,
        "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc,
        "lds_busy_frac": c["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc,
        "lds_bank_conflict_frac_of_lds_cycles": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
        "valu_wave_instructions_per_frame": c["SQ_INSTS_VALU"] / frames,
        "lds_wave_instructions_per_frame": c["SQ_INSTS_LDS"] / frames,
        "launches_per_pass": n,
    }
    json.dump({"command": command, "kernel": kernel, "counters": c, "derived": derived,
               "notes": __doc__.split("Counter units", 1)[1].strip()}, open(dest, "w"), indent=1)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
