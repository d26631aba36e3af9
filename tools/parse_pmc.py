#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/gpu_pmc.sh into profiles/pmc_traffic.json.

Units and corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE / WRITE_SIZE are in KiB and derive from the
L2's memory-side request counters; WRITE_SIZE is exact for 16-B-per-lane streaming stores (ours); FETCH_SIZE
under-reports wide reads by 2x on gfx950 and is "uncalibrated" for other widths, so it is calibrated here on
the RD_BURST=0 run, whose true read volume is known exactly (every u16 CFA sample is loaded once: W*H*2 bytes).
"""
import csv
import glob
import json
import os
import sys

W, H = 6016, 4016
TRUE_READ = W * H * 2


def mean_counter(d, name):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    vals = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "rd_develop_quads" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main(out_dir):
    res = {}
    for burst in (0, 1):
        f, nf = mean_counter(os.path.join(out_dir, f"b{burst}_FETCH_SIZE"), "FETCH_SIZE")
        w, nw = mean_counter(os.path.join(out_dir, f"b{burst}_WRITE_SIZE"), "WRITE_SIZE")
        res[burst] = dict(fetch_kib=f, write_kib=w, n=(nf, nw))
        print(f"RD_BURST={burst}: FETCH_SIZE {f} KiB over {nf} dispatches, WRITE_SIZE {w} KiB over {nw}")
    if res[0]["fetch_kib"] is None or res[0]["write_kib"] is None:
        print("missing counters")
        return 1
    cal = TRUE_READ / (res[0]["fetch_kib"] * 1024.0)          # bytes actually read per reported byte (4-B loads)
    print(f"calibration: true read {TRUE_READ} B / reported {res[0]['fetch_kib'] * 1024:.0f} B = x{cal:.3f}")
    out = {"note": "HBM bytes per rd_develop_quads launch from rocprofv3 PMC (separate FETCH_SIZE / WRITE_SIZE passes); "
                   "FETCH_SIZE calibrated on the RD_BURST=0 run (true read volume W*H*2), WRITE_SIZE exact for 16-B stores",
           "fetch_calibration": cal, "frame": [W, H]}
    for burst, key in ((1, "f32"), (0, "f32_noburst")):
        r = res[burst]
        if r["fetch_kib"] is None or r["write_kib"] is None:
            continue
        rd = r["fetch_kib"] * 1024.0 * cal
        wr = r["write_kib"] * 1024.0
        out[key] = {"read_bytes": round(rd), "write_bytes": round(wr), "hbm_bytes_per_launch": round(rd + wr),
                    "algorithmic_bytes": W * H * 18}
        print(f"{key}: read {rd / 1e6:.1f} MB + write {wr / 1e6:.1f} MB = {(rd + wr) / 1e6:.1f} MB per launch "
              f"(algorithmic {W * H * 18 / 1e6:.1f} MB)")
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    if os.access(os.path.dirname(dst), os.W_OK):
        json.dump(out, open(dst, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
