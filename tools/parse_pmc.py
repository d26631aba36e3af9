#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/gpu_pmc.sh into profiles/pmc_traffic.json + a readable summary.

Units and corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE / WRITE_SIZE are in KiB and derive from the L2's
memory-side (fabric) request counters; Infinity-Cache hits are counted, not excluded.
  * WRITE_SIZE is exact for 16-B-per-lane streaming stores (the f32 and f16 surfaces); the RGBA8 surface stores 8 B per
    lane, which the guide calls uncalibrated -- its figure is given raw, next to the known algorithmic bytes.
  * FETCH_SIZE under-reports on gfx950.  Two load shapes occur here and each gets its own factor:
      - the main loop's 4-B-per-lane CFA loads: calibrated on the run whose read volume is known exactly
        (one launch per frame, RD_BURST=0: every u16 sample is loaded once = W*H*2 bytes) -> cal4 (about 1.8);
      - the f32 kernel's read burst / per-frame sweep, 16-B-per-lane LDS-DMA: x2.0, the guide's figure for wide reads.
    For a run with the burst: reported = W*H*2 / cal4 + burst / 2  =>  burst = 2 * (reported - W*H*2 / cal4).
All figures are per FRAME (a multi-frame launch covers several frames; its counters are divided by that number).
"""
import csv
import glob
import json
import os
import sys


def bench_line(log):
    try:
        for line in open(log):
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    except OSError:
        pass
    return None


def counters(d):
    """mean per-dispatch value of every counter and the mean duration (us) of the export kernel's dispatches"""
    agg, dur = {}, []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rd_develop_batch" in r["Kernel_Name"] or "rd_develop_quads" in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rd_develop_batch" in r["Kernel_Name"] or "rd_develop_quads" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: sum(v) / len(v) for k, v in agg.items()}, (sum(dur) / len(dur) if dur else None), len(dur)


def run(out_dir, name):
    d = os.path.join(out_dir, name)
    b = bench_line(d + ".log")
    if not os.path.isdir(d) or b is None:
        return None
    c, dur, n = counters(d)
    fpl = b["config"]["frames_per_gpu"] / b["config"]["launches_per_step"]
    return {"counters": c, "dur_us": dur, "dispatches": n, "frames_per_launch": fpl,
            "W": b["config"]["width"], "H": b["config"]["height"], "bench_us_per_frame": b["roofline"].get("us_per_frame")}


def main(out_dir, tag):
    res, entries = {}, []
    def per_frame(name, ctr):
        r = run(out_dir, f"{name}_{ctr}")
        if r is None or ctr not in r["counters"]:
            return None, None
        return r["counters"][ctr] * 1024.0 / r["frames_per_launch"], r
    # calibration of the 4-B loads
    f_nb, r_nb = per_frame("f32_perframe_noburst", "FETCH_SIZE")
    w_nb, _ = per_frame("f32_perframe_noburst", "WRITE_SIZE")
    if f_nb is None:
        print("calibration run missing")
        return 1
    true_read = r_nb["W"] * r_nb["H"] * 2
    cal4 = true_read / f_nb
    print(f"calibration (one launch per frame, no burst): true CFA read {true_read} B / FETCH_SIZE {f_nb:.0f} B = x{cal4:.4f} for 4-B loads; "
          f"16-B LDS-DMA reads x2.0 (guide)")
    rows = [("f32_perframe_noburst", "f32", "per_frame_noburst", False), ("f32_perframe", "f32", "per_frame", True),
            ("f32_multi", "f32", "multi", True), ("f16_multi", "f16", "multi", False), ("u8_multi", "u8", "multi", False),
            ("c5_multi", "f16", "multi", False),
            ("f32_ragged_multi", "f32", "multi", True), ("f16_ragged_multi", "f16", "multi", False), ("u8_ragged_multi", "u8", "multi", False)]
    bpp = {"f32": 16, "f16": 8, "u8": 4}
    for name, fmt, mode, burst in rows:
        f, r = per_frame(name, "FETCH_SIZE")
        w, r2 = per_frame(name, "WRITE_SIZE")
        if f is None or w is None:
            print(f"{name}: missing")
            continue
        W, H = r["W"], r["H"]
        cfa = W * H * 2
        if burst:
            sweep = max(0.0, 2.0 * (f - cfa / cal4))
            read = cfa + sweep
        else:
            sweep, read = 0.0, f * cal4
        alg = W * H * (2 + bpp[fmt])
        ent = {"format": fmt, "frame": [W, H], "mode": mode, "frames_per_launch": r["frames_per_launch"],
               "fetch_size_reported_bytes": round(f), "read_bytes": round(read), "sweep_bytes": round(sweep),
               "write_bytes": round(w), "hbm_bytes_per_frame": round(read + w), "algorithmic_bytes": alg,
               "ratio": round((read + w) / alg, 4), "kernel_us_per_frame_profiled": round(r["dur_us"] / r["frames_per_launch"], 2),
               "write_note": "WRITE_SIZE exact (16-B-per-lane stores)" if fmt != "u8" else
                             "WRITE_SIZE raw: 8-B-per-lane stores are uncalibrated on gfx950; algorithmic surface bytes = %d" % (W * H * 4)}
        entries.append(ent)
        print(f"{name:22s} {W}x{H} {fmt}: read {read / 1e6:7.1f} MB (CFA {cfa / 1e6:.1f} + sweep {sweep / 1e6:.1f}) + write {w / 1e6:7.1f} MB "
              f"= {(read + w) / 1e6:7.1f} MB per frame vs {alg / 1e6:.1f} MB algorithmic (x{(read + w) / alg:.3f}); "
              f"profiled kernel {r['dur_us'] / r['frames_per_launch']:.1f} us per frame ({r['frames_per_launch']:g} frames per launch)")
    out = {"tag": tag, "note": __doc__.split("\n\n")[1].replace("\n", " "), "fetch_calibration_4B": cal4, "fetch_factor_16B": 2.0,
           "entries": entries}
    # SQ passes: everything per frame; the SQ counters tick in quad-cycles (MI355X_MICROARCH.md cycle-constants table)
    sq = {}
    for name in ("f32_multi_SQ1", "f32_multi_SQ2", "u8_multi_SQ1", "f16_multi_SQ1"):
        r = run(out_dir, name)
        if r is None:
            continue
        c = {k: v / r["frames_per_launch"] for k, v in r["counters"].items()}
        us = r["dur_us"] / r["frames_per_launch"]
        c["kernel_us_per_frame_profiled"] = us
        if "GRBM_GUI_ACTIVE" in c:                               # summed over the 8 XCDs (guide: DVFS give-back)
            c["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e-6) / 1e9
        sq[name] = {k: (round(v, 3) if v < 1e6 else round(v)) for k, v in c.items()}
        print(f"\n{name}: kernel {us:.1f} us per frame")
        for k in sorted(c):
            print(f"    {k:28s} {c[k]:.6g}")
        if {"SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"} <= set(c):
            wc = c["SQ_WAVE_CYCLES"]
            print(f"    -> of the wave-cycles: issuing {100 * c['SQ_ACTIVE_INST_ANY'] / wc:.1f} % (VALU {100 * c['SQ_ACTIVE_INST_VALU'] / wc:.1f} %), "
                  f"parked in s_waitcnt {100 * c['SQ_WAIT_ANY'] / wc:.1f} %, issue-stalled {100 * c['SQ_WAIT_INST_ANY'] / wc:.1f} %")
        if {"SQ_INSTS_VALU", "SQ_BUSY_CYCLES"} <= set(c) and "effective_clock_GHz" in c:
            # VALU time if it were the only thing running: instructions x 4 quad... a wave-instruction occupies its SIMD
            # for 2 cycles (64 lanes over a SIMD-32); 1024 SIMDs
            valu_us = c["SQ_INSTS_VALU"] * 2.0 / 1024.0 / (c["effective_clock_GHz"] * 1e3)
            print(f"    -> {c['SQ_INSTS_VALU']:.4g} VALU wave-instructions per frame x 2 cycles / 1024 SIMDs at "
                  f"{c['effective_clock_GHz']:.2f} GHz = {valu_us:.1f} us of pure VALU issue per frame")
    out["sq"] = sq
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    if os.access(os.path.dirname(dst), os.W_OK):
        json.dump(out, open(dst, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r02"))
