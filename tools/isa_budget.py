#!/usr/bin/env python3
"""tools/isa_budget.py -- per-stage VALU budget of the export kernel's main loop, from hipcc's own assembly.

    python tools/isa_budget.py [--kernel '<2,true,true,0,false>'] [--listing out.s] [--summary out.txt]

Compiles raweditor_amd/csrc/rawdev.hip for gfx950 with the product's flags plus -gline-tables-only -S (line tables do not
change code generation), takes one instance of rd_develop_batch, finds its main loop (the outermost backward branch that
encloses the most instructions), and attributes every instruction of the loop body to a STAGE of the pipeline by the
source line its .loc directive names (innermost inlined frame: rd_kernels.h / rd_math.h).  Each VALU instruction is
priced with the issue costs measured on this part by tools/valu_probe2.hip (gpurun_out/pk/valu3.txt; DESIGN.md 6b "VALU
calibration"), in cycles per wave-instruction per SIMD:

    2  "full rate": v_fma / v_fmac / v_fmaak / v_fmamk / v_mul / v_add / v_sub f32 and the simple integer VOP2s
       (v_and / v_or / v_xor / v_add_u32 / v_sub_u32 / shifts / v_mov), all sources VGPR, literal or inline constant
    4  "half rate": the same with an SGPR source; v_max / v_min (f32, u32, 3-operand), v_cvt_*, v_rndne, v_fract,
       v_cmp_*, v_cndmask, v_lshl_add / v_lshl_or / v_and_or / v_perm / v_bfe / v_mad_u32 (VOP3 integer), v_div_fixup,
       v_lshlrev_b32, every SDWA form, and the packed-f32 v_pk_mul / v_pk_add / v_pk_fma (two lane-operations in four
       cycles: NO gain over two full-rate scalar instructions on this part -- profiles/r03_valu_probe.txt)
    8  "quarter rate": v_exp_f32 / v_log_f32 / v_rcp_f32 / v_rsq / v_sqrt

Branch-conditional code inside the loop (the rare pinned-gamma fallback, the per-frame descriptor reload) is listed
separately and NOT charged to the per-tile budget: `--include-cold` shows it.  The hot / cold split is by basic block:
a block is cold when it is only reachable through a branch the source marks as rare (fallback lines of rd_q8_gamma /
rd_f16_gamma, adopt(), prefetch_frame()).
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

# ISA_BUDGET_ROOT: another checkout of the sources (e.g. the previous round's, for a before / after pair)
ROOT = os.environ.get("ISA_BUDGET_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raweditor_amd", "csrc", "rawdev.hip")
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17",
         "--cuda-device-only", "-gline-tables-only", "-S"]

QUARTER = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_")
HALF_OPS = re.compile(r"^v_(pk_|lshlrev_b32|max|min|med3|max3|min3|cvt|rndne|fract|floor|ceil|trunc|cmp|cmpx|cndmask|lshl_add|lshl_or|and_or|or3|"
                      r"add3|perm|bfe|bfi|mad_u32|mad_i32|mul_lo|mul_hi|div_fixup|div_scale|div_fmas|readlane|readfirstlane|"
                      r"writelane|alignbit|xad|add_lshl|mbcnt|sad|ldexp|frexp)")


def stage_of(fname, line, stages):
    for lo, hi, f, name in stages:
        if f == fname and lo <= line <= hi:
            return name
    return f"other ({fname}:{line})"


def build_stage_table():
    """Source-line ranges -> stage names, derived from marker text in the headers (so edits to the headers move with them)."""
    stages = []
    kpath = os.path.join(ROOT, "raweditor_amd", "csrc", "rd_kernels.h")
    mpath = os.path.join(ROOT, "raweditor_amd", "csrc", "rd_math.h")
    k = open(kpath).read().splitlines()
    m = open(mpath).read().splitlines()

    def find(lines, needle, start=0):
        """1-based line of the first line at or after `start` that contains `needle` (a string, or a tuple of alternatives:
        the round-3 spelling first, the round-2 one after it, so the tool also reads the previous round's sources)."""
        needles = needle if isinstance(needle, tuple) else (needle,)
        for n in needles:
            for i in range(start, len(lines)):
                if n in lines[i]:
                    return i + 1
        raise SystemExit(f"marker not found: {needles}")

    def rng(lines, first, last_excl, fname, name, start=0):
        a = find(lines, first, start)
        b = find(lines, last_excl, a) - 1
        stages.append((a, b, fname, name))
        return a, b
    rng(k, "rd_dot709(float r", "template <int M, int MATH>", "rd_kernels.h", "luma dot (highlights/shadows, saturation, vibrance)")
    rng(k, "template <int M, int MATH>", "rd_dot709_c(float r", "rd_kernels.h", "levels divide")
    a, _ = rng(k, ("rd_colour_front(const rd_ku &u", "rd_colour_n(const rd_ku &u"), "if (!(el & RD_EL_K))", "rd_kernels.h", "white balance")
    rng(k, "if (!(el & RD_EL_K))", "if (!(el & RD_EL_MAT))", "rd_kernels.h", "temperature / tint", a)
    rng(k, "if (!(el & RD_EL_MAT))", ("// The rest: exposure ... vibrance", "if (!(el & RD_EL_EM))"), "rd_kernels.h", "colour matrix", a)
    a = find(k, ("rd_colour_tail(const rd_ku &u", "rd_colour_n(const rd_ku &u"))
    rng(k, "if (!(el & RD_EL_EM))", "if ((el & (RD_EL_HL | RD_EL_SH))", "rd_kernels.h", "exposure", a)
    rng(k, "if ((el & (RD_EL_HL | RD_EL_SH))", "// :233-234", "rd_kernels.h", "highlights / shadows", a)
    a2 = find(k, "// :233-234", a)
    stages.append((a2 - 2, find(k, "if (!(el & RD_EL_BLK))", a2) - 1, "rd_kernels.h", "contrast"))
    rng(k, "if (!(el & RD_EL_BLK))", "if (!(el & RD_EL_SAT))", "rd_kernels.h", "levels (blacks, divide call)", a)
    rng(k, "if (!(el & RD_EL_SAT))", "if (!(el & RD_EL_VIB))", "rd_kernels.h", "saturation", a)
    rng(k, "if (!(el & RD_EL_VIB))", "if constexpr (GAMMA)", "rd_kernels.h", "vibrance", a)
    rng(k, "if constexpr (GAMMA)", "// The stack for a frame whose channel-mixing", "rd_kernels.h", "gamma call", a)
    rng(k, "rd_colour_separable(const rd_ku &u", "rd_colour_m(const rd_ku &u", "rd_kernels.h", "separable stack")
    rng(k, "rd_norm(uint32_t raw", "// Rgba8Unorm quantisation (pipeline.rs:322)", "rd_kernels.h", "unpack + convert (u16 -> f32 / 4096)")
    rng(k, "// Rgba8Unorm quantisation (pipeline.rs:322)", "#define RD_F16_KA", "rd_kernels.h", "gamma shortcut -> 8-bit code (rd_q8_gamma)")
    rng(k, "#define RD_F16_KA", "// Histogram: RD_HK private copies", "rd_kernels.h", "binary16 + code: shortcut (rd_f16_gamma) / tables (rd_f16_lut_*)")
    rng(k, "rd_hist_zero(uint32_t *lh)", "// Surface stores.  FMT is an rd_format", "rd_kernels.h", "histogram (addresses + LDS atomics)")
    rng(k, "auto adopt = [&]", "typedef uint32_t rd_u4 __attribute__", "rd_kernels.h", "frame change: uniforms (adopt)")
    rng(k, "auto split = [&]", "uint32_t unit, qt, fr_c, tin0;", "rd_kernels.h", "tile bookkeeping (tickets, tile -> row / column)")
    rng(k, "auto load_tile = [&]", "// demosaic + colour stack + histogram of one tile", "rd_kernels.h", "CFA loads")
    rng(k, "auto compute_tile = [&]", "rd_tile_out<FMT> r;", "rd_kernels.h", "demosaic select / triple assembly / unpack")
    rng(k, "rd_tile_out<FMT> r;", "// surface stores of one tile.", "rd_kernels.h", "pack")
    # the first / last unit's missing-row select sits behind a wave-uniform branch (2 of H/2+1 units): rare
    sa = find(k, "auto store_tile = [&]")
    for needle in ("first / last unit only (wave-uniform): a BRANCH", "first / last unit only (wave-uniform): a branch, not selects"):
        try:
            e0 = find(k, needle, sa)
            stages.append((e0, e0 + 3, "rd_kernels.h", "frame edge: missing-row select"))
        except SystemExit:
            pass
    rng(k, "auto store_tile = [&]", "if (tile < ntiles) {", "rd_kernels.h", "surface stores")
    rng(k, "auto prefetch_frame = [&]", "// Software pipeline, one tile deep", "rd_kernels.h", "frame change: sweep")
    rng(k, "// Software pipeline, one tile deep", "// One frame, or one row band of a frame", "rd_kernels.h", "tile bookkeeping (loop)")
    rng(m, "RD_HD float rd_log2f", "// 2^z for z in", "rd_math.h", "pinned log2 (fallback / f32 gamma)")
    rng(m, "// 2^z for z in", "// clamp(pow(x, 1/2.2), 0, 1)", "rd_math.h", "pinned exp2 (fallback / f32 gamma)")
    rng(m, "// clamp(pow(x, 1/2.2), 0, 1)", "RD_HD float rd_gamma_clamp(float x)", "rd_math.h", "pinned gamma glue")
    stages.append((find(m, "RD_HD float rd_gamma_clamp(float x)"), len(m), "rd_math.h", "pinned gamma glue"))
    stages.append((find(m, "RD_HD uint32_t rd_f2u"), find(m, "RD_HD uint32_t rd_f2u") + 1, "rd_math.h", "bit casts"))
    return stages


def price(op, operands):
    """(cycles, class) of one VALU instruction."""
    if QUARTER.match(op):
        return 8, "quarter"
    if HALF_OPS.match(op) or op.endswith("_sdwa"):
        return 4, "half"
    # an SGPR (or vcc / exec / m0) SOURCE halves the rate of the full-rate ops
    srcs = operands.split(",")[1:] if "," in operands else []
    for s in srcs:
        s = s.strip().lstrip("-|").rstrip("|")
        if re.match(r"^(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|m0|ttmp)", s):
            return 4, "half (SGPR source)"
    return 2, "full"


def extract_function(asm_lines, mangled_prefix):
    start = None
    for i, ln in enumerate(asm_lines):
        if ln.startswith(mangled_prefix) and ln.rstrip().split(":")[0].startswith(mangled_prefix) and ":" in ln:
            start = i
            break
    if start is None:
        raise SystemExit(f"kernel {mangled_prefix} not found in the assembly")
    end = start
    for j in range(start + 1, len(asm_lines)):
        if asm_lines[j].startswith(".Lfunc_end"):
            end = j
            break
    return asm_lines[start:end]


def mangle(template_args):
    """'<2,true,1,0,false>' -> _Z16rd_develop_batchILi2ELb1ELi1ELi0ELb0ELb0EE  (FMT, HIST, TILES, MATH, BURST[, STAMP]; the round-4
    spelling '<2,true,true,0,false>' -- TILES was a bool then -- still names the abutting-tiles instance)"""
    parts = [p.strip() for p in template_args.strip("<>").split(",")]
    if len(parts) > 2 and parts[2] in ("true", "false"):
        parts[2] = "1" if parts[2] == "true" else "0"
    if len(parts) == 5:
        parts.append("false")                  # round 6: the sixth parameter, STAMP (the clock-stamping diagnostic instance), defaults to false
    out = "_Z16rd_develop_batchI"
    for p in parts:
        if p in ("true", "false"):
            out += "Lb%dE" % (p == "true")
        else:
            out += "Li%sE" % p
    return out + "E"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="<2,true,true,0,false>", help="template arguments of rd_develop_batch<FMT,HIST,TILES,MATH,BURST> (TILES: 0 masked, 1 whole, 2 overlapped last tile)")
    ap.add_argument("--asm", help="an existing -gline-tables-only -S output (skips the compile)")
    ap.add_argument("--listing", help="write the annotated main-loop listing here")
    ap.add_argument("--summary", help="write the per-stage table here (also printed)")
    ap.add_argument("--define", action="append", default=[], help="extra -D for the compile (A/B builds)")
    ap.add_argument("--json", help="merge this kernel's hot-path totals into this JSON file (profiles/isa_budget.json: what bench.py's "
                                   "valu_issue_frac is computed from)")
    args = ap.parse_args()

    if args.asm:
        asm = open(args.asm).read().splitlines()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "rawdev.s")
            cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-D" + d for d in args.define] + ["-o", out, SRC]
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            asm = open(out).read().splitlines()
    files = {}
    for ln in asm:
        mm = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln)
        if mm:
            files[int(mm.group(1))] = os.path.basename(mm.group(2))
    fn = extract_function(asm, mangle(args.kernel))
    stages = build_stage_table()

    # instructions with their block label and source position
    insts, labels, cur_label, cur_loc = [], {}, "<entry>", ("?", 0)
    for ln in fn:
        s = ln.strip()
        if not s or s.startswith(";"):
            continue
        mm = re.match(r"^(\.LBB\d+_\d+):", s)
        if mm:
            cur_label = mm.group(1)
            labels[cur_label] = len(insts)
            continue
        mm = re.match(r"^\.loc\s+(\d+)\s+(\d+)", s)
        if mm:
            cur_loc = (files.get(int(mm.group(1)), "?"), int(mm.group(2)))
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        body = s.split(";")[0].strip()
        if not body:
            continue
        op = body.split()[0]
        operands = body[len(op):].strip()
        insts.append({"op": op, "operands": operands, "label": cur_label, "loc": cur_loc, "text": body})

    # loops = backward branches on a WAVE-UNIFORM condition (scc / vcc); the main loop is the one enclosing the most
    # instructions.  Not s_branch (block layout) and not s_cbranch_exec*: those are the returns of out-of-line blocks that
    # hipcc places behind the loop -- the rare pinned evaluation of a few lanes jumps back INTO the body from there, which
    # would otherwise read as a "loop" from the body's top to the end of the kernel.
    best = None
    for i, ins in enumerate(insts):
        if ins["op"].startswith(("s_cbranch_scc", "s_cbranch_vcc")):
            tgt = ins["operands"].strip()
            if tgt in labels and labels[tgt] <= i:
                span = (labels[tgt], i)
                if best is None or span[1] - span[0] > best[1] - best[0]:
                    best = span
    if best is None:
        raise SystemExit("no loop found")
    lo, hi = best
    loop = insts[lo:hi + 1]

    # basic blocks of the loop: a new block starts at a label and after a branch.  A block is COLD when most of its VALU
    # instructions belong to a rarely executed source region (the pinned-gamma fallback of the shortcuts, the per-frame
    # descriptor reload, the per-frame sweep); everything in a cold block is listed apart, whatever line it came from
    # (the fallback's own rd_q8 pack lives in the same block as the pinned polynomials).
    # (the f32 surface evaluates the pinned pair for every value: there it is the hot path)
    f32_surface = args.kernel.strip("<>").split(",")[0].strip() == "0"
    cold_stage = re.compile(r"^(frame change|frame edge)" if f32_surface else r"^(frame change|frame edge|pinned )")
    blocks, cur, last_label = [], [], None
    for ins in loop:
        if ins["label"] != last_label and cur:
            blocks.append(cur); cur = []
        last_label = ins["label"]
        cur.append(ins)
        if ins["op"].startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
            blocks.append(cur); cur = []
    if cur:
        blocks.append(cur)
    for blk in blocks:
        valu = [i for i in blk if i["op"].startswith("v_")]
        # (instructions without a source position -- line 0: selects and copies the compiler adds when it merges blocks --
        # take the character of the attributable ones around them)
        placed = [i for i in valu if i["loc"][1] != 0]
        ncold = sum(1 for i in placed if cold_stage.match(stage_of(i["loc"][0], i["loc"][1], stages)))
        is_cold = bool(placed) and ncold * 2 > len(placed)
        for i in blk:
            i["cold"] = is_cold

    rows = collections.OrderedDict()
    tot = collections.Counter()
    listing = []
    for ins in loop:
        st = stage_of(ins["loc"][0], ins["loc"][1], stages)
        op = ins["op"]
        kind, cyc = "other", 0
        if op.startswith("v_"):
            cyc, kind = price(op, ins["operands"])
        elif op.startswith("s_") and (op.startswith("s_load") or op.startswith("s_atomic") or op.startswith("s_buffer") or op.startswith("s_dcache")):
            kind = "SMEM"
        elif op.startswith("s_"):
            kind = "SALU"
        elif op.startswith("ds_"):
            kind = "LDS"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            kind = "VMEM"
        cold = ins["cold"]
        if cold and not cold_stage.match(st):
            st = "fallback block: " + st
        r = rows.setdefault(st, collections.Counter())
        r["cold"] = cold
        if op.startswith("v_"):
            r["valu"] += 1
            r["cyc"] += cyc
            r[kind.split(" ")[0]] += 1
            if kind.endswith("(SGPR source)"):
                r["sgpr_src"] += 1
            if not cold:
                tot["valu"] += 1; tot["cyc"] += cyc; tot[kind.split(" ")[0]] += 1
                if kind.endswith("(SGPR source)"):
                    tot["sgpr_src"] += 1
        else:
            r[kind] += 1
            if not cold:
                tot[kind] += 1
        listing.append(f"{'C' if cold else ' '} {cyc if cyc else '':>2} {kind:<18} {ins['text']:<70} ; {ins['loc'][0]}:{ins['loc'][1]}  [{st}]")

    out = []
    out.append(f"rd_develop_batch{args.kernel}: main loop = {len(loop)} instructions ({lo}..{hi} of {len(insts)} in the kernel)")
    out.append("VALU priced with tools/valu_probe2.hip's issue costs (cycles per wave-instruction per SIMD): full 2, half 4 "
               "(v_max/min/cvt/cmp/cndmask/VOP3-int, or any SGPR source), quarter 8 (v_exp/v_log/v_rcp)")
    out.append("")
    out.append(f"{'stage':<58} {'VALU':>5} {'full':>5} {'half':>5} {'(sgpr)':>6} {'qtr':>4} {'cycles':>7} {'SALU':>5} {'LDS':>4} {'VMEM':>5}")
    for cold in (False, True):
        if cold:
            out.append("-- branch-conditional (rare) code inside the loop, not charged to a tile --")
        for st, r in sorted(rows.items(), key=lambda kv: -kv[1]["cyc"]):
            if bool(r["cold"]) != cold:
                continue
            out.append(f"{st:<58} {r['valu']:>5} {r['full']:>5} {r['half']:>5} {r['sgpr_src']:>6} {r['quarter']:>4} {r['cyc']:>7} "
                       f"{r['SALU']:>5} {r['LDS']:>4} {r['VMEM']:>5}")
        if not cold:
            out.append(f"{'TOTAL per tile (hot path)':<58} {tot['valu']:>5} {tot['full']:>5} {tot['half']:>5} {tot['sgpr_src']:>6} "
                       f"{tot['quarter']:>4} {tot['cyc']:>7} {tot['SALU']:>5} {tot['LDS']:>4} {tot['VMEM']:>5}")
    tiles = 94376
    cyc_frame = tot["cyc"] * tiles / 1024.0
    out.append("")
    out.append(f"hot-path VALU issue per 24 MP frame: {tot['cyc']} cycles x {tiles} tiles / 1024 SIMDs = {cyc_frame:,.0f} cycles per SIMD "
               f"= {cyc_frame / 1.9e3:.1f} us at 1.9 GHz, {cyc_frame / 2.4e3:.1f} us at 2.4 GHz "
               f"(static count: a wave-uniform branch not taken, e.g. an elided step, costs less)")
    text = "\n".join(out)
    print(text)
    if args.json:
        import json
        try:
            doc = json.load(open(args.json))
        except (OSError, ValueError):
            doc = {}
        surface = {"0": "f32", "1": "f16", "2": "u8", "3": "rgb8"}.get(args.kernel.strip("<>").split(",")[0].strip(), args.kernel)
        doc.setdefault("note", "tools/isa_budget.py --json: hot-path VALU of rd_develop_batch's main loop per TILE (one wave, 64 quads = 256 px), "
                               "static count from hipcc's assembly with the bench workload's elision flags pinned (-DRD_BUDGET_ELIDE=128u); "
                               "issue_cycles prices full-rate instructions at 2, half-rate at 4, quarter-rate at 8 (tools/valu_probe2.hip)")
        doc.setdefault("kernels", {})[surface] = {
            "kernel": "rd_develop_batch" + args.kernel, "defines": args.define, "valu_instructions": tot["valu"], "full_rate": tot["full"],
            "half_rate": tot["half"], "quarter_rate": tot["quarter"], "issue_cycles": tot["cyc"], "lds_instructions": tot["LDS"],
            "salu_instructions": tot["SALU"], "px_per_tile": 256}
        with open(args.json, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
            f.write("\n")
    if args.summary:
        open(args.summary, "w").write(text + "\n")
    if args.listing:
        open(args.listing, "w").write("\n".join(listing) + "\n")


if __name__ == "__main__":
    main()
