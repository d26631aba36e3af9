"""Build librawdev.so (the HIP kernels + C ABI) in-tree for gfx950.

`python -m raweditor_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
-ffp-contract=off is part of the numerical contract: the colour stack's operation
order is the WGSL text's, and only the explicit FMAs of the pinned pow pair are fused.
"""
from __future__ import annotations

import json
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "librawdev.so")
SOURCES = ["rawdev.hip"]
HEADERS = ["rd_math.h", "rd_uniforms.h", "rd_kernels.h", "rd_ljpeg.h", "rd_copy_pool.h", "rd_node_worker.h", "rd_host_pipeline.inl", "rd_host_batch.inl", "rd_host_diag.inl", os.path.join("..", "..", "include", "rawdev.h")]
# -fno-slp-vectorize: what the SLP vectoriser packs into v_pk_*_f32 costs more in SGPR pairs, register moves and spills
# than the packed issue rate returns (measured 94-98 us against 82-89 us per frame), and a packed
# operand pair that happens to contain a pending load's register makes hipcc drain the store queue mid-loop (false
# vmcnt dependency); scalar f32 code keeps the s_waitcnt placement exact.
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-std=c++17", "-Wall", "-Wextra", "-Rpass-analysis=kernel-resource-usage", "-pthread", "-ldl"]
RESOURCES_PATH = os.path.join(PKG_DIR, "kernel_resources.json")
# The export kernel runs two 1024-thread workgroups per CU (8 waves per SIMD).  That holds only while a wave needs
# <= 64 VGPRs and <= 80 SGPRs (the kernel carries amdgpu_num_sgpr(80), so the compiler spills rather than exceed it)
# and no scratch; hipcc reports "occupancy 8" even when the SGPR budget is blown, so the build checks the numbers itself.
LIMITS = {"vgprs": 64, "sgprs": 80, "scratch": 0}


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    cmd = [find_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    out = subprocess.run(cmd, capture_output=True, text=True)
    remarks, other, in_remark = [], [], False
    for line in out.stderr.splitlines():
        snippet = line.lstrip()[:1].isdigit() and " | " in line or line.lstrip().startswith("|")
        if "-Rpass-analysis=kernel-resource-usage" in line:
            if other and other[-1].startswith("In file included"):
                other.pop()                                      # include trail that introduces the remark
            remarks.append(line); in_remark = True
        elif in_remark and snippet:
            continue                                             # source excerpt that belongs to the remark
        else:
            other.append(line); in_remark = False
    if other:
        print("\n".join(other), file=sys.stderr)
    nwarn = sum(": warning:" in line for line in other)
    if nwarn:                                                    # last thing on stderr, so that a glance at the tail of a build log sees it
        print(f"raweditor_amd.build: {nwarn} compiler warning line(s) above -- the tree is kept clean under -Wall -Wextra", file=sys.stderr)
    if out.returncode != 0:
        raise subprocess.CalledProcessError(out.returncode, cmd)
    res = parse_resource_remarks(remarks)
    with open(RESOURCES_PATH, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    check_resources(res)
    return LIB_PATH


def parse_resource_remarks(lines) -> dict:
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {kernel: {sgprs, vgprs, scratch, occupancy, lds, ...}}."""
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
            "LDS Size [bytes/block]": "lds"}
    res, cur = {}, None
    for line in lines:
        body = line.split("remark:", 1)[-1].split("[-Rpass-analysis")[0].strip()
        if body.startswith("Function Name:"):
            cur = res.setdefault(body.split(":", 1)[1].strip(), {})
            continue
        if cur is None or ":" not in body:
            continue
        k, v = body.rsplit(":", 1)
        if k.strip() in keys:
            try:
                cur[keys[k.strip()]] = int(v)
            except ValueError:
                pass
    return res


def check_resources(res: dict) -> None:
    """Every instance of the export kernel must keep the two-workgroups-per-CU budget (see LIMITS)."""
    bad = []
    for name, r in res.items():
        if "rd_develop_quads" not in name and "rd_develop_batch" not in name:
            continue
        for k, lim in LIMITS.items():
            if r.get(k, 0) > lim:
                bad.append(f"{name}: {k} = {r[k]} > {lim}")
    if bad:
        raise RuntimeError("export kernel exceeds its register budget (second workgroup per CU would not fit):\n  " +
                           "\n  ".join(bad))


def load_resources() -> dict:
    build_library()
    if not os.path.exists(RESOURCES_PATH):
        build_library(force=True)
    with open(RESOURCES_PATH) as f:
        return json.load(f)


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
