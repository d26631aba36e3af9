"""-m gpu: bench.py's entry points exactly as a driver would call them, small workloads, on the one-GPU box.

* `python bench.py --gpus 2 ...` with NO launcher around it: the parent starts two fresh rank processes itself
  (RAWDEV_DIST_BACKEND=gloo: both ranks share device 0 and the histogram all-reduce goes over gloo -- RCCL refuses two
  ranks on one device; the driver's multi-GPU runs use nccl on distinct devices, which has never run on this pool's
  one-GPU boxes: N > 1 on distinct devices stays UNMEASURED).
* `python bench.py --host node --gpus 2 ...`: ONE process, rd_node_batch_* (RD_NODE_REDUCE=host: device 0 listed twice).
* N = 1 with `extra_configs` on a reduced batch.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_raw(argv, extra_env, timeout=900):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def _bench(argv, extra_env, timeout=900):
    out = _bench_raw(argv, extra_env, timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bare_multi_gpu_call_launches_its_own_ranks(gpu_lib):
    r = _bench(["--gpus", "2", "--frames", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
               {"RAWDEV_DIST_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["verified"] is True
    assert r["config"]["frames_per_gpu"] == 8 and r["scaling"] == "weak"
    assert r["value"] > 0 and r["roofline"]["frac"] > 0
    assert "extra_configs" not in r and "cpu_baseline" not in r      # N = 1 only
    assert r["allreduce_us"]["median"] > 0 and r["allreduce_us"]["bytes"] == 6144     # reported separately
    # the line says by itself what ran where
    assert r["world_size_seen"] == r["world_size_env"] == 2 and r["backend"] == "gloo"
    assert [x["rank"] for x in r["ranks"]] == [0, 1]
    for x in r["ranks"]:
        assert x["pci_bus_id"] and x["name"] and x["pid"] > 0
        assert x["ms_per_step"] > 0 and x["ms_per_step_hip_events"] > 0 and x["launches"] > 0 and x["us_per_frame"] > 0 and x["MP_per_s"] > 0
        assert x["box_copy_GBps"] > 1000 and x["box_fill_GBps"] > 1000
    assert r["distinct_devices"] == 1                      # the rehearsal: both ranks share device 0 (allowed under gloo only)
    for x in r["ranks"]:                                   # round 6: every rank narrowed its own view to ONE card before it touched the GPU
        assert x["visibility"]["mode"] == "own" and x["visibility"]["variable"] == "ROCR_VISIBLE_DEVICES" and x["visibility"]["value"] == "0", x
        assert x["devices_visible"] == 1 and x["device_index"] == 0
        assert x["launch_us"]["median"] > 0 and x["kernel_ms_per_step"] > 0 and 0.3 < x["clock_under_kernel_GHz"] < 3.5
    assert r["roofline"]["launches"]["steps"] == 6 and r["kernel_ms_per_step"] > 0
    assert 0 < r["per_gpu_MPps_min"] <= r["per_gpu_MPps_max"]
    assert r["env"]["RAWDEV_DIST_BACKEND"] == "gloo" and "HSA_ENABLE_IPC_MODE_LEGACY" in r["env"]
    assert "distinct_devices" in r["config"]["host"]


def test_two_ranks_on_one_device_is_an_invalid_nccl_run(gpu_lib):
    """Under nccl (= RCCL) two ranks on one GPU did not measure two GPUs: bench.py must refuse to print a result.  The
    rule is exercised on the gloo rehearsal with RAWDEV_DIAG_ASSUME_NCCL=1 (RCCL itself refuses to start that way)."""
    out = _bench_raw(["--gpus", "2", "--frames", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-box"],
                     {"RAWDEV_DIST_BACKEND": "gloo", "RAWDEV_DIAG_ASSUME_NCCL": "1"})
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], out.stdout      # no result line on stdout
    assert "INVALID RUN" in out.stderr and "distinct device" in out.stderr


def test_node_host_mode_one_process(gpu_lib):
    r = _bench(["--host", "node", "--gpus", "2", "--frames", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
               {"RD_NODE_REDUCE": "host"})
    assert r["n_gpus"] == 2 and r["verified"] is True
    assert "rd_node_batch" in r["config"]["host"] and "REHEARSAL" in r["config"]["host"]
    assert r["histogram_call_us"]["median"] > 0 and "host fold" in r["histogram_call_us"]["reduction"]
    assert [d["device_index"] for d in r["devices"]] == [0, 0] and r["distinct_devices"] == 1 and r["devices"][0]["pci_bus_id"]
    r1 = _bench(["--host", "node", "--gpus", "1", "--frames", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {})
    assert r1["n_gpus"] == 1 and r1["verified"] is True and "REHEARSAL" not in r1["config"]["host"]


def test_single_gpu_line_carries_the_other_configs(gpu_lib):
    r = _bench(["--frames", "16", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"], {})
    assert r["n_gpus"] == 1 and r["verified"] is True
    assert r["config"]["workload"].startswith("BASELINE configs[2]")
    assert "uploads its descriptors" in r["config"]["descriptors"]
    ex = r["extra_configs"]
    assert "error" not in ex, ex
    for key in ("single_frame_f32", "batch_rgba8", "config5_shape_f16", "config5_shape_f16_tiled"):
        e = ex[key]
        assert e["verified"] is True, (key, e)
        assert e["ms"] > 0 and e["MP_per_s"] > 0
        assert 0 < e["roofline"]["frac"] < 1 and e["roofline"]["achieved"] > 0
    assert ex["config5_shape_f16"]["row_bands_effective"] == 1 and ex["config5_shape_f16_tiled"]["row_bands_effective"] == 8
    assert ex["config5_shape_f16_tiled"]["launches_per_step"] == 16 * 8
    # the reference's own metric-path entry, as its caller invokes it, and the export ring
    fr = ex["full_res_to_bytes"]
    assert fr["verified"] is True, fr
    assert fr["pcie_floor_ms"] > 0 and "pipeline.rs:525" in fr["reference_published"]
    for k in ("pinned_dst", "pageable_dst_reused", "pageable_dst_fresh", "borrowed_surface"):
        assert fr[k]["ms"] > 0 and fr[k]["GBps_over_pcie"] > 0 and fr[k]["limiter"]
    assert fr["pinned_dst"]["ms"] <= 2.5 and fr["borrowed_surface"]["ms"] <= 2.5, fr    # the bar: a 24 MP export costs the PCIe transfer
    for k in ("rgb8", "rgba8", "rgb8_from_pinned_host", "rgb8_from_pageable_host"):     # (the last two: rd_exporter_submit_host)
        assert ex["export_ring"][k]["verified"] is True and ex["export_ring"][k]["ms_per_frame"] > 0
    # the narrow surfaces carry the VALU roofline of their threshold-table kernels (round 4: no transcendental left in either)
    assert ex["batch_rgba8"]["roofline"]["bound"] == "valu" and ex["batch_rgba8"]["roofline"]["valu_issue_cycles_per_tile"] == 698
    assert ex["config5_shape_f16"]["roofline"]["bound"] == "valu" and ex["config5_shape_f16"]["roofline"]["valu_issue_cycles_per_tile"] == 876
    # round 5: a frame width that is not a multiple of the 128-px tile beside the headline's, all four surfaces, oracle-checked
    rw = ex["ragged_width"]
    assert rw["verified"] is True and "6000x4000" in rw["config"]
    for k in ("f32", "f16", "u8", "rgb8"):
        assert rw[k]["verified"] is True and 0.8 < rw[k]["ns_per_px_ratio"] < 1.25, (k, rw[k])
    ow = ex["odd_width"]                                          # round 6: an odd width through the batch path, timed and oracle-checked
    assert ow["verified"] is True and "6001x4001" in ow["config"] and "ODD" in ow["config"]
    for k in ("f32", "f16", "u8", "rgb8"):
        assert ow[k]["verified"] is True and 0.8 < ow[k]["ns_per_px_ratio"] < 1.6, (k, ow[k])
    # round 6: whole frames of the batch path against checksums of the reference's shader text, evaluated (no oracle involved)
    tp = ex["text_pin"]
    assert tp["verified"] is True and len(tp["frames"]) == 7, tp
    assert all(v["sha256_matches"] and v["histogram_matches"] for v in tp["frames"].values())
    assert tp["frames"]["full_11648x8736_mild"]["surface"] == "f16" and tp["frames"]["full_6001x4001_mild"]["surface"] == "f32"
    assert r["diagnosis"]["extras"]["text_pin_verified"] is True
    # the f32 kernel's issue budget (round 5: scalar-base store addresses; parking the slider uniforms measured negative and is off),
    # and the bound as this run's fractions say it
    assert r["roofline"]["valu_issue_cycles_per_tile"] == 1360 and r["roofline"]["bound"] == "hbm"
    assert r["roofline"]["bound_measured"] in ("hbm", "valu", "hbm+valu") and "bound_measured_note" in r["roofline"]
    assert r["config"]["buffers"]["plane_stagger"] == -1 and len(r["config"]["buffers"]["cfa_addr_mod_2MiB_first8"]) == 8
    # the box's own ceilings, measured in this run (item 4)
    rf = r["roofline"]
    assert rf["box_copy_GBps"] > 3000 and rf["box_fill_GBps"] > 3000 and 0 < rf["frac_of_box_copy"] < 1.2
    assert r["ranks"][0]["pci_bus_id"] and r["distinct_devices"] == 1 and r["world_size_seen"] == 1
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and "march=native" in cb["build"] and cb["value_portable_O2_build"] > 0
    # round 6: the line explains its own speed
    la = rf["launches"]
    assert la["steps"] == 6 and la["launches_per_step"] == r["config"]["launches_per_step"] == len(la["launch_us_by_position"]) == 2
    assert 0 < la["launch_us"]["min"] <= la["launch_us"]["median"] <= la["launch_us"]["max"] and la["step_boundary_gap_us"]["median"] > 0
    assert r["kernel_ms_per_step"] > 0 and r["gap_ms_per_step"] is not None and 0 < rf["frac_kernel"] < 1
    assert abs(r["kernel_ms_per_step"] + r["gap_ms_per_step"] - r["ms_per_step"]) < 1e-3
    assert rf["box_before"]["copy"] > 3000 and rf["box_after"]["copy"] > 3000
    assert rf["box_before"]["valu_effective_GHz"] > 0.5 and rf["box_after"]["valu_effective_GHz"] > 0.5
    ck = r["clock_under_kernel_GHz"]
    assert 0.3 < ck["GHz_min"] <= ck["GHz_median"] <= ck["GHz_max"] < 3.5, ck
    assert (r["clocks"] and r["clocks"]["samples"] > 0 and r["clocks"]["source"]) or r["clocks_reason"], (r["clocks"], r["clocks_reason"])
    assert rf["box_pattern_GBps"] > 1000 and 0 < rf["frac_of_box_pattern"] < 1.3 and rf["box_pattern"]["launch_us"]["median"] > 0
    ab = r["descriptor_upload_ab"]
    assert ab["rotating_ms_per_step"] > 0 and ab["static_ms_per_step"] > 0 and len(ab["rotating_all"]) == 3
    assert ex["batch_rgba8"]["launches"]["launch_us"]["median"] > 0 and ex["batch_rgba8"]["kernel_ms_per_step"] > 0
    assert ex["config5_shape_f16"]["launches"]["launches_per_step"] == ex["config5_shape_f16"]["launches_per_step"]
    # ... and says it where a record that keeps `roofline` whole, or only the tail of stdout, still has it
    assert list(r.keys())[-1] == "diagnosis"
    dg = r["diagnosis"]
    assert dg["value_MPps"] == r["value"] and dg["frac"] == rf["frac"] and dg["kernel_ms_per_step"] == r["kernel_ms_per_step"]
    assert len(dg["launch_us_by_position"]) == 2 and dg["clock_under_kernel_GHz"] == ck["GHz_median"] and dg["box_pattern_GBps"] == rf["box_pattern_GBps"]
    assert dg["extras"]["all_verified"] is True and dg["extras"]["config5_shape_f16"]["frac"] == ex["config5_shape_f16"]["roofline"]["frac"]
    assert dg["batch_rgba8"]["us_per_frame"] == ex["batch_rgba8"]["us_per_frame"] and dg["verified"] is True
    assert rf["kernel_ms_per_step"] == r["kernel_ms_per_step"] and rf["clock_under_kernel_GHz"] == ck["GHz_median"]
    assert len(json.dumps(dg)) < 6000                              # fits the tail a driver keeps
    q = _bench(["--frames", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--no-diagnose"], {})
    assert q["verified"] is True and q["kernel_ms_per_step"] is None and q["roofline"]["launches"] is None and q["clocks"] is None


def test_measure_hbm_sizes(gpu_lib):
    """rd_measure_hbm's kernels have no tail handling: the size is rounded down to whole 8-KiB steps of the larger grid
    (512 MiB units), so 768 MiB -- a size whose copy waves used to run 4 KiB past their range and the last one past the
    allocation -- measures 512 MiB, and a size below one unit is refused instead of measuring nothing."""
    ra = gpu_lib
    for nbytes in (768 << 20, 512 << 20, (1 << 30) + 12345):
        c, f, r, m = ra.measure_hbm(0, nbytes, 2)
        assert 1000.0 < c < 8000.0 and 1000.0 < f < 8000.0 and 1000.0 < r < 8000.0 and m > 500.0, (nbytes, c, f, r, m)
    for nbytes in (64 << 20, 256 << 20, (512 << 20) - 16):
        with pytest.raises(ra.RawdevError) as ei:
            ra.measure_hbm(0, nbytes, 2)
        assert ei.value.code == -1 and "512 MiB" in ei.value.message
