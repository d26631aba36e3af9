#!/usr/bin/env python3
"""bench.py -- megapixels/sec through the fused demosaic + 10-slider develop kernel (BASELINE.json).

Workload (configs[2] of BASELINE.json, the configuration the metric is quoted on; per GPU it is
also config 4's share): a batch of 256 distinct synthetic 24 MP RGGB frames (6016 x 4016 u16,
uniform 12-bit, seed 0x52415745) resident in HBM, one randomised slider stack per frame drawn from
the UI ranges, wb = (2, 1, 1.5), non-identity colour matrix, RGBA-f32 surface written to a ring of
output buffers, fused 3x256 histogram accumulated in u64.  One "step" = one pass over the batch:
the fused launches of rd_batch_develop (up to 8 frames each) + the histogram fold (+ one RCCL all-reduce of 768 x i64 when N > 1).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` counts output pixels of all ranks over the max-over-ranks
wall time of exactly K steps, inputs already resident in HBM.  `roofline` is the dominant kernel
(rd_develop_quads) against the 8 TB/s HBM3E peak with the algorithmic 18 B/px of BASELINE.md section 2;
its launch duration is measured here with HIP events on the launch stream.  `cpu_baseline` is the
oracle (oracle/develop_ref.c, a port -- the reference has no CPU path) on the host cores, rank 0,
N = 1 only, reported-only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x52415745
WB = (2.0, 1.0, 1.5, 1.0)
CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
BYTES_PER_PX = {"f32": 18, "f16": 10, "u8": 6}   # 2 B CFA read + surface write


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--width", type=int, default=6016)
    ap.add_argument("--height", type=int, default=4016)
    ap.add_argument("--format", choices=["f32", "f16", "u8"], default="f32")
    ap.add_argument("--ring", type=int, default=8, help="output buffers cycled through")
    ap.add_argument("--row-bands", type=int, default=1)
    ap.add_argument("--math", choices=["strict", "contracted"], default="strict",
                    help="arithmetic of the colour stack (rd_math_mode); strict = literal WGSL order (default)")
    ap.add_argument("--no-alt-math", action="store_true",
                    help="skip the short extra run in the other math mode (reported under 'alt_math', N=1 only)")
    ap.add_argument("--no-hist", action="store_true")
    ap.add_argument("--data", choices=["uniform", "gradient"], default="uniform",
                    help="uniform: i.i.d. 12-bit samples (SURVEY 8d, the headline); gradient: smooth ramp + 1 %% noise "
                         "(SURVEY 8d's second distribution: flat regions, same-bin histogram atomics, less bit toggling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget")
    return ap.parse_args()


def usable_cores():
    """Threads worth starting: the CPUs this process may run on, capped by the cgroup's CPU quota (a GPU box gives a
    one-GPU job a share of the host -- 256 hardware threads are visible, far fewer can run at once)."""
    hw = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = hw
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, f"{n} of {hw} hardware threads usable (affinity / cgroup CPU quota)"


def cpu_baseline(width, height, budget_s):
    """The oracle (a port: the reference has no CPU path) on the host cores, whole frames of the same workload.
    oracle/develop_ref.c::ref_bench_mt: one persistent thread per core, each owning a row band and a band buffer it
    first-touched itself (no per-frame thread creation, no remote-node page placement), frames separated by barriers."""
    import ctypes as C
    import numpy as np
    from oracle import ref_c
    from raweditor_amd import EditParams, FIELDS
    cores, cores_note = usable_cores()
    rng = np.random.default_rng([SEED, 0])
    cfa = rng.integers(0, 4096, (height, width), dtype=np.uint16)
    p = EditParams.random(np.random.default_rng([SEED, 1]))
    u = ref_c.make_uniforms({f: getattr(p, f) for f in FIELDS}, WB, CM)
    L = ref_c.lib()
    cp = cfa.ctypes.data_as(C.POINTER(C.c_uint16))
    fr, sec = C.c_int(), C.c_double()
    L.ref_bench_mt(cp, width, height, C.byref(u), cores, float(budget_s), 4096, C.byref(fr), C.byref(sec))
    frames, el = fr.value, sec.value
    # one-thread figure on a 256-row band of the same frame
    band_h = min(256, height)
    L.ref_bench_mt(cp, width, band_h, C.byref(u), 1, min(2.0, float(budget_s)), 64, C.byref(fr), C.byref(sec))
    one_thread = fr.value * width * band_h / 1e6 / sec.value
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    mpps = frames * width * height / 1e6 / el
    return {"value": round(mpps, 2), "unit": "MP/s", "cores": cores, "kind": "port",
            "one_thread_MPps": round(one_thread, 2),
            "parallel_efficiency": round(mpps / (one_thread * cores), 3),
            "sample": f"{frames} x {width}x{height} frame(s), randomised stack, f32 surface, {el:.1f} s on {cores} "
                      f"persistent threads with first-touch row bands ({model}; {cores_note}); scalar f32 port of the shader, no SIMD: "
                      f"arithmetic-bound (one thread: {one_thread:.1f} MP/s), reported-only"}


def verify_outputs(ra, fmt_name, W, H, cfas, params, ring, n_frames, row_bands, math_name):
    """Outside the timed region: download two surfaces of the output ring as the timed passes left them and compare
    four row bands of each with the oracle, bit for bit (the histogram sum alone would pass for garbage pixels).
    Slot s of the ring was last written by frame n_frames - len(ring) + s (frames go to slot i % len(ring))."""
    import numpy as np
    from oracle import ref_c                       # the checker, never the thing measured
    from raweditor_amd import FIELDS
    math_mode = ref_c.MATH_CONTRACTED if math_name == "contracted" else ref_c.MATH_STRICT
    mid = (H // 2) | 1
    bands = [(0, min(6, H)), (max(0, min(1001, H - 6)), min(1007, H)), (mid, min(mid + 2, H)), (max(0, H - 6), H)]
    nring = len(ring)
    checked = []
    for slot in sorted({0, nring - 1}):
        owners = [i for i in range(n_frames) if i % nring == slot]
        if not owners:
            continue
        i = owners[-1]
        cfa = cfas[i].cpu().numpy().view(np.uint16)
        u = ref_c.make_uniforms({f: getattr(params[i], f) for f in FIELDS}, WB, CM, math_mode=math_mode)
        raw = ring[slot].cpu().numpy()
        for r0, r1 in bands:
            exp = ref_c.render_band(cfa, u, r0, r1)
            if fmt_name == "f32":
                got = raw.view(np.float32).reshape(H, W, 4)[r0:r1]
                ok = np.array_equal(got.view(np.uint32), exp.view(np.uint32))
            elif fmt_name == "f16":
                got = raw.view(np.uint16).reshape(H, W, 4)[r0:r1]
                ok = np.array_equal(got, ref_c.pack_f16(exp).view(np.uint16))
            else:
                got = raw.reshape(H, W, 4)[r0:r1]
                ok = np.array_equal(got, ref_c.pack_u8(exp))
            if not ok:
                return False, f"frame {i} (ring slot {slot}) rows {r0}..{r1} differ from the oracle"
        checked.append(i)
    return True, f"frames {checked}: {len(bands)} row bands each bit-identical to the oracle"


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    # RCCL / device-tensor sharing across processes needs dmabuf IPC on this pool's host driver (already exported by the
    # image; kept here so that a bare `python -m torch.distributed.run ... bench.py` works from any shell)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # RAWDEV_DIST_BACKEND=gloo is a rehearsal mode for a 1-GPU box: several ranks share device 0 and the
    # histogram all-reduce goes over gloo (RCCL refuses two ranks on one device).  The driver's runs use nccl.
    backend = os.environ.get("RAWDEV_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import raweditor_amd as ra
    from raweditor_amd.batch import allreduce_histogram

    W, H, F = args.width, args.height, args.frames
    fmt = {"f32": ra.FMT_RGBA_F32, "f16": ra.FMT_RGBA_F16, "u8": ra.FMT_RGBA_U8}[args.format]
    bpp_out = ra.BYTES_PER_PIXEL[fmt]
    with_hist = not args.no_hist

    # ---- synthetic batch, generated on the device, keyed by (seed, global frame index) ------------
    cfas, params = [], []
    for f in range(F):
        gidx = rank + f * world                           # frame i -> rank i mod N
        g = torch.Generator(device=dev)
        g.manual_seed(SEED + gidx)
        if args.data == "uniform":
            cfas.append(torch.randint(0, 4096, (H, W), generator=g, device=dev, dtype=torch.int16))
        else:                                             # a diagonal ramp whose slope and offset vary per frame, +-1 % noise
            yy = torch.arange(H, device=dev, dtype=torch.float32)[:, None] / H
            xx = torch.arange(W, device=dev, dtype=torch.float32)[None, :] / W
            a = 0.25 + 0.5 * ((gidx * 37) % 16) / 16.0
            ramp = (a * xx + (1.0 - a) * yy) * 3600.0 + 200.0
            noise = (torch.rand((H, W), generator=g, device=dev) - 0.5) * 2.0 * 40.96
            cfas.append((ramp + noise).clamp_(0, 4095).to(torch.int16))
            del yy, xx, ramp, noise
        params.append(ra.EditParams.random(np.random.default_rng([SEED, gidx])))
    ring = [torch.empty(H * W * bpp_out, dtype=torch.uint8, device=dev) for _ in range(max(1, args.ring))]
    hist = torch.zeros(768, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    math_mode = ra.MATH_CONTRACTED if args.math == "contracted" else ra.MATH_STRICT
    be = ra.BatchExporter(dev_index, W, H, fmt, with_hist, math_mode=math_mode)
    frames = be.make_frames([c.data_ptr() for c in cfas], [ring[i % len(ring)].data_ptr() for i in range(F)],
                            params, WB, CM)
    stream = torch.cuda.Stream(device=dev)

    def step():
        be.develop(frames, row_bands=args.row_bands, stream=stream.cuda_stream)
        if with_hist:
            be.histogram(hist.data_ptr(), stream=stream.cuda_stream)
            if world > 1:
                allreduce_histogram(hist)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            step()
        ev1.record(stream)
        barrier()
        elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)                         # HIP events on the launch stream

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_px = float(world) * F * W * H * args.steps
    if with_hist:                                          # sanity: the global histogram counts every pixel
        got = int(hist.sum().item())
        assert got == 3 * world * F * W * H, f"histogram sum {got} != {3 * world * F * W * H}"

    verified, verified_note = None, "not checked"
    if rank == 0:                                          # outside the timed region
        try:
            verified, verified_note = verify_outputs(ra, args.format, W, H, cfas, params, ring, F, args.row_bands, args.math)
        except Exception as e:  # noqa: BLE001  (oracle not built / not shipped: say so, do not claim)
            verified, verified_note = None, f"oracle check unavailable: {e}"

    # One rd_batch_develop call = `lpc` fused launches (the library packs up to 8 consecutive frames into one launch;
    # RD_BATCH_PERSISTENT=0 gives one launch per frame / row band).  Algorithmic bytes per launch = SURVEY 8(d)'s
    # per-pixel figure x the pixels one launch processes.
    lpc = max(1, be.last_launch_count())
    launches = args.steps * lpc
    launch_us = dev_ms * 1e3 / launches                    # avg fused-launch period incl. gaps and folds
    frame_us = dev_ms * 1e3 / (args.steps * F)
    alg_bytes = BYTES_PER_PX[args.format] * W * H * F / lpc
    achieved = alg_bytes / (launch_us * 1e-6) / 1e9        # GB/s
    # HBM bytes per launch: NOT measured by this run.  It is the PMC figure (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    # passes, tools/gpu_pmc.sh + tools/parse_pmc.py) of the committed profile for this surface format, when the
    # profile was taken on the same frame size / band count; otherwise null.
    traffic, traffic_source = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            prof = json.load(fh)
        mode = "multi" if lpc < F * max(1, args.row_bands) else "per_frame"
        for ent in prof.get("entries", []):
            if ent.get("format") == args.format and list(ent.get("frame", [])) == [W, H] and ent.get("mode") == mode:
                traffic = int(ent["hbm_bytes_per_frame"] * F / lpc)             # per launch, like `achieved`
                traffic_source = (f"profiles/pmc_traffic.json [{args.format}, {W}x{H}, {mode}] (rocprofv3 PMC passes of "
                                  f"{prof.get('tag', 'a committed profile')}), not measured by this run")
                break
    except (OSError, ValueError):
        pass

    if (W, H) == (6016, 4016):
        cfg_label = "BASELINE configs[2]" if world == 1 else f"BASELINE configs[3] ({F} frames per GPU)"
    elif (W, H) == (11648, 8736):
        cfg_label = "BASELINE configs[4] shape (100 MP; per GPU)"
    else:
        cfg_label = "custom frame size"
    result = {
        "metric": "megapixels/sec through demosaic+10-slider pipeline; 24MP batch",
        "value": round(total_px / 1e6 / elapsed, 1),
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",                                    # the arithmetic type of the path for EVERY surface format
        "surface_dtype": {"f32": "f32", "f16": "f16", "u8": "u8"}[args.format],   # what is stored (`--format`)
        "data": "synthetic" if args.data == "uniform" else "synthetic (gradient + 1 % noise)",
        "verified": verified,
        "verified_note": verified_note,
        "config": {
            "workload": f"{cfg_label}: batch {F} x {W}x{H} synthetic RGGB u16 per GPU, randomised "
                        f"10-slider stacks, RGBA-{args.format} surface, fused histogram={'on' if with_hist else 'off'}, "
                        f"{args.math} f32 arithmetic",
            "frames_per_gpu": F, "width": W, "height": H, "surface": f"rgba_{args.format}",
            "row_bands": args.row_bands, "out_ring": len(ring), "math_mode": args.math, "launches_per_step": lpc,
            "parallelism": f"frames sharded round-robin over {world} GPU(s); RCCL all-reduce of i64[768] histogram only",
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": "rd_develop_batch" if lpc < F * max(1, args.row_bands) else "rd_develop_quads",
            "launch_us": round(launch_us, 2), "frames_per_launch": round(F / lpc, 3), "us_per_frame": round(frame_us, 2),
            "launch_us_note": "HIP-event time of the timed region / fused launches: an average launch PERIOD that "
                              "includes inter-launch gaps and the histogram folds (conservative)",
            "algorithmic_bytes_per_launch": int(alg_bytes),
        },
    }
    if world == 1 and not args.no_alt_math:
        # Reported-only: the same workload in the other arithmetic, 5 steps.
        other = "contracted" if args.math == "strict" else "strict"
        be2 = ra.BatchExporter(dev_index, W, H, fmt, with_hist,
                               math_mode=ra.MATH_CONTRACTED if other == "contracted" else ra.MATH_STRICT)
        with torch.cuda.stream(stream):
            be2.develop(frames, row_bands=args.row_bands, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(5):
                be2.develop(frames, row_bands=args.row_bands, stream=stream.cuda_stream)
                if with_hist:
                    be2.histogram(hist.data_ptr(), stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (5 * F)         # per frame
        ach = BYTES_PER_PX[args.format] * W * H / (us * 1e-6) / 1e9
        result["alt_math"] = {"math_mode": other, "value": round(W * H / us, 1), "unit": "MP/s",
                              "us_per_frame": round(us, 2), "achieved_GBps": round(ach, 1),
                              "frac": round(ach / HBM_PEAK_GBPS, 4)}
        be2.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(W, H, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(result), flush=True)
    be.close()
    if world > 1:
        dist.barrier()                                     # rank 0 spent a second on the oracle check: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
