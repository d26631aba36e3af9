#!/usr/bin/env python3
"""bench.py -- megapixels/sec through the fused demosaic + 10-slider develop kernel (BASELINE.json).

Workload (configs[2] of BASELINE.json, the configuration the metric is quoted on; per GPU it is
also config 4's share): a batch of 256 distinct synthetic 24 MP RGGB frames (6016 x 4016 u16,
uniform 12-bit, seed 0x52415745) resident in HBM, one randomised slider stack per frame drawn from
the UI ranges, wb = (2, 1, 1.5), non-identity colour matrix, RGBA-f32 surface written to a ring of
output buffers, fused 3x256 histogram accumulated in u64.  One "step" = one pass over the batch:
the fused launches of rd_batch_develop (up to 8 frames each) + the histogram fold (+ one RCCL
all-reduce of 768 x i64 when N > 1).  Consecutive steps submit DIFFERENT frame descriptors: the
steps rotate through THREE frame arrays (the slider stacks rotated by 0, 1/3 and 2/3 of the batch;
librawdev caches the last two arrays it saw), so every step pays the descriptor upload a real
export pays (--static-descriptors restores the resubmission of one array).

The line explains its own speed (round 6).  After the timed region, outside it: an instrumented
pass with a HIP event pair around every fused launch (`roofline.launches`: launch_us_by_position,
min / median / max, `kernel_ms_per_step`, `gap_ms_per_step`, `roofline.frac_kernel`), the shader
clock the part holds UNDER the kernel (`clock_under_kernel_GHz`: cycle / real-time stamps of a
diagnostic kernel instance), board clocks / power / temperature while the same steps run
(`clocks`: amdsmi, else sysfs, else null with the reason), the box probes before AND after the
region (`roofline.box_before` / `box_after`), the same launches with the kernel's arithmetic
removed on the same buffers (`roofline.box_pattern_GBps`, `frac_of_box_pattern`: the memory
pattern's own ceiling on this box), and an alternating A/B of rotating against static descriptors
(`descriptor_upload_ab`).  --no-diagnose skips all of it.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...                     # N > 1 without a launcher: starts N rank processes itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --host node --gpus N ...         # ONE process, rd_node_batch_* (what a Rust / C host calls)

Prints ONE JSON line on rank 0.  `value` counts output pixels of all ranks over the max-over-ranks
wall time of exactly K steps, inputs already resident in HBM.  `roofline` is the dominant kernel
(rd_develop_batch) against the 8 TB/s HBM3E peak with the algorithmic 18 B/px of BASELINE.md section 2;
its launch duration is measured here with HIP events on the launch stream.  `cpu_baseline` is the
oracle (oracle/develop_ref.c, a port -- the reference has no CPU path) on the host cores, rank 0,
N = 1 only, reported-only.  `extra_configs` (N = 1, after the headline and outside its timed region):
BASELINE config 1 (one 24 MP frame, launch + synchronise latency), the reference's own RGBA8
surface on the batch workload, and the config-5 shape (100 MP frames, f16 surface), each with its
own roofline figures and oracle check.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x52415745
WB = (2.0, 1.0, 1.5, 1.0)
CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
HBM_COPY_GBPS = 6290.0           # MI355X_MICROARCH.md: what a streaming copy reaches on this part ("~6.3 TB/s achievable")
BYTES_PER_PX = {"f32": 18, "f16": 10, "u8": 6, "rgb8": 5}   # 2 B CFA read + surface write


def parse_args(argv=None):
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
    ap.add_argument("--host", choices=["ranks", "node"], default="ranks",
                    help="ranks: one process per GPU + torch.distributed (the driver's contract; default).  node: ONE "
                         "process drives all GPUs through rd_node_batch_* (the C-ABI entry a Rust / C host uses; the "
                         "histogram all-reduce happens inside librawdev over RCCL)")
    ap.add_argument("--static-descriptors", action="store_true",
                    help="resubmit ONE frame array every step (librawdev then skips the descriptor upload); default: the "
                         "steps rotate through three arrays so every step uploads its descriptors")
    ap.add_argument("--no-alt-math", action="store_true",
                    help="skip the short extra run in the other math mode (reported under 'alt_math', N=1 only)")
    ap.add_argument("--no-extra", action="store_true", help="skip 'extra_configs' (config 1, RGBA8, config-5 shape; N=1 only)")
    ap.add_argument("--no-hist", action="store_true")
    ap.add_argument("--no-box", action="store_true", help="skip the box's own copy / fill ceiling (roofline.box_*, ~0.1 s before the headline)")
    ap.add_argument("--plane-stagger", type=int, default=-1,
                    help="A/B (profiles/r05_plane_stagger.txt, the launch-position question): -1 (default) = one torch allocation per CFA "
                         "plane; >= 0 = the planes are views into ONE arena at a pitch of (plane rounded up to 2 MiB) + this many bytes "
                         "(a multiple of 16; 0 = every plane starts 2 MiB-aligned)")
    ap.add_argument("--ring-arena", type=int, default=0, help="A/B: 1 = the output ring is one allocation too; 0 (default) = one torch allocation per surface")
    ap.add_argument("--data", choices=["uniform", "gradient"], default="uniform",
                    help="uniform: i.i.d. 12-bit samples (SURVEY 8d, the headline); gradient: smooth ramp + 1 %% noise "
                         "(SURVEY 8d's second distribution: flat regions, same-bin histogram atomics, less bit toggling)")
    ap.add_argument("--no-diagnose", action="store_true",
                    help="skip the self-diagnosis after the timed region (per-launch event pairs, clock under the kernel, board clocks, "
                         "box probes after the region, pattern probe, descriptor A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with N > 1 and no launcher environment
# ------------------------------------------------------------------------------------------------
VISIBILITY_VARS = ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")     # in the order the stack applies them


def physical_gpu_count():
    """GPUs THIS process could open, WITHOUT touching HIP or torch.cuda: the KFD topology nodes that have SIMDs (CPU nodes have
    none) and whose DRM render node can be opened -- sysfs shows every card of the host, a box that is a one-GPU slice of an
    eight-GPU machine must count one (ROCr skips the nodes it cannot open in the same way, so ROCR_VISIBLE_DEVICES indexes
    exactly these, in this order).  0 when nothing can be read (no driver: this container)."""
    import glob
    n = 0
    for path in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"), key=lambda q: int(q.split("/")[-2]) if q.split("/")[-2].isdigit() else 0):
        try:
            props = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
            if minor < 0:
                continue
            fd = os.open("/dev/dri/renderD%d" % minor, os.O_RDWR)     # a DRM file descriptor, closed at once: no HIP, no KFD queue
            os.close(fd)
            n += 1
        except (OSError, ValueError):
            continue
    return n


def narrow_visibility(env, local_rank, n_gpus=None):
    """One process per GPU, and every process sees ONLY its own GPU (the rule of round 6).

    `python -m torch.distributed.run --nproc-per-node N bench.py` starts N ranks that each see all N cards, and the first HIP
    call of every rank opens every card it sees: N processes on each GPU.  The pool's boxes kill a run with more than six
    processes on a card (the six-rank rehearsal of round 5 died that way), so before anything touches the GPU a rank narrows
    its own view to the card it will use -- the LOCAL_RANK-th entry of the visibility list that is already set
    (ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES, whichever the stack applies first), or ROCR_VISIBLE_DEVICES=<LOCAL_RANK> when none
    is -- and then uses device index 0.  A list with one entry is a launcher that has narrowed already: left alone.  With more
    ranks than cards (a gloo rehearsal on a one-GPU box) the entry is taken modulo the number of cards.
    RAWDEV_RANK_VISIBILITY=all keeps every card visible to every rank (rank r then uses index r): the usual torchrun
    arrangement, for a node whose RCCL refuses peers it cannot see.  The one-process host (`--host node`) is the fallback
    that puts exactly one process on every card whatever the launcher does.
    Mutates `env`; returns what it did (it goes into the bench line)."""
    mode = (env.get("RAWDEV_RANK_VISIBILITY") or "own").strip().lower()
    info = {"mode": mode, "variable": None, "value": None, "local_rank": local_rank}
    if mode != "own":
        info["note"] = "RAWDEV_RANK_VISIBILITY=%s: every rank sees every card the launcher shows it" % mode
        return info
    if env.get("RAWDEV_RANK_NARROWED") == "1":               # the self-launcher (child_env) has done it for this process
        for var in VISIBILITY_VARS:
            if env.get(var):
                info.update(variable=var, value=env[var])
                break
        info["note"] = "narrowed by the launcher (child_env)"
        return info
    for var in VISIBILITY_VARS:
        items = [x.strip() for x in (env.get(var) or "").split(",") if x.strip()]
        if not items:
            continue
        if len(items) == 1:
            info.update(variable=var, value=items[0], note="the launcher shows this rank one device already")
            return info
        env[var] = items[local_rank % len(items)]
        info.update(variable=var, value=env[var], note="entry %d of the launcher's list of %d" % (local_rank % len(items), len(items)))
        env["RAWDEV_RANK_NARROWED"] = "1"
        return info
    n = physical_gpu_count() if n_gpus is None else n_gpus
    idx = local_rank % n if n > 0 else local_rank
    env["ROCR_VISIBLE_DEVICES"] = str(idx)
    env["RAWDEV_RANK_NARROWED"] = "1"
    info.update(variable="ROCR_VISIBLE_DEVICES", value=str(idx), physical_gpus=n,
                note="no visibility list was set: card LOCAL_RANK" + (" modulo the machine's %d" % n if n > 0 else ""))
    return info


def child_env(base_env, rank, world, port, n_gpus=None):
    """Environment of rank `rank` of `world` single-node rank processes (what torch.distributed.run would export), narrowed
    to the rank's own GPU (narrow_visibility) unless RAWDEV_RANK_VISIBILITY=all."""
    env = dict(base_env)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "GROUP_RANK": "0", "ROLE_RANK": str(rank)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what this pool's host driver supports
    env.setdefault("OMP_NUM_THREADS", "1")
    env.pop("RAWDEV_RANK_NARROWED", None)
    narrow_visibility(env, rank, n_gpus)
    return env


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(world, argv, script=None):
    """Start `world` fresh rank processes of this script (never a re-exec of this process, which has not touched the GPU
    or torch.cuda), relay rank 0's stdout, return non-zero if any rank fails.  Ranks > 0 write to stderr."""
    port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(cmd, env=child_env(os.environ, r, world, port),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    import threading

    def relay():                                              # rank 0 prints the JSON line
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    rc, deadline, pending = 0, None, list(procs)
    try:
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or (code if code > 0 else 1)
                    if deadline is None:
                        deadline = time.time() + 20.0         # a rank died: the others cannot finish; stop them soon
            if deadline is not None and time.time() > deadline:
                for p in pending:
                    p.kill()                                  # exactly the PIDs started here
                deadline = float("inf")
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    t.join(timeout=5.0)
    return rc


# ------------------------------------------------------------------------------------------------
class quiet_gc:
    """No cyclic-GC pass inside a timed region: a generation-2 collection of a process that has torch imported takes
    20-30 ms (measured: one `--host node` step in ~13 took 40 ms instead of 20 -- the collection happened to fall into it;
    with one process per GPU the same pause hides behind queued work).  Collect before, disable inside, restore after."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


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
    # SURVEY 8d builds the baseline "-O3 -march=native, contraction off": that flavour is compiled on THIS machine
    # (oracle/Makefile `native`; -march=native code does not travel) and is bit-identical to the portable -O2 checker
    # (tests/test_oracle_kat.py).  If it cannot be built here the portable build is timed and the line says so.
    flavour = "-O3 -march=native -ffp-contract=off -fno-fast-math (built on this host)"
    try:
        L = ref_c.lib_native()
    except Exception as e:  # noqa: BLE001
        L = ref_c.lib()
        flavour = f"-O2 -mfma -ffp-contract=off (portable checker build; the native build failed: {e})"
    cp = cfa.ctypes.data_as(C.POINTER(C.c_uint16))
    fr, sec = C.c_int(), C.c_double()
    L.ref_bench_mt(cp, width, height, C.byref(u), cores, float(budget_s), 4096, C.byref(fr), C.byref(sec))
    frames, el = fr.value, sec.value
    # one-thread figure on a 256-row band of the same frame
    band_h = min(256, height)
    L.ref_bench_mt(cp, width, band_h, C.byref(u), 1, min(2.0, float(budget_s)), 64, C.byref(fr), C.byref(sec))
    one_thread = fr.value * width * band_h / 1e6 / sec.value
    # the portable -O2 build (what rounds 1-3 reported) beside it, briefly
    portable = None
    if L is not ref_c.lib():
        ref_c.lib().ref_bench_mt(cp, width, height, C.byref(u), cores, min(3.0, float(budget_s)), 4096, C.byref(fr), C.byref(sec))
        portable = round(fr.value * width * height / 1e6 / sec.value, 2)
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
            "build": flavour, "value_portable_O2_build": portable,
            "sample": f"{frames} x {width}x{height} frame(s), randomised stack, f32 surface, {el:.1f} s on {cores} "
                      f"persistent threads with first-touch row bands ({model}; {cores_note}); scalar f32 port of the shader, no SIMD: "
                      f"arithmetic-bound (one thread: {one_thread:.1f} MP/s), reported-only"}


def check_bands(fmt_name, W, H, cfa_t, p, surf_t, math_name, label):
    """Four row bands of one surface (a torch uint8 tensor on the device, sliced there) against the oracle, bit for bit."""
    import numpy as np
    from oracle import ref_c                       # the checker, never the thing measured
    from raweditor_amd import FIELDS
    math_mode = ref_c.MATH_CONTRACTED if math_name == "contracted" else ref_c.MATH_STRICT
    mid = (H // 2) | 1
    bands = [(0, min(6, H)), (max(0, min(1001, H - 6)), min(1007, H)), (mid, min(mid + 2, H)), (max(0, H - 6), H)]
    bpp = {"f32": 16, "f16": 8, "u8": 4, "rgb8": 3}[fmt_name]
    cfa = cfa_t.cpu().numpy().view(np.uint16)
    u = ref_c.make_uniforms({f: getattr(p, f) for f in FIELDS}, WB, CM, math_mode=math_mode)
    rows = surf_t.view(H, W * bpp)
    for r0, r1 in bands:
        exp = ref_c.render_band(cfa, u, r0, r1)
        raw = rows[r0:r1].cpu().numpy()
        if fmt_name == "f32":
            ok = np.array_equal(raw.view(np.uint32).reshape(r1 - r0, W, 4), exp.view(np.uint32))
        elif fmt_name == "f16":
            ok = np.array_equal(raw.view(np.uint16).reshape(r1 - r0, W, 4), ref_c.pack_f16(exp).view(np.uint16))
        elif fmt_name == "rgb8":
            ok = np.array_equal(raw.reshape(r1 - r0, W, 3), ref_c.pack_u8(exp)[..., :3])
        else:
            ok = np.array_equal(raw.reshape(r1 - r0, W, 4), ref_c.pack_u8(exp))
        if not ok:
            return False, f"{label} rows {r0}..{r1} differ from the oracle"
    return True, len(bands)


def verify_outputs(fmt_name, W, H, cfas, params_last, ring, n_frames, math_name):
    """Outside the timed region: two surfaces of the output ring as the timed passes left them, four row bands each,
    against the oracle (the histogram sum alone would pass for garbage pixels).  Slot s of the ring was last written by
    frame n_frames - len(ring) + s (frames go to slot i % len(ring)) with the stacks of the LAST step (`params_last`)."""
    nring = len(ring)
    checked, nb = [], 0
    for slot in sorted({0, nring - 1}):
        owners = [i for i in range(n_frames) if i % nring == slot]
        if not owners:
            continue
        i = owners[-1]
        ok, info = check_bands(fmt_name, W, H, cfas[i], params_last[i], ring[slot], math_name, f"frame {i} (ring slot {slot})")
        if not ok:
            return False, info
        nb = info
        checked.append(i)
    return True, f"frames {checked}: {nb} row bands each bit-identical to the oracle"


def pmc_traffic(fmt_name, W, H, mode):
    """HBM bytes per FRAME from the committed PMC profile for this surface format / frame size / launch mode, or None.
    NOT measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_pmc.sh + tools/parse_pmc.py)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            prof = json.load(fh)
        for ent in prof.get("entries", []):
            if ent.get("format") == fmt_name and list(ent.get("frame", [])) == [W, H] and ent.get("mode") == mode:
                return ent["hbm_bytes_per_frame"], (f"profiles/pmc_traffic.json [{fmt_name}, {W}x{H}, {mode}] (rocprofv3 PMC "
                                                    f"passes of {prof.get('tag', 'a committed profile')}), not measured by this run")
    except (OSError, ValueError, KeyError):
        pass
    return None, None



# ------------------------------------------------------------------------------------------------
# what ran where: per-rank identity + timings, gathered on rank 0 (the first real multi-GPU run cannot be repeated
# interactively, so its line must say by itself which devices the ranks were on and how each of them did)
# ------------------------------------------------------------------------------------------------
DIAG_ENV = ("HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "NCCL_DEBUG",
            "RCCL_MSCCL_ENABLE", "NCCL_SOCKET_IFNAME", "NCCL_P2P_DISABLE", "RAWDEV_DIST_BACKEND", "RAWDEV_RANK_VISIBILITY", "RD_BATCH_MAX_FRAMES",
            "RD_BATCH_PERSISTENT")


def rank_record(rank, local_rank, dev_index, ident, elapsed_s, dev_ms, steps, frames, width, height, launches_per_step, box):
    """One rank's own account of the timed region (its own wall clock and its own HIP events)."""
    wall_ms = elapsed_s * 1e3 / steps
    return {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "pci_bus_id": ident.get("pci_bus_id"),
            "name": ident.get("name"), "host": socket.gethostname(), "pid": os.getpid(),
            "ms_per_step": round(wall_ms, 4), "ms_per_step_hip_events": round(dev_ms / steps, 4),
            "launches": launches_per_step * steps, "us_per_frame": round(dev_ms * 1e3 / (steps * frames), 3),
            "MP_per_s": round(frames * width * height / 1e3 / wall_ms, 1),
            "box_copy_GBps": box.get("copy"), "box_fill_GBps": box.get("fill")}


def summarize_ranks(records, world_env, world_seen, backend):
    """rank 0's view of all ranks: (fields for the JSON line, error text or None).  Under nccl (= RCCL) two ranks on one
    device mean the run did not measure N GPUs: that is an error, not a number."""
    recs = sorted(records, key=lambda r: r["rank"])
    ids = [(r.get("host"), r.get("pci_bus_id") or f"index:{r.get('device_index')}") for r in recs]
    distinct = len(set(ids))
    mp = [r["MP_per_s"] for r in recs]
    out = {"ranks": recs, "world_size_seen": world_seen, "world_size_env": world_env, "distinct_devices": distinct,
           "per_gpu_MPps_min": min(mp), "per_gpu_MPps_max": max(mp),
           "per_gpu_MPps_spread": round((max(mp) - min(mp)) / max(mp), 4) if max(mp) > 0 else None,
           "env": {k: os.environ.get(k) for k in DIAG_ENV}}
    err = None
    if world_seen != world_env or len(recs) != world_env:
        err = f"world size mismatch: WORLD_SIZE={world_env}, process group reports {world_seen}, {len(recs)} rank records gathered"
    elif backend == "nccl" and distinct < len(recs):
        dup = sorted({i for i in ids if ids.count(i) > 1})
        err = (f"{len(recs)} ranks ran on {distinct} distinct device(s) (shared: {dup}): under nccl every rank must own its GPU; "
               "this run did not measure N GPUs")
    return out, err


def device_identity(dev_index):
    import ctypes as C
    from raweditor_amd import _lib
    buf = C.create_string_buffer(64)
    name = C.create_string_buffer(128)
    try:
        _lib.check(_lib.lib().rd_device_identity(dev_index, buf, len(buf), name, len(name)))
        return {"pci_bus_id": buf.value.decode() or None, "name": name.value.decode() or None}
    except Exception as e:  # noqa: BLE001
        return {"pci_bus_id": None, "name": f"unknown ({e})"}


def measure_box(ra, dev_index):
    """The box's own streaming ceilings, before the headline and outside its timed region ('measured
    hipMemcpyDtoD/stream-triad ceiling on the box'): librawdev's float4 copy / nt fill / read kernels over 1 GiB, median of 5."""
    try:
        t0 = time.perf_counter()
        c, f, r, m = ra.measure_hbm(dev_index, 1 << 30, 5)
        return {"copy": round(c, 1), "fill": round(f, 1), "read": round(r, 1), "memset": round(m, 1),
                "seconds": round(time.perf_counter() - t0, 2)}
    except Exception as e:  # noqa: BLE001
        return {"copy": None, "fill": None, "read": None, "memset": None, "error": str(e)}


def isa_budget(fmt_name):
    """Static VALU budget of the export kernel's main loop for this surface (profiles/isa_budget.json, written by
    tools/isa_budget.py --json from hipcc's own assembly of the bench workload's path), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "isa_budget.json")) as fh:
            return json.load(fh)["kernels"].get(fmt_name)
    except (OSError, ValueError, KeyError):
        return None


def measure_valu_ns(ra, dev_index):
    try:
        return ra.measure_valu(dev_index)
    except Exception:  # noqa: BLE001
        return None


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return 0.0 if not n else (xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2]))


def launch_summary(timeline, alg_bytes_per_launch=None, with_note=False):
    """[(call, start_us, end_us), ...] of rd_batch_launch_timeline -> what the launches of a step cost, by position in
    the step (a step = one develop call; the median over the kept calls), and what lies between them."""
    calls = {}
    for c, st, en in timeline:
        calls.setdefault(c, []).append((st, en))
    keys = sorted(calls)
    if not keys:
        return None
    npos = min(len(calls[k]) for k in keys)
    by_pos = [_median([calls[k][i][1] - calls[k][i][0] for k in keys]) for i in range(npos)]
    durs = [en - st for _, st, en in timeline]
    inner = [calls[k][i][0] - calls[k][i - 1][1] for k in keys for i in range(1, len(calls[k]))]
    boundary = [calls[keys[j]][0][0] - calls[keys[j - 1]][-1][1] for j in range(1, len(keys))]
    period = [calls[keys[j]][0][0] - calls[keys[j - 1]][0][0] for j in range(1, len(keys))]
    out = {"steps": len(keys), "launches_per_step": npos,
           "launch_us_by_position": [round(x, 1) for x in by_pos],
           "launch_us": {"min": round(min(durs), 1), "median": round(_median(durs), 1), "max": round(max(durs), 1),
                         "mean": round(sum(durs) / len(durs), 1)},
           "kernel_ms_per_step": round(sum(by_pos) / 1e3, 4),
           "gap_us_between_launches": {"median": round(_median(inner), 2), "max": round(max(inner), 2)} if inner else None,
           "step_boundary_gap_us": {"median": round(_median(boundary), 1), "max": round(max(boundary), 1)} if boundary else None,
           "instrumented_ms_per_step": round(_median(period) / 1e3, 4) if period else None}
    if with_note:
        out["note"] = ("an untimed pass AFTER the timed region with a HIP event pair around every fused launch (rd_batch_set_launch_timing): "
                       "launch_us_by_position = median over the steps of each launch's own duration; step_boundary_gap_us = last launch of a "
                       "step -> first launch of the next (histogram fold, all-reduce when N > 1, descriptor upload and the event packets).  A "
                       "pair's duration includes the command processor's handling of its two marker packets (about gap_us_between_launches), "
                       "which an un-instrumented stream hides under the previous launch: kernel_ms_per_step can therefore exceed ms_per_step "
                       "by ~1 % (a NEGATIVE gap_ms_per_step means: no idle time between launches worth the name)")
    if alg_bytes_per_launch:
        out["GBps_median_launch"] = round(alg_bytes_per_launch / (_median(durs) * 1e-6) / 1e9, 1)
    return out


def instrumented_pass(be, step_fn, sync_fn, n_steps, alg_bytes_per_launch=None, with_note=False):
    """n_steps of the caller's step with an event pair around every launch; None when the library refuses."""
    try:
        be.set_launch_timing(n_steps)
        for _ in range(n_steps):
            step_fn()
        sync_fn()
        tl = be.launch_timeline()
        be.set_launch_timing(0)
        return launch_summary(tl, alg_bytes_per_launch, with_note)
    except Exception as e:  # noqa: BLE001  (a diagnosis must not cost the line)
        try:
            be.set_launch_timing(0)
        except Exception:  # noqa: BLE001
            pass
        return {"error": f"{type(e).__name__}: {e}"}


# ------------------------------------------------------------------------------------------------
# board clocks / power / temperature while the workload runs (amdsmi if it initialises, else sysfs, else the reason)
# ------------------------------------------------------------------------------------------------
def _num(v):
    return v if isinstance(v, (int, float)) and not isinstance(v, bool) and 0 < v < 65535 else None


def open_clock_source(pci_bus_id):
    """-> (sample() -> {name: number}, source text) or (None, why not).  Never a new dependency: amdsmi is imported only
    if this image ships it, and every call is allowed to fail."""
    reasons = []
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        handles = list(amdsmi.amdsmi_get_processor_handles())
        h = None
        for cand in handles:
            try:
                if pci_bus_id and str(amdsmi.amdsmi_get_gpu_device_bdf(cand)).lower() == pci_bus_id.lower():
                    h = cand
            except Exception:  # noqa: BLE001
                continue
        if h is None and len(handles) == 1:
            h = handles[0]
        if h is None:
            raise RuntimeError(f"{len(handles)} amdsmi handles, none with BDF {pci_bus_id}")
        gfx_type = getattr(amdsmi.AmdSmiClkType, "GFX", amdsmi.AmdSmiClkType.SYS)

        def sample():
            out = {}
            try:
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                xs = [x for x in (m.get("current_gfxclks") or []) if _num(x)]
                if xs:
                    out["sclk_MHz"] = _median(xs)
                for src, dst in (("current_gfxclk", "sclk_MHz"), ("current_uclk", "mclk_MHz"), ("current_socket_power", "power_W"),
                                 ("average_socket_power", "power_W"), ("temperature_hotspot", "temp_hotspot_C"),
                                 ("temperature_mem", "temp_mem_C"), ("average_gfx_activity", "gfx_activity_pct"),
                                 ("average_umc_activity", "mem_activity_pct")):
                    if dst not in out and _num(m.get(src)) is not None:
                        out[dst] = m[src]
            except Exception:  # noqa: BLE001
                pass
            for dst, fn in (("sclk_MHz", lambda: amdsmi.amdsmi_get_clock_info(h, gfx_type).get("clk")),
                            ("mclk_MHz", lambda: amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.MEM).get("clk")),
                            ("power_W", lambda: (lambda d: d.get("current_socket_power") if _num(d.get("current_socket_power")) else d.get("average_socket_power"))(amdsmi.amdsmi_get_power_info(h))),
                            ("temp_hotspot_C", lambda: amdsmi.amdsmi_get_temp_metric(h, amdsmi.AmdSmiTemperatureType.HOTSPOT, amdsmi.AmdSmiTemperatureMetric.CURRENT))):
                if dst in out:
                    continue
                try:
                    v = fn()
                    if _num(v) is not None:
                        out[dst] = v
                except Exception:  # noqa: BLE001
                    pass
            return out
        if not sample():
            raise RuntimeError("amdsmi initialised but returned no clock / power / temperature metric")
        return sample, "amdsmi"
    except Exception as e:  # noqa: BLE001
        reasons.append(f"amdsmi: {type(e).__name__}: {' '.join(str(e).split())[:160]}")
    try:
        import glob
        dev = None
        for d in sorted(glob.glob("/sys/class/drm/card*/device")):
            try:
                ue = open(os.path.join(d, "uevent")).read()
            except OSError:
                continue
            if pci_bus_id and f"PCI_SLOT_NAME={pci_bus_id}".lower() in ue.lower():
                dev = d
        if dev is None:
            raise RuntimeError(f"no /sys/class/drm/card*/device with PCI_SLOT_NAME={pci_bus_id}")

        def dpm(name):
            for line in open(os.path.join(dev, name)).read().splitlines():
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            return None

        def first(patterns, scale):
            for pat in patterns:
                for f in sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*", pat))):
                    try:
                        return float(open(f).read().strip()) * scale
                    except (OSError, ValueError):
                        continue
            return None

        def sample():
            out = {}
            for dst, fn in (("sclk_MHz", lambda: dpm("pp_dpm_sclk")), ("mclk_MHz", lambda: dpm("pp_dpm_mclk")),
                            ("power_W", lambda: first(("power1_input", "power1_average"), 1e-6)),
                            ("temp_hotspot_C", lambda: first(("temp2_input", "temp1_input"), 1e-3))):
                try:
                    v = fn()
                    if _num(v) is not None:
                        out[dst] = v
                except Exception:  # noqa: BLE001
                    pass
            return out
        if not sample():
            raise RuntimeError("sysfs files present but unreadable")
        return sample, "sysfs (/sys/class/drm/card*/device: pp_dpm_sclk, pp_dpm_mclk, hwmon)"
    except Exception as e:  # noqa: BLE001
        reasons.append(f"sysfs: {type(e).__name__}: {' '.join(str(e).split())[:160]}")
    return None, "; ".join(reasons)


class ClockSampler:
    """Samples the source every few milliseconds on a thread while the caller keeps the GPU busy (the librawdev calls
    release the GIL)."""

    def __init__(self, sample_fn, period_s=0.004):
        import threading
        self.fn, self.period, self.rows = sample_fn, period_s, []
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            try:
                r = self.fn()
                if r:
                    self.rows.append(r)
            except Exception:  # noqa: BLE001
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join(timeout=2.0)
        return False

    def summary(self):
        out = {"samples": len(self.rows)}
        for k in sorted({k for r in self.rows for k in r}):
            xs = [r[k] for r in self.rows if k in r]
            out[k] = {"median": round(_median(xs), 1), "min": round(min(xs), 1), "max": round(max(xs), 1)}
        return out


def valu_fields(fmt_name, W, H, us_per_frame, valu_ns, n_simd=1024):
    """The OTHER roofline of the export kernel: VALU issue.  issue time per frame = the kernel's static issue budget per tile
    (profiles/isa_budget.json, in cycles where a full-rate instruction costs 2) x tiles per frame / SIMDs, priced with what
    a full-rate instruction costs a SIMD of this device in this run (rd_measure_valu, `valu_ns`); valu_issue_frac = that /
    the measured time per frame.  Reported BESIDE the HBM fraction, never instead of it."""
    b = isa_budget(fmt_name)
    if not b or not valu_ns:
        return {"valu_issue_frac": None, "valu_note": "profiles/isa_budget.json or rd_measure_valu unavailable"}
    # tiles per frame as librawdev cuts them (rawdev.hip, rd_tiles_per_unit): 64-quad tiles; 62 owned quads per tile for the
    # f32 surface's shifted-window tiling (W % 4 != 0, W >= 128)
    tiles = (H // 2 + 1) * ((W // 2 + 61) // 62 if (fmt_name == "f32" and W % 4 and W >= 128) else (W // 2 + 63) // 64)
    issue_us = b["issue_cycles"] / 2.0 * valu_ns * tiles / n_simd / 1e3
    return {"valu_issue_frac": round(issue_us / us_per_frame, 4), "valu_issue_us_per_frame": round(issue_us, 2),
            "valu_issue_cycles_per_tile": b["issue_cycles"], "valu_instructions_per_tile": b["valu_instructions"],
            "valu_ns_per_full_rate_instruction": round(valu_ns, 4), "valu_effective_GHz": round(2.0 / valu_ns, 3),
            "valu_source": f"profiles/isa_budget.json [{b['kernel']}] (static count from hipcc's assembly, bench workload's path) priced with "
                           "rd_measure_valu of this run (a v_mul / v_add loop at full occupancy, right after the timed region); a static "
                           "issue bound: stalls, LDS and memory waits come on top, and the part may clock a memory-bound kernel higher "
                           "than the calibration loop: read the fraction as +-5 % (it can exceed 1 by a few per cent), for the f32 surface as an upper estimate"}


def bound_of(fmt_name, hbm_frac, valu):
    """`roofline.bound` as DECLARED: the roofline this surface has been found on over four rounds of A/B runs -- the narrow
    surfaces are VALU-issue-bound (their HBM traffic is ~1.0x algorithmic and they speed up with fewer instructions, not with
    fewer bytes): "valu", as VERDICT round 3 asked; the f32 surface -- the headline, whose achieved / peak are HBM figures -- is
    "hbm".  What THIS run's two fractions say is `bound_measured` (bound_measured_of below)."""
    return "hbm" if fmt_name == "f32" else "valu"


def bound_measured_of(achieved_GBps, box_copy_GBps, valu):
    """Derived from the run, not declared: the HBM fraction against what this box's memory system delivers (its measured copy
    rate; the guide's 6290 GB/s when the run has no probe) beside the VALU issue fraction.  The larger one names the bound;
    within 5 % of each other both do."""
    ceiling = box_copy_GBps or HBM_COPY_GBPS
    h = achieved_GBps / ceiling
    v = (valu or {}).get("valu_issue_frac")
    if v is None:
        return {"bound_measured": None, "bound_measured_note": "no VALU figure in this run"}
    which = "hbm+valu" if abs(h - v) <= 0.05 else ("hbm" if h > v else "valu")
    return {"bound_measured": which,
            "bound_measured_note": f"hbm {h:.3f} of {'this box copy rate' if box_copy_GBps else 'the guide copy figure'} "
                                   f"({ceiling:.0f} GB/s) vs valu issue {v:.3f}; `bound` is the declared roofline"}


def alloc_ring(torch, dev, n, nbytes, arena=False):
    """n surfaces of nbytes each: n torch allocations (default), or views into one allocation at a 2 MiB-rounded pitch."""
    if not arena or n <= 1:
        return [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(n)]
    pitch = ((nbytes + (2 << 20) - 1) // (2 << 20)) * (2 << 20)
    buf = torch.empty(n * pitch + (2 << 20), dtype=torch.uint8, device=dev)
    base = (-buf.data_ptr()) % (2 << 20)
    return [buf[base + i * pitch: base + i * pitch + nbytes] for i in range(n)]


def make_batch(torch, np, ra, dev, W, H, F, first_index, stride, data="uniform", stagger=-1):
    """Synthetic frames generated on the device, keyed by (seed, global frame index = first_index + f * stride).
    stagger >= 0: the planes are views into one arena, plane f at f * (plane bytes rounded up to 2 MiB + stagger)."""
    cfas, params = [], []
    arena, pitch, base = None, 0, 0
    if stagger >= 0:
        assert stagger % 16 == 0, "--plane-stagger must be a multiple of 16 bytes"
        pitch = ((W * H * 2 + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + stagger
        arena = torch.empty(F * pitch + (2 << 20), dtype=torch.uint8, device=dev)
        base = (-arena.data_ptr()) % (2 << 20)                       # the arena's first 2 MiB boundary
    for f in range(F):
        gidx = first_index + f * stride
        g = torch.Generator(device=dev)
        g.manual_seed(SEED + gidx)
        def place(t):                                     # into the arena (same values), or as its own allocation
            if arena is None:
                return t
            view = arena[base + f * pitch: base + f * pitch + W * H * 2].view(torch.int16).view(H, W)
            view.copy_(t)
            return view
        if data == "uniform":
            cfas.append(place(torch.randint(0, 4096, (H, W), generator=g, device=dev, dtype=torch.int16)))
        else:                                             # a diagonal ramp whose slope and offset vary per frame, +-1 % noise
            yy = torch.arange(H, device=dev, dtype=torch.float32)[:, None] / H
            xx = torch.arange(W, device=dev, dtype=torch.float32)[None, :] / W
            a = 0.25 + 0.5 * ((gidx * 37) % 16) / 16.0
            ramp = (a * xx + (1.0 - a) * yy) * 3600.0 + 200.0
            noise = (torch.rand((H, W), generator=g, device=dev) - 0.5) * 2.0 * 40.96
            cfas.append(place((ramp + noise).clamp_(0, 4095).to(torch.int16)))
            del yy, xx, ramp, noise
        params.append(ra.EditParams.random(np.random.default_rng([SEED, gidx])))
    return cfas, params


def swapped_halves(params):
    """The same stacks, the two halves of the batch exchanged: different descriptors, identical total arithmetic."""
    n = len(params)
    return [params[(i + n // 2) % n] for i in range(n)] if n > 1 else list(params)


def rotated_stacks(params, k, m=3):
    """The same stacks, rotated by k/m of the batch: different descriptors, identical total arithmetic.  Three arrays in
    rotation defeat librawdev's descriptor cache (it keeps the last TWO arrays it was given and skips the upload when a call
    repeats one of them): round 5 found that rounds 3-4's "two alternating arrays" were both cached after the second step,
    so no step of their timed regions uploaded anything."""
    n = len(params)
    r = (k * n) // m
    return [params[(i + r) % n] for i in range(n)] if n > 1 else list(params)


def workload_label(W, H, world, F):
    if (W, H) == (6016, 4016):
        return "BASELINE configs[2]" if world == 1 else f"BASELINE configs[3] ({F} frames per GPU)"
    if (W, H) == (11648, 8736):
        return "BASELINE configs[4] shape (100 MP; per GPU)"
    return "custom frame size"


def result_line(args, world, F, W, H, elapsed, dev_ms, lpc, ring_len, verified, verified_note, host_note, descriptors_note, box=None,
                valu_ns=None, diag=None):
    total_px = float(world) * F * W * H * args.steps
    launches = args.steps * lpc
    launch_us = dev_ms * 1e3 / launches                    # avg fused-launch period incl. gaps and folds
    frame_us = dev_ms * 1e3 / (args.steps * F)
    alg_bytes = BYTES_PER_PX[args.format] * W * H * F / lpc
    achieved = alg_bytes / (launch_us * 1e-6) / 1e9        # GB/s
    multi = lpc < F * max(1, args.row_bands)
    per_frame, traffic_source = pmc_traffic(args.format, W, H, "multi" if multi else "per_frame")
    traffic = int(per_frame * F / lpc) if per_frame is not None else None       # per launch, like `achieved`
    box = box or {}
    diag = diag or {}
    valu = valu_fields(args.format, W, H, frame_us, valu_ns)
    ms_per_step = elapsed * 1e3 / args.steps
    launches = diag.get("launches") if isinstance(diag.get("launches"), dict) else None
    kernel_ms = (launches or {}).get("kernel_ms_per_step")
    med_us = ((launches or {}).get("launch_us") or {}).get("median")
    pattern = diag.get("pattern") if isinstance(diag.get("pattern"), dict) else None
    pattern_GBps = (pattern or {}).get("GBps")

    def ghz(ns):
        return round(2.0 / ns, 3) if ns else None
    box_before = dict({k: box.get(k) for k in ("copy", "fill", "read", "memset")}, valu_ns_per_full_rate_instruction=diag.get("valu_before_ns"),
                      valu_effective_GHz=ghz(diag.get("valu_before_ns")))
    ba = diag.get("box_after") or {}
    box_after = dict({k: ba.get(k) for k in ("copy", "fill", "read", "memset")},
                     valu_ns_per_full_rate_instruction=diag.get("valu_after_ns"), valu_effective_GHz=ghz(diag.get("valu_after_ns")),
                     valu_effective_GHz_after_the_probes=ghz(diag.get("valu_after_probes_ns")),
                     note="valu_*: rd_measure_valu right after the timed region (the clocks are up); copy / fill / read / memset: rd_measure_hbm "
                          "after the diagnostic passes that follow it")
    return {
        "metric": "megapixels/sec through demosaic+10-slider pipeline; 24MP batch",
        "value": round(total_px / 1e6 / elapsed, 1),
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        # where a step's time goes: the sum of its launches' own durations (instrumented pass after the region) and the rest
        "kernel_ms_per_step": kernel_ms,
        "gap_ms_per_step": round(ms_per_step - kernel_ms, 4) if kernel_ms else None,
        "clock_under_kernel_GHz": diag.get("clock_under_kernel"),
        "clocks": diag.get("clocks"), "clocks_reason": diag.get("clocks_reason"),
        "descriptor_upload_ab": diag.get("descriptor_upload_ab"),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",                                    # the arithmetic type of the path for EVERY surface format
        "surface_dtype": args.format,                      # what is stored (`--format`)
        "data": "synthetic" if args.data == "uniform" else "synthetic (gradient + 1 % noise)",
        "verified": verified,
        "verified_note": verified_note,
        "config": {
            "workload": f"{workload_label(W, H, world, F)}: batch {F} x {W}x{H} synthetic RGGB u16 per GPU, randomised "
                        f"10-slider stacks, RGBA-{args.format} surface, fused histogram={'off' if args.no_hist else 'on'}, "
                        f"{args.math} f32 arithmetic",
            "frames_per_gpu": F, "width": W, "height": H, "surface": f"rgba_{args.format}",
            "row_bands": args.row_bands, "out_ring": ring_len, "math_mode": args.math, "launches_per_step": lpc,
            "host": host_note, "descriptors": descriptors_note,
            "parallelism": f"frames sharded round-robin over {world} GPU(s); RCCL all-reduce of the u64[768] histogram only",
        },
        "roofline": {
            "bound": bound_of(args.format, achieved / HBM_PEAK_GBPS, valu), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, **valu,
            **bound_measured_of(achieved, box.get("copy"), valu),
            "traffic_source": traffic_source,
            "frac_of_copy_ceiling": round(achieved / HBM_COPY_GBPS, 4),      # reported beside `frac`, never instead of it
            # this box, this run, before the headline (rd_measure_hbm: float4 copy / nt fill / read of 1 GiB, median of 5)
            "box_copy_GBps": box.get("copy"), "box_fill_GBps": box.get("fill"), "box_read_GBps": box.get("read"),
            "box_memset_GBps": box.get("memset"),
            "frac_of_box_copy": round(achieved / box["copy"], 4) if box.get("copy") else None,
            "frac_of_box_fill": round(achieved / box["fill"], 4) if box.get("fill") else None,
            "box_note": "the guide's 6290 GB/s copy figure stays in frac_of_copy_ceiling; box_* are measured on this device in this run "
                        "(copy counts bytes read + written; this kernel writes 16 of its 18 B/px, so it sits between the copy and the fill ceiling)",
            "kernel": "rd_develop_batch" if multi else "rd_develop_quads",
            "launch_us": round(launch_us, 2), "frames_per_launch": round(F / lpc, 3), "us_per_frame": round(frame_us, 2),
            "launch_us_note": "HIP-event time of the timed region / fused launches: an average launch PERIOD that "
                              "includes inter-launch gaps, descriptor uploads and the histogram folds (conservative)",
            "algorithmic_bytes_per_launch": int(alg_bytes),
            # the same fraction from the MEDIAN launch's own duration (event pair around each launch, instrumented pass): what
            # the kernel does when nothing stands between launches; `frac` above is the timed region's average launch PERIOD
            "frac_kernel": round(alg_bytes / (med_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if med_us else None,
            "launches": launches if launches is not None else diag.get("launches"),
            "box_before": box_before, "box_after": box_after,
            # the memory pattern's own ceiling: this kernel's launches with the arithmetic removed, same buffers (rd_batch_probe_pattern)
            "box_pattern_GBps": pattern_GBps,
            "frac_of_box_pattern": round(achieved / pattern_GBps, 4) if pattern_GBps else None,
            "box_pattern": pattern if pattern is not None else diag.get("pattern"),
            # (copies of the line's top-level diagnosis keys: a record that keeps `roofline` whole keeps them too)
            "kernel_ms_per_step": kernel_ms, "gap_ms_per_step": round(ms_per_step - kernel_ms, 4) if kernel_ms else None,
            "clock_under_kernel_GHz": (diag.get("clock_under_kernel") or {}).get("GHz_median"),
            "clocks": {k: (v.get("median") if isinstance(v, dict) else v) for k, v in (diag.get("clocks") or {}).items()
                       if k in ("sclk_MHz", "mclk_MHz", "power_W", "temp_hotspot_C", "temp_mem_C", "gfx_activity_pct", "samples", "source")} or None,
            "clocks_reason": diag.get("clocks_reason"),
            "descriptor_upload_cost_ms_per_step": (diag.get("descriptor_upload_ab") or {}).get("upload_cost_ms_per_step"),
        },
    }


# ------------------------------------------------------------------------------------------------
# extra_configs (N = 1): the other BASELINE configurations, in the driver-timed record
# ------------------------------------------------------------------------------------------------
def roofline_of(fmt_name, W, H, us_per_frame, mode, kernel, valu_ns=None):
    alg = BYTES_PER_PX[fmt_name] * W * H
    ach = alg / (us_per_frame * 1e-6) / 1e9
    per_frame, src = pmc_traffic(fmt_name, W, H, mode)
    valu = valu_fields(fmt_name, W, H, us_per_frame, valu_ns) if kernel == "rd_develop_batch" else {}
    return {"bound": bound_of(fmt_name, ach / HBM_PEAK_GBPS, valu) if valu else ("hbm" if fmt_name == "f32" else "valu"),
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), **valu,
            **(bound_measured_of(ach, None, valu) if valu else {}),
            "traffic": int(per_frame) if per_frame is not None else None, "traffic_source": src,
            "traffic_unit": "HBM bytes per frame", "kernel": kernel, "algorithmic_bytes_per_frame": alg}


def extra_single_frame(torch, np, ra, dev, dev_index, cfa_t, p, stream, iters=60):
    """BASELINE configs[1]: ONE 24 MP frame, full 10-slider develop, f32 surface + fused histogram: rd_render_device
    (one fused launch + the histogram fold) + synchronise per iteration -- the latency an interactive caller sees."""
    import statistics
    H, W = cfa_t.shape
    out = torch.empty(H * W * 16, dtype=torch.uint8, device=dev)
    hist = torch.zeros(768, dtype=torch.int32, device=dev)
    pipe = ra.RenderPipeline.from_device(1, cfa_t.data_ptr(), W, H, p, WB, CM, device=dev_index)
    host_us, kern_us = [], []
    with torch.cuda.stream(stream):
        for it in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            pipe.render_device(W, H, ra.FMT_RGBA_F32, out.data_ptr(), hist.data_ptr(), stream.cuda_stream)
            e1.record(stream)
            stream.synchronize()
            host_us.append((time.perf_counter() - t0) * 1e6)
            kern_us.append(e0.elapsed_time(e1) * 1e3)
    host_us, kern_us = host_us[10:], kern_us[10:]
    med, kmed = statistics.median(host_us), statistics.median(kern_us)
    ok, info = check_bands("f32", W, H, cfa_t, p, out, "strict", "single frame")
    ok = ok and int(hist.sum().item()) == 3 * W * H
    # "randomised + default stacks" -- the same loop with EditParams::default() (edit.rs:81-95)
    pipe.update_uniforms(ra.EditParams())
    dflt = []
    with torch.cuda.stream(stream):
        for it in range(iters):
            t0 = time.perf_counter()
            pipe.render_device(W, H, ra.FMT_RGBA_F32, out.data_ptr(), hist.data_ptr(), stream.cuda_stream)
            stream.synchronize()
            dflt.append((time.perf_counter() - t0) * 1e6)
    dmed = statistics.median(dflt[10:])
    ok_d, _ = check_bands("f32", W, H, cfa_t, ra.EditParams(), out, "strict", "single frame, default stack")
    ok = ok and ok_d
    pipe.close()
    return {"config": "BASELINE configs[1]: single 24 MP RGGB frame, full 10-slider develop (randomised stack; the default stack beside it), "
                      "f32 surface + fused histogram, rd_render_device + synchronise per iteration",
            "iterations": len(host_us), "ms": round(med / 1e3, 5), "ms_min": round(min(host_us) / 1e3, 5),
            "MP_per_s": round(W * H / med, 1), "enqueue_us": round(kmed, 2),
            "ms_default_stack": round(dmed / 1e3, 5), "MP_per_s_default_stack": round(W * H / dmed, 1),
            "latency_note": "ms = median host-side time of launch + histogram fold + stream synchronise; enqueue_us = median HIP-event "
                            "time of the same enqueue on its stream: the fused launch + the histogram fold + the gap between them (the roofline figure uses it: conservative; the launch alone is in profiles/r03_kernel_stats.csv)",
            "roofline": roofline_of("f32", W, H, kmed, "per_frame", "rd_develop_quads"),
            "verified": bool(ok), "verified_note": f"{info} row bands bit-identical to the oracle, histogram counts every pixel" if ok else str(info)}


def extra_batch(torch, np, ra, dev, dev_index, fmt_name, cfas, params, W, H, ring_n, row_bands, steps, stream, label, kernel_mode,
                tiled=False, valu_ns=None, launch_diag=False):
    """A batch workload on another surface format / frame size, timed like the headline (HIP events, descriptors alternate).
    tiled: RD_BATCH_PERSISTENT=0 for this context -- every frame is `row_bands` separate row-band launches (BASELINE
    config 5's "tiled multi-launch per frame"); by default the multi-frame launch needs no bands of its own and ignores them."""
    fmt = {"f32": ra.FMT_RGBA_F32, "f16": ra.FMT_RGBA_F16, "u8": ra.FMT_RGBA_U8, "rgb8": ra.FMT_RGB_U8}[fmt_name]
    bpp = ra.BYTES_PER_PIXEL[fmt]
    F = len(cfas)
    ring = alloc_ring(torch, dev, ring_n, H * W * bpp)
    hist = torch.zeros(768, dtype=torch.int64, device=dev)
    saved = {k: os.environ.get(k) for k in ("RD_BATCH_PERSISTENT", "RD_BATCH_STREAMS")}
    if tiled:
        os.environ["RD_BATCH_PERSISTENT"] = "0"              # read by rd_batch_create
        os.environ.setdefault("RD_BATCH_STREAMS", "2")       # the band launches alternate between two streams, one workgroup per CU each: they run side by side (round 6: -6 %)
    try:
        be = ra.BatchExporter(dev_index, W, H, fmt, True)
    finally:
        if tiled:
            for k, v in saved.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
    variants = [rotated_stacks(params, k) for k in range(3)]
    arrays = [be.make_frames([c.data_ptr() for c in cfas], [ring[i % ring_n].data_ptr() for i in range(F)], v, WB, CM)
              for v in variants]
    with torch.cuda.stream(stream):
        k = 0
        for _ in range(2):
            be.develop(arrays[k % 3], row_bands=row_bands, stream=stream.cuda_stream); k += 1
        be.histogram(hist.data_ptr(), stream=stream.cuda_stream)
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(steps):
            be.develop(arrays[k % 3], row_bands=row_bands, stream=stream.cuda_stream); k += 1
            be.histogram(hist.data_ptr(), stream=stream.cuda_stream)
        e1.record(stream)
        stream.synchronize()
    ms = e0.elapsed_time(e1)
    us = ms * 1e3 / (steps * F)
    lpc = max(1, be.last_launch_count())
    launches = None
    if launch_diag:                                          # the same per-launch account as the headline's (outside the timed steps)
        kk = [k]

        def one_step():
            be.develop(arrays[kk[0] % 3], row_bands=row_bands, stream=stream.cuda_stream)
            kk[0] += 1
            be.histogram(hist.data_ptr(), stream=stream.cuda_stream)
        with torch.cuda.stream(stream):
            launches = instrumented_pass(be, one_step, stream.synchronize, 4, BYTES_PER_PX[fmt_name] * W * H * F / lpc)
        k = kk[0]
    ok = int(hist.sum().item()) == 3 * F * W * H
    note = "histogram does not count every pixel"
    if ok:
        ok, note = verify_outputs(fmt_name, W, H, cfas, variants[(k - 1) % 3], ring, F, "strict")
    be.close()
    del ring
    return {"config": label, "frames": F, "steps": steps, "ms": round(ms / steps, 4), "us_per_frame": round(us, 2),
            "MP_per_s": round(W * H / us, 1), "launches_per_step": lpc, "out_ring": ring_n, "row_bands_requested": row_bands,
            # what ran: the multi-frame launch sweeps a frame in row order and cuts no bands of its own
            "row_bands_effective": row_bands if lpc >= F * max(1, row_bands) else 1,
            "frames_per_launch": round(F / lpc, 3),
            "kernel_ms_per_step": (launches or {}).get("kernel_ms_per_step"),
            "gap_ms_per_step": round(ms / steps - launches["kernel_ms_per_step"], 4) if (launches or {}).get("kernel_ms_per_step") else None,
            "launches": launches,
            "roofline": roofline_of(fmt_name, W, H, us, kernel_mode, "rd_develop_quads" if lpc >= F else "rd_develop_batch", valu_ns),
            "verified": bool(ok), "verified_note": note}



def extra_full_res_to_bytes(torch, np, ra, dev, dev_index, cfa_t, p, iters=24):
    """The reference's own metric-path ENTRY as its caller invokes it: RenderPipeline::render_full_res_to_bytes
    (pipeline.rs:526-606, called from export_image_async, main.rs:1749-1754) -- kernel + read-back of the 96.6 MB RGBA8
    surface into host memory, per call, PCIe included.  Three destinations: page-locked (rd_host_alloc: direct DMA), a
    pageable buffer the caller reuses, and a fresh pageable buffer per call (what returning a new Vec<u8> costs: its
    first-touch page faults).  The PCIe floor beside them is measured here: one pinned D2H copy of the same bytes."""
    import statistics
    H, W = cfa_t.shape
    nbytes = W * H * 4
    pipe = ra.RenderPipeline.from_device(2, cfa_t.data_ptr(), W, H, p, WB, CM, device=dev_index)
    pin = ra.PinnedBytes(nbytes, dev_index)
    reused = np.empty(nbytes, np.uint8)
    reused[:] = 0                                             # touched once: its pages exist

    def run(make_dst):
        ms = []
        for it in range(iters + 4):
            dst = make_dst()
            t0 = time.perf_counter()
            out = pipe.render_full_res_to_bytes(out=dst)
            ms.append((time.perf_counter() - t0) * 1e3)
        ms = ms[4:]
        return out, statistics.median(ms), min(ms)

    res, ok_all, notes = {}, True, []
    for name, make_dst, limiter in (
            ("pinned_dst", lambda: pin.array, "PCIe (the DMA engine writes the caller's page-locked buffer directly)"),
            ("pageable_dst_reused", lambda: reused, "PCIe or the staging memcpy, whichever is slower on this host (8 MiB pinned slots -> caller's "
                                                    "buffer on RD_COPY_THREADS helper threads, overlapped with the next chunk's DMA)"),
            ("pageable_dst_fresh", lambda: None, "first-touch page faults of the new 96.6 MB buffer (taken inside the staging memcpy) -- the cost of "
                                                 "returning a fresh Vec<u8> per call, as the reference's signature does")):
        out, med, mn = run(make_dst)
        ok, info = check_bands("u8", W, H, cfa_t, p, torch.from_numpy(out), "strict", name)
        ok_all = ok_all and ok
        if not ok:
            notes.append(str(info))
        res[name] = {"ms": round(med, 3), "ms_min": round(mn, 3), "GBps_over_pcie": round(nbytes / med / 1e6, 1),
                     "MP_per_s": round(W * H / med / 1e3, 1), "limiter": limiter}
    # the allocation-free form: the pipeline lends a page-locked surface (rd_render_full_res_borrow)
    ms = []
    for it in range(iters + 4):
        t0 = time.perf_counter()
        with pipe.render_full_res_borrowed() as sfc:
            ms.append((time.perf_counter() - t0) * 1e3)
            if it == iters + 3:
                ok, info = check_bands("u8", W, H, cfa_t, p, torch.from_numpy(np.array(sfc.array, copy=True)), "strict", "borrowed")
                ok_all = ok_all and ok
                if not ok:
                    notes.append(str(info))
    ms = ms[4:]
    med = statistics.median(ms)
    res["borrowed_surface"] = {"ms": round(med, 3), "ms_min": round(min(ms), 3), "GBps_over_pcie": round(nbytes / med / 1e6, 1),
                               "MP_per_s": round(W * H / med / 1e3, 1),
                               "limiter": "PCIe; rd_render_full_res_borrow: the pipeline lends its own page-locked surface (no allocation, "
                                          "no page faults, no host copy on the caller's side)"}
    # the floor: the same bytes, device -> page-locked host, one hipMemcpyAsync on a stream, nothing else
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    fl = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        fl.append((time.perf_counter() - t0) * 1e3)
    floor = statistics.median(fl[2:])
    pipe.close()
    pin.free()
    del src, dst
    return {"config": "RenderPipeline::render_full_res_to_bytes as export_image_async calls it (pipeline.rs:526-606, main.rs:1749-1754): "
                      f"one {W}x{H} frame, RGBA8, kernel + read-back into host memory per call (PCIe-inclusive: never `value`)",
            "iterations": iters, "bytes": nbytes, **res,
            "pcie_floor_ms": round(floor, 3), "pcie_floor_GBps": round(nbytes / floor / 1e6, 1),
            "pcie_floor_note": "one page-locked D2H copy of the same 96.6 MB, measured in this run",
            "reference_published": "\"SLOW (1-2 seconds for 24MP)\", pipeline.rs:525 / main.rs:1752-1753, unspecified hardware",
            "verified": bool(ok_all), "verified_note": "4 row bands of every destination bit-identical to the oracle" if ok_all else "; ".join(notes)}


def extra_export_ring(torch, np, ra, dev, dev_index, cfas, params, n_frames=48):
    """the export feed for a STREAM of frames (rd_exporter_*: kernel of frame i+1 under the D2H of frame i,
    pinned ring), RGB8 = the reference's JPEG path with its CPU alpha strip (main.rs:1777-1786) fused into the kernel."""
    H, W = cfas[0].shape
    out = {}
    for name, fmt, fname in (("rgb8", ra.FMT_RGB_U8, "rgb8"), ("rgba8", ra.FMT_RGBA_U8, "u8")):
        ex = ra.Exporter(dev_index, W, H, fmt, n_slots=2)          # two slots: the copy of frame i under the kernel of frame i+1, nothing queued deeper (deeper rings measured slower)
        frames = [ex.frame(cfas[i % len(cfas)].data_ptr(), params[i % len(cfas)], WB, CM) for i in range(n_frames)]
        for _ in ex.export(frames[:6]):
            pass
        keep = None
        t0 = time.perf_counter()
        n, dt = 0, None
        for i, surf in ex.export(frames):
            n += 1
            if i == n_frames - 1:
                dt = time.perf_counter() - t0                # the last surface is in host memory: the clock stops here,
                keep = np.array(surf, copy=True)             # the copy for the oracle check is not part of the ring's work
        i_last = (n_frames - 1) % len(cfas)
        ok, info = check_bands(fname, W, H, cfas[i_last], params[i_last], torch.from_numpy(keep.reshape(-1)), "strict", f"export ring {name}")
        nbytes = W * H * ra.BYTES_PER_PIXEL[fmt]
        out[name] = {"ms_per_frame": round(dt / n * 1e3, 3), "frames_per_s": round(n / dt, 1), "MP_per_s": round(n * W * H / dt / 1e6, 1),
                     "GBps_over_pcie": round(n * nbytes / dt / 1e9, 1), "frames": n, "slots": 2,
                     "verified": bool(ok), "verified_note": f"{info} row bands of the last frame bit-identical to the oracle" if ok else str(info)}
        ex.close()
    # The same ring fed from HOST memory (rd_exporter_submit_host: the decoded files of a folder export, raw/loader.rs:11-19):
    # upload of frame i+1, kernel of frame i+1 and read-back of frame i overlap; page-locked planes and pageable ones.
    pins, ex = [], None
    try:
        ex = ra.Exporter(dev_index, W, H, ra.FMT_RGB_U8, n_slots=2)
        k = min(4, len(cfas))
        host = [np.ascontiguousarray(cfas[i].cpu().numpy()).view(np.uint16) for i in range(k)]
        for _ in range(k):
            pins.append(ra.PinnedBytes(W * H * 2, device=dev_index))
        for pin, a in zip(pins, host):
            pin.array.view(np.uint16)[:] = a.reshape(-1)
        for name, planes in (("rgb8_from_pinned_host", [pin.array.view(np.uint16) for pin in pins]), ("rgb8_from_pageable_host", host)):
            feed = [(planes[i % k], ex.frame(0, params[i % k], WB, CM)) for i in range(n_frames)]
            for _ in ex.export_host(feed[:6]):
                pass
            keep, dt, n = None, None, 0
            t0 = time.perf_counter()
            for i, surf in ex.export_host(feed):
                n += 1
                if i == n_frames - 1:
                    dt = time.perf_counter() - t0
                    keep = np.array(surf, copy=True)
            i_last = (n_frames - 1) % k
            ok, info = check_bands("rgb8", W, H, cfas[i_last], params[i_last], torch.from_numpy(keep.reshape(-1)), "strict", f"export ring {name}")
            out[name] = {"ms_per_frame": round(dt / n * 1e3, 3), "frames_per_s": round(n / dt, 1),
                         "GBps_up_plus_down": round(n * (W * H * 2 + W * H * 3) / dt / 1e9, 1), "frames": n, "slots": 2,
                         "verified": bool(ok), "verified_note": f"{info} row bands of the last frame bit-identical to the oracle" if ok else str(info)}
    except Exception as exc:                                      # an extra must not cost the line
        out["rgb8_from_host_error"] = f"{type(exc).__name__}: {exc}"
    finally:                                                      # on every path: the ring, then the page-locked planes it read
        if ex is not None:
            ex.close()
        for pin in pins:
            pin.free()
    out["config"] = (f"export ring (rd_exporter_*): {n_frames} x {W}x{H} frames resident in HBM -> fused develop -> pinned host ring, "
                     "PCIe-inclusive; RGB8 = JPEG feed (alpha strip fused), RGBA8 = PNG feed; *_from_*_host: the CFA planes "
                     "start in host memory too (48 MB up + 72 MB down per frame)")
    return out

def extra_ragged_width(torch, np, ra, dev, dev_index, cfas, params, stream, valu_ns=None, Wr=6000, Hr=4000):
    """A frame width that is not a multiple of the export kernel's 128-pixel tile -- 6000 x 4000, what most 24 MP cameras
    make (the reference renders any size, shaders.rs:181-187) -- beside 6016 x 4016: the same 64-frame batch per surface,
    the two sizes alternating three times on this box, best of each; `ns_per_px_ratio` = ragged / aligned time per PIXEL.
    Round 6: the same for an ODD width (6001 x 4001: whole quads by the export kernel, the last column by rd_develop_lastcol;
    CFA rows start on odd 16-bit boundaries; f32 rows off the 64-byte blocks: shifted store windows)."""
    nf = min(64, len(cfas))
    cr, pr = make_batch(torch, np, ra, dev, Wr, Hr, nf, 1 << 21, 1)
    ca, pa = cfas[:nf], params[:nf]
    out = {"config": f"{nf} x {Wr}x{Hr} (W % 128 = {Wr % 128}: every row pair ends in a pulled-back, overlapping tile; RGBA8 / RGB8 rows are "
                     f"{Wr * 4} / {Wr * 3} bytes, not whole 128-byte lines" + ("; ODD width: one more launch per multi-frame launch for the last "
                     "column, 2-byte aligned CFA rows; the f32 surface stores through 64-byte-aligned shifted windows (RD_TILES_SHIFT)" if Wr % 2 else "") + f") beside {nf} x 6016x4016, "
                     "randomised stacks, fused histogram, strict f32 arithmetic, multi-frame launches"}
    worst = 0.0
    for fmt_name in ("f32", "f16", "u8", "rgb8"):
        ring = 8 if fmt_name == "f32" else 16
        best = {}
        for rep in range(3):
            for tag, (cc, pp, W, H) in (("aligned", (ca, pa, 6016, 4016)), ("ragged", (cr, pr, Wr, Hr))):
                r = extra_batch(torch, np, ra, dev, dev_index, fmt_name, cc, pp, W, H, ring, 1, 8, stream, f"{fmt_name} {W}x{H}", "multi",
                                valu_ns=valu_ns)
                if tag not in best or r["us_per_frame"] < best[tag]["us_per_frame"]:
                    best[tag] = r
        a, b = best["aligned"], best["ragged"]
        ratio = (b["us_per_frame"] / (Wr * Hr)) / (a["us_per_frame"] / (6016 * 4016))
        worst = max(worst, ratio)
        out[fmt_name] = {"aligned_us_per_frame": a["us_per_frame"], "ragged_us_per_frame": b["us_per_frame"],
                         "aligned_MP_per_s": a["MP_per_s"], "ragged_MP_per_s": b["MP_per_s"], "ns_per_px_ratio": round(ratio, 4),
                         "ragged_hbm_frac": b["roofline"]["frac"], "verified": bool(a["verified"] and b["verified"]),
                         "verified_note": b["verified_note"]}
    out["worst_ns_per_px_ratio"] = round(worst, 4)
    out["verified"] = all(out[f]["verified"] for f in ("f32", "f16", "u8", "rgb8"))
    del cr
    return out


def extra_text_pin(torch, np, ra, dev, dev_index, stream):
    """The batch path against the reference's SHADER TEXT, executed: tests/golden/wgsl_fullsize.json holds checksums of whole frames
    evaluated from /root/reference/src/gpu/shaders.rs by this repository's WGSL evaluator (tools/make_wgsl_fullsize.py, made in
    the build container; only the checksums travel).  The frames' inputs are re-drawn from their seeded PCG64 streams (checked
    against their own hash), developed by ONE rd_batch_develop call per size -- f32 surface for the 24 MP frames (aligned, ragged,
    odd width), binary16 for the 100 MP frame -- and the surfaces' SHA-256 and the accumulated histogram compared.  No oracle
    code runs here: hashes of what the device wrote against hashes in a data file."""
    import hashlib
    import zlib
    with open(os.path.join(ROOT, "tests", "golden", "wgsl_fullsize.json")) as fh:
        full = json.load(fh)
    groups = {}
    for fr in full["frames"]:
        groups.setdefault((fr["w"], fr["h"]), []).append(fr)
    out = {"source": "tests/golden/wgsl_fullsize.json (shader sha256 " + full["shader_sha256"][:16] + "...)", "frames": {}}
    results, skipped = [], False
    for (w, h), group in sorted(groups.items()):
        wide = w * h > 30_000_000
        fmt, key, dt = (ra.FMT_RGBA_F16, "sha256_f16", torch.uint8) if wide else (ra.FMT_RGBA_F32, "sha256_f32", torch.uint8)
        planes = []
        for fr in group:
            rng = np.random.default_rng([full["seed"], zlib.crc32(fr["name"].encode())])
            cfa = rng.integers(0, 4096, (h, w), dtype=np.uint16)
            if hashlib.sha256(cfa.tobytes()).hexdigest() != fr["sha256_cfa"]:
                out["frames"][fr["name"]] = "inputs differ (this numpy draws another plane): not compared"
                skipped, planes = True, None
                break
            planes.append(torch.from_numpy(cfa.view(np.int16)).to(dev))
        if planes is None:
            continue
        bpp = ra.BYTES_PER_PIXEL[fmt]
        outs = [torch.empty(h * w * bpp, dtype=dt, device=dev) for _ in group]
        hist = torch.zeros(768, dtype=torch.int64, device=dev)
        be = ra.BatchExporter(dev_index, w, h, fmt, True)
        frames = be.make_frames([c.data_ptr() for c in planes], [o.data_ptr() for o in outs], [ra.EditParams(**fr["params"]) for fr in group],
                                full["wb"], group[0]["cm"])
        for f, fr in zip(frames, group):
            f.color_matrix[:] = fr["cm"]
        with torch.cuda.stream(stream):
            be.develop(frames, stream=stream.cuda_stream)
            be.histogram(hist.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
        want_hist = np.sum([np.asarray(fr["histogram"], np.int64) for fr in group], axis=0)
        hist_ok = bool(np.array_equal(hist.cpu().numpy(), want_hist))
        for fr, o in zip(group, outs):
            ok = hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest() == fr[key]
            out["frames"][fr["name"]] = {"surface": "f16" if wide else "f32", "sha256_matches": bool(ok), "histogram_matches": hist_ok}
            results.append(bool(ok) and hist_ok)
        be.close()
        del planes, outs
    out["verified"] = (all(results) if results and not skipped else (False if results and not all(results) else None))
    return out


def extra_configs(torch, np, ra, dev, dev_index, cfas, params, stream, valu_ns=None):
    out = {}
    t0 = time.perf_counter()
    try:
        out["text_pin"] = extra_text_pin(torch, np, ra, dev, dev_index, stream)
    except Exception as e:  # noqa: BLE001  (a data file or numpy difference must not cost the other extras)
        out["text_pin"] = {"verified": None, "error": f"{type(e).__name__}: {e}"}
    out["single_frame_f32"] = extra_single_frame(torch, np, ra, dev, dev_index, cfas[0], params[0], stream)
    out["full_res_to_bytes"] = extra_full_res_to_bytes(torch, np, ra, dev, dev_index, cfas[1], params[1])
    out["export_ring"] = extra_export_ring(torch, np, ra, dev, dev_index, cfas[:8], params[:8])
    n8 = len(cfas)
    out["batch_rgba8"] = extra_batch(torch, np, ra, dev, dev_index, "u8", cfas[:n8], params[:n8], 6016, 4016, 32, 1, 6, stream,
                                     f"the reference's own surface (Rgba8Unorm, pipeline.rs:322) on the batch workload: {n8} x 6016x4016, "
                                     "randomised stacks, fused histogram, strict f32 arithmetic", "multi", valu_ns=valu_ns, launch_diag=True)
    W5, H5 = 11648, 8736
    # BASELINE configs[4] is 512 frames over 8 GPUs = 64 per GPU: that many when the headline batch is the full one
    # (13 GB of planes + a ring of 4 surfaces; the reduced batches of the tests keep 16)
    n5 = 64 if len(cfas) >= 256 else 16
    c5, p5 = make_batch(torch, np, ra, dev, W5, H5, n5, 1 << 20, 1)
    out["config5_shape_f16"] = extra_batch(torch, np, ra, dev, dev_index, "f16", c5, p5, W5, H5, 4, 8, 3 if n5 > 16 else 6, stream,
                                           f"BASELINE configs[4]'s per-GPU share on one GPU: {n5} x 11648x8736 (100 MP) frames, RGBA-f16 surface, "
                                           "randomised stacks, fused histogram, strict f32 arithmetic; default launch mode: multi-frame "
                                           "launches (4 frames each, capped by the ring of 4), which sweep a frame in row order and "
                                           "need no row bands of their own", "multi", valu_ns=valu_ns, launch_diag=True)
    out["config5_shape_f16_tiled"] = extra_batch(torch, np, ra, dev, dev_index, "f16", c5, p5, W5, H5, 4, 8, 2 if n5 > 16 else 3, stream,
                                                 f"the same {n5} x 100 MP frames as BASELINE configs[4] words it: 'tiled multi-launch per frame' -- "
                                                 "8 row-band launches per frame (RD_BATCH_PERSISTENT=0), alternating between two streams that share the CU's two workgroup slots (RD_BATCH_STREAMS=2)", "per_frame", tiled=True)
    del c5
    out["ragged_width"] = extra_ragged_width(torch, np, ra, dev, dev_index, cfas, params, stream, valu_ns=valu_ns)
    out["odd_width"] = extra_ragged_width(torch, np, ra, dev, dev_index, cfas, params, stream, valu_ns=valu_ns, Wr=6001, Hr=4001)
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


# ------------------------------------------------------------------------------------------------
# --host ranks: one process per GPU (the driver's contract)
# ------------------------------------------------------------------------------------------------
def run_ranks(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    # RCCL / device-tensor sharing across processes needs dmabuf IPC on this pool's host driver (already exported by the
    # image; kept here so that a bare `python -m torch.distributed.run ... bench.py` works from any shell)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one process per GPU AND one GPU per process: narrowed BEFORE torch / HIP are imported or anything touches a card
    visibility = narrow_visibility(os.environ, local_rank) if world > 1 else {"mode": "single rank: untouched"}
    import numpy as np
    import torch
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # RAWDEV_DIST_BACKEND=gloo is a rehearsal mode for a 1-GPU box: several ranks share device 0 and the
    # histogram all-reduce goes over gloo (RCCL refuses two ranks on one device).  The driver's runs use nccl.
    backend = os.environ.get("RAWDEV_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    # nccl: one GPU per rank.  With the rank's view narrowed to its own card (narrow_visibility above, or a launcher that did
    # it) one device is visible and index 0 is that card; with every card visible rank r uses index r.  Whether two ranks ended
    # up on ONE physical device is checked from the PCI bus ids below.
    dev_index = local_rank if local_rank < ndev else local_rank % max(1, ndev)
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

    ident = device_identity(dev_index)
    box = measure_box(ra, dev_index) if not args.no_box else {}       # before the headline, outside its timed region (~0.1 s)
    valu_before = measure_valu_ns(ra, dev_index) if (rank == 0 and not args.no_box) else None
    clock_fn, clock_src = (None, "skipped (--no-diagnose)")
    clocks_idle = None
    if rank == 0 and not args.no_diagnose:
        clock_fn, clock_src = open_clock_source(ident.get("pci_bus_id"))
        if clock_fn:
            try:
                clocks_idle = {k: round(v, 1) for k, v in clock_fn().items()}
            except Exception:  # noqa: BLE001
                clocks_idle = None
    cfas, params = make_batch(torch, np, ra, dev, W, H, F, rank, world, args.data, stagger=args.plane_stagger)     # frame i -> rank i mod N
    ring = alloc_ring(torch, dev, max(1, args.ring), H * W * bpp_out, arena=bool(args.ring_arena))
    hist = torch.zeros(768, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    math_mode = ra.MATH_CONTRACTED if args.math == "contracted" else ra.MATH_STRICT
    be = ra.BatchExporter(dev_index, W, H, fmt, with_hist, math_mode=math_mode)
    variants = [params] if args.static_descriptors else [rotated_stacks(params, k) for k in range(3)]
    arrays = [be.make_frames([c.data_ptr() for c in cfas], [ring[i % len(ring)].data_ptr() for i in range(F)], v, WB, CM)
              for v in variants]
    stream = torch.cuda.Stream(device=dev)
    nstep = [0]

    def step(develop=None, array=None):
        """One step: the fused launches of one develop call over the batch + the histogram fold (+ the all-reduce).  `develop`
        replaces be.develop for the diagnostic passes (same frames, same order); `array` pins the frame array (descriptor A/B)."""
        arr = array if array is not None else arrays[nstep[0] % len(arrays)]
        if develop is None:
            be.develop(arr, row_bands=args.row_bands, stream=stream.cuda_stream)
        else:
            develop(arr)
        if array is None:
            nstep[0] += 1
        if with_hist:
            be.histogram(hist.data_ptr(), stream=stream.cuda_stream)
            if world > 1:
                allreduce_histogram(hist)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream), quiet_gc():
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
    elapsed_own = elapsed
    valu_ns = measure_valu_ns(ra, dev_index) if (rank == 0 and not args.no_box) else None     # right after the region: the clocks are up

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    lpc = max(1, be.last_launch_count())

    # ---- self-diagnosis, all of it OUTSIDE the timed region (round 6: the line must explain its own speed).  Every rank
    # runs the passes that contain steps (their all-reduce keeps the ranks in lockstep); nothing here may cost the line.
    diag = {"valu_before_ns": valu_before, "valu_after_ns": valu_ns, "clocks_reason": None if clock_fn else clock_src}
    if not args.no_diagnose:
        alg_launch = BYTES_PER_PX[args.format] * W * H * F / lpc
        with torch.cuda.stream(stream), quiet_gc():
            # (a) an event pair around every fused launch, six steps
            diag["launches"] = instrumented_pass(be, step, barrier, 6, alg_launch, with_note=True)
            # (b) the shader clock UNDER the kernel: two ordinary steps through the instance that stamps its clocks
            try:
                if not (args.format == "f32" and with_hist and args.math == "strict" and args.row_bands <= 1):
                    raise NotImplementedError("the stamped instance exists for the headline's kernel only (f32 surface, histogram, strict arithmetic)")
                got_clock = []
                for _ in range(2):
                    step(develop=lambda arr: got_clock.append(be.measure_clock(arr, stream=stream.cuda_stream)))
                barrier()
                g = got_clock[-1]
                diag["clock_under_kernel"] = {"GHz_median": round(g[0], 3), "GHz_min": round(g[1], 3), "GHz_max": round(g[2], 3),
                                              "workgroup_busy_us_median": round(g[3], 1),
                                              "note": "rd_batch_measure_clock: shader cycles / 100 MHz real-time ticks stamped by thread 0 of every "
                                                      "workgroup of the last launch of an ordinary step, in a diagnostic instance of the kernel "
                                                      "(no stamp executes in the timed region's instance)"}
            except NotImplementedError as e:
                diag["clock_under_kernel"] = {"skipped": str(e)}
            except Exception as e:  # noqa: BLE001
                diag["clock_under_kernel"] = {"error": f"{type(e).__name__}: {e}"}
            # (c) board clocks / power / temperature while five more steps run
            try:
                if clock_fn:
                    with ClockSampler(clock_fn) as smp:
                        for _ in range(5):
                            step()
                        barrier()
                    diag["clocks"] = dict(smp.summary(), source=clock_src, idle_before_the_run=clocks_idle,
                                          note="sampled every ~4 ms on a host thread while five untimed steps of the same workload ran, "
                                               "right after the timed region")
                else:
                    for _ in range(5):                     # the other ranks keep step with rank 0's sampled steps
                        step()
                    barrier()
            except Exception as e:  # noqa: BLE001
                diag["clocks"] = None
                diag["clocks_reason"] = f"sampling failed: {type(e).__name__}: {e}"
        # (d) the box probes again, after the region
        if not args.no_box:
            diag["box_after"] = measure_box(ra, dev_index)
            if rank == 0:
                diag["valu_after_probes_ns"] = measure_valu_ns(ra, dev_index)
        # (e) rotating vs static descriptors, alternating, N = 1 only (the upload's cost on THIS box)
        if world == 1 and len(arrays) > 1:
            try:
                ab = {"rotating": [], "static": []}
                with torch.cuda.stream(stream), quiet_gc():
                    for rep in range(3):
                        for name in ("rotating", "static"):
                            step(array=arrays[0] if name == "static" else None)      # settle: the static array is cached from here on
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(stream)
                            for _ in range(4):
                                step(array=arrays[0] if name == "static" else None)
                            e1.record(stream)
                            stream.synchronize()
                            ab[name].append(e0.elapsed_time(e1) / 4.0)
                    step()                                 # leave the ring as a regular step leaves it (the oracle check below reads it)
                    stream.synchronize()
                r_ms, s_ms = _median(ab["rotating"]), _median(ab["static"])
                diag["descriptor_upload_ab"] = {"rotating_ms_per_step": round(r_ms, 4), "static_ms_per_step": round(s_ms, 4),
                                                "rotating_all": [round(x, 4) for x in ab["rotating"]], "static_all": [round(x, 4) for x in ab["static"]],
                                                "upload_cost_ms_per_step": round(r_ms - s_ms, 4),
                                                "note": "3 x (4 steps rotating through the three frame arrays | 4 steps resubmitting ONE array, whose upload "
                                                        "librawdev skips), alternating, HIP events, after the timed region"}
            except Exception as e:  # noqa: BLE001
                diag["descriptor_upload_ab"] = {"error": f"{type(e).__name__}: {e}"}

    if with_hist:                                          # sanity: the global histogram counts every pixel
        got = int(hist.sum().item())
        assert got == 3 * world * F * W * H, f"histogram sum {got} != {3 * world * F * W * H}"

    verified, verified_note = None, "not checked"
    if rank == 0:                                          # outside the timed region
        try:
            last = variants[(nstep[0] - 1) % len(variants)]
            verified, verified_note = verify_outputs(args.format, W, H, cfas, last, ring, F, args.math)
        except Exception as e:  # noqa: BLE001  (oracle not built / not shipped: say so, do not claim)
            verified, verified_note = None, f"oracle check unavailable: {e}"

    # (f) the memory pattern's own ceiling on this box, in these buffers: the same launches with the arithmetic removed
    # (AFTER the check above: the surfaces receive raw samples)
    if not args.no_diagnose and args.format == "f32":
        try:
            with torch.cuda.stream(stream), quiet_gc():
                be.probe_pattern(arrays[0], stream=stream.cuda_stream)
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for k in range(4):
                    be.probe_pattern(arrays[(k + 1) % len(arrays)], stream=stream.cuda_stream)
                e1.record(stream)
                stream.synchronize()
                p_ms = e0.elapsed_time(e1) / 4.0
                be.set_launch_timing(3)
                for k in range(3):
                    be.probe_pattern(arrays[k % len(arrays)], stream=stream.cuda_stream)
                stream.synchronize()
                p_tl = launch_summary(be.launch_timeline(), BYTES_PER_PX["f32"] * W * H * F / lpc)
                be.set_launch_timing(0)
            diag["pattern"] = {"ms_per_step": round(p_ms, 4), "us_per_frame": round(p_ms * 1e3 / F, 2),
                               "GBps": round(BYTES_PER_PX["f32"] * W * H * F / (p_ms * 1e-3) / 1e9, 1),
                               "launch_us": (p_tl or {}).get("launch_us"), "GBps_median_launch": (p_tl or {}).get("GBps_median_launch"),
                               "note": "rd_batch_probe_pattern: rd_develop_batch's own loads, LDS-DMA sweeps, tile tickets, LDS store stage and "
                                       "non-temporal stores with the colour stack, gamma and histogram removed, over the SAME planes and the SAME "
                                       "output ring, 4 steps back to back (HIP events); what this memory pattern costs on this box"}
        except Exception as e:  # noqa: BLE001
            diag["pattern"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                be.set_launch_timing(0)
            except Exception:  # noqa: BLE001
                pass

    # every rank's own account of the region, gathered on rank 0 (N = 1: the one record)
    mine = rank_record(rank, local_rank, dev_index, ident, elapsed_own, dev_ms, args.steps, F, W, H, lpc, box)
    la = diag.get("launches") or {}
    mine.update({"visibility": visibility, "devices_visible": ndev, "launch_us": la.get("launch_us"), "kernel_ms_per_step": la.get("kernel_ms_per_step"),
                 "step_boundary_gap_us": la.get("step_boundary_gap_us"),
                 "clock_under_kernel_GHz": (diag.get("clock_under_kernel") or {}).get("GHz_median"),
                 "box_pattern_GBps": (diag.get("pattern") or {}).get("GBps"),
                 "box_after_copy_GBps": (diag.get("box_after") or {}).get("copy"), "box_after_fill_GBps": (diag.get("box_after") or {}).get("fill")})
    records, diag_err = [mine], None
    world_seen = 1
    if world > 1:
        world_seen = dist.get_world_size()
        try:
            gathered = [None] * world_seen
            dist.all_gather_object(gathered, mine)
            records = [r for r in gathered if r is not None]
        except Exception as e:  # noqa: BLE001  (the headline must survive a failed gather; the line then says so)
            diag_err = f"rank records could not be gathered: {type(e).__name__}: {e}"

    # One rd_batch_develop call = `lpc` fused launches (the library packs up to 8 consecutive frames into one launch;
    # RD_BATCH_PERSISTENT=0 gives one launch per frame / row band).  Algorithmic bytes per launch = SURVEY 8(d)'s
    # per-pixel figure x the pixels one launch processes.
    unmeasured = ("; N > 1 on DISTINCT devices had never run before this line was produced on a multi-GPU node -- check "
                  "`distinct_devices` == n_gpus" if world > 1 else "")
    result = result_line(args, world, F, W, H, elapsed, dev_ms, lpc, len(ring), verified, verified_note,
                         "one process per GPU (torch.distributed, backend " + (backend if world > 1 else "none: single rank") + ")" + unmeasured,
                         "one frame array resubmitted every step (upload skipped)" if args.static_descriptors else
                         "steps rotate through three frame arrays (the slider stacks rotated by a third of the batch); librawdev caches "
                         "the last two arrays it saw, so every step uploads its descriptors", box=box, valu_ns=valu_ns, diag=diag)
    if rank == 0:
        # as-nccl: the duplicate-device rule applies to this run (nccl always; RAWDEV_DIAG_ASSUME_NCCL=1 lets the gloo
        # rehearsal on a one-GPU box prove that the rule fires)
        rule_backend = "nccl" if (backend == "nccl" or os.environ.get("RAWDEV_DIAG_ASSUME_NCCL") == "1") else backend
        diag, err2 = summarize_ranks(records, world, world_seen, rule_backend)
        result.update(diag)
        # where the planes and surfaces of the first launch live (the launch-position question of profiles/HISTORY.md)
        result["config"]["buffers"] = {"plane_stagger": args.plane_stagger, "ring_arena": args.ring_arena,
                                       "cfa_addr_mod_2MiB_first8": [c.data_ptr() % (2 << 20) for c in cfas[:8]],
                                       "ring_addr_mod_2MiB": [r.data_ptr() % (2 << 20) for r in ring[:8]]}
        result["backend"] = backend if world > 1 else None
        try:
            result["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version()) if world > 1 else None
        except Exception:  # noqa: BLE001
            result["rccl_version"] = None
        diag_err = diag_err or err2
        if diag_err:
            result["invalid"] = diag_err
    if world > 1 and with_hist:
        # the all-reduce latency on its own (outside the timed region): 768 x i64 over RCCL, median of 20
        lat = []
        scratch = torch.zeros_like(hist)                   # not `hist`: 25 in-place sums would grow it by world^25
        with torch.cuda.stream(stream):
            for _ in range(25):
                scratch.zero_()
                barrier()
                t1 = time.perf_counter()
                allreduce_histogram(scratch)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t1) * 1e6)
        lat = sorted(lat[5:])
        result["allreduce_us"] = {"median": round(lat[len(lat) // 2], 1), "min": round(lat[0], 1), "bytes": 768 * 8, "backend": backend,
                                  "note": "one all-reduce of the 768 x i64 histogram + synchronise, host-timed on rank 0, after a barrier"}
    if world == 1 and not args.no_alt_math:
        # Reported-only: the same workload in the other arithmetic, 5 steps.
        other = "contracted" if args.math == "strict" else "strict"
        be2 = ra.BatchExporter(dev_index, W, H, fmt, with_hist,
                               math_mode=ra.MATH_CONTRACTED if other == "contracted" else ra.MATH_STRICT)
        with torch.cuda.stream(stream):
            be2.develop(arrays[0], row_bands=args.row_bands, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for k in range(5):
                be2.develop(arrays[(k + 1) % len(arrays)], row_bands=args.row_bands, stream=stream.cuda_stream)
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
    if rank == 0 and world == 1 and not args.no_extra and (W, H) == (6016, 4016) and args.data == "uniform":
        del ring
        try:
            with quiet_gc():                               # host-timed medians in there
                result["extra_configs"] = extra_configs(torch, np, ra, dev, dev_index, cfas, params, stream, valu_ns=valu_ns)
        except Exception as e:  # noqa: BLE001  (never lose the headline line to an extra)
            result["extra_configs"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(W, H, args.cpu_seconds)
    if rank == 0 and not args.no_diagnose:
        # LAST key of the line (a record that keeps only the tail of stdout keeps this): the self-diagnosis in one screenful
        rf, ex = result["roofline"], result.get("extra_configs") or {}
        la = rf.get("launches") or {}
        u8 = ex.get("batch_rgba8") if isinstance(ex.get("batch_rgba8"), dict) else {}
        result["diagnosis"] = {
            "value_MPps": result["value"], "frac": rf["frac"], "frac_kernel": rf.get("frac_kernel"), "frac_of_box_pattern": rf.get("frac_of_box_pattern"),
            "ms_per_step": result["ms_per_step"], "kernel_ms_per_step": result.get("kernel_ms_per_step"), "gap_ms_per_step": result.get("gap_ms_per_step"),
            "launch_us": la.get("launch_us"), "launch_us_by_position": la.get("launch_us_by_position"),
            "gap_us_between_launches": la.get("gap_us_between_launches"), "step_boundary_gap_us": la.get("step_boundary_gap_us"),
            "clock_under_kernel_GHz": rf.get("clock_under_kernel_GHz"), "clocks": rf.get("clocks"), "clocks_reason": rf.get("clocks_reason"),
            "box_before": {k: (rf.get("box_before") or {}).get(k) for k in ("copy", "fill", "read", "valu_effective_GHz")},
            "box_after": {k: (rf.get("box_after") or {}).get(k) for k in ("copy", "fill", "read", "valu_effective_GHz")},
            "box_pattern_GBps": rf.get("box_pattern_GBps"), "box_pattern_launch_us": ((rf.get("box_pattern") or {}).get("launch_us") or {}).get("median"),
            "descriptor_upload_ab_ms_per_step": {k: (result.get("descriptor_upload_ab") or {}).get(k) for k in ("rotating_ms_per_step", "static_ms_per_step")},
            "batch_rgba8": {"us_per_frame": u8.get("us_per_frame"), "frac": (u8.get("roofline") or {}).get("frac"),
                            "launch_us": (u8.get("launches") or {}).get("launch_us"), "kernel_ms_per_step": u8.get("kernel_ms_per_step"),
                            "gap_ms_per_step": u8.get("gap_ms_per_step"), "ms_per_step": u8.get("ms")},
            # the other configurations in one line each (their full objects are further up, in extra_configs)
            "extras": {
                "config5_shape_f16": {k: (ex.get("config5_shape_f16") or {}).get(k) for k in ("us_per_frame", "MP_per_s", "launches_per_step")}
                                     | {"frac": ((ex.get("config5_shape_f16") or {}).get("roofline") or {}).get("frac")},
                "config5_shape_f16_tiled": {k: (ex.get("config5_shape_f16_tiled") or {}).get(k) for k in ("us_per_frame", "MP_per_s", "launches_per_step")}
                                           | {"frac": ((ex.get("config5_shape_f16_tiled") or {}).get("roofline") or {}).get("frac")},
                "single_frame_f32_ms": (ex.get("single_frame_f32") or {}).get("ms"),
                "full_res_to_bytes_pinned_ms": ((ex.get("full_res_to_bytes") or {}).get("pinned_dst") or {}).get("ms"),
                "pcie_floor_ms": (ex.get("full_res_to_bytes") or {}).get("pcie_floor_ms"),
                "export_ring_ms_per_frame": {k: ((ex.get("export_ring") or {}).get(k) or {}).get("ms_per_frame") for k in ("rgb8", "rgba8")},
                "ragged_width_ns_per_px_ratio": {k: ((ex.get("ragged_width") or {}).get(k) or {}).get("ns_per_px_ratio") for k in ("f32", "f16", "u8", "rgb8")},
                "odd_width_ns_per_px_ratio": {k: ((ex.get("odd_width") or {}).get(k) or {}).get("ns_per_px_ratio") for k in ("f32", "f16", "u8", "rgb8")},
                "all_verified": all(bool((ex.get(k) or {}).get("verified")) for k in ("single_frame_f32", "full_res_to_bytes", "batch_rgba8", "config5_shape_f16",
                                                                                   "config5_shape_f16_tiled", "ragged_width", "odd_width")) if ex and "error" not in ex else None,
                # whole frames against checksums of the reference's shader text, evaluated (tests/golden/wgsl_fullsize.json)
                "text_pin_verified": (ex.get("text_pin") or {}).get("verified"),
            },
            "verified": result.get("verified"),
            "cpu_baseline_MPps": (result.get("cpu_baseline") or {}).get("value"),
        }
    failed = False
    if rank == 0:
        if result.get("invalid"):                          # not a measurement of N GPUs: stderr + non-zero exit, no stdout line
            print("bench.py: INVALID RUN: " + result["invalid"], file=sys.stderr)
            print(json.dumps(result), file=sys.stderr, flush=True)
            failed = True
        else:
            print(json.dumps(result), flush=True)
    be.close()
    if world > 1:
        dist.barrier()                                     # rank 0 spent a second on the oracle check: leave together
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


# ------------------------------------------------------------------------------------------------
# --host node: ONE process, all GPUs through rd_node_batch_* 
# ------------------------------------------------------------------------------------------------
def run_node(args):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import raweditor_amd as ra

    assert torch.cuda.is_available(), "bench.py needs a GPU"
    N = args.gpus
    ndev = torch.cuda.device_count()
    if N > ndev:
        # a rehearsal on a smaller box: a device listed several times; librawdev accepts that only with RD_NODE_REDUCE=host
        # (histograms folded on the host) or with the test stand-in for librccl
        if os.environ.get("RD_NODE_REDUCE") not in ("host", "standin"):
            sys.exit(f"--host node --gpus {N}: only {ndev} device(s) visible (RD_NODE_REDUCE=host rehearses N > devices on one GPU)")
    devices = [d % ndev for d in range(N)]
    W, H, F = args.width, args.height, args.frames
    fmt = {"f32": ra.FMT_RGBA_F32, "f16": ra.FMT_RGBA_F16, "u8": ra.FMT_RGBA_U8}[args.format]
    bpp_out = ra.BYTES_PER_PIXEL[fmt]
    with_hist = not args.no_hist
    math_mode = ra.MATH_CONTRACTED if args.math == "contracted" else ra.MATH_STRICT

    # (the box probe first, as in the ranks host: what is allocated and freed before the batch's buffers decides where the
    #  driver places them, and a launch's time follows that placement by +-1.5 % -- profiles/r05_plane_stagger.txt; the two
    #  hosts are compared on the same order of allocations)
    box = measure_box(ra, devices[0]) if not args.no_box else {}
    # frame i of the call belongs to devices[i mod N]: F frames per device, interleaved
    per_dev = []
    for r, d in enumerate(devices):
        dev = torch.device("cuda", d)
        cfas, params = make_batch(torch, np, ra, dev, W, H, F, r, N, args.data, stagger=args.plane_stagger)
        ring = alloc_ring(torch, dev, max(1, args.ring), H * W * bpp_out, arena=bool(args.ring_arena))
        per_dev.append((cfas, params, ring))
    for d in set(devices):
        torch.cuda.synchronize(d)
    nb = ra.NodeBatch(devices, W, H, fmt, with_hist, math_mode=math_mode)
    idents = [device_identity(d) for d in devices]

    def frame_array(swap):
        cp, op, pp = [], [], []
        for i in range(F * N):
            r, f = i % N, i // N
            cfas, params, ring = per_dev[r]
            pv = rotated_stacks(params, swap)
            cp.append(cfas[f].data_ptr()); op.append(ring[f % len(ring)].data_ptr()); pp.append(pv[f])
        return ra.BatchExporter.make_frames(cp, op, pp, WB, CM)

    arrays = [frame_array(0)] if args.static_descriptors else [frame_array(k) for k in range(3)]
    nstep = [0]
    hist = None

    drain = os.environ.get("RD_NODE_HIST_SYNC", "") == "1"   # A/B: rounds 2-4's step (histogram() synchronises every step)

    def step():
        nonlocal hist
        nb.develop(arrays[nstep[0] % len(arrays)], row_bands=args.row_bands)
        nstep[0] += 1
        if with_hist:
            # per-device fold + all-reduce + read-back, ENQUEUED (round 5): like the one-process-per-GPU host, whose fold and
            # all-reduce are stream-ordered, the steps queue up on the devices' streams; the last result is fetched after the region
            if drain:
                hist = nb.histogram()
            else:
                nb.histogram_enqueue()

    for _ in range(args.warmup):
        step()
    nb.synchronize()
    if with_hist and not drain and args.warmup:
        nb.histogram_fetch()                               # the warm-up's counts: fetch returns everything enqueued since the last fetch
    s0 = torch.cuda.ExternalStream(nb.stream(0), device=torch.device("cuda", devices[0]))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with quiet_gc():
        t0 = time.perf_counter()
        ev0.record(s0)
        step_ms = []
        for _ in range(args.steps):
            t1 = time.perf_counter()
            step()
            step_ms.append((time.perf_counter() - t1) * 1e3)   # with a histogram every step ends synchronised: its own wall time
        ev1.record(s0)
        nb.synchronize()
        elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)                         # HIP events on device 0's launch stream
    hist_steps = 1
    if with_hist and not drain:
        hist = nb.histogram_fetch()                        # every timed step's interval, summed (ABI 5; the read-backs landed before the synchronise)
        hist_steps = args.steps

    if with_hist:
        got = int(hist.sum())
        assert got == 3 * N * F * W * H * hist_steps, f"histogram sum {got} != {3 * N * F * W * H * hist_steps}"
    verified, verified_note = None, "not checked"
    try:
        cfas, params, ring = per_dev[0]
        last = rotated_stacks(params, (nstep[0] - 1) % len(arrays))
        verified, verified_note = verify_outputs(args.format, W, H, cfas, last, ring, F, args.math)
    except Exception as e:  # noqa: BLE001
        verified, verified_note = None, f"oracle check unavailable: {e}"
    lpc = max(1, nb.last_launch_count(0))
    dup = len(set(devices)) < N
    result = result_line(args, N, F, W, H, elapsed, dev_ms, lpc, len(per_dev[0][2]), verified, verified_note,
                         f"ONE process, rd_node_batch_* over devices {devices} (one rd_batch + stream + host thread per device); "
                         f"histogram reduction: {nb.reduce_kind()}; each step = develop + histogram " + ("(synchronises)" if drain else "fold / all-reduce / read-back enqueued (no drain between steps)") +
                         ("; REHEARSAL: a device is listed more than once, the ranks share one GPU" if dup else ""),
                         "one frame array resubmitted every step (upload skipped)" if args.static_descriptors else
                         "steps rotate through three frame arrays (librawdev caches two): every step uploads its descriptors", box=box)
    result["devices"] = [{"slot": i, "device_index": d, "pci_bus_id": idents[i].get("pci_bus_id"), "name": idents[i].get("name"),
                          "launches": nb.last_launch_count(i) * args.steps} for i, d in enumerate(devices)]
    result["distinct_devices"] = len({(i.get("pci_bus_id") or f"index:{d}") for i, d in zip(idents, devices)})
    if with_hist and drain:
        result["step_ms"] = {"min": round(min(step_ms), 4), "median": round(sorted(step_ms)[len(step_ms) // 2], 4), "max": round(max(step_ms), 4),
                             "all": [round(x, 3) for x in step_ms] if len(step_ms) <= 32 else None,
                             "note": "host wall time of each step (develop + histogram, which synchronises)"}
    result["env"] = {k: os.environ.get(k) for k in DIAG_ENV + ("RD_NODE_REDUCE", "RAWDEV_RCCL_LIB")}
    if with_hist:
        # the histogram call on its own (per-device fold + all-reduce over RCCL when N > 1 + read-back + synchronise), idle devices
        lat = []
        for _ in range(25):
            t1 = time.perf_counter()
            nb.histogram()
            lat.append((time.perf_counter() - t1) * 1e6)
        lat = sorted(lat[5:])
        result["histogram_call_us"] = {"median": round(lat[len(lat) // 2], 1), "min": round(lat[0], 1), "reduction": nb.reduce_kind(),
                                       "note": "rd_node_batch_histogram on idle devices: fold kernels + reduction + 6 KiB read-back + synchronise"}
    if N == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(W, H, args.cpu_seconds)
    print(json.dumps(result), flush=True)
    nb.close()


def main():
    args = parse_args()
    in_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.host == "node":
        if in_launcher and int(os.environ["WORLD_SIZE"]) > 1:
            sys.exit("--host node is ONE process for all GPUs: start it without torch.distributed.run")
        return run_node(args)
    if not in_launcher and args.gpus > 1:
        # the driver's N = 1 shape of the command, with N > 1: become the launcher.  Nothing has touched the GPU, torch
        # or HIP in this process; the ranks are fresh children (never an exec of this process).
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    return run_ranks(args)


if __name__ == "__main__":
    main()
