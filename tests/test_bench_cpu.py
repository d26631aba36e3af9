"""bench.py's launcher logic, no GPU: `python bench.py --gpus N` (N > 1, no torch.distributed.run around it) must start
N fresh rank processes with the launcher's environment, relay rank 0's line and fail when a rank fails -- without the
parent importing torch or touching HIP."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_child_environment():
    base = {"PATH": "/bin", "RANK": "7", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    envs = [bench.child_env(base, r, 4, 29511) for r in range(4)]
    for r, e in enumerate(envs):
        assert e["RANK"] == e["LOCAL_RANK"] == str(r)
        assert e["WORLD_SIZE"] == e["LOCAL_WORLD_SIZE"] == "4"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511"
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin"
    assert base["RANK"] == "7"                               # the parent's environment is not modified
    assert bench.child_env({}, 0, 2, 1)["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"   # set when the shell did not export it


def test_every_rank_sees_only_its_own_gpu():
    """VERDICT round 5, item 2 (the process guard): under the default RAWDEV_RANK_VISIBILITY=own the children of the
    self-launcher -- and a rank started by torch.distributed.run, through the same function -- narrow their own view to one
    card before anything touches the GPU; `all` keeps the launcher's view."""
    base = {"PATH": "/bin"}
    envs = [bench.child_env(base, r, 8, 29511, n_gpus=8) for r in range(8)]
    assert [e["ROCR_VISIBLE_DEVICES"] for e in envs] == [str(r) for r in range(8)]           # no list set: card LOCAL_RANK
    assert all(e["RAWDEV_RANK_NARROWED"] == "1" and "HIP_VISIBLE_DEVICES" not in e for e in envs)
    assert "ROCR_VISIBLE_DEVICES" not in base
    # a rehearsal with more ranks than cards (four gloo ranks on a one-GPU box): every rank gets the one card
    assert [bench.child_env(base, r, 4, 1, n_gpus=1)["ROCR_VISIBLE_DEVICES"] for r in range(4)] == ["0"] * 4
    # a list the launcher has set is indexed, whichever variable carries it; the lowest layer wins
    e = bench.child_env({"HIP_VISIBLE_DEVICES": "4,5,6,7"}, 2, 4, 1, n_gpus=8)
    assert e["HIP_VISIBLE_DEVICES"] == "6" and "ROCR_VISIBLE_DEVICES" not in e
    e = bench.child_env({"ROCR_VISIBLE_DEVICES": "1,3", "HIP_VISIBLE_DEVICES": "0,1"}, 1, 2, 1, n_gpus=8)
    assert e["ROCR_VISIBLE_DEVICES"] == "3" and e["HIP_VISIBLE_DEVICES"] == "0,1"
    # a launcher that narrowed already (one entry) is left alone
    e = bench.child_env({"HIP_VISIBLE_DEVICES": "5"}, 3, 8, 1, n_gpus=8)
    assert e["HIP_VISIBLE_DEVICES"] == "5" and "ROCR_VISIBLE_DEVICES" not in e and "RAWDEV_RANK_NARROWED" not in e
    # the switch: every rank keeps the launcher's view
    e = bench.child_env({"RAWDEV_RANK_VISIBILITY": "all"}, 3, 8, 1, n_gpus=8)
    assert not any(v in e for v in bench.VISIBILITY_VARS) and "RAWDEV_RANK_NARROWED" not in e
    # the LOCAL_RANK branch of run_ranks calls the same function on os.environ: a child the self-launcher narrowed is not
    # narrowed twice (its one-entry list would be indexed with its rank otherwise), a torchrun rank is
    env = dict(envs[5])
    info = bench.narrow_visibility(env, 5, n_gpus=8)
    assert env["ROCR_VISIBLE_DEVICES"] == "5" and info["variable"] == "ROCR_VISIBLE_DEVICES" and info["value"] == "5"
    env = {"LOCAL_RANK": "5"}
    info = bench.narrow_visibility(env, 5, n_gpus=8)
    assert env["ROCR_VISIBLE_DEVICES"] == "5" and info["mode"] == "own" and info["physical_gpus"] == 8
    assert bench.physical_gpu_count() >= 0                        # no KFD topology in this container: 0, never an exception


def test_launch_summary_splits_kernel_time_from_gaps():
    """The per-launch account of round 6: launches of 100 us at positions 0 and 1 of three steps, 5 us between them, 40 us
    across a step boundary."""
    tl, t = [], 0.0
    for call in range(3):
        for pos in range(2):
            dur = 100.0 + pos                                     # position 1 is 1 us slower, step after step
            tl.append((call, t, t + dur))
            t += dur + (5.0 if pos == 0 else 40.0)
    su = bench.launch_summary(tl, alg_bytes_per_launch=1e6)
    assert su["steps"] == 3 and su["launches_per_step"] == 2
    assert su["launch_us_by_position"] == [100.0, 101.0]
    assert su["launch_us"] == {"min": 100.0, "median": 100.5, "max": 101.0, "mean": 100.5}
    assert abs(su["kernel_ms_per_step"] - 0.201) < 1e-9
    assert su["gap_us_between_launches"]["median"] == 5.0 and su["step_boundary_gap_us"]["median"] == 40.0
    assert abs(su["instrumented_ms_per_step"] - 0.246) < 1e-9
    assert su["GBps_median_launch"] == round(1e6 / 100.5e-6 / 1e9, 1)
    assert bench.launch_summary([]) is None


def test_clock_source_reports_why_it_is_unavailable():
    """No GPU driver here: neither amdsmi nor sysfs can serve; the line then carries null and the reasons, never an exception
    (and never a new dependency)."""
    fn, why = bench.open_clock_source("0000:c1:00.0")
    assert fn is None and "sysfs" in why
    smp = bench.ClockSampler(lambda: {"sclk_MHz": 2100.0, "power_W": 700.0}, period_s=0.001)
    with smp:
        import time
        time.sleep(0.02)
    out = smp.summary()
    assert out["samples"] >= 2 and out["sclk_MHz"]["median"] == 2100.0 and out["power_W"]["max"] == 700.0


def test_argument_defaults_and_modes():
    a = bench.parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.frames, a.host, a.static_descriptors) == (1, 20, 3, 256, "ranks", False)
    a = bench.parse_args(["--gpus", "8", "--host", "node", "--static-descriptors", "--no-extra"])
    assert (a.gpus, a.host, a.static_descriptors, a.no_extra) == (8, "node", True, True)


def test_swapped_halves_is_a_permutation_that_changes_every_descriptor():
    p = list(range(8))
    q = bench.swapped_halves(p)
    assert sorted(q) == p and all(a != b for a, b in zip(p, q))
    assert bench.swapped_halves([5]) == [5]


def test_three_rotated_arrays_defeat_a_two_entry_descriptor_cache():
    """librawdev keeps the last two frame arrays it was given and skips the upload when a call repeats one (rd_host_batch.inl);
    bench.py's steps therefore rotate through THREE arrays (round 5: two alternating ones were both cached after the second
    step).  The rotations are permutations of the same stacks, pairwise different in every position, and in a rotation of
    three no step's array equals either of the two before it."""
    p = list(range(12))
    v = [bench.rotated_stacks(p, k) for k in range(3)]
    assert v[0] == p and all(sorted(x) == p for x in v)
    for a in range(3):
        for b in range(a + 1, 3):
            assert all(x != y for x, y in zip(v[a], v[b]))
    cache = []                                                    # a two-entry cache, as the library keeps it
    for step in range(9):
        cur = v[step % 3]
        assert cur not in cache, "a cached array: no upload would happen"
        cache = (cache + [cur])[-2:]
    assert bench.rotated_stacks([7], 2) == [7]


def _fake_rank_script(tmp_path, fail_rank=None):
    script = tmp_path / "fake_rank.py"
    script.write_text(textwrap.dedent(f"""
        import json, os, sys, time
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        assert "torch" not in sys.modules
        if r == {fail_rank!r}:
            sys.exit(3)
        if {fail_rank!r} is not None and r != {fail_rank!r}:
            time.sleep(60)                                    # a rank that would wait for the dead one
        if r == 0:
            print(json.dumps({{"n_gpus": w, "argv": sys.argv[1:], "local_rank": os.environ["LOCAL_RANK"]}}), flush=True)
        else:
            print("rank", r, "says hello")                    # must not reach the parent's stdout
    """))
    return str(script)


def _run_parent(script, world, extra_env=None):
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"rc = bench.self_launch({world}, ['--gpus', '{world}', '--steps', '2'], script={script!r}); "
            "assert 'torch' not in sys.modules, 'the launcher parent imported torch'; sys.exit(rc)")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)


def test_self_launch_starts_one_process_per_rank_and_relays_rank0(tmp_path):
    out = _run_parent(_fake_rank_script(tmp_path), 3)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout                       # ONE JSON line on stdout: rank 0's
    got = json.loads(lines[0])
    assert got == {"n_gpus": 3, "argv": ["--gpus", "3", "--steps", "2"], "local_rank": "0"}
    assert "rank 1 says hello" in out.stderr and "rank 2 says hello" in out.stderr


def test_self_launch_fails_when_a_rank_fails_and_stops_the_others(tmp_path):
    import time
    t0 = time.time()
    out = _run_parent(_fake_rank_script(tmp_path, fail_rank=1), 2)
    assert out.returncode == 3, (out.returncode, out.stdout, out.stderr)
    assert time.time() - t0 < 50.0                           # the surviving rank was stopped, not waited for (it sleeps 60 s)


def test_main_routes_a_bare_multi_gpu_call_into_the_launcher(monkeypatch):
    seen = {}
    monkeypatch.setattr(bench, "self_launch", lambda world, argv, script=None: seen.update(world=world, argv=list(argv)) or 0)
    monkeypatch.setattr(bench, "run_ranks", lambda args: seen.update(ranks=True))
    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen == {"world": 4, "argv": ["--gpus", "4", "--steps", "3"]}
    # inside a launcher (RANK / WORLD_SIZE exported) the same command is a rank, not a launcher
    seen.clear()
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.main()
    assert seen == {"ranks": True}


# ---- the N > 1 line describes itself ---------------------------------------------------------
def _records(bus_ids, mpps=None):
    recs = []
    for r, bus in enumerate(bus_ids):
        rec = bench.rank_record(r, r, r, {"pci_bus_id": bus, "name": "AMD Instinct MI355X (gfx950)"}, 2.0 + 0.01 * r, 1990.0, 20, 256,
                                6016, 4016, 32, {"copy": 5100.0, "fill": 6200.0})
        if mpps:
            rec["MP_per_s"] = mpps[r]
        recs.append(rec)
    return recs


def test_rank_record_holds_the_ranks_own_clocks():
    rec = _records(["0000:05:00.0"])[0]
    assert rec["rank"] == 0 and rec["device_index"] == 0 and rec["pci_bus_id"] == "0000:05:00.0"
    assert rec["ms_per_step"] == 100.0 and rec["ms_per_step_hip_events"] == 99.5            # 2.0 s wall / 1990 ms of HIP events over 20 steps
    assert rec["launches"] == 32 * 20 and abs(rec["us_per_frame"] - 1990e3 / (20 * 256)) < 1e-3
    assert abs(rec["MP_per_s"] - 256 * 6016 * 4016 / 1e3 / 100.0) < 0.1
    assert rec["box_copy_GBps"] == 5100.0 and rec["pid"] > 0 and rec["host"]


def test_summarize_ranks_counts_devices_and_refuses_shared_ones_under_nccl():
    eight = [f"0000:{i:02x}:00.0" for i in range(8)]
    d, err = bench.summarize_ranks(list(reversed(_records(eight, mpps=[300.0 + i for i in range(8)]))), 8, 8, "nccl")
    assert err is None and d["distinct_devices"] == 8 and d["world_size_seen"] == 8
    assert [r["rank"] for r in d["ranks"]] == list(range(8))                             # sorted by rank, whatever the gather order
    assert (d["per_gpu_MPps_min"], d["per_gpu_MPps_max"]) == (300.0, 307.0) and 0 < d["per_gpu_MPps_spread"] < 0.03
    assert set(d["env"]) >= {"HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES", "NCCL_DEBUG"}
    shared = eight[:7] + [eight[0]]
    d, err = bench.summarize_ranks(_records(shared), 8, 8, "nccl")
    assert d["distinct_devices"] == 7 and err and "7 distinct device" in err and "0000:00:00.0" in err
    d, err = bench.summarize_ranks(_records(shared), 8, 8, "gloo")                       # the one-GPU rehearsal may share
    assert err is None and d["distinct_devices"] == 7
    d, err = bench.summarize_ranks(_records(eight[:4]), 8, 4, "nccl")                    # the group is smaller than the launcher said
    assert err and "world size mismatch" in err
    d, err = bench.summarize_ranks(_records([None, None]), 2, 2, "nccl")                 # no bus ids: fall back to the device index
    assert err is None and d["distinct_devices"] == 2


def _gather_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = bench.rank_record(rank, rank, 0, {"pci_bus_id": "0000:c1:00.0", "name": "one shared GPU"}, 1.0 + rank, 900.0, 10, 8,
                                 640, 480, 1, {"copy": 1.0, "fill": 2.0})
        got = [None] * dist.get_world_size()
        dist.all_gather_object(got, mine)                    # what run_ranks does after the timed region
        if rank == 0:
            for rule in ("gloo", "nccl"):
                d, err = bench.summarize_ranks(got, world, dist.get_world_size(), rule)
                with open(os.path.join(out_dir, f"diag_{rule}.json"), "w") as f:
                    json.dump({"diag": d, "err": err}, f)
    finally:
        dist.destroy_process_group()


def test_rank_records_gather_over_gloo_world_size_2(tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gather_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = json.load(open(tmp_path / "diag_gloo.json"))
    assert g["err"] is None and [r["rank"] for r in g["diag"]["ranks"]] == [0, 1]
    assert g["diag"]["ranks"][1]["ms_per_step"] == 200.0 and g["diag"]["distinct_devices"] == 1
    assert len({r["pid"] for r in g["diag"]["ranks"]}) == 2                                # two processes really reported
    n = json.load(open(tmp_path / "diag_nccl.json"))
    assert n["err"] and "under nccl every rank must own its GPU" in n["err"]


def test_bound_measured_follows_the_runs_own_fractions():
    """roofline.bound is declared (the roofline achieved / peak refer to); roofline.bound_measured is derived from the two
    fractions of the run: the larger names it, within 5 % of each other both do; no VALU figure, no claim."""
    valu = {"valu_issue_frac": 0.92}
    assert bench.bound_measured_of(5489.2, 5938.8, valu)["bound_measured"] == "hbm+valu"          # 0.924 vs 0.92
    assert bench.bound_measured_of(3747.7, 5938.8, {"valu_issue_frac": 0.97})["bound_measured"] == "valu"
    assert bench.bound_measured_of(5700.0, 5750.0, {"valu_issue_frac": 0.80})["bound_measured"] == "hbm"
    assert bench.bound_measured_of(5489.2, None, valu)["bound_measured"] in ("hbm", "hbm+valu", "valu")   # the guide's copy figure
    assert "6290" in bench.bound_measured_of(5489.2, None, valu)["bound_measured_note"]
    assert bench.bound_measured_of(5489.2, 5938.8, {"valu_issue_frac": None})["bound_measured"] is None
    assert bench.bound_of("f32", 0.69, valu) == "hbm" and bench.bound_of("u8", 0.46, valu) == "valu"      # declared, as before


def test_buffers_can_be_views_into_one_arena():
    """--plane-stagger / --ring-arena (A/B switches, profiles/r05_plane_stagger.txt): planes and surfaces as views into one
    allocation each, 2 MiB-aligned pitch + stagger, same contents as separate allocations; the defaults allocate separately."""
    import numpy as np
    import torch
    a = bench.parse_args([])
    assert a.plane_stagger == -1 and a.ring_arena == 0
    ring = bench.alloc_ring(torch, "cpu", 3, 1000, arena=True)
    ptrs = [r.data_ptr() for r in ring]
    assert all(r.numel() == 1000 for r in ring) and ptrs[1] - ptrs[0] == ptrs[2] - ptrs[1] == 2 << 20 and ptrs[0] % (2 << 20) == 0
    assert len({r.data_ptr() for r in bench.alloc_ring(torch, "cpu", 3, 1000)}) == 3

    class FakeRa:
        class EditParams:
            @staticmethod
            def random(rng):
                return float(rng.random())
    c0, p0 = bench.make_batch(torch, np, FakeRa, "cpu", 16, 6, 3, 0, 1)
    c1, p1 = bench.make_batch(torch, np, FakeRa, "cpu", 16, 6, 3, 0, 1, stagger=4096)
    assert p0 == p1 and all(torch.equal(x, y) for x, y in zip(c0, c1))
    d = [c.data_ptr() for c in c1]
    assert d[0] % (2 << 20) == 0 and d[1] - d[0] == (2 << 20) + 4096 == d[2] - d[1]
