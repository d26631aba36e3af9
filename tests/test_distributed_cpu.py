"""CPU, world_size 2 over gloo: the N>1 data path of the batch export.  Frames are dealt round-robin
(no pixel data is exchanged); the only collective is the u64[768] histogram all-reduce.  The per-rank
histograms here come from the oracle (the checker) -- the function under test is the sharding +
reduction plumbing in raweditor_amd.batch, which is device-agnostic."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_c
        from raweditor_amd.batch import allreduce_histogram, shard_frames
        from tests.helpers import CM_TEST, WB_DAYLIGHT, random_params
        mine = shard_frames(n_frames, rank, world)
        local = np.zeros(768, np.int64)
        for f in mine:
            rng = np.random.default_rng([0x52415745, f])          # frame content keyed by (seed, frame)
            cfa = rng.integers(0, 4096, (16, 24), dtype=np.uint16)
            u = ref_c.make_uniforms(random_params(rng), WB_DAYLIGHT, CM_TEST)
            local += ref_c.histogram(ref_c.pack_u8(ref_c.render_f32(cfa, u))).reshape(-1).astype(np.int64)
        # counts beyond u32 must survive the reduction: add a large per-rank offset in one bin
        local[5] += (1 << 33) * (rank + 1)
        t = torch.from_numpy(local.copy())
        allreduce_histogram(t)
        np.save(os.path.join(out_dir, f"hist_{rank}.npy"), t.numpy())
        np.save(os.path.join(out_dir, f"frames_{rank}.npy"), np.array(mine))
    finally:
        dist.destroy_process_group()


def test_sharded_histogram_allreduce_gloo(tmp_path, refc):
    from tests.helpers import CM_TEST, WB_DAYLIGHT, random_params
    world, n_frames = 2, 7
    mp.spawn(_worker, args=(world, _free_port(), n_frames, str(tmp_path)), nprocs=world, join=True)
    expect = np.zeros(768, np.int64)
    for f in range(n_frames):
        rng = np.random.default_rng([0x52415745, f])
        cfa = rng.integers(0, 4096, (16, 24), dtype=np.uint16)
        u = refc.make_uniforms(random_params(rng), WB_DAYLIGHT, CM_TEST)
        expect += refc.histogram(refc.pack_u8(refc.render_f32(cfa, u))).reshape(-1).astype(np.int64)
    expect[5] += (1 << 33) * 3
    h0, h1 = (np.load(tmp_path / f"hist_{r}.npy") for r in range(world))
    assert np.array_equal(h0, h1) and np.array_equal(h0, expect)
    frames = sorted(np.concatenate([np.load(tmp_path / f"frames_{r}.npy") for r in range(world)]).tolist())
    assert frames == list(range(n_frames))
    assert expect[:5].sum() + expect[6:].sum() + (expect[5] - (1 << 33) * 3) == 3 * n_frames * 16 * 24


def test_allreduce_is_noop_without_process_group():
    from raweditor_amd.batch import allreduce_histogram
    t = torch.arange(768, dtype=torch.int64)
    assert torch.equal(allreduce_histogram(t.clone()), t)
