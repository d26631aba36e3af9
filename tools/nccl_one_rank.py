#!/usr/bin/env python3
"""The collective calls bench.py makes when N > 1, on a ONE-rank nccl (= RCCL) process group: init_process_group with
device_id, the i64[768] histogram all-reduce (counts beyond 2^33 survive), barrier, the f64 MAX reduce of the elapsed
time, the all_gather_object of the per-rank records.  What a one-GPU box can rehearse of the RCCL leg of `bench.py --gpus N`; the multi-rank path itself is covered over
gloo (tests/test_distributed_cpu.py, `RAWDEV_DIST_BACKEND=gloo`)."""
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from raweditor_amd.batch import allreduce_histogram
    h = torch.arange(768, dtype=torch.int64, device=dev) * (2 ** 33)
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        allreduce_histogram(h)                    # world size 1: returns without a collective, by design
        dist.all_reduce(h, op=dist.ReduceOp.SUM)  # the collective itself, on one rank
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        got = [None]
        dist.all_gather_object(got, {"rank": 0, "pci_bus_id": "x", "MP_per_s": 1.0})   # the per-rank records of the N > 1 line
    ok = int(h[767].item()) == 767 * 2 ** 33 and float(t.item()) == 1.5 and got[0]["pci_bus_id"] == "x"
    dist.destroy_process_group()
    print("nccl one-rank rehearsal", "ok" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
