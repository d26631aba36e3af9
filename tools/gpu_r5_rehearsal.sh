#!/bin/bash
# round 5: the driver's N > 1 command shapes on the one-GPU box, full size.
#   * `python -m torch.distributed.run --nproc-per-node 4 ... bench.py --gpus 4` over gloo: the launcher's rendezvous on
#     127.0.0.1, 4 x 256 x 24 MP resident (~75 GB of the 288 GB), the record gather, the duplicate-device diagnostic.
#     FOUR ranks, not eight: the pool's GPU boxes kill a run that has more than 6 processes on the card ("process guard";
#     six ranks were killed with "7 processes had the GPU open" -- the launcher counts -- gpurun_out/r5rehearsal, first try).
#   * the same with RAWDEV_DIAG_ASSUME_NCCL=1: the exit-3 rule must fire (ranks share one bus id) and print no result line.
#   * `bench.py --host node --gpus 8` with RD_NODE_REDUCE=host: ONE process, eight rd_batch handles + worker threads on
#     device 0, 8 x 256 x 24 MP resident (~124 GB).
#   bash tools/gpu_r5_rehearsal.sh [tag]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r5rehearsal}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
P=$((20000 + RANDOM % 20000))
echo "== torch.distributed.run, 4 gloo ranks on one GPU, full size"
RAWDEV_DIST_BACKEND=gloo timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $P \
    bench.py --gpus 4 --steps 5 --warmup 1 > "$OUT/ranks4_gloo.json" 2> "$OUT/ranks4_gloo.err"
rc=$?; echo "rc=$rc"; [ $rc -ge 124 ] && exit $rc
python3 - "$OUT/ranks4_gloo.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("n_gpus", d["n_gpus"], "world_size_seen", d.get("world_size_seen"), "distinct_devices", d.get("distinct_devices"), "ranks", len(d.get("ranks", [])),
      "value", d["value"], "ms_per_step", d["ms_per_step"], "verified", d.get("verified"), "allreduce_us", d.get("allreduce_us", {}).get("median"))
PY
echo "== the same under RAWDEV_DIAG_ASSUME_NCCL=1: must exit 3 without a result line"
P=$((20000 + RANDOM % 20000))
RAWDEV_DIST_BACKEND=gloo RAWDEV_DIAG_ASSUME_NCCL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $P \
    bench.py --gpus 4 --steps 2 --warmup 1 --frames 32 --no-box > "$OUT/ranks4_assume_nccl.out" 2> "$OUT/ranks4_assume_nccl.err"
rc=$?; echo "rc=$rc (non-zero expected); result lines on stdout: $(grep -c '^{' "$OUT/ranks4_assume_nccl.out"); $(grep -m1 -o 'INVALID RUN[^\"]*' "$OUT/ranks4_assume_nccl.err" | cut -c1-160)"
[ $rc -ge 124 ] && exit $rc
echo "== one process, 8 handles on device 0 (RD_NODE_REDUCE=host), full size"
RD_NODE_REDUCE=host timeout -k 10 900 python bench.py --host node --gpus 8 --steps 5 --warmup 1 > "$OUT/node8_host.json" 2> "$OUT/node8_host.err"
rc=$?; echo "rc=$rc"; [ $rc -ge 124 ] && exit $rc
python3 - "$OUT/node8_host.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("n_gpus", d["n_gpus"], "devices", [x["device_index"] for x in d["devices"]], "distinct_devices", d["distinct_devices"], "value", d["value"],
      "ms_per_step", d["ms_per_step"], "verified", d.get("verified"))
PY
