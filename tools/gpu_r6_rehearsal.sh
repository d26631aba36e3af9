#!/bin/bash
# round 6: the driver's N > 1 command shape on the one-GPU box with every rank narrowed to its own card (VERDICT round 5,
# item 2: the pool kills a run with more than six processes on a card; `python -m torch.distributed.run --nproc-per-node N`
# ranks that each see all N cards put N processes on every card).
#   * four gloo ranks under torch.distributed.run, full size, RAWDEV_RANK_VISIBILITY unset (= own): each rank must report
#     ONE visible device, index 0, ROCR_VISIBLE_DEVICES=0 (one card in this box: LOCAL_RANK modulo 1); distinct_devices 1.
#   * the same under RAWDEV_DIAG_ASSUME_NCCL=1: the duplicate-device rule must still fire (exit 3, no result line) -- the
#     PCI bus ids tell the ranks apart, not the device index, which is 0 for every narrowed rank.
#   * RAWDEV_RANK_VISIBILITY=all, two ranks: the launcher's view is kept (devices_visible = what the box shows).
#   bash tools/gpu_r6_rehearsal.sh [tag]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r6rehearsal}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
summ() {
python3 - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("n_gpus", d["n_gpus"], "world_size_seen", d.get("world_size_seen"), "distinct_devices", d.get("distinct_devices"), "value", d["value"],
      "ms_per_step", d["ms_per_step"], "verified", d.get("verified"), "allreduce_us", (d.get("allreduce_us") or {}).get("median"))
for r in d.get("ranks", []):
    print("  rank", r["rank"], "pid", r["pid"], "device_index", r["device_index"], "devices_visible", r.get("devices_visible"), "bus", r["pci_bus_id"],
          "visibility", {k: (r.get("visibility") or {}).get(k) for k in ("mode", "variable", "value", "physical_gpus")},
          "launch_us", (r.get("launch_us") or {}).get("median"), "clock_GHz", r.get("clock_under_kernel_GHz"), "pattern_GBps", r.get("box_pattern_GBps"))
PY
}
P=$((20000 + RANDOM % 20000))
echo "== torch.distributed.run, 4 gloo ranks on one GPU, full size, every rank narrowed to its own card (default)"
RAWDEV_DIST_BACKEND=gloo timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $P \
    bench.py --gpus 4 --steps 5 --warmup 1 > "$OUT/ranks4_gloo_own.json" 2> "$OUT/ranks4_gloo_own.err"
rc=$?; echo "rc=$rc"; [ $rc -ge 124 ] && exit $rc
[ $rc -eq 0 ] && summ "$OUT/ranks4_gloo_own.json" || tail -5 "$OUT/ranks4_gloo_own.err"
echo "== the same under RAWDEV_DIAG_ASSUME_NCCL=1: must exit 3 without a result line"
P=$((20000 + RANDOM % 20000))
RAWDEV_DIST_BACKEND=gloo RAWDEV_DIAG_ASSUME_NCCL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $P \
    bench.py --gpus 4 --steps 2 --warmup 1 --frames 32 --no-box --no-diagnose > "$OUT/ranks4_assume_nccl.out" 2> "$OUT/ranks4_assume_nccl.err"
rc=$?; echo "rc=$rc (non-zero expected); result lines on stdout: $(grep -c '^{' "$OUT/ranks4_assume_nccl.out"); $(grep -m1 -o 'INVALID RUN[^\"]*' "$OUT/ranks4_assume_nccl.err" | cut -c1-200)"
[ $rc -ge 124 ] && exit $rc
echo "== RAWDEV_RANK_VISIBILITY=all, two self-launched gloo ranks: the launcher's view is kept"
RAWDEV_DIST_BACKEND=gloo RAWDEV_RANK_VISIBILITY=all timeout -k 10 600 python bench.py --gpus 2 --frames 64 --steps 3 --warmup 1 --no-cpu-baseline \
    > "$OUT/ranks2_gloo_all.json" 2> "$OUT/ranks2_gloo_all.err"
rc=$?; echo "rc=$rc"; [ $rc -ge 124 ] && exit $rc
[ $rc -eq 0 ] && summ "$OUT/ranks2_gloo_all.json" || tail -5 "$OUT/ranks2_gloo_all.err"
echo "== done"
