#!/bin/bash
# HBM traffic and SQ activity of the export kernels from rocprofv3 PMC passes, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), --kernel-trace only, the program itself
# after "--".  Every pass is one short bench.py run (32 frames per step, ring of 8: multi-frame launches of exactly 8
# frames; config 5: 8 frames, ring of 4).  tools/parse_pmc.py turns the result into profiles/pmc_traffic.json and a
# summary; the gfx950 corrections (FETCH_SIZE x2 for 16-B LDS-DMA reads, calibrated on a known volume for 4-B loads)
# are applied there.
#   tools/gpu_pmc.sh [tag] [only]      (run from the repository root on the GPU box; only = "f16": just the f16 surface's passes,
#                                       e.g. after a change to that kernel -- the other passes of the tag stay as they are)
set -u
TAG=${1:-r02}
ONLY=${2:-all}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="--steps 2 --warmup 1 --frames 32 --no-cpu-baseline --no-alt-math --no-extra"
C5="--width 11648 --height 8736 --format f16 --row-bands 8 --frames 8 --ring 4 --steps 2 --warmup 1 --no-cpu-baseline --no-alt-math --no-extra"
pass() {   # name, env assignments ("-" for none), counters (comma separated), bench arguments...
  local name=$1 envs=$2 ctrs=$3; shift 3
  if [ "$ONLY" = f16 ]; then case $name in f16_*|c5_*) ;; *) return 0;; esac; fi
  if [ "$ONLY" = ragged ]; then case $name in *_ragged_*|f32_perframe_noburst_*) ;; *) return 0;; esac; fi
  echo "== $name [$envs] $ctrs"
  ( [ "$envs" != "-" ] && export $envs
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc ${ctrs//,/ } --output-format csv -d "$OUT/$name" -- \
        python3 "$ROOT/bench.py" "$@" > "$OUT/$name.log" 2>&1 )
  local rc=$?; echo "rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
}
for CTR in FETCH_SIZE WRITE_SIZE; do
  pass f32_multi_$CTR      -                                    $CTR $BENCH
  pass f32_perframe_noburst_$CTR "RD_BATCH_PERSISTENT=0 RD_BURST=0" $CTR $BENCH
  pass f32_perframe_$CTR   "RD_BATCH_PERSISTENT=0"              $CTR $BENCH
  pass f16_multi_$CTR      -                                    $CTR $BENCH --format f16
  pass u8_multi_$CTR       -                                    $CTR $BENCH --format u8
  pass c5_multi_$CTR       -                                    $CTR $C5
  # round 5: a width that is not a multiple of the 128-px tile (the overlapped-last-tile instances): no re-read, and the
  # duplicate stores of the overlap are 16 of 6000 px per row
  pass f32_ragged_multi_$CTR -                                  $CTR $BENCH --width 6000 --height 4000
  pass f16_ragged_multi_$CTR -                                  $CTR $BENCH --width 6000 --height 4000 --format f16
  pass u8_ragged_multi_$CTR  -                                  $CTR $BENCH --width 6000 --height 4000 --format u8
done
SQ1=SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,SQ_INST_CYCLES_VMEM_WR,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY,GRBM_GUI_ACTIVE
SQ2=SQ_INSTS_VALU_TRANS_F32,SQ_INSTS_VMEM_WR,SQ_INSTS_VMEM_RD,SQ_INSTS_LDS,SQ_INSTS_SALU,SQ_INSTS_SMEM,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_VMEM,GRBM_GUI_ACTIVE
pass f32_multi_SQ1 - $SQ1 $BENCH
pass f32_multi_SQ2 - $SQ2 $BENCH
pass u8_multi_SQ1  - $SQ1 $BENCH --format u8
pass f16_multi_SQ1 - $SQ1 $BENCH --format f16
# (The six-counter "DRAM destination" pass of round 3 -- profiles/r03_dram_counters.txt -- is not repeated: rocprofv3 refuses six
#  counters in one pass on this box, and in round 4 the attempt sat until its 300 s limit.  tools/gpu_r3_dram.sh holds the split form.)
cd "$ROOT" && python3 tools/parse_pmc.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1; cat "$OUT/summary.txt"
