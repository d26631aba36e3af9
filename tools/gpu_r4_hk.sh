#!/bin/bash
# round 4: private histogram copies per bin (RD_HK = 4 / 8 / 16) under the RGBA8 kernel with the LDS threshold table, and the f32 kernel
set -u
OUT=gpurun_out/${1:-r4hk}; mkdir -p $OUT
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["roofline"]["us_per_frame"], "us/frame  verified", d["verified"])'
for i in 1 2; do
  for lib in tools/librawdev_hk4.so raweditor_amd/librawdev.so tools/librawdev_hk16.so; do
    for data in uniform gradient; do
      RAWDEV_LIB=$lib timeout -k 10 300 python bench.py --format u8 --ring 32 --data $data --no-cpu-baseline --no-alt-math --no-extra --no-box --steps 10 2>>$OUT/err.txt | python -c "$pick" "u8 $data $(basename $lib)" | tee -a $OUT/ab.txt
    done
  done
done
for lib in tools/librawdev_hk4.so raweditor_amd/librawdev.so tools/librawdev_hk16.so; do
  RAWDEV_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-math --no-extra --no-box --steps 10 2>>$OUT/err.txt | python -c "$pick" "f32 uniform $(basename $lib)" | tee -a $OUT/ab.txt
done
