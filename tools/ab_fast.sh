#!/bin/bash
# on the GPU box: the S0 bench line with the library as built and with the variants in build_variants/ (same box, alternating)
cd "$(dirname "$0")/.." || exit 1
for round in 1 2; do
  for v in default "$@"; do
    lib=""; opts=""
    case $v in
      default) ;;
      nosplit) opts="--opts fast_split=0" ;;
      nosplit_*) opts="--opts fast_split=0"; lib=$PWD/build_variants/librt_${v#nosplit_}.so ;;
      *) lib=$PWD/build_variants/librt_$v.so ;;
    esac
    RT_AMD_LIB=$lib timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 $opts > gpurun_out/ab_${v}_$round.json 2> gpurun_out/ab_${v}_$round.err || { echo "$v failed"; tail -3 gpurun_out/ab_${v}_$round.err; exit 1; }
    python3 - gpurun_out/ab_${v}_$round.json $v <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-10s %.2f ms/frame  fast kernels %.3f ms/launch" % (sys.argv[2], r["ms_per_step"], r["roofline"]["ms_per_launch"]))
PY
  done
done
