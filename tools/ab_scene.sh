#!/bin/bash
# on the GPU box: tools/ab_scene.sh "<bench args>" variant... -- the bench line of a workload with the library as built ("default") and with
# build_variants/librt_<variant>.so, alternating, twice
cd "$(dirname "$0")/.." || exit 1
args=$1; shift
for round in 1 2; do
  for v in default "$@"; do
    lib=""
    [ "$v" != default ] && lib=$PWD/build_variants/librt_$v.so
    RT_AMD_LIB=$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline $args > gpurun_out/abs_${v}_$round.json 2> gpurun_out/abs_${v}_$round.err || { echo "$v failed"; tail -3 gpurun_out/abs_${v}_$round.err; exit 1; }
    python3 - gpurun_out/abs_${v}_$round.json $v <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-10s %8.2f ms/frame  %.3e rays/s" % (sys.argv[2], r["ms_per_step"], r["value"]))
PY
  done
done
