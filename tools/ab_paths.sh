#!/bin/bash
# on the GPU box: S0 bench line (frame time, rays by path) for the in-tree library and the variants named
cd "$(dirname "$0")/.." || exit 1
for v in default "$@"; do
  lib=""; [ "$v" != default ] && lib=$PWD/build_variants/librt_$v.so
  RT_AMD_LIB=$lib timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$v', round(r['ms_per_step'], 2), 'ms', r['config']['primary_rays_per_step_by_path'])
" || exit 1
done
