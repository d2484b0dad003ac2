#!/bin/bash
# on the GPU box: S4 1080p / 4 spp frame time (bench line) with the in-tree library; optional RT_AMD_LIB variants as arguments
cd "$(dirname "$0")/.." || exit 1
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export RT_AMD_LIB=$PWD/$lib; fi
  timeout -k 10 200 python3 bench.py --scene S4 --width 1920 --height 1080 --spp 4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('S4 ${lib:-in-tree}: ms/step', round(r['ms_per_step'], 1))
" || exit 1
done
