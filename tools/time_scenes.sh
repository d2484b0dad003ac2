#!/bin/bash
# on the GPU box: frame times of the named scenes with the in-tree library (S4 at its own size): tools/time_scenes.sh S0 S1 S2 S4 MESH10M
cd "$(dirname "$0")/.." || exit 1
for sc in "$@"; do
  args="--scene $sc"
  [ "$sc" = S4 ] && args="--scene S4 --width 1920 --height 1080 --spp 4"
  [ "$sc" = Venera ] && args="--scene Venera --width 1920 --height 1080 --spp 4"
  timeout -k 10 300 python3 bench.py $args --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['config']['scene'], round(r['ms_per_step'], 2), 'ms')
" || exit 1
done
