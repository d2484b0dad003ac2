#!/bin/bash
# on the GPU box: bench lines of one scene with option sets, e.g. tools/ab_opts.sh S2 "" "tile_aperture=0"
cd "$(dirname "$0")/.." || exit 1
scene=$1; shift
for o in "$@"; do
  extra=""; [ -n "$o" ] && extra="--opts $o"
  timeout -k 10 300 python3 bench.py --scene $scene --no-cpu-baseline --steps 3 --warmup 1 $extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$scene [$o]', round(r['ms_per_step'], 2), 'ms', r['config'].get('primary_rays_per_step_by_path'))
" || exit 1
done
