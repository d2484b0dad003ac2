#!/bin/bash
# on the GPU box: S4 1080p / 4 spp frame time with option sets, e.g. tools/ab_s4_opts.sh "" "direct_pct_deep=20"
cd "$(dirname "$0")/.." || exit 1
for o in "$@"; do
  extra=""; [ -n "$o" ] && extra="--opts $o"
  timeout -k 10 200 python3 bench.py --scene S4 --width 1920 --height 1080 --spp 4 --steps 4 --warmup 2 --no-cpu-baseline $extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('S4 [$o]', round(r['ms_per_step'], 1), 'ms')
" || exit 1
done
