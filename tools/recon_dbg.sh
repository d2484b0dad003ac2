#!/bin/bash
# on the GPU box: where k_splat_rows_need spends its time (debug build: wrong pictures) -- 0 all, 1 no filter loop, 2 no gather, 3 neither
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for d in ${RECON_DBG_MODES:-0 1 2 3}; do
  rm -rf /tmp/prof_rd$d
  RT_RECON_DEBUG=$d RT_AMD_LIB=$PWD/build_variants/librt_rdbg.so timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/prof_rd$d -o run -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
  python3 tools/rocpd_kernel_stats.py "$(find /tmp/prof_rd$d -name '*.db' | head -1)" /tmp/rd$d.csv
  python3 -c "
import csv,sys
for r in csv.reader(open('/tmp/rd$d.csv')):
    if r and r[0].startswith('k_splat_rows_need'): print('debug $d: calls', r[1], 'avg ms', float(r[3])/1e6)
"
done
