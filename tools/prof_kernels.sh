#!/bin/bash
# rocprofv3 kernel summary of a bench.py run on the GPU box: tools/prof_kernels.sh <tag> [bench args...]
# writes gpurun_out/<tag>_kernel_stats.csv and gpurun_out/<tag>_bench.json
tag=$1; shift
cd "$(dirname "$0")/.." || exit 1
out=$PWD/gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o run -- python3 bench.py --no-cpu-baseline "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
python3 tools/rocpd_kernel_stats.py "$db" $out/${tag}_kernel_stats.csv
python3 - $out/${tag}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[:14]:
    print("%-90s %6s %14s %12s %7s" % (r[0][:90], r[1], r[2], r[3], r[4]))
PY
