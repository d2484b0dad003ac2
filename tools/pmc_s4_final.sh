#!/bin/bash
# FETCH_SIZE / WRITE_SIZE over three frames of S4 (the third has its levels' memory: direct, sorted): tools/pmc_s4_final.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
timeout -s KILL 170 rocprofv3 --pmc $c -d gpurun_out/pmc_${tag}_$c -o p --output-format csv -- python3 bench.py --scene S4 --width 1920 --height 1080 --spp 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${tag}_$c.log 2>&1 || exit 1
done
python3 tools/pmc_table.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE "S4 1920x1080 4 spp, THREE frames (3 x 126 M rays; final state: pool, sorted deep levels, twelve frames per root)"
