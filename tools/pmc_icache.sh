#!/bin/bash
# instruction-cache counters of a scene's kernels (one frame after one warm-up): tools/pmc_icache.sh <tag> <scene>   (RT_AMD_LIB picks a variant)
tag=$1; scene=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 170 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d gpurun_out/pmc_${tag}_ic -o p --output-format csv -- python3 bench.py --scene $scene --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${tag}_ic.log 2>&1 || exit 1
python3 tools/pmc_sum.py gpurun_out/pmc_${tag}_ic > gpurun_out/pmc_${tag}_icache.txt
