#!/bin/bash
# usage (on the GPU box): tools/pmc_k.sh <tag> <counters...> -- collects one PMC pass over a small bench run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc "$@" -d gpurun_out/pmc_$tag -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --width 1920 --height 1080 --spp 16 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
