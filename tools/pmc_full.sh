#!/bin/bash
# usage (on the GPU box): tools/pmc_full.sh <tag> <counters...> -- one PMC pass over one frame of the bench workload (S0 4K 64 spp)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 170 rocprofv3 --pmc "$@" -d gpurun_out/pmc_$tag -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
