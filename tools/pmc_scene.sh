#!/bin/bash
# usage (on the GPU box): tools/pmc_scene.sh <tag> <scene> <width> <height> <spp> <counters...> -- one PMC pass over one frame of a bench workload
tag=$1; scene=$2; w=$3; h=$4; spp=$5; shift 5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 170 rocprofv3 --pmc "$@" -d gpurun_out/pmc_$tag -o p --output-format csv -- python3 bench.py --scene $scene --width $w --height $h --spp $spp --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
