#!/bin/bash
# on the GPU box: tools/pmc_traffic.sh <tag> <kernel> -- FETCH_SIZE and WRITE_SIZE passes (runs of their own) over one frame of the
# default bench workload, then gpurun_out/<tag>_pmc_traffic_all_kernels.txt and gpurun_out/<tag>_pmc_traffic.json
tag=$1; kernel=$2
cd "$(dirname "$0")/.." || exit 1
tools/pmc_full.sh ${tag}_fetch FETCH_SIZE && tools/pmc_full.sh ${tag}_write WRITE_SIZE || exit 1
timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_bench.json 2>/dev/null || exit 1
python3 tools/pmc_traffic.py gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/${tag}_pmc_bench.json $kernel "profiles/${tag}_pmc_traffic_all_kernels.txt" gpurun_out/${tag}_pmc_traffic.json gpurun_out/${tag}_pmc_traffic_all_kernels.txt
