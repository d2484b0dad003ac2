#!/bin/bash
# texture-addresser / vector-L1 / L2 counters of the S4 frame's kernels (two frames): tools/pmc_s4_mem.sh <tag>   (RT_AMD_LIB picks a variant)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --scene S4 --width 1920 --height 1080 --spp 4 --steps 1 --warmup 1 --no-cpu-baseline"
timeout -s KILL 170 rocprofv3 --pmc TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -d gpurun_out/pmc_${tag}_m1 -o p --output-format csv -- $B > gpurun_out/pmc_${tag}_m1.log 2>&1 || exit 1
timeout -s KILL 170 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_FLAT_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d gpurun_out/pmc_${tag}_m2 -o p --output-format csv -- $B > gpurun_out/pmc_${tag}_m2.log 2>&1 || exit 1
python3 tools/pmc_sum.py gpurun_out/pmc_${tag}_m1 gpurun_out/pmc_${tag}_m2 > gpurun_out/pmc_${tag}_mem.txt
