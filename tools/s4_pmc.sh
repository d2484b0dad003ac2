#!/bin/bash
# SQ counters over the steady state of S4 (frames after the first two, which carry no memory of their levels): tools/s4_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pmc_s4s -o p --output-format csv -- python3 bench.py --scene S4 --width 960 --height 540 --spp 4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_s4s.log 2>&1
python3 tools/pmc_sum.py gpurun_out/pmc_s4s
