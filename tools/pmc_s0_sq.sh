#!/bin/bash
# SQ instruction mix of the default bench frame's kernels: tools/pmc_s0_sq.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS -d gpurun_out/pmc_${tag}_a -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${tag}_a.log 2>&1 || exit 1
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAVES -d gpurun_out/pmc_${tag}_b -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${tag}_b.log 2>&1 || exit 1
python3 tools/pmc_sum.py gpurun_out/pmc_${tag}_a gpurun_out/pmc_${tag}_b > gpurun_out/pmc_${tag}_sq.txt
