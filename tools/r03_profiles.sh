#!/bin/bash
# on the GPU box: kernel summaries, bench lines and PMC traffic of round 3's final state -> gpurun_out/<tag>_*
tag=${1:-r03}
cd "$(dirname "$0")/.." || exit 1
tools/prof_kernels.sh ${tag}_S0_4k64 --steps 5 --warmup 2 &&
tools/prof_kernels.sh ${tag}_S1_4k64 --scene S1 --steps 3 --warmup 2 &&
tools/prof_kernels.sh ${tag}_S2_4k64 --scene S2 --steps 3 --warmup 2 &&
tools/prof_kernels.sh ${tag}_S4_1080p4 --scene S4 --width 1920 --height 1080 --spp 4 --steps 3 --warmup 2 &&
tools/prof_kernels.sh ${tag}_MESH10M_4k64 --scene MESH10M --steps 2 --warmup 1 &&
tools/pmc_traffic.sh $tag k_fast &&
cp gpurun_out/${tag}_pmc_traffic.json ray-tracer_amd/data/pmc_traffic.json &&
timeout -k 10 300 python3 bench.py > gpurun_out/${tag}_S0_4k64_bench_with_cpu_baseline.json 2> gpurun_out/${tag}_bench_default.err &&
tools/pmc_s0_sq.sh ${tag}_s0 && cp gpurun_out/pmc_${tag}_s0_sq.txt gpurun_out/${tag}_pmc_sq_S0_4k64_two_frames.txt &&
tools/pmc_scene.sh ${tag}_s4_fetch S4 1920 1080 4 FETCH_SIZE && tools/pmc_scene.sh ${tag}_s4_write S4 1920 1080 4 WRITE_SIZE &&
python3 tools/pmc_sum.py gpurun_out/pmc_${tag}_s4_fetch gpurun_out/pmc_${tag}_s4_write > gpurun_out/${tag}_pmc_S4_1080p4.txt &&
tools/pmc_scene.sh ${tag}_mesh_fetch MESH10M 1920 1080 16 FETCH_SIZE && tools/pmc_scene.sh ${tag}_mesh_write MESH10M 1920 1080 16 WRITE_SIZE &&
python3 tools/pmc_sum.py gpurun_out/pmc_${tag}_mesh_fetch gpurun_out/pmc_${tag}_mesh_write > gpurun_out/${tag}_pmc_MESH10M_1080p16.txt &&
timeout -k 10 300 python3 bench.py --scene MESH10M --width 1920 --height 1080 --spp 16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_MESH10M_1080p16_bench.json 2>/dev/null
