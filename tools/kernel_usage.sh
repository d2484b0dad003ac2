#!/bin/bash
# VGPRs / private segment / LDS of every kernel of librt_amd.so as hipcc builds it (no GPU needed): tools/kernel_usage.sh [pattern]
cd "$(dirname "$0")/../ray-tracer_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -c -Rpass-analysis=kernel-resource-usage '-DRT_TABLES_PATH="x"' $RT_EXTRA_DEFS rt_kernels.hip -o /tmp/rt_usage.o 2> /tmp/rt_usage.txt
python3 - "$1" <<'PY'
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
txt = open("/tmp/rt_usage.txt").read()
for blk in txt.split("Function Name: ")[1:]:
    name = blk.split()[0]
    if pat and pat not in name: continue
    g = lambda k: (re.search(k + r": (\d+)", blk) or [0, "?"])[1]
    print("%-70s VGPR %4s AGPR %3s scratch %6s B/lane  LDS %6s  occupancy %s" % (name[:70], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")))
PY
