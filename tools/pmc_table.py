#!/usr/bin/env python3
"""HBM bytes per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_scene.sh) over one frame.
usage: pmc_table.py <fetch_dir> <write_dir> [title...]   (KiB counters; FETCH_SIZE doubled on gfx950, see pmc_traffic.py)"""
import collections, csv, glob, sys


def sums(d):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


fetch, nf = sums(sys.argv[1])
write, nw = sums(sys.argv[2])
if len(sys.argv) > 3:
    print(" ".join(sys.argv[3:]))
print("%-38s %8s %18s %18s" % ("kernel", "launches", "read B (2xFETCH)", "written B"))
rows = sorted(set(fetch) | set(write), key=lambda k: -(2048.0 * fetch.get(k, 0) + 1024.0 * write.get(k, 0)))
tr = tw = 0.0
for k in rows:
    rd, wr = 2048.0 * fetch.get(k, 0.0), 1024.0 * write.get(k, 0.0)
    tr += rd; tw += wr
    if rd + wr >= 1e6:
        print("%-38s %8d %18.0f %18.0f" % (k[:38], max(nf.get(k, 0), nw.get(k, 0)), rd, wr))
print("%-38s %8s %18.0f %18.0f" % ("all kernels", "", tr, tw))
