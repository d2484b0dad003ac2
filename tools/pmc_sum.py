#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: pmc_sum.py dir [dir...]"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(agg.items()):
    print(k)
    for c, x in sorted(v.items()):
        print('   %-32s %16.0f' % (c, x))
