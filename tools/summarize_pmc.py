#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: per kernel name, mean of each counter per dispatch."""
import csv, sys, collections, glob, os
for path in sys.argv[1:]:
    for f in glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        print('#', f)
        for k, cs in acc.items():
            if not any(t in k for t in ('spd_pdist', 'vec_', 'product_pair')): continue
            print(k, {c: (sum(v) / len(v)) for c, v in cs.items()}, 'dispatches', len(next(iter(cs.values()))))
