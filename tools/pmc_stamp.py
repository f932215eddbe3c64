#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes of `python3 bench.py` (separate passes for SQ counters, FETCH_SIZE and WRITE_SIZE, as
MI355X_MICROARCH.md prescribes) into profiles/pmc_head.json, stamped with the hash of the kernel sources they were
measured on; bench.py quotes `roofline.traffic` from it only while the hash matches the tree.
    python3 tools/pmc_stamp.py <dir with *counter_collection.csv> [...] --source "profiles/rXX_..." [--pairs 12497500]
FETCH_SIZE is reported in KiB and counts HALF the bytes of the kernels' streaming reads on gfx950 (calibrated with
tools/micro/load_bw.hip on the backward's own access pattern: 49.99 MB read -> 24 420 KiB for 16-B and 4-B per lane
alike; the triangular pair-vector pattern reads 1.46x its bytes because its 256-B wave segments start at arbitrary 4-B
offsets and touch three 128-B lines): traffic = 2 x FETCH_SIZE + WRITE_SIZE."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KERNELS = {'spd_pdist_bwd': 'spd_pdist_bwd_kernel<float, 3', 'spd_pdist_fwd': 'spd_pdist_fwd_kernel<float, 3'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dirs', nargs='+')
    ap.add_argument('--source', required=True)
    ap.add_argument('--pairs', type=int, default=12497500)
    args = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in args.dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                for key, pat in KERNELS.items():
                    if pat in r['Kernel_Name']:
                        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    out = {'kernel_source_hash': bench.kernel_source_hash(), 'source': args.source, 'pairs_per_launch': args.pairs, 'kernels': {}}
    for key, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        rec = {'counters_per_launch': m}
        if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
            rec['traffic_bytes'] = (2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024
            rec['fetch_bytes_x2'] = 2 * m['FETCH_SIZE'] * 1024
            rec['write_bytes'] = m['WRITE_SIZE'] * 1024
        if 'SQ_INSTS_VALU' in m:
            rec['valu_insts_per_64_pairs'] = m['SQ_INSTS_VALU'] / (args.pairs / 64)
        if 'SQ_ACTIVE_INST_ANY' in m:
            rec['insts_any_per_64_pairs'] = m['SQ_ACTIVE_INST_ANY'] / (args.pairs / 64)
        out['kernels'][key] = rec
    path = os.path.join(ROOT, 'profiles', 'pmc_head.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:600])


if __name__ == '__main__':
    main()
