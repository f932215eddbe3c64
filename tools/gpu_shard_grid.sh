#!/bin/bash
# backward kernel of one rank's shard (N = 8 and N = 4, rank 0) against the number of workgroups of its launch
cd /tmp && export TMPDIR=/tmp
for W in 8 4; do
for G in 128 192 256 384 512 768 1024; do
  MM_SPD_BWD_GRID=$G rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sg${W}_$G -o s -- python3 /root/repo/tools/shard_case.py $W 0 > /dev/null 2>&1
  python3 - /tmp/sg${W}_$G "N=$W grid=$G" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pdist_bwd" in r["Name"]: print(sys.argv[2], "bwd avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
done
