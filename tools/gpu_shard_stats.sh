#!/bin/bash
# per-kernel rocprofv3 averages of ONE rank's shard of the headline problem (N = 8: first and last rank), on one GPU
cd /tmp && export TMPDIR=/tmp
for R in 0 7; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sc$R -o s -- python3 /root/repo/tools/shard_case.py 8 $R > /dev/null 2>&1
  echo "== N=8 rank $R"
  python3 - /tmp/sc$R <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spd_" in r["Name"]: print("  ", r["Name"][9:60], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
