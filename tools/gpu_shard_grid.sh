#!/bin/bash
# backward of ONE rank's shard of the headline problem (N = 8, rank 3) for several workgroup counts (MM_SPD_BWD_GRID), rocprofv3 averages
cd /tmp && export TMPDIR=/tmp
for G in ${@:-0 256 384 512 768 1024}; do
  if [ "$G" = 0 ]; then unset MM_SPD_BWD_GRID; else export MM_SPD_BWD_GRID=$G; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sg$G -o s -- python3 /root/repo/tools/shard_case.py 8 3 > /dev/null 2>&1
  python3 - /tmp/sg$G $G <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pdist_bwd" in r["Name"]: print("grid", sys.argv[2], "bwd avg %.1f us min %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
