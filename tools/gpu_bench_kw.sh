#!/bin/bash
# the headline's sensitivity to the driver's K / W (steps / warm-up)
cd $GRAFT_REPO_ROOT
for k in "50 10" "5 1" "20 0" "200 10"; do
  set -- $k
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps', d['steps'], 'warmup', d['warmup'], 'us/step', round(d['ms_per_step']*1e3,1), 'G pairs/s', round(d['value']/1e9,1))"
done
