#!/bin/bash
# same-box A/B of ENVIRONMENT switches of one library build:
#   tools/gpu_ab_env.sh "NAME1:VAR=VAL VAR2=VAL" "NAME2:VAR=VAL" ...      ("NAME:" alone = no variable set)
# alternates the settings MM_AB_ROUNDS (2) times over the cases in MM_AB_CASES (';'-separated argument lists of
# tools/profile_case.py) and prints the rocprofv3 averages per kernel.
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_env
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra CASES <<< "${MM_AB_CASES:-pdist 3 5000 f32 0.1;pdist 3 5000 f32 0.35}"
for ROUND in $(seq 1 ${MM_AB_ROUNDS:-2}); do
  for SPEC in "$@"; do
    V=${SPEC%%:*}; SETTINGS=${SPEC#*:}
    for CASE in "${CASES[@]}"; do
      NAME=$(echo $CASE | tr ' .' '__')_${V}_$ROUND
      ( for KV in $SETTINGS; do export "$KV"; done
        rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$NAME -o s -- python3 /root/repo/tools/profile_case.py $CASE 60 > /dev/null 2>&1 )
      python3 - $OUT/$NAME/s_kernel_stats.csv "$V round $ROUND: $CASE" <<'PY'
import csv, sys
out, tot = [], 0.0
for r in csv.DictReader(open(sys.argv[1])):
    nm = r['Name']
    if 'pdist_bwd' in nm or 'pdist_fwd' in nm or 'gram' in nm or 'fused_step' in nm or 'prep' in nm or 'finalize' in nm:
        short = nm.split('(')[0].replace('void mm::', '').replace('spd_pdist_', '').replace('_kernel', '')[:44]
        out.append('%s avg %.1f min %.1f (x%s)' % (short, float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, r['Calls']))
        tot += float(r['AverageNs']) / 1e3
print(sys.argv[2], '| sum %.1f |' % tot, '; '.join(out))
PY
    done
  done
done
