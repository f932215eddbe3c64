#!/bin/bash
# Copy the judged summaries of an evidence run (tools/gpu_evidence.sh TAG, merged back under gpurun_out/TAG) into profiles/:
#   tools/collect_profiles.sh TAG [ROUND]      # ROUND = prefix of the committed files, default r06
# profiles/pmc_head.json is the stamp written on the GPU box (same kernel-source hash as the tree that ran); the counter
# summary it cites is regenerated here from the same CSVs.
set -e
TAG=$1; R=${2:-r06}
cd "$(dirname "$0")/.."
G=gpurun_out/$TAG
cp $G/summary.txt profiles/${R}_summary.txt
cp $G/bench.json profiles/${R}_bench.json
cp $G/configs.json profiles/${R}_configs.json
cp $G/shard_kernel_times.json profiles/${R}_shard_kernel_times.json
cp $G/bench_gloo2_dryrun.json profiles/${R}_bench_gloo2_dryrun.json 2>/dev/null || true
for f in eager_euclid eager_spd timeline_cold timeline_warm product_timeline; do [ -s $G/$f.txt ] && cp $G/$f.txt profiles/${R}_$f.txt; done
for d in $G/*_stats; do
  n=$(basename $d _stats)
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f profiles/${R}_${n}_kernel_stats.csv
done
grep -a "passed\|failed" $G/pytest_gpu.log | tail -1 > profiles/${R}_pytest_gpu_tail.txt
cp $G/pmc_head.json profiles/pmc_head.json
rm -f profiles/${R}v*_bench_pmc_summary.txt
python3 tools/summarize_pmc.py $G/bench_pmc_sq $G/bench_pmc_fetch $G/bench_pmc_write > profiles/${TAG}_bench_pmc_summary.txt
python3 - <<PY
import json, sys
sys.path.insert(0, '.')
import bench
h = json.load(open('profiles/pmc_head.json'))['kernel_source_hash']
print('pmc stamp', h, 'tree', bench.kernel_source_hash(), 'MATCH' if h == bench.kernel_source_hash() else 'STALE')
PY
