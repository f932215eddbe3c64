#!/bin/bash
# Issue-side counters of the headline backward (and of tools/micro/bwd_rate's arithmetic-only loop) in separate --pmc passes:
#   tools/gpu_pmc_issue.sh TAG          -> gpurun_out/TAG/pmc_issue_{case,micro}_P*/ + pmc_issue.txt
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CU_CYCLES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_LDS"
P4="SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_THREAD_CYCLES_VALU"
P5="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
P6="SQ_INSTS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_issue_case_P$i -o p -- python3 /root/repo/tools/profile_case.py pdist 3 5000 f32 0.1 3 > /dev/null 2>&1
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_issue_micro_P$i -o p -- /root/repo/tools/micro/bwd_rate > /dev/null 2>&1
done
python3 - $OUT <<'PY' > $OUT/pmc_issue.txt
import csv, glob, os, sys, collections
out = sys.argv[1]
for which in ('case', 'micro'):
    tot = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(out, f'pmc_issue_{which}_P*'))):
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                nm = r['Kernel_Name'].split('(')[0].replace('void mm::', '')
                if which == 'case' and 'pdist_bwd' not in nm and 'pdist_fwd' not in nm:
                    continue
                tot[nm][r['Counter_Name']].append(float(r['Counter_Value']))
    for nm, cs in tot.items():
        print(f'== {which}: {nm[:90]}')
        for c, v in sorted(cs.items()):
            print(f'   {c:28s} {sum(v) / len(v):16.0f}   (x{len(v)})')
PY
cat $OUT/pmc_issue.txt
