#!/usr/bin/env python3
"""Table of per-kernel resources from a hipcc log made with -Rpass-analysis=kernel-resource-usage (tools/snap_make.sh NAME
writes /tmp/build_NAME.log):  python tools/kernel_resources.py /tmp/build_NAME.log [substring ...]"""
import re
import subprocess
import sys


def main():
    log = open(sys.argv[1]).read().splitlines()
    pats = sys.argv[2:]
    rows, cur = [], None
    for ln in log:
        m = re.search(r'remark: (?:\s*)(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)', ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == 'Function Name':
            cur = {'name': v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(' [')[0]] = v
    names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'VGPR':>5} {'SGPR':>5} {'occ':>3} {'sSpill':>6} {'vSpill':>6} {'scratch':>7} {'LDS':>6}  kernel")
    for r, nm in zip(rows, names):
        nm = re.sub(r'\(.*', '', nm).replace('void mm::', '')
        if pats and not all(p in nm for p in pats):
            continue
        print(f"{r.get('VGPRs', '?'):>5} {r.get('TotalSGPRs', '?'):>5} {r.get('Occupancy', '?'):>3} {r.get('SGPRs Spill', '?'):>6} {r.get('VGPRs Spill', '?'):>6} "
              f"{r.get('ScratchSize', '?'):>7} {r.get('LDS Size', '?'):>6}  {nm}")


if __name__ == '__main__':
    main()
