#!/usr/bin/env python3
"""Static instruction budget of the SPD backward's per-pair phases (tools/micro/phase_budget.hip): compiles the phase kernels
with the library's flags, disassembles them and prints, per phase, the vector / scalar instruction counts minus those of the
kernel that only moves the same records (`base_*`).  Runs anywhere hipcc does (no GPU):  python tools/phase_budget.py"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = '-O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-but-set-variable -Wno-unused-value'.split()


def main():
    with tempfile.TemporaryDirectory() as d:
        obj = os.path.join(d, 'x.o')
        subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-c', os.path.join(ROOT, 'tools/micro/phase_budget.hip'), '-o', obj],
                       check=True, stderr=subprocess.DEVNULL)
        asm = subprocess.run(['bash', os.path.join(ROOT, 'tools/devasm.sh'), obj], check=True, capture_output=True, text=True).stdout
    counts, cur = {}, None
    for ln in asm.splitlines():
        m = re.match(r'^[0-9a-f]+ <(\w+)>:', ln)
        if m:
            cur = collections.Counter()
            counts[m.group(1)] = cur
            continue
        m = re.match(r'^\s+(\S+)\s', ln)
        if m and cur is not None:
            op = m.group(1)
            if op.startswith('v_'):
                cur['valu'] += 1
                if 'f64' in op:
                    cur['f64'] += 1
                if re.match(r'v_(rcp|rsq|sqrt|log|exp|sin|cos)', op):
                    cur['trans'] += 1
                if op.startswith(('v_permlane', 'v_readlane', 'v_writelane')) or 'dpp' in ln:
                    cur['cross'] += 1
            elif op.startswith('s_') and not op.startswith(('s_waitcnt', 's_nop', 's_endpgm', 's_load', 's_branch', 's_cbranch')):
                cur['salu'] += 1
            elif op.startswith('ds_'):
                cur['lds'] += 1
    rows = [('congr', 'A = L_i^-1 X_j L_i^-T (congr_chol)'), ('gate', 'close-pair gate ||A - I||_F^2'),
            ('logclose', 'log A, close-pair series (log_series3/4)'), ('logcentred', 'log A, recentred series (fp32)'),
            ('logcayley', 'log A, Cayley-transform form'), ('colcongr', 'column side L_i^T M L_j^T into the accumulators'),
            ('rowred', 'row side: transposing reduction of M, per wavefront row (64 pairs)')]
    for tn, tname in (('f', 'fp32'), ('d', 'fp64')):
        for D in (3, 4):
            base = counts[f'base_{tn}{D}']
            print(f'SPD({D}) {tname}   (record mover: {base["valu"]} vector, {base["salu"]} scalar instructions, subtracted)')
            for key, label in rows:
                c = counts.get(f'{key}_{tn}{D}')
                if c is None:
                    continue
                extra = f', {c["f64"]} of them 64-bit' if tn == 'd' else ''
                print(f'  {label:72s} {c["valu"] - base["valu"]:4d} vector{extra} ({c["trans"]} transcendental, '
                      f'{c["cross"]} cross-lane, {c["lds"]} LDS), {max(c["salu"] - base["salu"], 0):3d} scalar')
    return 0


if __name__ == '__main__':
    sys.exit(main())
