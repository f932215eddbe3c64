#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel in the BUILT library, from the code objects' metadata notes (no compile, no
GPU): llvm-objdump --offloading extracts the gfx950 code objects of lib/libmm_manifolds.so, llvm-readelf --notes lists per kernel
.vgpr_count / .vgpr_spill_count / .sgpr_spill_count / .private_segment_fixed_size.
    python tools/kernel_meta.py [substring of the demangled name]          # table
`kernels()` is what tests/test_kernel_budget.py asserts on."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
LIB = os.path.join(ROOT, 'matrix-manifolds_amd', 'lib', 'libmm_manifolds.so')


def kernels(lib=LIB):
    """{demangled kernel name: dict(vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds)}"""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(lib, os.path.join(d, 'x.so'))
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', 'x.so'], cwd=d, check=True, capture_output=True)
        names, rows = [], []
        for f in sorted(os.listdir(d)):
            if 'hipv4-amdgcn' not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', f], cwd=d, capture_output=True, text=True).stdout
            cur = None
            for ln in notes.splitlines():
                m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)$', ln)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip().strip("'")
                if key == 'agpr_count' or (key == 'group_segment_fixed_size'):
                    if key == 'agpr_count':
                        cur = {'agpr': int(val)}
                        rows.append(cur)
                    elif cur is not None:
                        cur['lds'] = int(val)
                    if key == 'group_segment_fixed_size' and cur is not None and 'lds' not in cur:
                        cur['lds'] = int(val)
                elif cur is not None and key in ('name', 'private_segment_fixed_size', 'sgpr_count', 'sgpr_spill_count', 'vgpr_count',
                                                 'vgpr_spill_count', 'group_segment_fixed_size'):
                    cur[key] = val if key == 'name' else int(val)
        mangled = [r['name'] for r in rows if 'name' in r]
        dem = subprocess.run(['c++filt'], input='\n'.join(mangled), capture_output=True, text=True).stdout.splitlines()
        for r, nm in zip([r for r in rows if 'name' in r], dem):
            nm = re.sub(r'^void ', '', nm).split('(')[0].replace('mm::', '').replace('(anonymous namespace)::', '')
            out[nm] = dict(vgpr=r.get('vgpr_count', 0), agpr=r.get('agpr', 0), sgpr=r.get('sgpr_count', 0),
                           vgpr_spill=r.get('vgpr_spill_count', 0), sgpr_spill=r.get('sgpr_spill_count', 0),
                           scratch=r.get('private_segment_fixed_size', 0), lds=r.get('group_segment_fixed_size', r.get('lds', 0)))
    return out


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else ''
    ks = kernels()
    print(f'{"VGPR":>5} {"AGPR":>5} {"SGPR":>5} {"vSpill":>6} {"sSpill":>6} {"scratch":>7} {"LDS":>6}  kernel')
    for nm in sorted(ks):
        if key in nm:
            r = ks[nm]
            print(f'{r["vgpr"]:5d} {r["agpr"]:5d} {r["sgpr"]:5d} {r["vgpr_spill"]:6d} {r["sgpr_spill"]:6d} {r["scratch"]:7d} {r["lds"]:6d}  {nm}')
    print(len(ks), 'kernels')


if __name__ == '__main__':
    main()
