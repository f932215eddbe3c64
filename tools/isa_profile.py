#!/usr/bin/env python3
"""Instruction mix of one kernel's basic blocks from an llvm-objdump -d listing of the gfx950 code object
(tools/snap_make.sh leaves build/var_NAME/spd.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 next to the object):
  llvm-objdump -d <code object> > /tmp/x.s;  python tools/isa_profile.py /tmp/x.s <mangled-name substring> [min block size]
Blocks are split at branch targets and branches; per block: instruction count and the classes that matter for the pair
kernels (v_readlane / v_writelane = spilled scalar registers, v_mov = copies, s_waitcnt, 64-bit FMAs, transcendentals)."""
import re
import sys
from collections import Counter


def main():
    path, key = sys.argv[1], sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if re.match(r'^[0-9a-f]+ <.*' + re.escape(key), ln))
    body = []
    for ln in lines[start + 1:]:
        if re.match(r'^[0-9a-f]+ <', ln):
            break
        m = re.match(r'^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):', ln)
        if m:
            body.append((int(m.group(3), 16), m.group(1), m.group(2)))
    targets = set()
    for addr, op, args in body:
        if op.startswith('s_cbranch') or op == 's_branch':
            m = re.search(r'<.*\+0x([0-9a-f]+)>', args)
            if m:
                targets.add(m.group(1))
    base = body[0][0]
    blocks, cur = [], []
    for addr, op, args in body:
        if format(addr - base, 'x') in targets and cur:
            blocks.append(cur); cur = []
        cur.append((addr, op, args))
        if op.startswith('s_cbranch') or op == 's_branch' or op == 's_endpgm':
            blocks.append(cur); cur = []
    if cur:
        blocks.append(cur)
    print(f'{len(body)} instructions, {len(blocks)} blocks; blocks of >= {minsz} instructions:')
    print(f"{'offset':>8} {'n':>5} {'valu':>5} {'salu':>5} {'f64fma':>6} {'trans':>5} {'rdlane':>6} {'wrlane':>6} {'v_mov':>5} {'waits':>5} {'smem':>4} {'vmem':>4} {'lds':>4}")
    for b in blocks:
        if len(b) < minsz:
            continue
        c = Counter()
        for addr, op, args in b:
            if op.startswith('v_'):
                c['valu'] += 1
            elif op.startswith('s_load') or op.startswith('s_buffer'):
                c['smem'] += 1
            elif op.startswith('s_'):
                c['salu'] += 1
            if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
                c["vmem"] += 1
            if op.startswith('ds_'):
                c['lds'] += 1
            if re.match(r'v_(fma|mul|add|fmac)_f64', op):
                c['f64'] += 1
            if re.match(r'v_(rcp|rsq|sqrt|log|exp|sin|cos)_', op):
                c['trans'] += 1
            if op.startswith('v_readlane') or op.startswith('v_readfirstlane'):
                c['rd'] += 1
            if op.startswith('v_writelane'):
                c['wr'] += 1
            if op.startswith('v_mov') or op.startswith('v_accvgpr'):
                c['mov'] += 1
            if op == 's_waitcnt':
                c['wait'] += 1
        print(f"{b[0][0] - base:>8x} {len(b):>5} {c['valu']:>5} {c['salu']:>5} {c['f64']:>6} {c['trans']:>5} {c['rd']:>6} {c['wr']:>6} {c['mov']:>5} {c['wait']:>5} {c['smem']:>4} {c['vmem']:>4} {c['lds']:>4}")


if __name__ == '__main__':
    main()
