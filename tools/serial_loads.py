#!/usr/bin/env python3
"""Serial memory round trips in a kernel's straight-line code: counts, per kernel of an llvm-objdump -d listing
(tools/devasm.sh), the groups of vector-memory loads that are separated by a FULL drain (`s_waitcnt vmcnt(0)`) with further
loads behind it — each such group is one exposed round trip.  Found the six-deep prologue of the mixed-manifold pair kernel
(round 4).  Reported: total groups, and groups before the first backward branch (the prologue).
    tools/devasm.sh build/x.o > /tmp/x.s; python tools/serial_loads.py /tmp/x.s [name substring] [min groups]"""
import re
import sys


def main():
    path = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else ''
    min_groups = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    name, rows = None, []
    out = []

    def flush():
        if name is None or key not in name:
            return
        groups, prologue, pending, seen_back = 0, 0, False, False
        for addr, op, args in rows:
            if op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')):
                pending = True
            elif op == 's_waitcnt' and re.search(r'vmcnt\(0\)', args) and pending:
                groups += 1
                if not seen_back:
                    prologue += 1
                pending = False
            elif op.startswith('s_cbranch') or op == 's_branch':
                m = re.search(r'(?:^|\s)(\d+)\s*$', args.split('//')[0])
                if m and int(m.group(1)) > 32767:
                    seen_back = True
        if groups >= min_groups:
            out.append((prologue, groups, len(rows), name))

    for ln in open(path):
        m = re.match(r'^[0-9a-f]+ <(.*)>:', ln)
        if m:
            flush()
            name, rows = m.group(1), []
            continue
        m = re.match(r'^\s+(\S+)\s*(.*)$', ln)
        if m and name is not None:
            rows.append((0, m.group(1), m.group(2)))
    flush()
    import subprocess
    for prologue, groups, n, nm in sorted(out, reverse=True):
        try:
            nm = subprocess.run(['c++filt', nm], capture_output=True, text=True).stdout.strip() or nm
        except OSError:
            pass
        print(f'{prologue:3d} drained load groups before the first loop ({groups:3d} in all, {n:6d} instructions)  {nm[:150]}')


if __name__ == '__main__':
    main()
