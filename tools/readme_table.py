#!/usr/bin/env python3
"""README's configuration table from a tools/bench_configs.py output:  python tools/readme_table.py profiles/r02_configs.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print('| config | n | dtype | fwd µs | fwd+bwd µs |')
print('|---|---|---|---|---|')
h = d.get('_host')
for name, r in d.items():
    if name.startswith('_'):
        continue
    if 'step_us' in r:
        print(f"| {name} | {r.get('n', '')} | {r.get('dtype', '')} | | step {r['step_us']:.0f} |")
    else:
        print(f"| {name} | {r.get('n', '')} | {r.get('dtype', '')} | {r['fwd_us']:.0f} | {r['fwd_bwd_us']:.0f} |")
if h:
    print(f"\nHost of this run: one eager torch launch {h['launch_us']:.1f} µs, one C-ABI call through ctypes {h['cabi_us']:.2f} µs, "
          f"1000 iterations of a python loop {h['python_us']:.0f} µs ({h['host_cpus']} CPUs) — the yardstick for the eager rows.")
