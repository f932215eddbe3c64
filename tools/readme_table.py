#!/usr/bin/env python3
"""README's configuration table from a tools/bench_configs.py output:  python tools/readme_table.py profiles/r02_configs.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print('| config | n | dtype | fwd µs | fwd+bwd µs |')
print('|---|---|---|---|---|')
for name, r in d.items():
    if 'step_us' in r:
        print(f"| {name} | {r.get('n', '')} | {r.get('dtype', '')} | | step {r['step_us']:.0f} |")
    else:
        print(f"| {name} | {r.get('n', '')} | {r.get('dtype', '')} | {r['fwd_us']:.0f} | {r['fwd_bwd_us']:.0f} |")
