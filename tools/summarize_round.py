#!/usr/bin/env python3
"""Text summary of a tools/gpu_evidence.sh output directory: per case and kernel the average duration
(rocprofv3 --kernel-trace --stats), the PMC counters per launch, HBM traffic (2 x FETCH_SIZE + WRITE_SIZE, KiB) and the
algorithmic-bytes roofline fraction of the pair kernels."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
HBM = 8000.0  # GB/s
WARM_GHZ = 2.38  # shader clock of the timed regime (tools/stamp_timeline.py --warm)


def stats(d):
    out = {}
    for f in glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'mm::' in r['Name']:
                out[r['Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void mm::', '')] = (int(r['Calls']), float(r['AverageNs']) / 1e3,
                                                                          float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3)
    return out


def pmc(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'mm::' in r['Kernel_Name']:
                    acc[r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void mm::', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


CASES = {'bench': ('SPD(3) f32 n=5000 reference init (headline; python3 bench.py)', 3, 5000, 4),
         'case_pdist_3_5000_f64_0_1': ('SPD(3) f64 n=5000 reference init', 3, 5000, 8),
         'case_pdist_3_5000_f32_0_35': ('SPD(3) f32 n=5000 mid-training spread (||log X|| = 0.35)', 3, 5000, 4),
         'case_pdist_3_5000_f64_0_35': ('SPD(3) f64 n=5000 mid-training spread (||log X|| = 0.35)', 3, 5000, 8),
         'case_pdist_4_2274_f32_0_1': ('SPD(4) f32 n=2274 (BASELINE config 5, small graph)', 4, 2274, 4),
         'case_pdist_4_16384_f32_0_1': ('SPD(4) f32 n=16384 pdist fwd + bwd, reference init', 4, 16384, 4),
         'case_pdist_4_16384_f32_0_35': ('SPD(4) f32 n=16384 pdist fwd + bwd, mid-training spread (||log X|| = 0.35)', 4, 16384, 4),
         'case_loss_4_16384_f32': ('SPD(4) f32 n=16384 fused QuotientLoss step (BASELINE config 5)', 4, 16384, 4),
         'case_pdist_6_2000_f32_0_1': ('SPD(6) f32 n=2000 pdist fwd + bwd (matrix series since round 5; Jacobi before)', 6, 2000, 4),
         'case_pdist_6_2000_f32_0_35': ('SPD(6) f32 n=2000 pdist fwd + bwd, mid-training spread (recentred matrix series)', 6, 2000, 4),
         'case_pdist_6_2000_f64_0_1': ('SPD(6) f64 n=2000 pdist fwd + bwd (matrix series, degree 19)', 6, 2000, 8),
         'case_pdist_9_2000_f32_0_1': ('SPD(9) f32 n=2000 pdist fwd + bwd (matrix series; the reference\'s largest test size)', 9, 2000, 4),
         'case_step_3_5000_f32': ('SPD(3) f32 n=5000 full training step through mm_train_step_run (pair kernel + fused finalize/update/tables)', 3, 5000, 4),
         'case_vec_11_4039_f32_lorentz': ('Lorentz(11) f32 n=4039 (BASELINE config 2) pdist fwd + bwd (default: matrix-core forward, symmetric VALU backward)', 11, 4039, 4),
         'case_vecgram_11_4039_f32_lorentz': ('Lorentz(11) f32 n=4039 pdist fwd + bwd with MM_VEC_BWD=gram (matrix-core backward)', 11, 4039, 4),
         'case_vstep_11_4039_f32_lorentz': ('Lorentz(11) f32 n=4039 (BASELINE config 2) full training step through mm_train_step_run (pair kernel + one per-point kernel)', 11, 4039, 4),
         'case_c5_spd4_minibatch512_step_n16384_f32_native_graph': ('SPD(4) f32 n=16384, node minibatches of 512 through mm_train_step_run (batch_idx): launches of a step', 4, 512, 4),
         'case_lorentz24_minibatch512_step_n4039_f32_native_graph': ('Lorentz(24) f32 n=4039, node minibatches of 512 through mm_train_step_run', 24, 512, 4),
         'case_product_1025': ('BASELINE config 4: H^5 x S^5 x SPD(2) f32 n=1025 training step (mixed-manifold pair kernel)', 2, 1025, 4),
         'case_product_5000': ('H^5 x S^5 x SPD(2) f32 n=5000 training step (symmetric mixed-manifold pair kernel)', 2, 5000, 4)}
for key, (title, d, n, esz) in CASES.items():
    st = stats(os.path.join(root, key + '_stats'))
    if not st:
        continue
    pm = pmc([os.path.join(root, key + s) for s in ('_pmc_sq', '_pmc_fetch', '_pmc_write', '_pmc_mfma')])
    pairs = n * (n - 1) // 2
    npk = d * (d + 1) // 2
    print(f'== {title}: {pairs} pairs')
    for name, (calls, avg, mn, mx) in sorted(st.items(), key=lambda kv: -kv[1][1]):
        line = f'  {name[:58]:58s} calls {calls:3d}  avg {avg:8.1f} us  (min {mn:.1f}, max {mx:.1f})'
        c = pm.get(name, {})
        if 'vec_gram' in name or 'vec_pdist_bwd_sym' in name:   # vector-manifold pair kernels: d is the vector dimension
            alg = pairs * esz + n * d * esz * 2
            line += f'  | algorithmic {alg / 1e6:.1f} MB -> {alg / avg / 1e3:.0f} GB/s = {alg / avg / 1e3 / HBM:.3f} of HBM peak'
            if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
                tr = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
                line += f'; traffic {tr / 1e6:.1f} MB (2 x FETCH {2 * c["FETCH_SIZE"] * 1024 / 1e6:.1f} + WRITE {c["WRITE_SIZE"] * 1024 / 1e6:.1f}) = {tr / alg:.2f} x algorithmic'
            if c.get('SQ_INSTS_VALU_MFMA_F32', 0) > 0 and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
                # matrix-core utilisation: cycles the MFMA pipe of a SIMD is busy, summed over the 1024 SIMDs, against the
                # kernel's duration at the warm shader clock (2.38 GHz, in-kernel stamps: profiles/r03_timeline_warm.txt)
                util = c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (avg * WARM_GHZ * 1e3)
                tf = c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) * 512 / (avg * 1e-6) / 1e12
                line += (f"; MFMA: {c.get('SQ_INSTS_VALU_MFMA_F32', 0):.0f} instructions, pipe busy {util:.1%} of the kernel's cycles "
                         f"({c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(c.get('SQ_INSTS_VALU_MFMA_F32', 1), 1):.0f} cycles each), "
                         f"{tf:.1f} TFLOP/s = {tf / 157.3:.1%} of the 157.3 TFLOP/s fp32 matrix peak")
            if 'sym' in name and 'SQ_INSTS_VALU' in c:
                line += f'; VALU {c["SQ_INSTS_VALU"] / (pairs / 64):.0f} / all {c.get("SQ_ACTIVE_INST_ANY", 0) / (pairs / 64):.0f} instructions per 64 pairs'
                if c.get('SQ_WAVE_CYCLES'):
                    line += f'; wait {c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]:.0%} issue-wait {c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]:.0%} of wave cycles'
        elif 'product_sym_kernel' in name:
            alg = pairs * esz
            line += f'  | algorithmic {alg / 1e6:.1f} MB (4 B target per pair) -> {alg / avg / 1e3:.0f} GB/s = {alg / avg / 1e3 / HBM:.4f} of HBM peak'
            if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
                tr = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
                line += f'; traffic {tr / 1e6:.1f} MB = {tr / alg:.2f} x algorithmic'
            if 'SQ_INSTS_VALU' in c:
                line += f'; VALU {c["SQ_INSTS_VALU"] / (pairs / 64):.0f} instructions per 64 UNORDERED pairs'
            if 'SQ_WAVE_CYCLES' in c and c['SQ_WAVE_CYCLES']:
                line += f'; wait {c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]:.0%} of wave cycles; LDS bank conflicts {c.get("SQ_LDS_BANK_CONFLICT", 0):.0f}'
        elif 'product_pair_kernel' in name:
            alg = pairs * esz
            line += f'  | algorithmic {alg / 1e6:.1f} MB (4 B target per pair) -> {alg / avg / 1e3:.0f} GB/s = {alg / avg / 1e3 / HBM:.4f} of HBM peak'
            if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
                tr = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
                line += f'; traffic {tr / 1e6:.1f} MB = {tr / alg:.2f} x algorithmic'
            if 'SQ_INSTS_VALU' in c:
                line += f'; VALU {c["SQ_INSTS_VALU"] / (pairs * 2 / 64):.0f} instructions per 64 ORDERED pairs'
            if 'SQ_WAVE_CYCLES' in c and c['SQ_WAVE_CYCLES']:
                line += f'; wait {c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]:.0%} of wave cycles; LDS bank conflicts {c.get("SQ_LDS_BANK_CONFLICT", 0):.0f}'
        elif 'pdist_bwd' in name or 'pdist_fwd' in name:
            alg = pairs * esz + n * (4 if 'bwd' in name else 2) * npk * esz
            line += f'  | algorithmic {alg / 1e6:.1f} MB -> {alg / avg / 1e3:.0f} GB/s = {alg / avg / 1e3 / HBM:.3f} of HBM peak'
            if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
                tr = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
                line += f'; traffic {tr / 1e6:.1f} MB (2 x FETCH {2 * c["FETCH_SIZE"] * 1024 / 1e6:.1f} + WRITE {c["WRITE_SIZE"] * 1024 / 1e6:.1f}) = {tr / alg:.2f} x algorithmic'
            if 'SQ_INSTS_VALU' in c:
                line += f'; VALU {c["SQ_INSTS_VALU"] / (pairs / 64):.0f} / all {c.get("SQ_ACTIVE_INST_ANY", 0) / (pairs / 64):.0f} instructions per 64 pairs'
            if 'SQ_WAVE_CYCLES' in c and c['SQ_WAVE_CYCLES']:
                line += f'; wait {c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]:.0%} issue-wait {c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]:.0%} of wave cycles'
        print(line)
