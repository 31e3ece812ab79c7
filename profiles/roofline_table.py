#!/usr/bin/env python3
"""Per-kernel roofline table of the headline step from a rocprofv3 --kernel-trace --stats summary:
    python profiles/roofline_table.py [profiles/r05_bench_kernel_stats.csv] > profiles/r05_roofline_table.md
Shape = BASELINE configs[1] (B=64, T=2000, F=513, N=2000, K=25); algorithmic work per launch as DESIGN.md 4
counts it; peaks from MI355X_MICROARCH.md (157.3 TFLOP/s fp32 MFMA, 8 TB/s HBM).  Durations are the TRACED
averages (the instrumentation adds ~0.9 us to a 4-us kernel; the bench line's `roofline` uses HIP events of an
unprofiled run)."""
import csv
import sys

B, T, F, N, K = 64, 2000, 513, 2000, 25
PEAK_TF, PEAK_GBS = 157.3, 8000.0
path = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r05_bench_kernel_stats.csv'
rows = list(csv.DictReader(open(path)))
chain = 2.0 * B * F * N                      # one B x F x N contraction
BT = B * T
work = [   # (substring, label, flops per launch, bytes per launch)
    ('cell_b_kernel', 'cell_b (x^ = h Dn^T; forward and BPTT)', chain, None),
    ('cell_a_kernel', 'cell_a (g = r Dn, fused update)', chain, None),
    ('bwd_a_kernel', 'bwd_a (dh = dz - dr Dn)', chain, None),
    ('bwd_edge_kernel', 'bwd_edge (last layer of a frame)', None, None),
    ('gemm_tn_kernel<(anonymous namespace)::EpiP1', 'gemm_tn EpiP1 (dD += r^T g, all frames)', 2.0 * BT * F * N, None),
    ('gemm_tn_kernel<(anonymous namespace)::EpiP2', 'gemm_tn EpiP2 (dD -= dr^T h, all frames)', 2.0 * BT * F * N, None),
    ('colreduce4_kernel', 'colreduce4 (column sums of three B T x N streams)', None, 3.0 * BT * N * 4),
    ('gemm_nt_kernel<(anonymous namespace)::EpiHead', 'head GEMMs (A, Bn, mask)', 2.0 * BT * (N // 2) * F, None),
    ('adam_flat_kernel', 'Adam over the flat buffer', None, None),
]
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('| kernel | calls | avg us (traced) | % of GPU time | work per launch | rate | of peak |')
print('|---|---|---|---|---|---|---|')
for sub, label, fl, by in work:
    sel = [r for r in rows if sub in r['Name']]
    if not sel:
        continue
    calls = sum(int(r['Calls']) for r in sel)
    dur = sum(float(r['TotalDurationNs']) for r in sel)
    avg = dur / calls
    if fl:
        rate = fl / avg / 1e3
        cell = '%.1f MFLOP | %.1f TFLOP/s | %.1f %% of fp32 MFMA' % (fl / 1e6, rate, 100 * rate / PEAK_TF)
    elif by:
        rate = by / avg
        cell = '%.0f MB | %.0f GB/s | %.1f %% of HBM' % (by / 1e6, rate, 100 * rate / PEAK_GBS)
    else:
        cell = '-- | -- | --'
    print('| %s | %d | %.2f | %.1f | %s |' % (label, calls, avg / 1e3, 100 * dur / tot, cell))
print()
print('Source: %s (%d kernels, %.1f ms of GPU time).' % (path, len(rows), tot / 1e6))
