"""Per-kernel table from the two PMC passes of tools/x3_pmc.sh: duration, clock, matrix-pipe busy, LDS conflicts."""
import csv, collections, sys, os
out, mode = sys.argv[1], sys.argv[2]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(out, 'pmc_sq_%s/p_kernel_trace.csv' % mode))):
    if 'gemm' in r['Kernel_Name']:
        dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ('sq', 'lds'):
    f = os.path.join(out, 'pmc_%s_%s/p_counter_collection.csv' % (p, mode))
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name']:
            cnt[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in dur:
    c = {n: sum(v) / len(v) for n, v in cnt[k].items()}
    d = sum(dur[k]) / len(dur[k])
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8
    name = k.replace('(anonymous namespace)::', '')[:90]
    print('%s %s: %.1f us, %.2f GHz, MFMA busy %.1f %% of SIMD cycles, VALU/MFMA insts %.2f, LDS active %.1f %% (conflict share %.1f %%), wave cycles: issue-stall %.0f %% wait %.0f %% active %.0f %%' % (
        mode, name, d, cyc / d / 1e3 if d else 0, 100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc) if cyc else 0,
        c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_INSTS_MFMA', 1), 1), 100 * c.get('SQ_LDS_IDX_ACTIVE', 0) / 256 / cyc if cyc else 0,
        100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1),
        100 * c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1), 100 * c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1),
        100 * c.get('SQ_ACTIVE_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1)))
