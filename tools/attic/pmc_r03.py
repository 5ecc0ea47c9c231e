#!/usr/bin/env python3
"""Per-kernel PMC table of gpurun_out/prof_<tag>/pmc*/ (tools/profile_r03.sh): mean per launch, derived ratios.
usage: python tools/pmc_r03.py r03b"""
import collections, csv, glob, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('gpurun_out/prof_%s/pmc*/*/*_counter_collection.csv' % tag):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if 'k_noddi' in k or 'k_nnls' in k or 'k_lasso' in k:
            acc[k.replace('void ', '').replace('amx::', '')[:44]][r['Counter_Name']].append(float(r['Counter_Value']))
mean = lambda v: sum(v) / len(v) if v else float('nan')
print('%-44s %9s %9s %7s %7s %7s %7s %9s %9s %8s' % ('kernel (mean per launch, 1 M voxels)', 'VALU/vox', 'SALU/vox', 'VALUbsy', 'waitany', 'MFMAbsy', 'LDS/vox', 'fetch MB', 'write MB', 'bankcnf'))
for k, c in sorted(acc.items(), key=lambda kv: -mean(kv[1]['SQ_WAVE_CYCLES'])):
    n = 1e6
    wc = mean(c['SQ_WAVE_CYCLES'])
    print('%-44s %9.0f %9.0f %7.2f %7.2f %7.2f %7.0f %9.1f %9.1f %8.3f' % (
        k, mean(c['SQ_INSTS_VALU']) / n, mean(c['SQ_INSTS_SALU']) / n, mean(c['SQ_ACTIVE_INST_VALU']) / wc if wc else 0,
        mean(c['SQ_WAIT_ANY']) / wc if wc else 0, mean(c['SQ_VALU_MFMA_BUSY_CYCLES']) / mean(c['SQ_BUSY_CU_CYCLES']) if c['SQ_BUSY_CU_CYCLES'] and mean(c['SQ_BUSY_CU_CYCLES']) else 0,
        mean(c['SQ_INSTS_LDS']) / n, 2 * mean(c['FETCH_SIZE']) / 1024, mean(c['WRITE_SIZE']) / 1024,
        mean(c['SQ_LDS_BANK_CONFLICT']) / mean(c['SQ_ACTIVE_INST_LDS']) if c['SQ_ACTIVE_INST_LDS'] and mean(c['SQ_ACTIVE_INST_LDS']) else 0))
    print('%-44s waves %.0f  wave-cycles/1e9 %.2f  busy_cycles/1e6 %.1f  fp64 FMA/MUL/ADD per voxel %.0f/%.0f/%.0f  cvt %.0f int32 %.0f  mfma_f64_mops %.3g' % (
        '', mean(c['SQ_WAVES']), wc / 1e9, mean(c['SQ_BUSY_CYCLES']) / 1e6, mean(c['SQ_INSTS_VALU_FMA_F64']) / n, mean(c['SQ_INSTS_VALU_MUL_F64']) / n,
        mean(c['SQ_INSTS_VALU_ADD_F64']) / n, mean(c['SQ_INSTS_VALU_CVT']) / n, mean(c['SQ_INSTS_VALU_INT32']) / n, mean(c['SQ_INSTS_VALU_MFMA_MOPS_F64'])))
