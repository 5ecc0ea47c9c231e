#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/r04/profile.sh) -> profiles/<tag>_kernel_stats_noddi_1M.txt, profiles/<tag>_pmc.txt and
profiles/pmc_traffic.json (per HIP-event group of bench.py: HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE, VALU wave-instructions).
usage: python tools/r04/summarise.py r04a"""
import collections, csv, glob, json, os, subprocess, sys
tag = sys.argv[1]
O = 'gpurun_out/prof_%s' % tag
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
g = lambda *a: subprocess.run(['git', '-C', ROOT] + list(a), capture_output=True, text=True).stdout.strip()
dirty = g('status', '--porcelain', '--', 'amico_amd', 'bench.py')
stamp = '# source tree: git %s%s\n' % (g('rev-parse', '--short', 'HEAD'), ' + uncommitted changes in: ' + ', '.join(l.split()[-1] for l in dirty.splitlines()) if dirty else '')
st = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rocpd_summary.py'), O + '/noddi/noddi_results.db'], capture_output=True, text=True).stdout
bench = [l for l in open(O + '/noddi_bench.log') if l.startswith('{')]
with open('profiles/%s_kernel_stats_noddi_1M.txt' % tag, 'w') as f:
    f.write(stamp + '# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs (NODDI, 1 M voxels)\n' + st)
    if bench:
        f.write('\n# bench.py line of the same run\n# ' + bench[-1].strip() + '\n')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(O + '/pmc*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if any(p in k for p in ('k_noddi', 'k_nnls', 'k_lasso', 'k_s2_prep')):
            acc[k.replace('void ', '').replace('amx::', '')][r['Counter_Name']].append(float(r['Counter_Value']))
mean = lambda v: sum(v) / len(v) if v else 0.0
out = [stamp.rstrip(), '# rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs; separate passes '
       '(never combined with other trace domains); mean per launch (1 M voxels)',
       '# FETCH_SIZE / WRITE_SIZE in KiB (gfx950: FETCH_SIZE under-reports wide coalesced reads 2x); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles']
for k in sorted(acc, key=lambda k: -mean(acc[k]['SQ_WAVE_CYCLES'])):
    out.append(k)
    for c, v in sorted(acc[k].items()):
        out.append('    %-30s n=%d mean=%.6g' % (c, len(v), mean(v)))
open('profiles/%s_pmc.txt' % tag, 'w').write('\n'.join(out) + '\n')
# groups = the HIP-event pairs of bench.py (amx_last_kernel_ms)
groups = {'1': ['k_noddi<1,'], '2': ['k_noddi<4,', 'k_noddi<2,'], '3': ['k_noddi<3,'],
          '5': ['k_noddi_gemm<false', 'k_noddi_project<', 'k_nnls_seed<1', 'k_nnls_gcert<1'],
          '6': ['k_noddi_gemm<true', 'k_s2_prep', 'k_noddi_project2', 'k_lasso_seed', 'k_lasso_gcert'], '7': ['k_nnls_seed<3', 'k_nnls_gcert<3'],
          '8': ['k_nnls_seed<1'], '9': ['k_lasso_seed']}
tr, valu, per_kernel = {}, {}, {}
for k, c in acc.items():
    per_kernel[k] = {'bytes': int(2 * mean(c['FETCH_SIZE']) * 1024 + mean(c['WRITE_SIZE']) * 1024), 'valu_insts': mean(c['SQ_INSTS_VALU']),
                     'valu_busy': mean(c['SQ_ACTIVE_INST_VALU']) / mean(c['SQ_WAVE_CYCLES']) if mean(c['SQ_WAVE_CYCLES']) else None,
                     'wait_any': mean(c['SQ_WAIT_ANY']) / mean(c['SQ_WAVE_CYCLES']) if mean(c['SQ_WAVE_CYCLES']) else None,
                     'mfma_f64_mops': mean(c['SQ_INSTS_VALU_MFMA_MOPS_F64'])}
for gk, pats in groups.items():
    ks = [k for k in acc if any(p in k for p in pats)]
    tr[gk] = sum(per_kernel[k]['bytes'] for k in ks)
    valu[gk] = sum(per_kernel[k]['valu_insts'] for k in ks)
try:
    t = json.load(open('profiles/pmc_traffic.json'))
except (OSError, ValueError):
    t = {}
sys.path.insert(0, ROOT)
from amico_amd import _capi
# the identity of the kernels that were measured: bench.py reports these counter figures only next to a library built from the same sources
t['csrc_hash'] = _capi.source_id()
t.update({'_source': 'profiles/%s_pmc.txt (rocprofv3 --pmc, separate passes, NODDI 1 M voxels, mean per launch); git %s' % (tag, g('rev-parse', '--short', 'HEAD')),
          '_correction': 'bytes = 2 * FETCH_SIZE[KiB] * 1024 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE[KiB] * 1024 (calibrated round 4: a 1 GiB fill counts 1 048 580 KiB of WRITE_SIZE, a 1 GiB copy 524 302 KiB of FETCH_SIZE -- profiles/r04a_counter_calibration.txt)',
          '_groups': 'keys = which of amx_last_kernel_ms: 1-3 stage kernels (incl. their re-run kernels), 5-7 GEMM + seed solver + Gram certificate ahead of stage 1 / 2 / 3, 8 / 9 = k_nnls_seed<1> / k_lasso_seed alone (already contained in 5 / 6)',
          'voxels_per_launch': 1000000, 'stage_bytes_per_launch': tr, 'stage_valu_insts_per_launch': valu, 'kernels': per_kernel})
json.dump(t, open('profiles/pmc_traffic.json', 'w'), indent=2)
print('bytes per fit %.3f GB' % (sum(v for k, v in tr.items() if k not in ('8', '9')) / 1e9), {k: round(v / 1e9, 3) for k, v in tr.items()})
print('VALU wave-instructions per voxel', {k: round(v / 1e6) for k, v in valu.items()})
