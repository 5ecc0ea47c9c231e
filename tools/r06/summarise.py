#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/r06/profile.sh) -> profiles/<tag>_kernel_stats_noddi_1M.txt, profiles/<tag>_pmc.txt and
profiles/pmc_traffic.json (per HIP-event group of bench.py: HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE, VALU wave-instructions).
usage: python tools/r06/summarise.py r06a
Round 6: the exact DRAM byte counters of gfx950 (TCC_EA0_RDREQ_DRAM_32B / TCC_EA0_WRREQ_WRITE_DRAM_32B, 32-byte units) ride along; they agree with
2 x FETCH_SIZE + WRITE_SIZE on every access pattern of profiles/r06_counter_calibration.txt, and the summary says so per kernel."""
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
                     'bytes_read_dram_32B_counter': int(32 * mean(c['TCC_EA0_RDREQ_DRAM_32B_sum'])) if c.get('TCC_EA0_RDREQ_DRAM_32B_sum') else None,
                     'bytes_written_dram_32B_counter': int(32 * mean(c['TCC_EA0_WRREQ_WRITE_DRAM_32B_sum'])) if c.get('TCC_EA0_WRREQ_WRITE_DRAM_32B_sum') else None,
                     'l2_hit_rate': (mean(c['TCC_HIT_sum']) / (mean(c['TCC_HIT_sum']) + mean(c['TCC_MISS_sum']))) if c.get('TCC_HIT_sum') and (mean(c['TCC_HIT_sum']) + mean(c['TCC_MISS_sum'])) > 0 else None,
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
          '_correction': 'bytes = 2 * FETCH_SIZE[KiB] * 1024 + WRITE_SIZE[KiB] * 1024.  CALIBRATED IN ROUND 6 ON EVERY ACCESS CLASS OF THIS LIBRARY (profiles/r06_counter_calibration.txt, tools/probes/counter_probe.hip, 2 GiB buffers, every byte touched once): '
                         'wide 16-B and 8-B coalesced reads, one 8-B read per 64-B sector, one per 128-B line, one per 512 B, the certificates\' row gathers of a 176 x 512 B table block, global_load_lds dword / dwordx4 -- '
                         'in ALL of them every DRAM read request is 128 bytes (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ), FETCH_SIZE tallies it at 64, and the exact counter TCC_EA0_RDREQ_DRAM_32B x 32 equals 2 x FETCH_SIZE to four digits: '
                         'a lone 8-byte gather moves a 128-byte line (16 x its useful bytes).  WRITE_SIZE is exact: 64-B requests for full sectors, 32-B requests for partial ones (an 8-byte scattered store costs 32 bytes).  '
                         'per kernel: bytes_read_dram_32B_counter / bytes_written_dram_32B_counter = the exact counters of the same build, beside `bytes`',
          '_groups': 'keys = which of amx_last_kernel_ms: 1-3 stage kernels (incl. their re-run kernels), 5-7 GEMM + seed solver + Gram certificate ahead of stage 1 / 2 / 3, 8 / 9 = k_nnls_seed<1> / k_lasso_seed alone (already contained in 5 / 6)',
          'voxels_per_launch': 1000000, 'stage_bytes_per_launch': tr, 'stage_valu_insts_per_launch': valu, 'kernels': per_kernel})
exact = sum((v.get('bytes_read_dram_32B_counter') or 0) + (v.get('bytes_written_dram_32B_counter') or 0) for v in per_kernel.values())
t['bytes_whole_fit_exact_dram_counters'] = exact or None
json.dump(t, open('profiles/pmc_traffic.json', 'w'), indent=2)
print('exact DRAM counters, all kernels of a fit: %.3f GB' % (exact / 1e9))
print('bytes per fit %.3f GB' % (sum(v for k, v in tr.items() if k not in ('8', '9')) / 1e9), {k: round(v / 1e9, 3) for k, v in tr.items()})
print('VALU wave-instructions per voxel', {k: round(v / 1e6) for k, v in valu.items()})
