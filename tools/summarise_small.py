#!/usr/bin/env python3
"""Turn gpurun_out/small_<tag>/ (tools/profile_small.sh) into committed summaries under profiles/:
<tag>_kernel_stats_{freewater_2M,sandi_1M,lut,prep}.txt and <tag>_pmc_small.txt.
usage: python tools/summarise_small.py r02a"""
import collections, csv, glob, os, subprocess, sys
tag = sys.argv[1]
O = 'gpurun_out/small_%s' % tag
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def head_stamp():
    """git HEAD (+ 'dirty' when amico_amd/ or bench.py differ from it) of the tree the profile was taken from: the collection
    and this summary run back to back on the same working tree, so a stale summary is visible"""
    g = lambda *a: subprocess.run(['git', '-C', ROOT] + list(a), capture_output=True, text=True).stdout.strip()
    dirty = g('status', '--porcelain', '--', 'amico_amd', 'bench.py')
    return '# source tree: git %s%s\n' % (g('rev-parse', '--short', 'HEAD'), ' + uncommitted changes in: ' + ', '.join(l.split()[-1] for l in dirty.splitlines()) if dirty else '')


def stats(db):
    return subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rocpd_summary.py'), db], capture_output=True, text=True).stdout


def bench_line(log):
    for ln in open(log):
        if ln.startswith('{'):
            return ln.strip()
    return ''


for key, name, cmd in (('fw', 'freewater_2M', '--model freewater --voxels 2000000'), ('sandi', 'sandi_1M', '--model sandi --voxels 1000000'), ('czb', 'czb_500k', '--model czb --voxels 500000'),
                       ('lut', 'lut', '--model lut'), ('prep', 'prep', '--model prep')):
    db = '%s/%s/%s_results.db' % (O, key, key)
    if not os.path.exists(db):
        continue
    with open('profiles/%s_kernel_stats_%s.txt' % (tag, name), 'w') as f:
        f.write(head_stamp())
        f.write('# rocprofv3 --kernel-trace --stats -- python bench.py %s --steps 5 --warmup 1\n' % cmd)
        f.write(stats(db))
        f.write('\n# bench.py line of the same run\n# ' + bench_line('%s/%s_bench.log' % (O, key)) + '\n')
out = [head_stamp().rstrip(), '# rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python bench.py --model {freewater --voxels 2000000 | sandi --voxels 1000000 | czb --voxels 500000 | lut} '
       '--steps 2 --warmup 1; separate passes (never combined with other trace domains); mean per launch',
       '# FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE under-reports wide coalesced reads 2x on gfx950); SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles']
for m, pats in (('freewater', ['k_freewater', 'k_fw_project']), ('sandi', ['k_sandi']), ('czb', ['k_czb']), ('lut', ['k_lut_resample']), ('prep', ['k_prep_', 'k_mean_b0', 'k_scatter', 'k_dti_dirs'])):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob('%s/pmc_%s_*/*/*_counter_collection.csv' % (O, m)):
        for r in csv.DictReader(open(fn)):
            if any(p in r['Kernel_Name'] for p in pats):
                acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    out.append('## --model %s' % m)
    for k in acc:
        out.append(k)
        for c, v in sorted(acc[k].items()):
            out.append('    %-30s n=%d mean=%.6g' % (c, len(v), sum(v) / len(v)))
open('profiles/%s_pmc_small.txt' % tag, 'w').write('\n'.join(out) + '\n')
# per-voxel HBM traffic of the lane kernels -> profiles/pmc_traffic.json (read by bench.py for roofline.traffic)
import json
small = {}
for m, pats, n in (('freewater', ('k_fw_project', 'k_freewater'), 2000000), ('sandi', ('k_sandi',), 1000000), ('czb', ('k_czb',), 500000)):
    # a fit may be several kernels (FreeWater: projection + solver): per-launch means per kernel, summed over the kernels
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    names = set()                                       # the kernels that RAN (the summary listed its search prefixes)
    for fn in glob.glob('%s/pmc_%s_*/*/*_counter_collection.csv' % (O, m)):
        for r in csv.DictReader(open(fn)):
            for pat in pats:
                if pat in r['Kernel_Name']:
                    per[pat][r['Counter_Name']].append(float(r['Counter_Value']))
                    names.add(r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').replace('amx::', '').split('(')[0])
    mean = lambda v: sum(v) / len(v)
    tot = collections.defaultdict(float)
    for pat in per:
        for c, v in per[pat].items():
            tot[c] += mean(v)
    if 'FETCH_SIZE' in tot and 'WRITE_SIZE' in tot:
        small[m] = {'voxels_per_launch': n, 'bytes_per_voxel_measured': (2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024 / n,
                    'valu_insts_per_voxel': tot['SQ_INSTS_VALU'] / n, 'kernels': sorted(names),
                    'bytes_per_voxel_exact_dram_counters': (32 * (tot.get('TCC_EA0_RDREQ_DRAM_32B_sum', 0) + tot.get('TCC_EA0_WRREQ_WRITE_DRAM_32B_sum', 0)) / n) if 'TCC_EA0_RDREQ_DRAM_32B_sum' in tot else None,
                    'mfma_busy_quad_cycles': tot.get('SQ_VALU_MFMA_BUSY_CYCLES'), 'mfma_f64_mops': tot.get('SQ_INSTS_VALU_MFMA_MOPS_F64'),
                    '_source': 'profiles/%s_pmc_small.txt; bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB), see _correction' % tag}
try:
    t = json.load(open('profiles/pmc_traffic.json'))
except (OSError, ValueError):
    t = {}
t['small_models'] = small
json.dump(t, open('profiles/pmc_traffic.json', 'w'), indent=2)
print(json.dumps(small, indent=1))
print('\n'.join(out))
