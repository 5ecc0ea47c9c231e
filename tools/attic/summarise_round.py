#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/:
<tag>_kernel_stats_noddi_1M.txt, <tag>_kernel_stats_dti_prep.txt, <tag>_pmc.txt and pmc_traffic.json.
usage: python tools/summarise_round.py r01g [gpurun_out/mix]"""
import collections, csv, glob, json, os, subprocess, sys
tag = sys.argv[1]
O = 'gpurun_out/prof_%s' % tag
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def head_stamp():
    """git HEAD (+ 'dirty' when amico_amd/ or bench.py differ from it) of the tree the profile was taken from: the collection
    and this summary run back to back on the same working tree, so a stale summary is visible"""
    g = lambda *a: subprocess.run(['git', '-C', ROOT] + list(a), capture_output=True, text=True).stdout.strip()
    dirty = g('status', '--porcelain', '--', 'amico_amd', 'bench.py')
    return '# source tree: git %s%s\n' % (g('rev-parse', '--short', 'HEAD'), ' + uncommitted changes in: ' + ', '.join(l.split()[-1] for l in dirty.splitlines()) if dirty else '')


def stats(db):
    return subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rocpd_summary.py'), db], capture_output=True, text=True).stdout


with open('profiles/%s_kernel_stats_noddi_1M.txt' % tag, 'w') as f:
    f.write(head_stamp())
    f.write('# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline (NODDI 1 M voxels)\n')
    f.write(stats(O + '/noddi/noddi_results.db'))
with open('profiles/%s_kernel_stats_dti_prep.txt' % tag, 'w') as f:
    f.write(head_stamp())
    f.write('# rocprofv3 --kernel-trace --stats -- python bench.py --model dti --steps 5 --warmup 1\n')
    f.write(stats(O + '/dti/dti_results.db'))
    f.write('\n# rocprofv3 --kernel-trace --stats -- python bench.py --model prep --steps 5 --warmup 1 (F order first, then C order)\n')
    f.write(stats(O + '/prep/prep_results.db'))
out = [head_stamp().rstrip(), '# rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python bench.py [--model ...] --steps 3 --warmup 1; '
       'separate passes; mean per launch']


def summarize(files, pats):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in files:
        for r in csv.DictReader(open(fn)):
            if any(p in r['Kernel_Name'] for p in pats):
                acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in acc:
        out.append(k)
        for c, v in sorted(acc[k].items()):
            out.append('    %-24s n=%d mean=%.6g' % (c, len(v), sum(v) / len(v)))
    return acc


files = sum([glob.glob('%s/pmc%d/*/*_counter_collection.csv' % (O, i)) for i in (1, 2, 3, 4, 11, 12)], [])
if len(sys.argv) > 2:
    files += glob.glob(sys.argv[2] + '/*/*_counter_collection.csv')
a = summarize(files, ['k_noddi<1, 2, 3, 8, 16, false', 'k_noddi<4, 2, 3, 20, 16, false', 'k_noddi<3, 2, 3, 8, 16, false'])
out.append('# --model dti (FETCH_SIZE, WRITE_SIZE passes)')
summarize(sum([glob.glob('%s/pmc%d/*/*_counter_collection.csv' % (O, i)) for i in (5, 6)], []), ['k_dti_dirs'])
out.append('# --model prep (FETCH_SIZE, WRITE_SIZE passes; Fortran-order and C-order launches averaged together)')
summarize(sum([glob.glob('%s/pmc%d/*/*_counter_collection.csv' % (O, i)) for i in (7, 8)], []), ['k_prep_gather'])
open('profiles/%s_pmc.txt' % tag, 'w').write('\n'.join(out) + '\n')
mean = lambda v: sum(v) / len(v)
tr, valu, act = {}, {}, {}
for k, v in a.items():
    st = '1' if 'k_noddi<1' in k else ('2' if 'k_noddi<4' in k else '3')
    tr[st] = int(2 * mean(v['FETCH_SIZE']) * 1024 + mean(v['WRITE_SIZE']) * 1024)
    valu[st] = mean(v['SQ_INSTS_VALU'])
    act[st] = 4.0 * mean(v['SQ_ACTIVE_INST_VALU']) / 1024.0
    print(st, 'VALU/voxel %.0f' % (valu[st] / 1e6), 'VALU-active cycles per SIMD %.3g' % act[st], 'traffic %.3g GB' % (tr[st] / 1e9),
          {c: round(mean(v[c]) / 1e6) for c in v if c.startswith('SQ_INSTS_VALU_')})
json.dump({'_source': 'profiles/%s_pmc.txt (rocprofv3 --pmc, separate passes, NODDI 1 M voxels, mean per launch)' % tag,
           '_correction': 'bytes = 2 * FETCH_SIZE[KiB] * 1024 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE[KiB] * 1024 (uncalibrated)',
           'voxels_per_launch': 1000000, 'stage_bytes_per_launch': tr, 'stage_valu_insts_per_launch': valu,
           'stage_valu_active_cycles_per_simd': act}, open('profiles/pmc_traffic.json', 'w'), indent=2)
