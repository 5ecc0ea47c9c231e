#!/bin/bash
# does the rescue pass pay at 1 M voxels for the data sets with more left-overs (ex vivo, longer protocols, hard mix)?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 2000000 0; do
AMX_RESCUE_FROM=$r timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); o = d['other_configs']
print('AMX_RESCUE_FROM=$r headline %.1f' % (d['value'] / 1e6), ' '.join('%s %.1f (left %s)' % (k, o[k]['value'] / 1e6, [o[k]['seed_chain'][q] for q in ('leftover_stage1', 'leftover_lasso', 'leftover_stage3')]) for k in ('noddi_hard_mix', 'noddi_105vol', 'noddi_150vol', 'noddi_exvivo')))"
done
