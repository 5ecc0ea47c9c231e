#!/bin/bash
# the rescue pass decided on the device from the first pass's left-over count (AMX_RESCUE_PCT, default 7; stage 3: 4): headline and the legs with more left-overs
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for p in 7 -1 7 -1; do
AMX_RESCUE_PCT=$p timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); o = d['other_configs']
print('AMX_RESCUE_PCT=$p headline %.1f' % (d['value'] / 1e6), ' '.join('%s %.1f (left %s)' % (k, o[k]['value'] / 1e6, [o[k]['seed_chain'][q] for q in ('leftover_stage1', 'leftover_lasso', 'leftover_stage3')]) for k in ('noddi_hard_mix', 'noddi_105vol', 'noddi_150vol', 'noddi_exvivo')))"
done
