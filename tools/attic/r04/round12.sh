#!/bin/bash
# stage-1 seed scan on normalised atoms (default) against the tree before (variants/prev)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/ab.sh "50000 200000 1000000 4000000" prev default prev default 2>&1
for v in prev default; do
unset AMICO_AMD_LIB; [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); o = d['other_configs']; sc = d['seed_chain']
print('$v headline %.1f (left %s)' % (d['value'] / 1e6, [sc[q] for q in ('leftover_stage1', 'leftover_lasso', 'leftover_stage3')]), ' '.join('%s %.1f (left %s)' % (k, o[k]['value'] / 1e6, [o[k]['seed_chain'][q] for q in ('leftover_stage1', 'leftover_lasso', 'leftover_stage3')]) for k in ('noddi_hard_mix', 'noddi_105vol', 'noddi_150vol', 'noddi_exvivo')))"
done
unset AMICO_AMD_LIB
timeout 900 python -m pytest tests -m gpu -x -q -k "kkt or parity or multi" 2>&1 | grep "passed\|failed"
