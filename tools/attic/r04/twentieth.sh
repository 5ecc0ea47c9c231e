#!/bin/bash
# rescue pass gated by call size: tests with it forced on, size scan around the gate
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "hard_mix or diagnosis_switches or fullsize" 2>&1 | tail -3
bash tools/r04/ab.sh "1000000 2000000 4000000 8000000" default 2>&1
AMX_RESCUE_FROM=100000000 bash tools/r04/ab.sh "2000000 8000000" default 2>&1 | sed 's/^default/norescue/'
