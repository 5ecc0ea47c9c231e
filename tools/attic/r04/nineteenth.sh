#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export AMX_RESCUE_GLOBAL=1
bash tools/r04/trace1m.sh default 2>&1 | grep "gcert\|k_noddi<"
bash tools/r04/ab.sh "1000000 4000000" default 2>&1
