#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export AB_STEPS=2
echo "--- tree (wb words)"
timeout -s KILL 90 python tools/r06/fork_ab.py "100000 1000000" "AMX_FORK=0" 2>&1 | grep "^AMX\|fault\|Error" | head -5
echo "--- variant lam, AMX_DEBUG=1"
AMX_DEBUG=1 AMICO_AMD_LIB=$PWD/variants/lam/libamico_amd.so timeout -s KILL 90 python tools/r06/fork_ab.py "100000" "AMX_FORK=0" 2>&1 | grep "^AMX\|fault\|Error\|\[amx\]" | tail -12
