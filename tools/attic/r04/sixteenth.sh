#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 200000 1000000 4000000; do timeout 600 python tools/r04/split_ab.py $n 2>&1 | grep "part"; done
