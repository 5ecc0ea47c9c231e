#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/trace1m.sh default notouch s3np 2>&1 | grep "==\|seed<\|seed("
bash tools/r04/ab.sh "50000 200000 1000000 4000000" default notouch s3np 2>&1
