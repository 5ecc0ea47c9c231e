#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do bash tools/r04/ab.sh "1000000 2000000" default nosteal; done
