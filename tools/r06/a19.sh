#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r06a
bash tools/r06/profile.sh r06a
bash tools/r06/stress.sh
