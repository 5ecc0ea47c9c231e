cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/r05/proto_fit.py hcp 1000000 3 2>/dev/null | cut -c1-900
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_hcp -o t -- python tools/r05/proto_fit.py hcp 1000000 3 > /dev/null 2>&1
python tools/rocpd_summary.py gpurun_out/kt_hcp/t_results.db 2>/dev/null | head -22 | cut -c1-150
rm -rf gpurun_out/kt_hcp
