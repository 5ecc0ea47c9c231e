#!/bin/bash
# counter calibration: tools/probes/counter_probe.hip under rocprofv3, one counter set per run -> gpurun_out/r06_calib
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_calib
mkdir -p $O
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -o /tmp/counter_probe tools/probes/counter_probe.hip || exit 1
/tmp/counter_probe 2 3 > $O/probe_plain.txt 2>&1; cat $O/probe_plain.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUBBLE_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- /tmp/counter_probe 2 1 > $O/p$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $O/p$i.log)"
done
python tools/r06/calib_summary.py $O | tee $O/summary.txt
