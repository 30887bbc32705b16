#!/bin/bash
# Vector-memory-path counters (TA / TCP / TLB) for the bench kernels; separate rocprofv3 passes.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcm_$i -o p$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu $BENCH_ARGS > $R/gpurun_out/pmcm_$i.log 2>&1
  echo "pass $i rc=$?"
done
python $R/tools/rocpd_pmc.py $R/gpurun_out/pmcm_*/p*_results.db > $R/gpurun_out/pmcm_summary.txt 2>&1
