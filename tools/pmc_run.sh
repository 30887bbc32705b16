#!/bin/bash
# Collects PMC counters for the bench kernels in separate rocprofv3 passes (run on the GPU box).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$i -o p$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extras $BENCH_ARGS > $R/gpurun_out/pmc_$i.log 2>&1
  echo "pass $i rc=$?"
done
python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_2/p2_results.db $R/gpurun_out/pmc_3/p3_results.db $R/gpurun_out/pmc_traffic.json 100000000 enwik "$(cat $R/tools/var/commit.txt 2>/dev/null || echo unknown)"
