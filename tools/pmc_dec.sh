#!/bin/bash
# PMC counters of the decode kernels on the bench stream (run on the GPU box): tools/pmc_dec.sh [tag]
R=$GRAFT_REPO_ROOT; tag=${1:-dec}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $R/gpurun_out/pmcd_$tag -o p -- python $R/tools/dec_time.py > $R/gpurun_out/pmcd_$tag.log 2>&1
python $R/tools/rocpd_pmc.py $R/gpurun_out/pmcd_$tag/p_results.db | grep "dec_" | cut -c1-330
