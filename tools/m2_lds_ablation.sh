#!/bin/bash
# GPU box: LDS counters + kernel time of match_tiles_kernel for LDS-access ablation builds (tools/exp_build.sh <name> -DMLZ_ABL3=x).
# usage: tools/m2_lds_ablation.sh name1 name2 ...   -> gpurun_out/m2_lds_ablation.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/m2_lds_ablation.txt
: > $out
for v in "$@"; do
  echo "== $v" >> $out
  MINLZ_HIP_LIB=$R/tools/var/$v.so python $R/tools/enc_time.py 2>&1 | grep -v amdgpu.ids >> $out
  rm -rf /tmp/ab_$v
  MINLZ_HIP_LIB=$R/tools/var/$v.so ENC_TIME_REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/ab_$v -o p -- python $R/tools/enc_time.py > /tmp/ab_$v.log 2>&1
  python $R/tools/rocpd_pmc.py /tmp/ab_$v/p_results.db 2>&1 | grep -E "match_tiles|serialize" >> $out
done
cat $out
