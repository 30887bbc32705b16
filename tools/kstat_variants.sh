#!/bin/bash
# GPU box: rocprofv3 kernel stats of an encode-only loop for library variants (timing experiments). usage: tools/kstat_variants.sh <grep-pattern> name...
R=$GRAFT_REPO_ROOT; pat=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/kv_$v
  MINLZ_HIP_LIB=$R/tools/var/$v.so LEVEL=${LEVEL:-2} WL=${WL:-json} timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kv_$v -o t -- python $R/tools/enc_time.py > /tmp/kv_$v.log 2>&1
  echo "== $v"; python $R/tools/rocpd_summary.py /tmp/kv_$v/t_results.db | grep -E "$pat"
done
