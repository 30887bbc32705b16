#!/bin/bash
# usage: pmc_set.sh <tag> "<counters>" [bench args...]   — one rocprofv3 PMC pass over bench.py, summary to gpurun_out/pmc_<tag>.txt
R=$GRAFT_REPO_ROOT
tag=$1; set_=$2; shift 2
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc $set_ -d $R/gpurun_out/pmcs_$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extras "$@" > $R/gpurun_out/pmcs_$tag.log 2>&1
echo "$tag rc=$?"
python $R/tools/rocpd_pmc.py $R/gpurun_out/pmcs_$tag/p_results.db > $R/gpurun_out/pmc_$tag.txt 2>&1
rm -rf $R/gpurun_out/pmcs_$tag
