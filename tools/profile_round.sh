#!/bin/bash
# Collects everything profiles/ holds for one round tag (run on the GPU box): kernel-trace stats, PMC passes,
# HBM-side traffic per kernel, and the bench line.  usage: profile_round.sh <tag>
R=$GRAFT_REPO_ROOT; tag=$1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --no-extras > $R/gpurun_out/prof_$tag.log 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/prof_$tag/t_results.db > $R/gpurun_out/${tag}_kernel_stats.txt
bash $R/tools/pmc_run.sh
python $R/tools/rocpd_pmc.py $R/gpurun_out/pmc_1/p1_results.db $R/gpurun_out/pmc_4/p4_results.db > $R/gpurun_out/${tag}_pmc_counters.txt 2>&1
cd $R && python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench.json
# foreign (reference-algorithm) streams through the general-block path: kernel stats of the decode alone, 8 MiB and 2 MiB blocks
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_f8 -o t -- python $R/tools/foreign_time.py enwik 100 > $R/gpurun_out/${tag}_foreign_8MiB.log 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/prof_${tag}_f8/t_results.db > $R/gpurun_out/${tag}_foreign_kernel_stats.txt
BLOCK=2097152 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_f2 -o t -- python $R/tools/foreign_time.py enwik 100 > $R/gpurun_out/${tag}_foreign_2MiB.log 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/prof_${tag}_f2/t_results.db > $R/gpurun_out/${tag}_foreign_2MiB_kernel_stats.txt
tail -n 2 $R/gpurun_out/${tag}_foreign_8MiB.log $R/gpurun_out/${tag}_foreign_2MiB.log
