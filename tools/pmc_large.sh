#!/bin/bash
# GPU box, after tools/pmc_run.sh: the FETCH_SIZE / WRITE_SIZE passes once more on a 1 GiB stream (beyond the 256 MiB Infinity Cache) and the step's
# traffic there, merged into gpurun_out/pmc_traffic.json as "large".
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcL_$i -o q$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extras --bytes 1073741824 > $R/gpurun_out/pmcL_$i.log 2>&1
  echo "large pass $i rc=$?"
done
python $R/tools/pmc_traffic.py $R/gpurun_out/pmcL_1/q1_results.db $R/gpurun_out/pmcL_2/q2_results.db $R/gpurun_out/pmc_traffic_1GiB.json 1073741824 enwik "$(cat $R/tools/var/commit.txt 2>/dev/null || echo unknown)" > /dev/null
python - <<PY
import json
R="$R"
big=json.load(open(R+"/gpurun_out/pmc_traffic_1GiB.json"))
line=json.loads([l for l in open(R+"/gpurun_out/pmcL_1.log") if l.startswith("{")][-1])
N=line["config"]["bytes_per_gpu"]; C=int(line["config"]["ratio"]*N)
step=sum(k["traffic"] for k in big["kernels"].values())
small=json.load(open(R+"/gpurun_out/pmc_traffic.json"))
small["large"]={"workload_bytes":N,"compressed_bytes":C,"step_traffic":step,"step_algorithmic_bytes":2*(N+C),"step_traffic_over_algorithmic":round(step/(2*(N+C)),3),
                "kernels":{k:v["traffic"] for k,v in big["kernels"].items()}}
json.dump(small,open(R+"/gpurun_out/pmc_traffic.json","w"),indent=1)
print("large:", small["large"]["step_traffic_over_algorithmic"], "x algorithmic;", "small:", round(sum(k["traffic"] for k in small["kernels"].values())/ (2*(small["workload_bytes"]*1.4753)),3))
PY
