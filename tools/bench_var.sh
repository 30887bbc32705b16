#!/bin/bash
# usage: tools/bench_var.sh [bench args] -- name1 name2 ...   (GPU box): bench.py with tools/var/<name>.so instead of the product library
cd $GRAFT_REPO_ROOT
args=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done
shift
for v in "$@"; do
  MINLZ_HIP_LIB=$GRAFT_REPO_ROOT/tools/var/$v.so python bench.py --steps 10 --warmup 2 --no-cpu --no-extras "${args[@]}" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['config']['ratio'], d['config']['kernel_ms'])"
done
