#!/bin/bash
# usage: tools/bench_variants.sh v1 v2 ...  (run on GPU box): swaps in alternative builds of the library
cd $GRAFT_REPO_ROOT
cp minlz_amd/libminlz_hip.so /tmp/orig.so
for v in "$@"; do
  cp tools/libminlz_hip_$v.so minlz_amd/libminlz_hip.so
  python -m pytest tests/test_gpu_encode.py tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -1
  python bench.py --steps 10 --warmup 2 --no-cpu --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['config']['ratio'], d['config']['kernel_ms'], d['config']['encode_MBps'], d['config']['decode_MBps'])"
done
cp /tmp/orig.so minlz_amd/libminlz_hip.so
