cd $GRAFT_REPO_ROOT
for v in "$@"; do
MINLZ_HIP_LIB=$GRAFT_REPO_ROOT/tools/var/$v.so python bench.py --steps 10 --warmup 2 --no-cpu --no-extras --level 2 --workload json 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['ratio'], d['config']['encode_MBps'], d['config']['kernel_ms']['enc_far_build'], d['config']['kernel_ms']['enc_tiles'])"
done
