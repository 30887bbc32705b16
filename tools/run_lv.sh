cd $GRAFT_REPO_ROOT
for v in nobk noext; do KIND=text LEVEL=1 TAG=$v MINLZ_HIP_LIB=build_var/$v.so timeout 120 python tools/level_time.py 2>&1 | tail -1 | cut -c1-200; done
