cd $GRAFT_REPO_ROOT
for v in pf128 pf256; do KIND=text LEVEL=1 TAG=$v MINLZ_HIP_LIB=build_var/$v.so timeout 120 python tools/level_time.py 2>&1 | tail -1 | cut -c1-160; done
