cd $GRAFT_REPO_ROOT
KIND=text LEVEL=1 TAG=base timeout 120 python tools/level_time.py 2>&1 | tail -1 | cut -c1-330
for v in t14 t14s; do KIND=text LEVEL=1 TAG=$v MINLZ_HIP_LIB=build_var/$v.so timeout 120 python tools/level_time.py 2>&1 | tail -1 | cut -c1-330; done
