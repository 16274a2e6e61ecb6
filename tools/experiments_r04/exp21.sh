#!/bin/bash
# batch 21: K5w occupancy under 4 waves per workgroup (LDS padded so that 4 / 3 / 2 workgroups fit a CU), and 2 waves per workgroup
cd /root/repo
for v in wpw4 wpw4o4 wpw4o3 wpw4o2 wpw2 wpw4 wpw4o4 wpw4o3 wpw4o2 wpw2; do
  echo "== $v"
  SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_$v.so PATHS=wave:0 LS=4,3,5 python tools/bench_interp_paths.py 2>&1 | grep interpolate
done
