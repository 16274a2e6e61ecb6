#!/bin/bash
# batch 26b: K1m timeline (300th launch), all pieces as workgroups (old) against one piece workgroup per free CU (new)
cd /root/repo
for v in mfstamps_old mfstamps mfstamps_old mfstamps; do
  echo "== $v"
  SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_$v.so python tools/experiments_r04/k1m_stamps.py 4 2>&1 | grep "launch span\|end deciles\|duration mean"
done
