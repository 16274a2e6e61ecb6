#!/bin/bash
# nt loads in the VALU decimator / interpolator / short matrix-core cascades: A/B on one box
cd $GRAFT_REPO_ROOT
cp sdrdaemon_amd/libsdrhip.so /tmp/lib_nt.so
rm -f sdrdaemon_amd/csrc/build/*.hip.o; make -s -C sdrdaemon_amd/csrc EXTRA="-DSDRHIP_NT=0" > /dev/null 2>&1; cp sdrdaemon_amd/libsdrhip.so /tmp/lib_plain.so
for r in 1 2; do for v in nt plain; do cp /tmp/lib_$v.so sdrdaemon_amd/libsdrhip.so; echo "== $v"; python tools/bench_kernels.py decim interp 2>&1 | grep -v "^$"; done; done > gpurun_out/exp10.txt 2>&1
cp /tmp/lib_nt.so sdrdaemon_amd/libsdrhip.so
python - <<'PY'
import re,collections
d=collections.defaultdict(lambda: collections.defaultdict(list)); v=None
for l in open('gpurun_out/exp10.txt'):
    if l.startswith('=='): v=l.split()[1]; continue
    m=re.match(r'(\S+)\s+([\d.]+) ms',l)
    if m: d[m.group(1)][v].append(float(m.group(2)))
for k in d: print('%-20s nt %s  plain %s'%(k, ' '.join('%.3f'%x for x in d[k]['nt']), ' '.join('%.3f'%x for x in d[k]['plain'])))
PY
