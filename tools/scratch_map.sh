#!/bin/bash
# where does the compiler spill?  scratch_* instructions of k_phase by source line
# (tools/scratch_map.sh [extra hipcc flags]); spills inside the S5 main loop cost HBM
# traffic and sit on the critical path: keep that range clean
cd "$(dirname "$0")/../ahsoka_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -gline-tables-only --cuda-device-only "$@" -S k_phase.hip -o /tmp/k_phase_lines.s 2>/dev/null
python3 - <<'PY'
import re
lines=open('/tmp/k_phase_lines.s').read().split('\n')
src=open('k_phase.hip').read().split('\n')
lo=next(i for i,l in enumerate(src) if 'S5 main loop' in l)+1
hi=next(i for i,l in enumerate(src) if 'clusters: ordered by smallest member' in l)+1
f0=next(i for i,l in enumerate(src) if 'void rescan_rows2' in l)+1
f1=next(i for i,l in enumerate(src) if 'struct GreaterCnt' in l)+1
cur=None; hits={}
for l in lines:
    m=re.match(r'\s*\.loc\s+0\s+(\d+)',l)
    if m: cur=int(m.group(1)); continue
    if 'scratch_' in l:
        k=('store' if 'store' in l else 'load')
        hits.setdefault(cur,{'store':0,'load':0})[k]+=1
print(f"S5 main loop = source lines {lo}..{hi}; rescan helpers {f0}..{f1}")
for k in sorted(x for x in hits if x):
    tag = 'LOOP' if lo<=k<=hi else ('RESCAN' if f0<=k<=f1 else '')
    print(k, hits[k], tag)
print('line 0 (no location):', hits.get(0))
PY
