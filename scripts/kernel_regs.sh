#!/bin/bash
# register / scratch / LDS use of the kernels in one object file of the build: scripts/kernel_regs.sh mn_kernels [name filter]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$R/sqlite-muninn_amd/csrc/_obj/$1.o" $T/k.fatbin || exit 1
$L/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/k.fatbin --output=$T/k.co || exit 1
$L/llvm-readelf --notes $T/k.co | python3 -c "
import sys, re
flt = sys.argv[1] if len(sys.argv) > 1 else ''
cur = {}
def flush():
    if cur.get('name') and flt in cur['name']:
        print(cur.get('vgpr'), 'vgpr', cur.get('agpr'), 'agpr', cur.get('sgpr'), 'sgpr', cur.get('scratch'), 'scratch', cur.get('lds'), 'lds', cur['name'][:150])
for line in sys.stdin:
    line = line.strip()
    m = re.match(r'-?\s*\.(\w+):\s*(.*)', line)
    if not m: continue
    k, v = m.groups()
    if k == 'agpr_count': 
        if 'agpr' in cur: flush(); cur.clear()
        cur['agpr'] = v
    elif k == 'vgpr_count': cur['vgpr'] = v
    elif k == 'sgpr_count': cur['sgpr'] = v
    elif k == 'private_segment_fixed_size': cur['scratch'] = v
    elif k == 'group_segment_fixed_size': cur['lds'] = v
    elif k == 'name' and 'name' not in cur and v.startswith('_Z'): cur['name'] = v
flush()
" "${2:-}"
rm -rf $T
