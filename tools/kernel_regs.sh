#!/bin/bash
# tools/kernel_regs.sh <file.hip> <name pattern> [extra hipcc flags]: VGPRs / scratch / LDS of the matching gfx950 kernels
SRC=$1; PAT=$2; shift 2
D=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -I$(dirname $0)/../include -I/opt/rocm/include \
   --cuda-device-only -S "$@" $SRC -o $D/k.s 2>/dev/null
python3 - $D/k.s "$PAT" <<'P'
import re,sys
txt=open(sys.argv[1]).read(); pat=sys.argv[2]
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)(?=\n  - \.agpr_count|\Z)', txt, re.S): pass
# metadata block: entries with .name, .vgpr_count, .private_segment_fixed_size, .group_segment_fixed_size
ents=re.split(r'\n  - \.agpr_count', txt)
import subprocess
for e in ents[1:]:
    n=re.search(r'\.name:\s+(\S+)', e); v=re.search(r'\.vgpr_count:\s+(\d+)', e); s=re.search(r'\.private_segment_fixed_size:\s+(\d+)', e)
    sp=re.search(r'\.vgpr_spill_count:\s+(\d+)', e)
    if not n: continue
    name=subprocess.run(['c++filt', n.group(1)],capture_output=True,text=True).stdout.strip()
    if re.search(pat, name): print(v.group(1), 'vgpr', s.group(1), 'B scratch', (sp.group(1) if sp else '?'), 'spilled |', name[:150])
P
rm -rf $D
