#!/bin/bash
# VGPR / SGPR / LDS / scratch of every kernel in the built library (code-object notes): tools/kernel_resources.sh [lib.so]
set -e
LIB=${1:-clive2_amd/libclive2_amd.so}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$LIB" --output="$T/k.co" 2>/dev/null || \
  python3 - "$LIB" "$T/k.co" <<'P'
import sys
b=open(sys.argv[1],'rb').read()
i=b.find(b'\x7fELF',1)
# the embedded device ELF: find ELF headers with e_machine = 224 (AMDGPU)
import struct
while i!=-1:
    if struct.unpack_from('<H',b,i+18)[0]==224:
        open(sys.argv[2],'wb').write(b[i:]); break
    i=b.find(b'\x7fELF',i+1)
P
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$T/k.co" | python3 -c '
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+)", txt, re.S):
    lds,name,scr,sg,vg=m.groups()
    print(f"{int(vg):4d} VGPR {int(sg):4d} SGPR {int(lds):6d} B LDS {int(scr):5d} B scratch  {name[:110]}")
' | sort -k9
rm -rf "$T"
