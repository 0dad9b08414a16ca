#!/bin/bash
# compile the (row, channel)-per-lane unit with the resource report; prints name, VGPRs, scratch per kernel
cd "$(dirname "$0")/../clifford-group-equivariant-simplicial-message-passing-networks_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value -fno-slp-vectorize $EXTRA -Rpass-analysis=kernel-resource-usage -c ${1:-k_cl_n3.hip} -o ${2:-_build/k_cl_n3.o} 2>&1 | python3 -c "
import sys,re
name=None
for l in sys.stdin:
    if 'error' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1)
    m=re.search(r' VGPRs: (\d+)',l)
    if m: v=m.group(1)
    m=re.search(r'AGPRs: (\d+)',l)
    if m: a=m.group(1)
    m=re.search(r'ScratchSize \[bytes/lane\]: (\d+)',l)
    if m: s=m.group(1)
    m=re.search(r'Occupancy \[waves/SIMD\]: (\d+)',l)
    if m and name and 'reduce' not in name:
        print(re.sub(r'_ZN5csmpn\d+|INS_3AlgILi3ELj0EEE|EvNS_8DevCemlpENS_5RowIOE','',name), 'VGPR',v,'AGPR',a,'scratch',s,'occ',m.group(1))
"
