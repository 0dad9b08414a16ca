#!/bin/bash
# compile the channel-MFMA unit with the resource report; prints name, VGPRs, AGPRs, scratch per kernel
cd "$(dirname "$0")/../clifford-group-equivariant-simplicial-message-passing-networks_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value -fno-slp-vectorize $EXTRA -Rpass-analysis=kernel-resource-usage -c ${1:-k_cm_n3.hip} -o ${2:-_build/k_cm_n3.o} 2>&1 | python3 ../../tools/res_report.py
