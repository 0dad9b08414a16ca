#!/bin/bash
# Iteration aid: rebuild only the parity-lane units (EXTRA="-D..." for experiments) and relink with the
# other, already built objects. Use `make` before committing.
set -e
cd "$(dirname "$0")/../clifford-group-equivariant-simplicial-message-passing-networks_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value"
B=_build
hipcc $FLAGS $EXTRA -Rpass-analysis=kernel-resource-usage -c k_pl_n5m.hip -o $B/k_pl_n5m.o 2> /tmp/k_pl_n5m.log &
if [ -z "$ONLY_M" ]; then hipcc $FLAGS $EXTRA -c k_pl_n5.hip -o $B/k_pl_n5.o & fi
if [ "$1" = "capi" ]; then hipcc $FLAGS -c capi.hip -o $B/capi.o & fi
wait
grep -E "Name:|VGPRs:|ScratchSize" /tmp/k_pl_n5m.log | sed 's/.*remark: [^ ]* //; s/\[-Rpass.*//' | paste - - - | sed 's/_ZN5csmpn15cemlp_pl_kernelINS_3AlgILi5ELj16EEE//'
hipcc -shared -fPIC --offload-arch=gfx950 $B/capi.o $B/csr.o $B/glue.o $B/layers.o $B/k_n2.o $B/k_n3.o $B/k_n4.o $B/k_n4m.o $B/k_n5.o $B/k_n5m.o $B/k_pl_n5.o $B/k_pl_n5m.o $B/k_plw_n5.o $B/k_plw_n5m.o -o ../csmpn_hip/libcsmpn_hip.so
echo built
