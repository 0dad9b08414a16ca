#!/bin/bash
# Iteration aid: rebuild only the row-per-lane unit(s) (and capi.hip when asked) and relink with the
# other, already built objects. `STAMPS=1` builds libcsmpn_hip_stamps.so with per-phase s_memtime
# stamps in the row-per-lane kernels (diagnostic, never shipped). Use `make` before committing.
set -e
cd "$(dirname "$0")/../clifford-group-equivariant-simplicial-message-passing-networks_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value"
B=_build
OUT=../csmpn_hip/libcsmpn_hip.so
RLO=$B/k_rl_n3.o
VF="-mllvm -amdgpu-mfma-vgpr-form"
# (the vgpr-form rewrite pass of clang 22 crashes on the stamped kernels)
if [ -n "$STAMPS" ]; then EXTRA="$EXTRA -DCSMPN_STAMPS"; OUT=../csmpn_hip/libcsmpn_hip_stamps.so; RLO=$B/k_rl_n3_stamps.o; VF=""; fi
# (a failed compile must not be linked over: wait for every job by pid)
pids=()
hipcc $FLAGS $EXTRA $VF -c k_rl_n3.hip -o $RLO & pids+=($!)
if [ "$1" = "capi" ]; then hipcc $FLAGS -c capi.hip -o $B/capi.o & pids+=($!); hipcc $FLAGS -c csr.hip -o $B/csr.o & pids+=($!); fi
for p in "${pids[@]}"; do wait $p; done
hipcc -shared -fPIC --offload-arch=gfx950 $B/capi.o $B/csr.o $B/k_n2.o $B/k_n3.o $B/k_n4.o $B/k_n4m.o $B/k_n5.o $B/k_n5m.o $B/glue.o $B/layers.o $B/k_pl_n5.o $B/k_pl_n5m.o $B/k_plw_n5.o $B/k_plw_n5m.o $RLO -o $OUT
echo built $OUT
