#!/bin/bash
# Experiment aid: builds tools/_bin/libexp_<name>.so = the shipped objects with the (row, channel)-per-lane unit
# recompiled under extra -D flags. usage: build_variant_cl.sh name -DFOO ...   (run with CSMPN_LIB=tools/_bin/libexp_<name>.so)
set -e
name=$1; shift
cd "$(dirname "$0")/../clifford-group-equivariant-simplicial-message-passing-networks_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value -fno-slp-vectorize"
B=_build; O=../../tools/_bin
mkdir -p $O
hipcc $FLAGS "$@" -c k_cl_n3.hip -o $O/exp_$name.o
hipcc -shared -fPIC --offload-arch=gfx950 $B/capi.o $B/csr.o $B/k_n2.o $B/k_n3.o $B/k_n4.o $B/k_n4m.o $B/k_n5.o $B/k_n5m.o $B/glue.o $B/layers.o $B/k_pl_n5.o $B/k_pl_n5m.o $B/k_plw_n5.o $B/k_plw_n5m.o $B/k_cm_n3.o $O/exp_$name.o -o $O/libexp_$name.so
rm -f $O/exp_$name.o
echo built $O/libexp_$name.so
