#!/bin/bash
# builds scripts/slp_repro/libslp_repro.so: the same kernel source three times with different code generation
set -e
cd "$(dirname "$0")"
H=/opt/rocm/bin/hipcc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
$H $F -fslp-vectorize -DVICTIM_NAME=up_slp -c victim.hip -o v_slp.o
$H $F -fno-slp-vectorize -DVICTIM_NAME=up_noslp -c victim.hip -o v_noslp.o
$H $F -fslp-vectorize -mllvm -amdgpu-waitcnt-forcezero -DVICTIM_NAME=up_slp_wait0 -c victim.hip -o v_wait0.o
$H $F -fno-slp-vectorize -c host.hip -o host.o
$H --offload-arch=gfx950 -shared -fPIC -o libslp_repro.so v_slp.o v_noslp.o v_wait0.o host.o
for v in slp noslp wait0; do $H $F $( [ $v = noslp ] && echo -fno-slp-vectorize || echo -fslp-vectorize ) $( [ $v = wait0 ] && echo "-mllvm -amdgpu-waitcnt-forcezero" ) -DVICTIM_NAME=up_$v -S --cuda-device-only -o v_$v.s victim.hip 2>/dev/null; done
grep -c v_pk_ v_slp.s v_noslp.s v_wait0.s || true
