#!/bin/bash
# Builds timing-only variants of the library with parts of assemble_perm_kernel's V phase removed (PERM_ABL, see assemble_perm.hip)
# into build/vabl/libgdml_hip_<k>.so; run on the GPU box with tools/asm_perm_vabl_run.sh.  Build container: bash tools/asm_perm_vabl.sh 1 2 5 ...
set -e
cd "$(dirname "$0")/../sgdml_amd/csrc"
mkdir -p ../../build/vabl
OBJS=$(ls *.o | grep -v assemble_perm.o)
for k in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-everything -DPERM_ABL=$k -c assemble_perm.hip -o ../../build/vabl/assemble_perm_$k.o &
done
wait
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../build/vabl/assemble_perm_$k.o -ldl -o ../../build/vabl/libgdml_hip_$k.so
done
ls -la ../../build/vabl/*.so
