#!/bin/bash
# Round 6: A/B of compile-time variants of the trailing-update GEMM on ONE box: builds variant libraries next to the shipped one
# (build/libgdml_<name>.so, selected with GDML_HIP_LIB) and runs tools/chol_ab.py with each.
#   tools/gemm_variants.sh build        (build container: cross-compiles)
#   tools/gemm_variants.sh run [opts]   (GPU box)
set -e
cd "$(dirname "$0")/.."
HIPCC=/opt/rocm/bin/hipcc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value"
declare -A VAR=( [lp1]="-DGEMM_LP=1" [lp3]="-DGEMM_LP=3" )
if [ "$1" = build ]; then
  mkdir -p build/variants
  for v in "${!VAR[@]}"; do
    $HIPCC $FLAGS ${VAR[$v]} -c sgdml_amd/csrc/chol.hip -o build/variants/chol_$v.o
    objs=$(ls sgdml_amd/csrc/*.o | grep -v '/chol.o')
    $HIPCC --offload-arch=gfx950 -shared -fPIC $objs build/variants/chol_$v.o -ldl -o build/libgdml_$v.so
    echo built build/libgdml_$v.so
  done
else
  shift || true
  for rep in 1 2; do
    echo "== shipped"; python tools/chol_ab.py "$@"
    for v in "${!VAR[@]}"; do echo "== $v"; GDML_HIP_LIB=build/libgdml_$v.so python tools/chol_ab.py "$@"; done
  done
fi
