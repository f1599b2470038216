#!/bin/bash
# GPU box: times assemble_perm_kernel with every variant library under build/vabl (tools/asm_perm_vabl.sh builds the V-phase
# ablations, timing only) next to the shipped one.  Usage: bash tools/asm_perm_vabl_run.sh [shape list as python literal]
cd "$(dirname "$0")/.."
SHAPES=${1:-"[(42,300,'c3^3'),(100,120,'id')]"}
for so in sgdml_amd/libgdml_hip.so $(ls build/vabl/libgdml_hip_*.so | sort); do
  k=$(basename $so .so | sed 's/libgdml_hip_\?//'); [ -z "$k" ] && k=0
  GDML_HIP_LIB=$PWD/$so python - "$k" "$SHAPES" <<'PY' 2>&1 | grep -v "^$"
import sys
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
from asm_perm_check import time_case
k=sys.argv[1]
for N, M, kind in eval(sys.argv[2]):
    time_case(N, M, kind, {'asm.pts': 0}, label='var%-3s' % k)
PY
done
