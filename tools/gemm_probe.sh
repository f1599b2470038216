#!/bin/bash
# GEMM ablation: prints aggregated gemm TF/s of one factorization at n = 63*N_TRAIN for each debug mask
NT=${1:-500}
for d in 0 1 2 4 6 7 8 15; do
  GDML_GEMM_DEBUG=$d python bench.py --n-train $NT --steps 1 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
try:
  d=json.loads(sys.stdin.read()); print('dbg=$d gemm %.1f TF  factor %.1f ms' % (d['roofline']['achieved'], d['phases_ms']['factor']))
except Exception as e: print('dbg=$d failed', e)"
done
