# A/B of the late-phase CU-masked stream pair in the Cholesky factorisation (same box)
for cfg in "0 32" "16000 32" "24000 32" "32000 32" "24000 16" "24000 64" "40000 32" "0 32"; do
  set -- $cfg
  echo "== GDML_CHOL_MASK_ROWS=$1 GDML_CHOL_MASK_CUS=$2"
  GDML_CHOL_MASK_ROWS=$1 GDML_CHOL_MASK_CUS=$2 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.4f s  factor %.1f ms  resid %.1e  gemm %.1f TF' % (d['value'], d['phases_ms']['factor'], d['solve_rel_residual'], d['roofline']['achieved']))"
done
