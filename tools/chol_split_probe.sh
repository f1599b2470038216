GDML_CHOL_SPLIT=1 GDML_CHOL_NB=128 timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "chol or panel or solve or dropin" > gpurun_out/split_tests.txt 2>&1
grep -E "passed|failed" gpurun_out/split_tests.txt
run() {
  echo "== $*"
  env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.4f s  factor %.1f ms  resid %.1e  gemm %.1f TF' % (d['value'], d['phases_ms']['factor'], d['solve_rel_residual'], d['roofline']['achieved']))"
}
run GDML_CHOL_SPLIT=0
run GDML_CHOL_SPLIT=1 GDML_CHOL_AUX_CUS=32
run GDML_CHOL_SPLIT=1 GDML_CHOL_AUX_CUS=16
run GDML_CHOL_SPLIT=1 GDML_CHOL_AUX_CUS=48
run GDML_CHOL_SPLIT=1 GDML_CHOL_AUX_CUS=32 GDML_CHOL_PANEL_B=1.7e-4
run GDML_CHOL_SPLIT=1 GDML_CHOL_AUX_CUS=32 GDML_CHOL_PANEL_B=4e-5
run GDML_CHOL_SPLIT=0
