timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "chol or solve or dropin or full_size" > gpurun_out/t.txt 2>&1; grep -E "passed|failed" gpurun_out/t.txt
for i in 1 2; do
for L in libgdml_hip.so libgdml_hip_old.so; do
  echo "== $L"
  GDML_HIP_LIB=$PWD/sgdml_amd/$L timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.4f s  phases %s  resid %.1e' % (d['value'], {k: round(v, 1) for k, v in d['phases_ms'].items()}, d['solve_rel_residual']))"
done
done
