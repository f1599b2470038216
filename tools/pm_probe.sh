timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "predict or matvec or dropin or pcg or errors" > gpurun_out/pm_tests.txt 2>&1
grep -E "passed|failed" gpurun_out/pm_tests.txt
for a in "21 5000 5000" "21 1000 1000" "21 1000 4000" "9 1000 2000" "15 2000 2000" "23 1000 1000"; do
 timeout 120 python tools/predict_probe.py $a
done
