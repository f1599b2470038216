for a in "21 5000 5000" "21 1000 1000" "21 1000 4000" "9 1000 2000" "15 2000 2000"; do
 timeout 120 python tools/predict_probe.py $a
done
