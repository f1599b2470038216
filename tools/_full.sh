mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r6f_gpu_tests.txt
cat gpurun_out/r6f_gpu_tests.txt
