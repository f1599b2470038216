mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r4d_pytest.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r4d_bench.json 2> gpurun_out/r4d_bench.err
HIP_ENABLE_DEFERRED_LOADING=0 timeout 200 python tools/sweep_hostprof.py 2>/dev/null | grep "^sweep" > gpurun_out/r4d_sweep_eager_loading.txt
tail -6 gpurun_out/r4d_pytest.log; tail -c 6000 gpurun_out/r4d_bench.json; tail -5 gpurun_out/r4d_bench.err; cat gpurun_out/r4d_sweep_eager_loading.txt
