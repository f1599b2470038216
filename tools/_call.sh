mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r4j_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4j_smoke.log 2>&1
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r4j_bench.json 2> gpurun_out/r4j_bench.err
bash tools/profile_round.sh r4j > gpurun_out/r4j_profile_round.log 2>&1
grep -E "passed|failed" gpurun_out/r4j_pytest.log; cat gpurun_out/r4j_smoke.log; head -c 1500 gpurun_out/r4j_bench.json; echo; tail -3 gpurun_out/r4j_bench.err; head -12 gpurun_out/r4j_kernel_stats.txt; cat gpurun_out/r4j_pmc_hbm_traffic.txt | head -20
