mkdir -p gpurun_out
timeout 600 python -m pytest -q -m gpu tests/test_hip_r4.py::test_iterative_solver_with_permutation_group_vs_reference tests/test_hip_parity.py -k "permutation_group or analytic or cholesky" 2>&1 | tail -5 > gpurun_out/r4e_pytest.log
timeout 300 python tools/chol_ab.py - gemm.a4=1 - gemm.a4=1 > gpurun_out/r4e_gemm_a4_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cgprof; (cd $GRAFT_REPO_ROOT && timeout 400 rocprofv3 --kernel-trace -d /tmp/cgprof -- python tools/cg_step_probe.py 20) > /tmp/cgprof.log 2>&1
f=$(find /tmp/cgprof -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
cd $GRAFT_REPO_ROOT
{ echo "== rocprofv3 --kernel-trace -- python tools/cg_step_probe.py 20  (configs[2] step: N=21, N_train=5000, k=200: n=315000, m=12600; two steps)"; python tools/rocpd_stats.py $f | head -22; grep "^[01] " /tmp/cgprof.log; } > gpurun_out/r4e_cg_step_kernels.txt
timeout 900 python bench.py --gpus 2 --comm host --steps 1 --warmup 0 --cg-iters 10 > gpurun_out/r4e_bench_2ranks_host.json 2> gpurun_out/r4e_bench_2ranks_host.err
cat gpurun_out/r4e_pytest.log gpurun_out/r4e_gemm_a4_ab.txt gpurun_out/r4e_cg_step_kernels.txt; tail -c 2500 gpurun_out/r4e_bench_2ranks_host.json; tail -5 gpurun_out/r4e_bench_2ranks_host.err
