mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_hip_r4.py::test_column_modes_n24_p6_vs_reference --deselect tests/test_hip_r4.py::test_iterative_solver_with_permutation_group_vs_reference 2>&1 | tail -40 > gpurun_out/r4a_pytest.log
bash tools/pcg_trace.sh r4a build/libgdml_hip_r3.so
timeout 400 python tools/cfg_solve_probe.py cg 42 2000 c3x3 64 20 traj > gpurun_out/r4a_cfg3.log 2>&1
timeout 400 python tools/cfg_solve_probe.py cg 100 3000 - 128 20 traj > gpurun_out/r4a_cfg4.log 2>&1
tail -5 gpurun_out/r4a_pytest.log; cat gpurun_out/r4a_pcg_host_gaps.txt | head -50; tail -2 gpurun_out/r4a_cfg3.log; tail -2 gpurun_out/r4a_cfg4.log
