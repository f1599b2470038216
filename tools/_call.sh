mkdir -p gpurun_out
timeout 600 python -m pytest -q -m gpu tests/test_hip_r4.py::test_assemble_perm_mode_matrix tests/test_hip_r4.py::test_column_modes_n24_p6_vs_reference "tests/test_hip_scale.py::test_K_samples_at_config_shapes" 2>&1 | tail -8 > gpurun_out/r4h_pytest.log
{ echo "== x_j two chunks ahead; VU = 3 (library)"; timeout 300 python tools/asm_perm_ablate.py quick; echo "== VU = 2"; GDML_HIP_LIB=$PWD/build/libgdml_vu2.so timeout 300 python tools/asm_perm_ablate.py quick; echo "== VU = 4"; GDML_HIP_LIB=$PWD/build/libgdml_vu4.so timeout 300 python tools/asm_perm_ablate.py quick; } > gpurun_out/r4h_asm_perm_vu.txt 2>&1
cat gpurun_out/r4h_pytest.log gpurun_out/r4h_asm_perm_vu.txt
