# full GPU check: parity suite, smoke, bench (+ kernel trace of the bench)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.txt 2>&1
grep -E "passed|failed|error" gpurun_out/gpu_tests.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
(cd $GRAFT_REPO_ROOT && timeout 900 rocprofv3 --kernel-trace -d /tmp/prof -- python bench.py --steps 2 --warmup 1) > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $f $GRAFT_REPO_ROOT/gpurun_out/kernel_stats.txt > /dev/null
head -20 $GRAFT_REPO_ROOT/gpurun_out/kernel_stats.txt
