cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python tools/matvec_probe.py
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mv -o mv -- python tools/matvec_probe.py > gpurun_out/prof_mv.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof_mv/mv_results.db | head -20
rm -rf gpurun_out/prof_mv
