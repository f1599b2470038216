cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for c in cfg2 cfg3 cfg4; do
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mv -o mv -- python tools/matvec_probe.py $c > gpurun_out/prof_mv.log 2>&1
echo "== $c"; python tools/rocpd_stats.py gpurun_out/prof_mv/mv_results.db | head -9
rm -rf gpurun_out/prof_mv
done
