# Per-kernel time of the configs[0] sigma sweep (sigma-grid reuse question) and of a 27-permutation sweep:
#   bash tools/sweep_profile.sh <tag>   -> gpurun_out/<tag>_sweep_kernel_stats.txt
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/${tag}_sweep_kernel_stats.txt
: > $out
for case in "cfg0" "p27"; do
  rm -rf /tmp/sw_$case
  (cd $R && timeout 300 rocprofv3 --kernel-trace -d /tmp/sw_$case -- python tools/sweep_one.py $case) > /tmp/sw_$case.log 2>&1
  f=$(find /tmp/sw_$case -name "*.db" | head -1)
  { echo "== rocprofv3 --kernel-trace -- python tools/sweep_one.py $case"; grep "^SWEEP" /tmp/sw_$case.log; python $R/tools/rocpd_stats.py $f | head -16; } >> $out
done
cat $out
