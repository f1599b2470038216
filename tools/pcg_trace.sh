# PCG host-gap trace: rocprofv3 kernel trace of tools/cg_step_probe.py (50 iterations of the configs[2] step on one GPU),
# summarised by tools/pcg_gaps.py.  bash tools/pcg_trace.sh <tag> [old-library.so]
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() {  # $1 = label, $2 = library override or ""
  rm -rf /tmp/pcgprof_$1
  if [ -n "$2" ]; then export GDML_HIP_LIB=$2; else unset GDML_HIP_LIB; fi
  (cd $R && timeout 600 rocprofv3 --kernel-trace -d /tmp/pcgprof_$1 -- python tools/cg_step_probe.py 50) > /tmp/pcgprof_$1.log 2>&1
  f=$(find /tmp/pcgprof_$1 -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  python $R/tools/pcg_gaps.py $f "$1"
  grep "^1 " /tmp/pcgprof_$1.log
}
{
  echo "== rocprofv3 --kernel-trace -- python tools/cg_step_probe.py 50   (N=21, N_train=5000, k=200: n=315000, m=12600)"
  if [ -n "$2" ]; then run before_round3_library $R/$2; fi
  run after ""
} > $R/gpurun_out/${tag}_pcg_host_gaps.txt 2>&1
