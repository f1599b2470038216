# Where the MFMA pipes of the fused trailing-update launches idle (round-4 review, next #4b): two SQ passes of one benchmark
# step -- wave-cycle accounting (parked in s_waitcnt / barrier vs stalled at issue vs issuing, and the LDS / VMEM / VALU share of
# the issue cycles) and the pipe view (MFMA busy cycles, LDS bank conflicts).  bash tools/pmc_gemm_stalls.sh <tag>
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
CMD="python bench.py --steps 1 --warmup 0 --no-cpu --no-configs --no-profile"
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
out=$R/gpurun_out/${tag}_pmc_gemm_stalls.txt
: > $out
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/pmc_s$i
  (cd $R && timeout 600 rocprofv3 --pmc $P --output-format csv -d /tmp/pmc_s$i -- $CMD) > /tmp/pmc_s$i.log 2>&1
  f=$(find /tmp/pmc_s$i -name "*counter_collection.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  { echo "== rocprofv3 --pmc $P -- $CMD   (the 6 largest fused launches)"; python $R/tools/pmc_summary.py $f gemm_nt_sub_diag 6; echo; } >> $out
done
cat $out
