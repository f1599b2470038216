# PMC passes over one assembly shape:  bash tools/pmc_asm.sh <tag> N M kind [opts...]   -> gpurun_out/<tag>_pmc.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/${tag}_pmc.txt
: > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  (cd $R && timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -- python tools/asm_perm_one.py "$@") > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set" >> $out
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f assemble_p 2 >> $out; else tail -5 /tmp/pmc_$i.log >> $out; fi
done
cat $out
