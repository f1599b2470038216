cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pass() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  (cd $R && timeout 600 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-profile) > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  echo "== pass $name: $*"
  python $R/tools/pmc_summary.py $f "" 3 | grep -A12 "gemm_nt_sub\|assemble_wave\|predict_mfma\|negate_shift" | grep -v "^--"
}
pass busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pass wr WRITE_SIZE
pass rd FETCH_SIZE
rm -rf /tmp/pmc_pp
(cd $R && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_pp -- python tools/predict_probe.py 21 5000 5000) > /tmp/pp.log 2>&1
echo "== predict_probe 21 5000 5000"
python $R/tools/pmc_summary.py $(find /tmp/pmc_pp -name "*counter_collection.csv" | head -1) predict_mfma
