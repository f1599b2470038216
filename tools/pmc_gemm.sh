# MFMA-pipe occupancy of the fused GEMM launches (PMC pass of one benchmark step):  bash tools/pmc_gemm.sh <tag>
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_g
(cd $R && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_g -- python bench.py --steps 1 --warmup 0 --no-cpu --no-configs --no-profile) > /tmp/pmc_g.log 2>&1
f=$(find /tmp/pmc_g -name "*counter_collection.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
{ echo "== rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- bench.py --steps 1 (6 largest fused launches)"; python $R/tools/pmc_summary.py $f gemm_nt_sub_diag 6; } | tee $R/gpurun_out/${tag}_pmc_gemm_mfma.txt
