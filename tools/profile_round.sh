# Round profile: kernel trace of the benchmark step (per-kernel stats + factorisation timeline) and the PMC passes
# for HBM traffic (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes).
# Run on the GPU box from the repo root:  bash tools/profile_round.sh r02   -> gpurun_out/<tag>_*.txt
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
CMD="python bench.py --steps 3 --warmup 1 --no-cpu --no-configs --no-profile"
rm -rf /tmp/prof
(cd $R && timeout 900 rocprofv3 --kernel-trace -d /tmp/prof -- $CMD) > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)   # the largest: bench.py also starts a preflight child
{ echo "== rocprofv3 --kernel-trace -- $CMD"; python $R/tools/rocpd_stats.py $f; } > $R/gpurun_out/${tag}_kernel_stats.txt
{ echo "== tools/chol_timeline.py on the same trace (last of the 4 factorisations)"; TAIL_MS=${TAIL_MS:-0} python $R/tools/chol_timeline.py $f; } > $R/gpurun_out/${tag}_chol_timeline.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd $R && timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python bench.py --steps 1 --warmup 0 --no-cpu --no-configs --no-profile) > /tmp/pmc_$c.log 2>&1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2) /tmp/pmc_$c.csv
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv $R/gpurun_out/${tag}_hbm_traffic.json > $R/gpurun_out/${tag}_pmc_hbm_traffic.txt
tail -3 /tmp/prof.log
