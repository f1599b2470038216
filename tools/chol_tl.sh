cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rows in 0 32000; do
rm -rf /tmp/prof
(cd $R && GDML_CHOL_MASK_ROWS=$rows timeout 600 rocprofv3 --kernel-trace -d /tmp/prof -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-profile) > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
echo "== mask_rows=$rows"
python $R/tools/chol_timeline.py $f
done
