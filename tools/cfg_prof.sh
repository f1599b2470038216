#!/bin/bash
# rocprofv3 kernel trace of one iterative configuration run to solver_tol (GPU box):  bash tools/cfg_prof.sh cfg3
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
c=${1:-cfg3}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg -o t -- python tools/cg_form_probe.py $c 3 > gpurun_out/prof_cfg.log 2>&1
echo "== rocprofv3 --kernel-trace -- python tools/cg_form_probe.py $c 3"
python tools/rocpd_stats.py gpurun_out/prof_cfg/t_results.db | head -28
rm -rf gpurun_out/prof_cfg
