#!/bin/bash
# rocprofv3 kernel trace of tools/perm_match_probe.py (GPU box); summary: python tools/rocpd_stats.py gpurun_out/prof_pm/pm_results.db
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pm -o pm -- python tools/perm_match_probe.py > gpurun_out/prof_pm.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof_pm/pm_results.db
