#!/bin/bash
# A/B of two BUILDS of libgdml_hip.so on ONE box (round 6: the boxes of the pool differ by up to 7 % in the factorisation, so a
# compile-time change cannot be judged across gpurun calls).  Build the baseline and the experiment here, keep both next to this
# script, and let the GPU box alternate them under the same command:
#     make -C sgdml_amd/csrc && cp sgdml_amd/libgdml_hip.so tools/_ab_exp.so        # experiment (working tree)
#     git stash && make -C sgdml_amd/csrc && cp sgdml_amd/libgdml_hip.so tools/_ab_base.so && git stash pop && make -C sgdml_amd/csrc
#     gpurun -- 'bash tools/lib_ab.sh python tools/chol_ab.py -'
# (tools/_ab_*.so are not tracked: remove them afterwards; the library in the tree is restored from the experiment at the end.)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
for v in base exp base exp; do
  cp tools/_ab_$v.so sgdml_amd/libgdml_hip.so
  echo "== $v"
  "$@" 2>&1 | tail -${AB_TAIL:-1}
done
cp tools/_ab_exp.so sgdml_amd/libgdml_hip.so
