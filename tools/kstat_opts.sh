#!/bin/bash
# usage: tools/kstat_opts.sh PATTERN "key=val,..." ...  -> rocprofv3 kernel-trace stats of bench.py (1 step) per option set,
# rows matching PATTERN (egrep) only
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
for o in "$@"; do
  [ "$o" = "-" ] && unset GDML_OPTIONS || export GDML_OPTIONS=$o
  rm -rf /tmp/ks_prof
  (cd $R && timeout 600 rocprofv3 --kernel-trace -d /tmp/ks_prof -- python bench.py --steps 1 --warmup 0 --no-cpu --no-configs --no-profile) > /tmp/ks.log 2>&1
  f=$(find /tmp/ks_prof -name "*.db" | head -1)
  echo "== $o"
  python $R/tools/rocpd_stats.py $f | egrep "calls|$pat"
done
