#!/bin/bash
# usage: tools/kstat_cmd.sh "key=val,..." <command...>  -> rocprofv3 kernel-trace stats (tools/rocpd_stats.py) of an arbitrary
# command under GDML_OPTIONS (first argument; '-' = none)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
o=$1; shift
cd /tmp && export TMPDIR=/tmp
[ "$o" = "-" ] && unset GDML_OPTIONS || export GDML_OPTIONS=$o
rm -rf /tmp/kc_prof
(cd $R && timeout 600 rocprofv3 --kernel-trace -d /tmp/kc_prof -- "$@") > /tmp/kc.log 2>&1
f=$(find /tmp/kc_prof -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
echo "== GDML_OPTIONS=$o $*"
tail -2 /tmp/kc.log
python $R/tools/rocpd_stats.py $f
