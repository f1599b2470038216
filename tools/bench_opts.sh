#!/bin/bash
# usage: tools/bench_opts.sh "key=val,..." ...   -> one summary line of bench.py (analytic headline) per option set
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for o in "$@"; do
  [ "$o" = "-" ] && unset GDML_OPTIONS || export GDML_OPTIONS=$o
  python bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu --no-configs 2>/tmp/bench_opts.err | tail -1 > /tmp/bench_opts.json
  python - "$o" <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/bench_opts.json'))
    r = d['roofline']
    print('%-40s value %.4f s | %s | gemm %.1f TF over %d launches | resid %.2e' % (sys.argv[1], d['value'],
          ' '.join('%s %.1f' % kv for kv in d['phases_ms'].items()), r['achieved'], r['launches'], d['solve_rel_residual']))
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('/tmp/bench_opts.err').read()[-600:])
PY
done
