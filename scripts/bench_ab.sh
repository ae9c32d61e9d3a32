#!/bin/bash
# Interleaved same-box A/B of two environments on one bench configuration: prints ms_per_step per run.
# usage: bash scripts/bench_ab.sh "<bench args>" "<env A>" "<env B>" [pairs]
args=$1; ea=$2; eb=$3; n=${4:-3}
one() { env $1 python bench.py $args --no-cpu-baseline --no-also --no-roofline 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]; d=json.loads(l); print('%-28s %8.3f ms  %8.2f %s' % ('$1' or 'default', d['ms_per_step'], d['value'], d['unit']))"; }
for i in $(seq $n); do one "$ea"; one "$eb"; done
