#!/bin/bash
# rocprofv3 kernel stats + ordered step trace of one bench configuration.  usage: bash scripts/r06_prof.sh <tag> <name> <bench args...>
tag=$1; name=$2; shift 2
R=$(pwd); out=$R/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$name -o kt -- python $R/bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-roofline \
    > $out/${tag}_${name}_line_profiled.json 2> /tmp/kt_$name.err
f=$(find /tmp/kt_$name -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $R/scripts/step_trace.py $f --adam ${ADAM:-adam_} > $out/${tag}_${name}_step_trace.txt
s=$(find /tmp/kt_$name -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && cp $s $out/${tag}_${name}_kernel_stats.csv
grep "^# step" $out/${tag}_${name}_step_trace.txt
