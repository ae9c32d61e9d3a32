#!/bin/bash
# One ordered step of launches per bench configuration (scripts/step_trace.py) + the aten ops torch still launches inside a step.
# usage (GPU box, repo root): bash scripts/r06_trace.sh r06a
tag=${1:-r06}
R=$(pwd); out=$R/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
tr() {  # name, bench args
    name=$1; shift
    rm -rf /tmp/kt_$name
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$name -o kt -- python $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-roofline \
        > $out/${tag}_${name}_line_traced.json 2> /tmp/kt_$name.err
    f=$(find /tmp/kt_$name -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python $R/scripts/step_trace.py $f > $out/${tag}_${name}_step_trace.txt
    tail -c 300 $out/${tag}_${name}_line_traced.json; echo
    grep "^# step" $out/${tag}_${name}_step_trace.txt
}
tr cfg2_fp16 --config 2 --fp16
tr train
cd $R
python scripts/trace_aten_ops.py --config 2 --fp16 > $out/${tag}_aten_cfg2_fp16.txt 2>&1
python scripts/trace_aten_ops.py --config 1 > $out/${tag}_aten_cfg1.txt 2>&1
tail -3 $out/${tag}_aten_cfg2_fp16.txt
