#!/bin/bash
# HBM traffic per launch from PMC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 passes with --kernel-trace only.   usage (GPU box, repo root): bash scripts/pmc_round.sh r04
tag=${1:-r04}
R=$(pwd); out=$R/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() {   # name counter cmd...
    name=$1; ctr=$2; shift 2
    rm -rf /tmp/pmc_$name_$ctr
    timeout 200 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${name}_$ctr --output-format csv -- "$@" > /dev/null 2>&1
    find /tmp/pmc_${name}_$ctr -name "*counter_collection.csv" | head -1
}
# the 1024-channel trunk layer of configs[1] (the dominant GEMM's layer)
f=$(pass conv FETCH_SIZE python $R/scripts/bench_conv.py --only bottleneck --iters 2)
w=$(pass conv WRITE_SIZE python $R/scripts/bench_conv.py --only bottleneck --iters 2)
python $R/scripts/pmc_traffic.py $f $w $out/${tag}_pmc_hbm_traffic.csv $R/profiles/traffic.json
# K1 / K2 at 4096 clips
f=$(pass mdct FETCH_SIZE python $R/bench.py --mode codec --steps 3 --warmup 1)
w=$(pass mdct WRITE_SIZE python $R/bench.py --mode codec --steps 3 --warmup 1)
python $R/scripts/pmc_traffic.py $f $w $out/${tag}_pmc_hbm_traffic_codec.csv $R/profiles/traffic.json "codec"
# the dominant kernels of the other bench lines (same symbols, other layer shapes): one section each
for spec in "configs[1] --fp16:c1h:bottleneck:--f16" "configs[2]:c2:trunk2048:" "configs[2] --fp16:c2h:trunk2048:--f16" "configs[4]:c4:bottleneck_b64:"; do
    section=${spec%%:*}; rest=${spec#*:}; key=${rest%%:*}; rest=${rest#*:}; layer=${rest%%:*}; flag=${rest#*:}
    f=$(pass $key FETCH_SIZE python $R/scripts/bench_conv.py --only $layer --iters 2 $flag)
    w=$(pass $key WRITE_SIZE python $R/scripts/bench_conv.py --only $layer --iters 2 $flag)
    python $R/scripts/pmc_traffic.py $f $w $out/${tag}_pmc_hbm_traffic_$key.csv $R/profiles/traffic.json "$section" > /dev/null
done
# configs[4]: the stride-2 ladder at the inference batch (conv_fwd_dma / conv_dgrad_dma: 30 % of that step) -- time, TFLOP/s and
# HBM bytes per launch of every rung (forward = the down rungs, data gradient = the transposed-convolution up rungs)
export MG_BENCH_BATCH=64
( cd $R; python scripts/bench_conv.py --only down512,down256,down128,down64 --iters 10 ) > $out/${tag}_ladder_b64_times.txt 2>/dev/null
f=$(pass lad FETCH_SIZE python $R/scripts/bench_conv.py --only down512,down256,down128,down64 --iters 2)
w=$(pass lad WRITE_SIZE python $R/scripts/bench_conv.py --only down512,down256,down128,down64 --iters 2)
PMC_BY_GRID=1 python $R/scripts/pmc_traffic.py $f $w $out/${tag}_pmc_hbm_traffic_c4_ladder.csv $R/profiles/traffic.json "configs[4] ladder" > /dev/null
unset MG_BENCH_BATCH
cp $R/profiles/traffic.json $out/${tag}_traffic.json
