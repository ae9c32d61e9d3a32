#!/bin/bash
# HBM traffic per launch from PMC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 passes with --kernel-trace only.   usage (GPU box, repo root): bash scripts/pmc_round.sh r03
tag=${1:-r03}
R=$(pwd); out=$R/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() {   # name counter cmd...
    name=$1; ctr=$2; shift 2
    rm -rf /tmp/pmc_$name_$ctr
    rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${name}_$ctr --output-format csv -- "$@" > /dev/null 2>&1
    find /tmp/pmc_${name}_$ctr -name "*counter_collection.csv" | head -1
}
# the 1024-channel trunk layer of configs[1] (the dominant GEMM's layer)
f=$(pass conv FETCH_SIZE python $R/scripts/bench_conv.py --only bottleneck --iters 2)
w=$(pass conv WRITE_SIZE python $R/scripts/bench_conv.py --only bottleneck --iters 2)
python $R/scripts/pmc_traffic.py $f $w $out/${tag}_pmc_hbm_traffic.csv $R/profiles/traffic.json
# K1 / K2 at 4096 clips
f=$(pass mdct FETCH_SIZE $R/scripts/ubench/mdct_bs_bench 4096)
w=$(pass mdct WRITE_SIZE $R/scripts/ubench/mdct_bs_bench 4096)
python $R/scripts/pmc_traffic.py $f $w $out/${tag}_pmc_hbm_traffic_codec.csv $R/profiles/traffic.json "codec"
cp $R/profiles/traffic.json $out/${tag}_traffic.json
