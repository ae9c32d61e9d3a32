#!/bin/bash
# Round evidence in one GPU call: rocprofv3 --kernel-trace --stats summaries of the bench lines (-> gpurun_out/rNN_*), to be
# copied into profiles/.   usage (GPU box, from the repo root): bash scripts/profile_round.sh r04
tag=${1:-r04}
R=$(pwd)
out=$R/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
prof() {   # name, bench args...
    name=$1; shift
    rm -rf /tmp/prof_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-also \
        > $out/${tag}_${name}_line_profiled.json 2> /tmp/prof_$name.err
    f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $out/${tag}_${name}_kernel_stats.csv
    a=$(find /tmp/prof_$name -name "*agent_info.csv" | head -1)
    [ -n "$a" ] && cp $a $out/${tag}_agent_info.csv
    tail -c 400 $out/${tag}_${name}_line_profiled.json; echo
}
prof bench_train --steps 10 --warmup 2
prof bench_cfg2_fp16 --config 2 --fp16 --steps 10 --warmup 2
prof bench_infer_cfg4 --config 4 --steps 10 --warmup 2
prof bench_codec --mode codec --steps 10 --warmup 2
cd $R
python bench.py --mode codec --steps 20 --warmup 5 > $out/${tag}_bench_codec_line.json 2>/dev/null
python bench.py --config 2 --steps 20 --warmup 5 --no-also > $out/${tag}_bench_cfg2_f32_line.json 2>/dev/null
python bench.py --fp16 --steps 20 --warmup 5 --no-also > $out/${tag}_bench_train_fp16_line.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_train_line.json 2>/dev/null
for f in bench_codec bench_cfg2_f32 bench_train_fp16 bench_train; do tail -c 300 $out/${tag}_${f}_line.json; echo; done

# K1 / K2 micro-benchmark (the factored-transform product kernels of mdct_ct.h beside the retired bf16 x 3 and f32-pipe dense-table
# kernels, which live on in scripts/ubench/ for this comparison) and their SQ / LDS counters
hipcc --version > /dev/null 2>&1
[ -x scripts/ubench/mdct_bs_bench ] && for b in 4096 64 8; do scripts/ubench/mdct_bs_bench $b; done > $out/${tag}_mdct_bs_ubench.log 2>&1
for b in 4096 1024 512 256 128 64 8; do echo "##### clips $b"; scripts/ubench/mdct_b3_bench $b; done > $out/${tag}_mdct_ct_ubench.log 2>&1
cd /tmp
rm -rf /tmp/pmc_bs /tmp/pmc_lds
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY \
    -d /tmp/pmc_bs --output-format csv -- $R/scripts/ubench/mdct_b3_bench 4096 > /dev/null 2>&1
f=$(find /tmp/pmc_bs -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python $R/scripts/pmc_sq.py $f $out/${tag}_pmc_sq_mdct.csv > /dev/null
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD \
    -d /tmp/pmc_lds --output-format csv -- $R/scripts/ubench/mdct_b3_bench 1024 > /dev/null 2>&1
f=$(find /tmp/pmc_lds -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" > $out/${tag}_pmc_lds_mdct_ct.txt <<'PY'
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print("# per dispatch at 1024 clips (4096 tiles x 8 waves = 32768 wave-tiles): counter sums over the chip")
for k, c in acc.items():
    if "mdct4_" in k:
        print(k, {n: round(v / cnt[k][n]) for n, v in sorted(c.items())})
PY
cd $R
# (the full-size step parity report is its own call: scripts/r05_step_parity.sh)
tail -5 $out/${tag}_mdct_bs_ubench.log

# float16 GEMM structure ablation on the dense twin of the 128-channel 64x128 layers (HISTORY.md section 3, --fp16): which part of
# hgemm_kernel's loop sets its time.  Binaries: hipcc ... -DHG_DEBUG_{NO_DMA,NO_LDS,NO_COMPUTE,NO_BARRIER} scripts/ubench/hgemm_bench.hip
( cd scripts/ubench; for v in "" _NO_DMA _NO_LDS _NO_COMPUTE _NO_BARRIER; do [ -x ./hgemm_bench$v ] || continue; echo "### hgemm_bench$v"; timeout 120 ./hgemm_bench$v "local128 as" 2>&1 | grep -E "==|splits  1 " | grep -v " rc"; done; echo "### trunk weight gradient"; timeout 120 ./hgemm_bench wgrad 2>&1 | grep -E "==|A-stat|128x128/4x2 " ) > $out/${tag}_hgemm_ablation.log 2>&1
