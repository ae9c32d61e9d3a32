#!/bin/bash
# Round-2 evidence batch (one gpurun call): bench lines, rocprofv3 kernel stats, PMC passes.  Outputs under gpurun_out/ev/.
set -u
R=/root/repo; O=$R/gpurun_out/ev; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 3 > $O/bench_train_line.json 2> $O/bench_train.err
python $R/bench.py --fp16 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_train_fp16_line.json 2>/dev/null
python $R/bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_cfg2_f32_line.json 2>/dev/null
python $R/bench.py --config 2 --fp16 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_cfg2_fp16_line.json 2>/dev/null
python $R/bench.py --config 4 --steps 5 --warmup 2 > $O/bench_infer_cfg4_line.json 2>/dev/null
python $R/bench.py --mode codec --steps 10 --warmup 2 > $O/bench_codec_line.json 2>/dev/null
MG_F32_SPLIT=1 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_train_f32split_line.json 2>/dev/null
MG_F32_SPLIT=1 python $R/bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_infer_cfg4_f32split_line.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_train -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_train_line_profiled.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_cfg2h -o t -- python $R/bench.py --config 2 --fp16 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg2_fp16_line_profiled.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_cfg4 -o t -- python $R/bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_infer_cfg4_line_profiled.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/scripts/bench_conv.py --only bottleneck --iters 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/scripts/bench_conv.py --only bottleneck --iters 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq -o s -- python $R/scripts/bench_conv.py --only bottleneck --iters 2 > /dev/null 2>&1
# HBM traffic of the dominant kernels of the other bench lines (same symbols, other layer shapes)
for spec in "c1h:bottleneck:--f16" "c2:trunk2048:" "c2h:trunk2048:--f16" "c4:bottleneck_b64:"; do
  tag=${spec%%:*}; rest=${spec#*:}; layer=${rest%%:*}; flag=${rest#*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$tag -o f -- python $R/scripts/bench_conv.py --only $layer --iters 2 $flag > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$tag -o w -- python $R/scripts/bench_conv.py --only $layer --iters 2 $flag > /dev/null 2>&1
done
find $O -name "*_kernel_trace.csv" -delete
ls -R $O | head -60
