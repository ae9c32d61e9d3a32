#!/bin/bash
# Per-kernel times (rocprofv3 --stats) of the float16 implicit GEMMs of one bench_conv shape under forced tiles / splits.
#   bash scripts/sweep_half_tiles.sh [shape] [nbuf]      (GPU box, repo root)
shape=${1:-local128}; nbuf=${2:-2}
R=$(pwd); cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/ps; env "$@" MG_HALF_NBUF=$nbuf rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o p -- python $R/scripts/bench_conv.py --f16 --only $shape --iters 5 > /dev/null 2>&1
  python - "$*" <<PY
import csv, glob, sys, re
f = glob.glob("/tmp/ps/**/*kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    n = re.sub(r"^void ", "", r["Name"]).replace("(anonymous namespace)::", "").split("(")[0]
    if any(k in n for k in ("dma_kernel", "splitk", "fold")):
        out.append("%s %.1f" % (n.replace("conv_", "").replace("_dma_kernel", "").replace(", true, ", ",n"), float(r["AverageNs"]) / 1e3))
print("%-32s %s" % (sys.argv[1], " | ".join(sorted(out))))
PY
}
run MG_X=1
for t in 64,64 64,128 128,64 128,128; do for sp in 1 4 8 16 32; do run MG_FORCE_CONV_DMA=$t,$sp; done; done
