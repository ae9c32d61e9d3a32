#!/bin/bash
# Memory-system / LDS counters of the float16 implicit GEMMs on one layer shape, one small --pmc pass per group, each under its
# own timeout (a derived TCP/TCC group once ran into gpurun's limit).   bash scripts/pmc_half_conv.sh   (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
echo "-- $ctrs"
rm -rf /tmp/pq; timeout 100 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pq --output-format csv -- python /root/repo/scripts/bench_conv.py --f16 --only local128 --iters 3 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/pq/**/*counter_collection.csv",recursive=True)
if not fs: print("no output for", "$ctrs"); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(fs[0])):
    k=r["Kernel_Name"]
    if "dma_kernel" not in k: continue
    k=k.split("(anonymous namespace)::")[1].split("(")[0]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in sorted(acc):
    print(k, {c: round(v/n[k][c]) for c,v in acc[k].items()})
PY
done
