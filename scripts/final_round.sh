#!/bin/bash
# End-of-round evidence that depends on the final code: the driver's command (headline + also lines, live PMC traffic), the PMC
# traffic tables, the GPU suite.   usage (GPU box, repo root): bash scripts/final_round.sh r04
tag=${1:-r04}
R=$(pwd); out=$R/gpurun_out; mkdir -p $out
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_train_line.json 2> $out/${tag}_bench_train_line.err
tail -c 600 $out/${tag}_bench_train_line.json; echo
python - "$out/${tag}_bench_train_line.json" "$out" "$tag" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
names = ["bench_cfg2_fp16_line", "bench_infer_cfg4_line", "bench_codec_line"]
for a, n in zip(d.get("also", []), names):
    json.dump(a, open("%s/%s_%s.json" % (sys.argv[2], sys.argv[3], n), "w"))
    print(n, a.get("value"), a.get("ms_per_step"), (a.get("roofline") or {}).get("frac"), (a.get("roofline") or {}).get("traffic"))
PY
bash scripts/pmc_round.sh $tag > $out/pmc_round.log 2>&1
python -m pytest tests -m gpu -q --durations=30 2>&1 | grep -v -i "rccl\|^HIP version\|^ROCm version\|^Hostname\|^Librar" | tail -60 > $out/${tag}_pytest.log
cat $out/${tag}_pytest.log | tail -4
