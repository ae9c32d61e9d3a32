#!/bin/bash
# Round-4 parity evidence (GPU box): the full-size step tests with their per-parameter report, twice (the numbers must repeat:
# deterministic yardstick + deterministic kernels), then the summary that goes to profiles/r04_fullsize_step_parity.txt.
out=${1:-gpurun_out/r04_step_report.jsonl}
rm -f "$out"
for rep in 1 2; do
  MG_STEP_REPORT="$out" python -m pytest tests/test_fullsize_step_gpu.py -q -k "gradients" --durations=0 2>&1 | tail -25
done
python -m pytest tests/test_fullsize_step_gpu.py -q -k "replay" --durations=0 2>&1 | tail -8
python - "$out" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
for d in rows:
    g = d["grads"]
    for pre in ("G.", "D."):
        w = max((v[0], k, v[1]) for k, v in g.items() if k.startswith(pre))
        print("%-24s worst %s rel-L2 %.3e (cpu yardstick %.3e) %s" % (d["case"], pre, w[0], w[2], w[1]))
    if d["case"].endswith("_fp16"):
        for k, v in g.items():
            if k.startswith("G.") and len(v) >= 5:
                print("    %-44s e_hip %.3f e_cpu16 %.3f cos_hip %.4f cos_cpu16 %.4f ratio %.3f" % (k, v[0], v[1], v[2], v[3], v[4]))
    st = d.get("steps") or {}
    ups = sorted(((v, k) for k, v in st.items() if k.startswith("update ")), reverse=True)[:5]
    for v, k in ups:
        print("    %-60s %.3e" % (k, v))
    for k, v in st.items():
        if k.startswith("loss_"):
            print("    %-30s hip %.7g oracle %.7g" % (k, v[0], v[1]))
# determinism: the two repetitions must carry identical numbers
half = len(rows) // 2
same = all(rows[i]["grads"] == rows[i + half]["grads"] for i in range(half)) if half and len(rows) == 2 * half else None
print("repeat run bit-identical report:", same)
PY
