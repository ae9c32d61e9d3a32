#!/bin/bash
# Full-size step parity under kernel-family switches: which family owns which part of the gradient error.
# usage (GPU box): bash scripts/diag_fullsize_step.sh gpurun_out/step_report.jsonl
out=${1:-gpurun_out/step_report.jsonl}
rm -f "$out"
run() { echo "== $*"; env "$@" MG_STEP_REPORT="$out" python -m pytest tests/test_fullsize_step_gpu.py -q -x -k "$K" 2>&1 | tail -3; }
K=configs1_f32 run MG_TAG=default
K=configs1_f32 run MG_TAG=no_wino4 MG_NO_WINOGRAD4=1 MG_NO_WINOGRAD42=1
K=configs1_f32 run MG_TAG=no_wino_all MG_NO_WINOGRAD4=1 MG_NO_WINOGRAD42=1 MG_NO_WINOGRAD=1
python - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    g = d["grads"]
    worstG = max((v[0], k, v[1]) for k, v in g.items() if k.startswith("G."))
    worstD = max((v[0], k, v[1]) for k, v in g.items() if k.startswith("D."))
    print(d["case"], d["env"].get("MG_TAG"), "worst G %.2e (cpu32 %.2e) %s | worst D %.2e (cpu32 %.2e) %s" % (worstG[0], worstG[2], worstG[1], worstD[0], worstD[2], worstD[1]))
    for k, v in g.items():
        if k.startswith("D."):
            print("    %-28s hip %.2e  cpu32 %.2e" % (k, v[0], v[1]))
PY
