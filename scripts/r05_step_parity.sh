#!/bin/bash
# Round-5 parity evidence (GPU box): the full-size step tests with their per-parameter report -- incl. the bench's batch 8 and the
# three --fp16 iterations -- then configs[1] once more with the discriminators' 4x4 layers on the DIRECT kernels (no F(2x2,4x4) /
# F(4x4,2x2) Winograd: VERDICT r4 weak 1 asks whether the 25-position families explain the D-layer error), then the summary that
# goes to profiles/r05_fullsize_step_parity.txt.
out=${1:-gpurun_out/r05_step_report.jsonl}
rm -f "$out" "$out.direct"
MG_STEP_REPORT="$out" python -m pytest tests/test_fullsize_step_gpu.py -q -k "gradients" --durations=0 2>&1 | tail -25
MG_NO_WINOGRAD4=1 MG_NO_WINOGRAD42=1 MG_STEP_REPORT="$out.direct" python -m pytest tests/test_fullsize_step_gpu.py -q -k "configs1_f32_batch2" 2>&1 | tail -5
python - "$out" "$out.direct" <<'PY'
import json, sys
def show(path, title):
    print("==", title)
    for d in [json.loads(l) for l in open(path)]:
        g = d["grads"]
        for pre in ("G.", "D."):
            w = max((v[0], k, v[1]) for k, v in g.items() if k.startswith(pre))
            print("%-30s worst %s rel-L2 %.3e (cpu yardstick %.3e) %s" % (d["case"], pre, w[0], w[2], w[1]))
        for k in ("D.scale0_layer0.0.weight", "D.scale0_layer1.0.weight", "D.scale0_layer2.0.weight", "D.scale0_layer3.0.weight"):
            if k in g:
                print("    %-40s hip %.3e  cpu-f32 %.3e" % (k, g[k][0], g[k][1]))
        for k, v in d["losses"].items():
            print("    loss %-12s hip %.7g  oracle-f64 %.7g  oracle-yardstick %.7g" % (k, v[0], v[1], v[2]))
        st = d.get("steps") or {}
        if "scales" in st:
            print("    loss-scale trajectory hip %s oracle %s" % (st["scales"][0], st["scales"][1]))
        sz = [v for k, v in st.items() if k.startswith("update-size ")]
        if sz:
            print("    update size |hip| / |oracle| over %d parameters: min %.3f max %.3f" % (len(sz), min(sz), max(sz)))
        for v, k in sorted(((v, k) for k, v in st.items() if k.startswith("update ")), reverse=True)[:3]:
            print("    %-60s %.3e" % (k, v))
        for k, v in st.items():
            if k.startswith("loss_"):
                print("    %-30s hip %.7g oracle %.7g" % (k, v[0], v[1]))
show(sys.argv[1], "default plans")
show(sys.argv[2], "MG_NO_WINOGRAD4=1 MG_NO_WINOGRAD42=1 (direct 4x4 kernels in the discriminators)")
PY
