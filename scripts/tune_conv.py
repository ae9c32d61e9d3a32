"""Empirical tile/split search for the fwd / dgrad passes (MG_FORCE_PLAN) on the configs[1] layer shapes.
Each configuration runs in a fresh process (the override is read per call, but keep it simple)."""
import os, subprocess, sys, re
shapes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bottleneck", "down512", "down256", "down128", "down64", "d256_512", "d128_256", "d64_128"]
cfgs = [(bm, bn, sp) for (bm, bn) in ((128, 128), (128, 64), (64, 64)) for sp in (1, 2, 3, 4, 6, 8, 12)]
WGRAD_ONLY = len(sys.argv) > 2 and sys.argv[2] == "--wgrad"
best = {}
for bm, bn, sp, k32 in ([] if WGRAD_ONLY else [(a, b, c, d) for (a, b, c) in cfgs for d in (1, 0)]):
    env = dict(os.environ, MG_FORCE_PLAN="%d,%d,%d" % (bm, bn, sp))
    if not k32:
        env["MG_NO_BK32"] = "1"
    out = subprocess.run([sys.executable, "scripts/bench_conv.py", "--only", ",".join(shapes), "--iters", "10"] +
                         (["--f16"] if os.environ.get("MG_TUNE_F16") else []), env=env,
                         capture_output=True, text=True).stdout
    for line in out.splitlines():
        m = re.match(r"(\S+)\s+[\d.]+ GF \| fwd\s+([\d.]+) us.*?\| dgrad\s+([\d.]+) us", line)
        if not m:
            continue
        name, tf, td = m.group(1), float(m.group(2)), float(m.group(3))
        for ps, t in (("fwd", tf), ("dgrad", td)):
            k = (name, ps)
            if k not in best or t < best[k][0]:
                best[k] = (t, bm, bn, sp, k32)
            best.setdefault((name, ps, "all"), []).append((t, bm, bn, sp, k32))
    print("done", bm, bn, sp, k32, flush=True)
for k in sorted(k for k in best if len(k) == 2):
    print("%-12s %-6s best %.1f us with %dx%d split %d k32=%d" % (k[0], k[1], *best[k]))
    top = sorted(best[(k[0], k[1], "all")])[:4]
    print("    runners-up: " + "; ".join("%.1f us %dx%d s%d k32=%d" % t for t in top))


def tune_wgrad(shapes):
    """MG_FORCE_WGRAD=big,splits sweep; prints a table and the lines for csrc/wgrad_plans.inc."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench_conv import SHAPES
    best = {}
    for big in (0, 1):
        for sp in [int(v) for v in os.environ.get('MG_TUNE_SPLITS', '1,2,3,4,6,8,12,16,24,32,48').split(',')]:
            env = dict(os.environ, MG_FORCE_WGRAD="%d,%d" % (big, sp))
            out = subprocess.run([sys.executable, "scripts/bench_conv.py", "--only", ",".join(shapes), "--iters", "10"] +
                                 (["--f16"] if os.environ.get("MG_TUNE_F16") else []),
                                 env=env, capture_output=True, text=True).stdout
            for line in out.splitlines():
                m = re.match(r"(\S+)\s+[\d.]+ GF \|.*\| wgrad\s+([\d.]+) us", line)
                if m:
                    name, tw = m.group(1), float(m.group(2))
                    best.setdefault(name, []).append((tw, big, sp))
    B16 = int(os.environ.get("MG_BENCH_BATCH", 0))
    for name in sorted(best):
        top = sorted(best[name])[:3]
        B, H, W, Ci, Co, k, s, p, refl = SHAPES[name]
        B = B16 or B
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        print("%-12s wgrad: %s" % (name, "; ".join("%.1f us big=%d s%d" % t for t in top)))
        print("    {%d, %d, %d, %d, %d},   // %s%s" % (Co, k * k * Ci, B * OH * OW, top[0][1], top[0][2], name,
                                                          " (batch %d)" % B if B16 else ""))


if len(sys.argv) > 2 and sys.argv[2] == "--wgrad":
    tune_wgrad(shapes)
