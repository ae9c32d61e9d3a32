"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, see profiles/README.md) over
scripts/bench_conv.py into per-kernel HBM bytes per launch: (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950 correction
of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE reports half the bytes of wide coalesced reads).
    python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv> <traffic.json> [section]
Without a section the top level of traffic.json is rewritten (the configs[1] float32 trunk layer) together with "_source",
the csv the numbers came from.  With one ("configs[2] --fp16", "configs[4]", ...: the same kernels are launched on other
layer shapes there) the numbers go under traffic.json[section], which also names its csv.  bench.py copies
traffic.json[section of the run][dominant kernel] into roofline.traffic."""
import csv, json, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


import os
BY_GRID = os.environ.get("PMC_BY_GRID", "0") == "1"      # one row per (kernel, grid size): the same symbol on several layer shapes


def collect(path, counter):
    acc, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r["Kernel_Name"])
        if BY_GRID:
            k += " @grid %s" % r.get("Grid_Size")
        acc[k] += float(r["Counter_Value"])
        n[k] += 1
    return acc, n


f, fn = collect(sys.argv[1], "FETCH_SIZE")
w, wn = collect(sys.argv[2], "WRITE_SIZE")
rows, out = [], {}
for k in f:
    if k not in w or not (k.startswith(("conv_", "dense_", "wino", "dgemm32", "hgemm", "h16_", "splitk", "norm_", "adam", "mdct4_", "imdct4_"))):
        continue
    fk, wk = f[k] / fn[k], w[k] / wn[k]
    b = int((2 * fk + wk) * 1024)
    rows.append((k, fn[k], round(fk, 1), round(wk, 1), b))
    out[k] = b
with open(sys.argv[3], "w") as fh:
    fh.write("kernel,dispatches,FETCH_SIZE_KB_raw,WRITE_SIZE_KB_raw,hbm_bytes_per_launch_corrected\n")
    for r in rows:
        fh.write('"%s",%d,%s,%s,%d\n' % r)
import os
out["_source"] = "profiles/" + os.path.basename(sys.argv[3])
if len(sys.argv) > 5:
    doc = json.load(open(sys.argv[4])) if os.path.exists(sys.argv[4]) else {}
    doc[sys.argv[5]] = out
else:
    old = json.load(open(sys.argv[4])) if os.path.exists(sys.argv[4]) else {}
    doc = dict(out)
    doc.update({k: v for k, v in old.items() if isinstance(v, dict)})      # keep the sections
json.dump(doc, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
