"""Reduce one rocprofv3 --pmc pass of SQ counters (GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS; its own run, --kernel-trace only) over scripts/bench_conv.py into a
per-kernel table:
  kernel cycles      GRBM_GUI_ACTIVE / 8              (the counter sums the 8 XCDs)
  MFMA busy          SQ_VALU_MFMA_BUSY_CYCLES / 1024  per SIMD (256 CUs x 4), and as a fraction of the kernel cycles
  waves per SIMD     SQ_WAVES / 1024
  wave residency     SQ_WAVE_CYCLES / SQ_WAVES / kernel cycles (SQ_WAVE_CYCLES counts in quad-cycles: x4)
  waiting            SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, of which on LDS SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
    python scripts/pmc_sq.py <counter_collection.csv> <out.csv>"""
import csv, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
rows = []
for k, c in acc.items():
    if "GRBM_GUI_ACTIVE" not in c or not k.startswith(("dgemm32", "conv_", "hgemm", "dense_", "mdct4_", "imdct4_")):
        continue
    n = cnt[k]["GRBM_GUI_ACTIVE"]
    cyc = c["GRBM_GUI_ACTIVE"] / n / 8.0
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n / 1024.0
    waves = c.get("SQ_WAVES", 0.0) / n
    wave_cyc = c.get("SQ_WAVE_CYCLES", 0.0) / n
    rows.append((k, n, round(cyc), round(mfma), round(mfma / cyc, 3) if cyc else 0, round(waves / 1024.0, 2),
                 round(4.0 * wave_cyc / waves / cyc, 3) if waves and cyc else 0,
                 round(c.get("SQ_WAIT_INST_ANY", 0.0) / n / wave_cyc, 3) if wave_cyc else 0,
                 round(c.get("SQ_WAIT_INST_LDS", 0.0) / n / wave_cyc, 3) if wave_cyc else 0,
                 round(c.get("SQ_WAIT_ANY", 0.0) / n / wave_cyc, 3) if wave_cyc else 0,
                 round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / n / wave_cyc, 3) if wave_cyc else 0,
                 round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / n / wave_cyc, 3) if wave_cyc else 0))
rows.sort(key=lambda r: -r[2] * r[1])
with open(sys.argv[2], "w") as fh:
    fh.write("kernel,dispatches,kernel_cycles,mfma_busy_cycles_per_simd,mfma_busy_frac,waves_per_simd,wave_residency_frac,"
             "wait_inst_any_frac_of_wave_cycles,wait_lds_frac_of_wave_cycles,wait_any_parked_frac,active_valu_frac,active_any_frac\n")
    for r in rows:
        fh.write('"%s",%s\n' % (r[0], ",".join(str(x) for x in r[1:])))
for r in rows:
    print(r)
