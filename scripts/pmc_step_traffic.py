"""Whole-step HBM view: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) over an EAGER bench run, reduced per
kernel symbol to launches per step, average duration (the dispatch timestamps of the same csv), corrected HBM bytes per launch
((2 * FETCH_SIZE + WRITE_SIZE) * 1024, MI355X_MICROARCH.md) and the rate they imply -- which kernels of the step sit at the HBM
roofline, and which move far more than their time explains.
    python scripts/pmc_step_traffic.py <fetch.csv> <write.csv> <steps in the trace> <out.csv>"""
import csv, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").replace("_GLOBAL__N_1", "")
    return name.split("(")[0].strip()[:90]


def collect(path, counter):
    acc, n, dur = defaultdict(float), defaultdict(int), defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r["Kernel_Name"])
        acc[k] += float(r["Counter_Value"])
        n[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return acc, n, dur


f, fn, fd = collect(sys.argv[1], "FETCH_SIZE")
w, wn, _ = collect(sys.argv[2], "WRITE_SIZE")
steps = float(sys.argv[3])
rows = []
for k in f:
    if k not in w:
        continue
    b = (2 * f[k] / fn[k] + w[k] / wn[k]) * 1024
    us = fd[k] / fn[k]
    rows.append((b * fn[k] / steps, k, fn[k] / steps, us, b, b / (us * 1e-6) / 1e12 if us > 0 else 0.0))
rows.sort(reverse=True)
tot_b = sum(r[0] for r in rows)
tot_us = sum(r[2] * r[3] for r in rows)
with open(sys.argv[4], "w") as fh:
    fh.write("kernel,launches_per_step,avg_us_under_pmc,hbm_bytes_per_launch,TB_per_s,share_of_step_bytes\n")
    for r in rows:
        fh.write('"%s",%.1f,%.1f,%d,%.2f,%.3f\n' % (r[1], r[2], r[3], r[4], r[5], r[0] / tot_b))
    fh.write('"TOTAL per step",%.0f,%.1f,%d,%.2f,1.000\n' % (sum(r[2] for r in rows), tot_us, tot_b, tot_b / (tot_us * 1e-6) / 1e12))
print("per step: %.2f GB in %.2f ms of kernels = %.2f TB/s" % (tot_b / 1e9, tot_us / 1e3, tot_b / (tot_us * 1e-6) / 1e12))
for r in rows[:28]:
    print("%-72s %5.1f x %7.1f us  %8.1f MB  %5.2f TB/s  %4.1f%%" % (r[1][:72], r[2], r[3], r[4] / 1e6, r[5], 100 * r[0] / tot_b))
