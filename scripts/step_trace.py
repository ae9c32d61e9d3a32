"""One training step's launches in order, from a rocprofv3 --kernel-trace CSV (not the --stats summary, which mixes start-up work
into per-step averages): finds the LAST complete step -- the launches between the second-to-last and the last pair of Adam
kernels -- and prints every launch (start offset, duration, gap to the previous launch's end, kernel name), then per-symbol
totals and the share of launches under 12 us.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --config 2 --fp16 --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-roofline
    python scripts/step_trace.py /tmp/kt/**/kt_kernel_trace.csv [--adam adam_dev_kernel] [--per-step 2] > gpurun_out/step_trace.txt"""
import argparse
import collections
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--adam", default="adam_")          # substring of the optimiser kernels that end a backward pass
ap.add_argument("--per-step", type=int, default=2)  # Adam launches per step (G and D)
ap.add_argument("--quiet", action="store_true")
a = ap.parse_args()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if m:
        k = int(m.group(1))
        rest = n[len(m.group(0)):]
        return rest[:k]
    return n.split("(")[0][:110]


rows = []
for r in csv.DictReader(open(a.csv)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
is_adam = [i for i, r in enumerate(rows) if a.adam in r[2] and "tick" not in r[2] and "prime" not in r[2]]
assert len(is_adam) >= 2 * a.per_step, "not enough optimiser launches in the trace"
end = is_adam[-1]
begin = is_adam[-1 - a.per_step] + 1
step = rows[begin:end + 1]
t0 = step[0][0]
tot = collections.OrderedDict()
prev_end = t0
small_n = small_t = 0
for s, e, n in step:
    d = (e - s) / 1000.0
    k = short(n)
    if not a.quiet:
        print("%9.1f us  %8.1f us  gap %6.1f  %s" % ((s - t0) / 1000.0, d, (s - prev_end) / 1000.0, k))
    prev_end = e
    c = tot.setdefault(k, [0, 0.0])
    c[0] += 1
    c[1] += d
    if d < 12.0:
        small_n += 1
        small_t += d
print("# step: %d launches, %.1f us of kernel time, %.1f us wall; %d launches under 12 us = %.1f us"
      % (len(step), sum(v[1] for v in tot.values()), (step[-1][1] - t0) / 1000.0, small_n, small_t))
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("# %8.1f us %4d x %7.1f us  %s" % (t, c, t / c, k))
