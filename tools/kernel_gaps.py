"""idle time between the kernels of a step, from a rocprofv3 --kernel-trace csv: per consecutive pair of the steady-state steps,
start(next) - end(previous). usage: kernel_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the steady part: from the first k_project after 200 kernels on
k = [(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("tfl::", "").replace("kz1::", "").replace("kz2::", "").split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
k = [x for x in k if x[0].startswith("k_")]
gaps = collections.defaultdict(list)
busy = 0
for a, b in zip(k[200:-50], k[201:-49]):
    g = b[1] - a[2]
    if g < 200000:      # (not the pause between two timed blocks)
        gaps[(a[0][:28], b[0][:28])].append(g)
tot = 0.0
for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:14]:
    m = sorted(v)[len(v) // 2]
    print("%-30s -> %-30s n %4d  median gap %6.2f us  mean %6.2f" % (a, b, len(v), m / 1e3, sum(v) / len(v) / 1e3))
    tot += m / 1e3
print("sum of the median gaps of the listed pairs: %.1f us" % tot)
