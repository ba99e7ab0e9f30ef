"""Reads the kernel trace of tools/cold_probe.py: per phase (idle gaps > 20 ms separate them) the mean duration of every kernel name."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
phases, cur, last = [], [], None
for r in rows:
    s = int(r["Start_Timestamp"])
    if last is not None and s - last > 20e6:
        phases.append(cur); cur = []
    cur.append(r); last = int(r["End_Timestamp"])
phases.append(cur)
def short(n):
    m = re.search(r"(conv_lw_kernel<[^>]*>|conv1x1_kernel<[^>]*>|abn_\w+kernel|wgrad\w*kernel<[^>]*>|wgrad\w*kernel)", n)
    return m.group(1) if m else re.sub(r"\(.*", "", n)[:50]
for i, ph in enumerate(phases):
    agg = collections.defaultdict(list)
    for r in ph[len(ph) // 3:]:                       # the last replays of the phase
        agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    t0, t1 = int(ph[0]["Start_Timestamp"]), int(ph[-1]["End_Timestamp"])
    print("phase %d: %d launches, span %.2f ms" % (i, len(ph), (t1 - t0) / 1e6))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        print("   %8.2f us mean  %8.2f median  x%-5d %s" % (sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, len(v), k))
