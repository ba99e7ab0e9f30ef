"""Average a rocprofv3 --pmc counter per kernel: pmc_summary.py <counter_collection.csv> <COUNTER>"""
import collections, csv, re, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]:
        continue
    n = r["Kernel_Name"]
    m = re.search(r"(pixcon16_\w+kernel|pixcon_\w+kernel|abn_\w+kernel|reduce_bands_kernel|seg_losses\w*kernel|gather_normalize_kernel)", n)
    if not m:
        continue
    agg[(m.group(1), r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    v = sorted(v)
    print("%-28s grid %-10s n=%3d  %s median %.4g  min %.4g max %.4g" % (k[0], k[1], len(v), sys.argv[2], v[len(v) // 2], v[0], v[-1]))
