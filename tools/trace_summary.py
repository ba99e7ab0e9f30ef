"""Summarise a rocprofv3 --kernel-trace CSV of bench.py into a small per-kernel table of the LAST two steps
(step boundaries = pixcon_reduce_kernel launches).  usage: trace_summary.py <kernel_trace.csv> <out.txt> [title]"""
import collections, csv, os, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pixcon_reduce_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-1]
win = rows[a:b]
t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
def short(n):
    if "ucd" in n and ("N_1" in n or "ucd::" in n):
        m = re.search(r"(pixcon16_\w+kernel|pixcon_\w+kernel|abn_\w+kernel|reduce_bands_kernel|plane_sum_kernel|prep_\w+kernel|"
                      r"gather_normalize_kernel|scatter_grad_kernel|seg_losses\w*kernel|attmap\w+|conv1x1_kernel<[^>]*>|conv_lw_kernel<[^>]*>|wgrad3_kernel<[^>]*>|wgrad_kernel<[^>]*>|wgrad_sum_kernel<[^>]*>|sgd_step_dev_kernel|conv1x1_wgrad_kernel|"
                      r"tile_stats_reduce_kernel|window_\w+kernel|flip_weights_kernel|transpose_bf16_kernel|wgrad_reduce_kernel|sgd_step_kernel)", n)
        t = "<bf16>" if "bfloat16" in n else ("<f32>" if "<float" in n or "IfE" in n else "")
        return "UCD   " + (m.group(1) if m else n[:60]) + t
    if "at::native" in n:
        m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", n)
        m2 = re.findall(r"native::(\w+?)(?:_kernel_cuda|Functor|_kernel)", n)
        return "ATEN  " + m.group(1) + " " + " ".join(m2[1:3]) + (" bf16" if "BFloat16" in n else "")
    if n.startswith("igemm") or "ck::" in n or "Cijk" in n or "naive_conv" in n or "conv" in n.lower() or "gemm" in n.lower() or "SubTensorOp" in n:
        return "CONV  " + re.sub(r"\(.*", "", n)[:70]
    return "OTHER " + re.sub(r"\(.*", "", n)[:60]
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    k = short(r["Kernel_Name"])
    agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
grp = collections.defaultdict(float)
for k, (d, c) in agg.items():
    grp[k.split()[0]] += d / 2e6
# GPU busy time = union of the kernel intervals (the teacher's stream overlaps the student's); idle = gaps between kernels
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s_, e_ in iv[1:]:
    if s_ > cur_e:
        busy += cur_e - cur_s; gaps.append(s_ - cur_e); cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
# the largest gaps with the kernels on either side (where does the GPU wait for the host?)
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in win), key=lambda t: t[0])
big, reach, last = [], ev[0][1], ev[0][2]
for s_, e_, n_ in ev[1:]:
    if s_ > reach:
        big.append((s_ - reach, last, n_))
    if e_ > reach:
        reach, last = e_, n_
big.sort(reverse=True)
gaps.sort()
with open(sys.argv[2], "w") as f:
    f.write("# %s\n" % (sys.argv[3] if len(sys.argv) > 3 else ""))
    f.write("# GPU busy (union of kernel intervals) %.3f ms/step, idle between kernels %.3f ms/step in %d gaps (median %.1f us, "
            "p90 %.1f us)\n" % (busy / 2e6, sum(gaps) / 2e6, len(gaps) // 2, gaps[len(gaps) // 2] / 1e3 if gaps else 0,
                                gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0))
    f.write("# last two timed steps: wall %.3f ms/step, sum of kernel durations %.3f ms/step, %d launches/step\n" %
            ((t1 - t0) / 2e6, sum(v[0] for v in agg.values()) / 2e6, len(win) // 2))
    f.write("# by group (ms/step): " + ", ".join("%s %.2f" % kv for kv in sorted(grp.items(), key=lambda kv: -kv[1])) + "\n")
    f.write("%10s %10s %10s  %s\n" % ("ms/step", "calls/step", "avg_us", "kernel"))
    for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("UCD_TRACE_TOP", "70"))]:
        f.write("%10.3f %10.1f %10.2f  %s\n" % (d / 2e6, c / 2, d / c / 1e3, k))
    # the HBM-stream ABN kernels by launch geometry (= by layer shape): where the small layers sit against the large ones
    bygrid = collections.defaultdict(lambda: [0, 0])
    for r in win:
        k = short(r["Kernel_Name"])
        if any(n in k for n in ("abn_apply_kernel", "abn_bwd_apply_kernel", "abn_bwd_reduce_kernel", "abn_apply_fast_kernel",
                                "abn_bwd_apply_fast_kernel", "abn_bwd_reduce_fast_kernel")):
            key = (k.split()[1], "x".join(str(r[c]) for c in sorted(r) if c.startswith("Grid_Size")))
            bygrid[key][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); bygrid[key][1] += 1
    f.write("# ABN stream kernels by grid size (threads): ms/step, calls/step, avg us\n")
    for (k, g_), (d, c) in sorted(bygrid.items(), key=lambda kv: -kv[1][0])[:24]:
        f.write("#   %8.3f %6.1f %8.2f  %-28s grid %s\n" % (d / 2e6, c / 2, d / c / 1e3, k, g_))
    # the GEMM / implicit-GEMM / weight-gradient kernels by launch geometry (= by layer shape; grid in threads)
    bygrid = collections.defaultdict(lambda: [0, 0])
    for r in win:
        k = short(r["Kernel_Name"])
        if any(n in k for n in ("conv1x1_kernel", "conv_lw_kernel", "wgrad3_kernel", "wgrad_kernel")):
            key = (k.split(None, 1)[1], "x".join(str(r[c]) for c in sorted(r) if c.startswith("Grid_Size")))
            bygrid[key][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); bygrid[key][1] += 1
    f.write("# GEMM kernels by grid size (threads): ms/step, calls/step, avg us\n")
    for (k, g_), (d, c) in sorted(bygrid.items(), key=lambda kv: -kv[1][0])[:48]:
        f.write("#   %8.3f %6.1f %8.2f  %-60s grid %s\n" % (d / 2e6, c / 2, d / c / 1e3, k, g_))
    f.write("# largest idle gaps of the two steps (us: after kernel -> before kernel)\n")
    for g_, a_, b_ in big[:14]:
        f.write("#   %8.1f  %s  ->  %s\n" % (g_ / 1e3, a_[:60], b_[:60]))
print(open(sys.argv[2]).read()[:14000])
