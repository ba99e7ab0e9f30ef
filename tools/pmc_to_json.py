"""Fold the two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE; separate runs) into profiles/rNN_pmc_bench.json:
HBM bytes per library call of the contrastive loss (sweep 1 + sweep 2 + finalize + reduce) and of the ABN kernels.
Units/corrections per MI355X_MICROARCH.md: both counters are in KiB; on gfx950 FETCH_SIZE counts at half rate (x2).
usage: pmc_to_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <global_batch>"""
import collections, csv, json, re, sys

def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(pixcon16p_\w+kernel|pixcon16_\w+kernel|pixcon_\w+kernel|abn_\w+kernel|reduce_bands_kernel|seg_losses\w*kernel|conv1x1_kernel|conv_lw_kernel|window_\w+kernel|tile_stats_reduce_kernel|wgrad3_kernel|wgrad_kernel|wgrad_sum_kernel|stem_\w+kernel|sgd_step_dev_kernel|sgd_step_kernel)", r["Kernel_Name"])
        if m:
            k = m.group(1)
            if k == "conv1x1_kernel" and re.search(r"conv1x1_kernel<\d+, \w+, \d+, true", r["Kernel_Name"]):
                k = "conv3x3_kernel"          # the CONV3 instances of the same template: the 3x3 implicit GEMM
            if k == "conv_lw_kernel":         # the loader-wave forms (round 4) of the same products: <BM, BN, OUT, CONV3, BK, NST>
                k = "conv3x3_kernel" if re.search(r"conv_lw_kernel<\d+, \d+, \d+, true", r["Kernel_Name"]) else "conv1x1_kernel"
            if k == "wgrad3_kernel":          # the three-tap form of the 3x3 weight gradient (round 4)
                k = "wgrad3x3_kernel"
            if k == "sgd_step_dev_kernel":
                k = "sgd_step_kernel"
            if k in ("abn_apply_fast_kernel", "abn_bwd_apply_fast_kernel", "abn_bwd_reduce_fast_kernel"):   # round 4: the packed-math forms
                k = k.replace("_fast", "")
            if k == "wgrad_kernel":
                k = "wgrad3x3_kernel" if re.search(r"wgrad_kernel<\d+, \d+, true", r["Kernel_Name"]) else "wgrad1x1_kernel"
            agg[k].append(float(r["Counter_Value"]))
    return agg

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_note": "bytes = FETCH_SIZE[KiB]*1024*2 + WRITE_SIZE[KiB]*1024, mean per dispatch; pixcon_loss = sum over its kernels per call"}
kern = {}
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [])))
    w = sum(write.get(k, [0])) / max(1, len(write.get(k, [])))
    kern[k] = {"dispatches_seen": len(fetch.get(k, [])), "fetch_bytes": f * 1024 * 2, "write_bytes": w * 1024, "bytes": f * 2048 + w * 1024}
out["kernels"] = kern
calls = [k for k in kern if k.startswith("pixcon")]
if calls:
    n = max(1, min(kern[k]["dispatches_seen"] for k in calls if "neg" in k or "pos" in k or "plan" in k))
    out["ucd_pixcon_loss"] = {"global_batch": int(sys.argv[4]), "bytes_per_launch": sum(kern[k]["bytes"] * kern[k]["dispatches_seen"] / n for k in calls),
                              "kernels": calls}
# HBM-stream calls: one dominant kernel each (the stage-2 reduce_bands launches move a few KB)
for call, k in (("ucd_abn_apply", "abn_apply_kernel"), ("ucd_abn_stats", "abn_stats_kernel"),
                ("ucd_abn_bwd_reduce", "abn_bwd_reduce_kernel"), ("ucd_abn_bwd_apply", "abn_bwd_apply_kernel"),
                ("ucd_seg_losses", "seg_losses_kernel"), ("ucd_conv1x1", "conv1x1_kernel"), ("ucd_conv3x3", "conv3x3_kernel"),
                ("ucd_conv1x1_wgrad", "wgrad1x1_kernel"), ("ucd_conv3x3_wgrad", "wgrad3x3_kernel"),
                ("ucd_stem_apply_pool", "stem_apply_pool_kernel"), ("ucd_sgd_step", "sgd_step_kernel")):
    if k in kern:
        out[call] = {"global_batch": int(sys.argv[4]), "bytes_per_launch": kern[k]["bytes"], "kernels": [k]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"})[:600])
for k, v in kern.items():
    print("%-28s n=%4d fetch %.3e B write %.3e B" % (k, v["dispatches_seen"], v["fetch_bytes"], v["write_bytes"]))
