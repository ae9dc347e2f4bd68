"""Summarise the rocprofv3 --pmc passes of scripts/gpu_pmc.sh for the sample-loop kernel: per-launch averages, the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (x2 for 16-B-per-lane streaming reads), and profiles/traffic_latest.json."""
import csv, glob, json, os, sys
out = sys.argv[1]
KERNEL = "wnv_ring_kernel"
vals = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if KERNEL not in row.get("Kernel_Name", ""):
                continue
            vals.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            vals[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
res = {}
for name, d in sorted(vals.items()):
    per = sorted(d.values())
    res[name] = {"launches": len(per), "mean_per_launch": sum(per) / len(per), "last_launch": list(d.values())[-1]}
    print(f"{name:24s} launches {len(per)}  mean/launch {sum(per)/len(per):.6g}  min {per[0]:.6g}  max {per[-1]:.6g}")
summary = {"kernel": KERNEL, "counters": res}
if "FETCH_SIZE" in res:
    kb = res["FETCH_SIZE"]["mean_per_launch"]
    summary["fetch_bytes_reported"] = kb * 1024
    summary["fetch_bytes_corrected_x2"] = kb * 1024 * 2        # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
if "WRITE_SIZE" in res:
    summary["write_bytes_reported"] = res["WRITE_SIZE"]["mean_per_launch"] * 1024
if "FETCH_SIZE" in res:
    summary["hbm_bytes_per_launch"] = summary["fetch_bytes_corrected_x2"] + summary.get("write_bytes_reported", 0.0)
if "TCC_HIT_sum" in res and "TCC_MISS_sum" in res:
    h, m = res["TCC_HIT_sum"]["mean_per_launch"], res["TCC_MISS_sum"]["mean_per_launch"]
    summary["l2_hit_rate"] = h / (h + m) if h + m else None
# ---- the issue side (round 6): shares of WAVE time.  SQ_WAVE_CYCLES = cycles summed over resident waves (a persistent workgroup holds 8
#      waves for the whole launch); SQ_WAIT_INST_ANY = of those, cycles a wave spent waiting for an instruction to issue or complete
#      (s_waitcnt, s_barrier, s_sleep and the polls' round trips all land here); SQ_ACTIVE_INST_VALU = cycles the VALU worked for a wave.
def m(n):
    return res[n]["mean_per_launch"] if n in res else None
wc = m("SQ_WAVE_CYCLES")
if wc:
    issue = {}
    if m("SQ_WAIT_INST_ANY") is not None:
        issue["wave_time_waiting_share"] = m("SQ_WAIT_INST_ANY") / wc
    if m("SQ_ACTIVE_INST_ANY") is not None:
        issue["wave_time_issuing_share"] = m("SQ_ACTIVE_INST_ANY") / wc
    if m("SQ_ACTIVE_INST_VALU") is not None:
        issue["wave_time_valu_busy_share"] = m("SQ_ACTIVE_INST_VALU") / wc
    if m("SQ_INSTS_VALU") and m("SQ_INSTS_VMEM_RD") is not None:
        issue["valu_insts_per_vmem_read"] = m("SQ_INSTS_VALU") / max(m("SQ_INSTS_VMEM_RD"), 1.0)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVES"):
        if m(k) is not None:
            issue[k.lower() + "_per_launch"] = m(k)
    if m("SQ_LDS_IDX_ACTIVE") is not None and m("SQ_BUSY_CYCLES"):
        issue["lds_active_over_sq_busy"] = m("SQ_LDS_IDX_ACTIVE") / m("SQ_BUSY_CYCLES")
    summary["issue_side"] = issue
print(json.dumps(summary, indent=1))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(summary, open(os.path.join(out, "traffic.json"), "w"), indent=1)
