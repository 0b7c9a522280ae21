"""Summarise a tools/profile_gpu.sh output directory into one JSON document (stdout)."""
import csv
import glob
import json
import os
import sys


def main(out):
    summary = {"kernels": {}, "counters_per_launch": {}}
    for f in glob.glob(os.path.join(out, "trace", "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Name"].split("(")[0].replace("void ", "")
            summary["kernels"][name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                        "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                        "pct": float(r["Percentage"])}
    for f in glob.glob(os.path.join(out, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            k = summary["kernels"].setdefault(name, {})
            k.update({"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                      "lds_bytes": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"]),
                      "workgroup": int(r["Workgroup_Size_X"]), "grid": int(r["Grid_Size_X"])})
    acc = {}
    for f in glob.glob(os.path.join(out, "pmc*", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "k_trace" not in name or "<1>" in name or "<1," in name:
                continue
            acc.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (name, ctr), v in sorted(acc.items()):
        summary["counters_per_launch"].setdefault(name, {})[ctr] = sum(v) / len(v)
    for name, c in summary["counters_per_launch"].items():
        d = {}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            # MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE under-reports wide streaming
            # reads by 2x (uncalibrated for other widths) -> report both the raw and the doubled read side.
            d["hbm_bytes_raw"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
            d["hbm_bytes_fetch_doubled"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c and c["SQ_ACTIVE_INST_VALU"]:
            d["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
            d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]):
            d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            d["wait_any_frac_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            d["active_valu_frac_of_wave_cycles"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
        summary.setdefault("derived", {})[name] = d
    print(json.dumps(summary, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1])
