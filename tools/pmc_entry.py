#!/usr/bin/env python3
"""tools/pmc_entry.py SUMMARY.json CONFIG KERNEL ORDER SOURCE [FRAMES_PER_LAUNCH] -- merge the counters of one tools/profile_gpu.sh run
into profiles/pmc_counters.json (what bench.py reads for roofline.achieved / traffic; keyed by config, kernel and
launch order, stamped with the commit and a hash of the device sources the counters were measured on)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_source_hash() -> str:
    h = hashlib.sha256()
    for rel in ("ray_tracing_octrees_amd/csrc/rto_device.hip.h", "ray_tracing_octrees_amd/csrc/rto_api.hip"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    summary_path, config, kernel, order, source = sys.argv[1:6]
    fpl = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    s = json.load(open(summary_path))
    name = next(k for k in s["counters_per_launch"] if (kernel + "<0") in k or (kernel + "<false>") in k or (kernel + "(") in k)     # exact kernel, colour mode
    c = s["counters_per_launch"][name]
    entry = {
        "config": config, "kernel": kernel, "order": order, "kernel_symbol": name, "frames_per_launch": fpl,
        "SQ_INSTS_VALU": int(round(c["SQ_INSTS_VALU"])), "SQ_THREAD_CYCLES_VALU": int(round(c["SQ_THREAD_CYCLES_VALU"])),
        "SQ_WAVES": int(round(c["SQ_WAVES"])),
        "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
        "hbm_bytes_per_launch": int(round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)),
        "rocprof_avg_us": s["kernels"].get(name, {}).get("avg_us"),
        "source": source,
        "commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
        "device_source_hash": device_source_hash(),
        "note": "rocprofv3 --pmc, separate passes (counters only), 12 plain-launch frames each; FETCH_SIZE/WRITE_SIZE in KiB, read side "
                "doubled per MI355X_MICROARCH.md (an upper bound for 8-byte gathers)",
    }
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    table = json.load(open(path)) if os.path.exists(path) else {"entries": []}
    table["entries"] = [e for e in table["entries"] if (e["config"], e["kernel"], e["order"]) != (config, kernel, order)] + [entry]
    table["entries"].sort(key=lambda e: (e["config"], e["kernel"], e["order"]))
    json.dump(table, open(path, "w"), indent=1)
    print(json.dumps(entry))


if __name__ == "__main__":
    main()
