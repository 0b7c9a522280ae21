#!/bin/bash
# tools/r3_all.sh -- the round-3 measurement pass (one GPU box, ~6 minutes): bench lines of every configuration, the driver's form,
# the rehearsed ranks of the split, and the auxiliary numbers.  Output under gpurun_out/r3_final/.
O=gpurun_out/r3_final; mkdir -p $O
python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_c2_driver_form.json 2>> $O/bench_c2.err
python3 bench.py --config 4 --steps 1000 --warmup 100 > $O/bench_c4.json 2> $O/bench_c4.err
python3 bench.py --config 4synthetic --steps 1000 --warmup 100 > $O/bench_c4s.json 2> $O/bench_c4s.err
python3 bench.py --config 5 --steps 200 --warmup 20 > $O/bench_c5.json 2> $O/bench_c5.err
python3 bench.py --force-comm --steps 800 --warmup 80 > $O/comm_one_rank.json 2> $O/comm.err
for w in 2 4 8; do for r in 0 1; do
  python3 bench.py --rehearse-world $w --rehearse-rank $r --steps 800 --warmup 80 > $O/rehearse_c2_w${w}_r$r.json 2>> $O/comm.err
done; done
for r in 0 1; do python3 bench.py --config 5 --rehearse-world 8 --rehearse-rank $r --steps 200 --warmup 20 > $O/rehearse_c5_w8_r$r.json 2>> $O/comm.err; done
python3 tools/r3_measure.py > $O/measure.txt 2>&1
python3 - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r3_final/*.json")):
    try:
        d = json.loads([ln for ln in open(f) if ln.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f), "FAILED", e); continue
    extra = ""
    if "dropin_call" in d: extra += f" dropin {d['dropin_call']['ms_per_call']:.5f}/{d['dropin_call']['ms_per_call_without_update']:.5f}"
    if "frames_per_launch" in d: extra += f" fpl4 {d['frames_per_launch']['ms_per_frame']:.5f}"
    if "orbit" in d: extra += f" orbit {d['orbit']['ms_per_frame']:.5f}/{d['orbit'].get('frames_per_launch', {}).get('ms_per_frame', 0):.5f}"
    if "roofline" in d: extra += f" kernel {d['roofline']['kernel_ms_avg']:.5f} frac {d['roofline']['frac']}"
    if "single_frame_latency" in d: extra += f" latency {d['single_frame_latency']['ms_per_frame']:.5f} payload {d['single_frame_latency']['payload_bytes_per_frame_and_rank']}"
    if "cpu_baseline" in d: extra += f" cpu {d['cpu_baseline']['value']} x{d.get('speedup_vs_cpu_all_cores')}"
    print(f"{os.path.basename(f):34s} value {d['value']:10.1f} ms/step {d['ms_per_step']:.5f}{extra}")
PY
cat $O/measure.txt
