set -x
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2_t6.log 2>&1; tail -4 gpurun_out/r2_t6.log
python bench.py > gpurun_out/r2_b6_c2.json 2> gpurun_out/r2_b6_c2.err; tail -c 1500 gpurun_out/r2_b6_c2.json; tail -3 gpurun_out/r2_b6_c2.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_b6_c2_driver.json 2> gpurun_out/r2_b6_c2_driver.err; python -c "
import json; j=json.loads(open('gpurun_out/r2_b6_c2_driver.json').read().strip().splitlines()[-1]); print('driver-style', j['value'], j['ms_per_step'], j.get('orbit'))"
python bench.py --force-comm --cpu-frames 0 --steps 400 --warmup 40 > gpurun_out/r2_b6_comm.json 2> gpurun_out/r2_b6_comm.err; tail -c 600 gpurun_out/r2_b6_comm.json; tail -3 gpurun_out/r2_b6_comm.err
python bench.py --force-comm --cpu-frames 0 --steps 400 --warmup 40 --frames-per-gather 1 > gpurun_out/r2_b6_comm1.json 2>> gpurun_out/r2_b6_comm.err; python -c "
import json; j=json.loads(open('gpurun_out/r2_b6_comm1.json').read().strip().splitlines()[-1]); print('comm fpg1', j['value'], j['ms_per_step'])"
python bench.py --config 4 --steps 500 --warmup 50 > gpurun_out/r2_b6_c4.json 2> gpurun_out/r2_b6_c4.err; tail -c 1200 gpurun_out/r2_b6_c4.json; tail -3 gpurun_out/r2_b6_c4.err
python bench.py --config 5 --steps 50 --warmup 5 > gpurun_out/r2_b6_c5.json 2> gpurun_out/r2_b6_c5.err; tail -c 1200 gpurun_out/r2_b6_c5.json; tail -3 gpurun_out/r2_b6_c5.err
