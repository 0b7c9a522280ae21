bash tools/profile_gpu.sh r02_c2 2 > gpurun_out/r2_prof_c2.log 2>&1; tail -4 gpurun_out/r2_prof_c2.log
bash tools/profile_gpu.sh r02_c4 4 --steps 500 --warmup 50 > gpurun_out/r2_prof_c4.log 2>&1; tail -4 gpurun_out/r2_prof_c4.log
bash tools/profile_gpu.sh r02_c5 5 --steps 50 --warmup 5 > gpurun_out/r2_prof_c5.log 2>&1; tail -4 gpurun_out/r2_prof_c5.log
