set -o pipefail
tools/profile_round.sh r02_c3 --steps 100 --warmup 10 > gpurun_out/prof_c3.log 2>&1 && \
tools/profile_round.sh r02_c2 --landmarks 256 --visible 16 --cov-dtype float64 --steps 100 --warmup 10 > gpurun_out/prof_c2.log 2>&1 && \
tools/profile_round.sh r02_c5 --landmarks 4096 --visible 64 --steps 50 --warmup 5 > gpurun_out/prof_c5.log 2>&1
echo rc=$?
tail -n 3 gpurun_out/prof_c3.log gpurun_out/prof_c2.log gpurun_out/prof_c5.log
