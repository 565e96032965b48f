#!/bin/bash
# rocprofv3 evidence for the four bench configurations of this round (GPU box):  tools/profile_all.sh
set -o pipefail
tools/profile_round.sh r03_c3 --steps 100 --warmup 10 > gpurun_out/prof_c3.log 2>&1 && \
tools/profile_round.sh r03_c2 --landmarks 256 --visible 16 --cov-dtype float64 --steps 100 --warmup 10 > gpurun_out/prof_c2.log 2>&1 && \
tools/profile_round.sh r03_c5 --landmarks 4096 --visible 64 --steps 50 --warmup 5 > gpurun_out/prof_c5.log 2>&1 && \
tools/profile_round.sh r03_rot --filter ekf_rotations --steps 100 --warmup 10 > gpurun_out/prof_rot.log 2>&1
echo rc=$?
