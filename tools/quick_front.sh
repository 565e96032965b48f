#!/bin/bash
# Quick look at a front-kernel change on the GPU box: in-kernel stamps (pipelined: all / role level only; serial), us per
# frame at the headline, and the bitwise fused-vs-stage / pipelined-vs-serial parity tests.   tools/quick_front.sh <tag>
set -o pipefail
tag=${1:-qf}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 120 python tools/front_stamps.py seq > $out/stamps_seq.log 2>&1 || exit 1
timeout -k 10 120 python tools/front_stamps.py seq light > $out/stamps_seq_light.log 2>&1 || exit 1
timeout -k 10 120 python tools/front_stamps.py > $out/stamps_serial.log 2>&1 || exit 1
timeout -k 10 200 python tools/pipe_probe.py 1024 32 2000 > $out/pipe_c3.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "bitwise or fused or pipelined or intermediates or k192 or g2_teacher" > $out/pytest.log 2>&1
tail -3 $out/pytest.log; grep -v amdgpu.ids $out/stamps_seq.log; grep -E "factor:|X_b in LDS|S-block|step 5|W/dx" $out/stamps_seq_light.log; cat $out/pipe_c3.log | grep us/frame
