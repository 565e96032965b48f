cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/la
python3 $ROOT/bench.py --cpu-frames 0 --steps 200 --warmup 10 --lookahead on > $ROOT/gpurun_out/la/bench_la.json 2>/dev/null
python3 $ROOT/bench.py --cpu-frames 0 --steps 200 --warmup 10 > $ROOT/gpurun_out/la/bench_serial.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/la/trace -- python3 $ROOT/bench.py --cpu-frames 0 --steps 100 --warmup 10 --lookahead on > $ROOT/gpurun_out/la/trace.json 2> $ROOT/gpurun_out/la/trace.err
python3 $ROOT/tools/trace_gaps.py $(ls $ROOT/gpurun_out/la/trace/*/*kernel_trace.csv | head -1) > $ROOT/gpurun_out/la/gaps.txt 2>&1
tail -30 $ROOT/gpurun_out/la/gaps.txt
