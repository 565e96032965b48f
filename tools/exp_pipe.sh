set -o pipefail
mkdir -p gpurun_out/exp1
for v in "" "EKF_LA_LATE=1" "EKF_PIPE_WPE=4" "EKF_PIPE_WPE=3" "EKF_PIPE_WPE=2" "EKF_LA_LATE=1 EKF_PIPE_WPE=4" "EKF_LA_LATE=1 EKF_PIPE_WPE=3" "EKF_LA_LDS_KB=0" ; do
  env $v timeout -k 10 120 python tools/pipe_probe.py 2>&1 | grep "n=" | tee -a gpurun_out/exp1/log.txt || exit 1
done
