// Fused front kernel, double covariance instantiations (see ekf_front_impl.h).
#include "ekf_front_impl.h"

template void ekf_launch_front<double>(const EkfFrame&, hipStream_t);
