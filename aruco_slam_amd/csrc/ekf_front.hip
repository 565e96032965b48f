// Fused front kernel, float covariance instantiations (see ekf_front_impl.h).
#include "ekf_front_impl.h"

template void ekf_launch_front<float>(const EkfFrame&, hipStream_t);
