// Covariance update  P <- P + Q - W^T W   (the dominant kernel of the path).
//
// Reference: P <- (I - K H) P after P += Q (extended_kalman_filter.py:95-105,
// 155-156), a dense N^3 GEMM there.  With S = L L^T and W = L^-1 H (P+Q)
// (k x N, k = 3 m) the same update is the symmetric rank-k downdate below:
// 2 N^2 k flops, P read once and written once (2 N^2 sizeof(T) bytes).
//
// W is a k-major panel [kpad][ldw]; rows >= k and columns >= N are zero, so
// the padded part of P stays exactly zero without any masking.
//
// Bitwise symmetry: element (i,j) and (j,i) run the same k-ordered fma chain
// on commuting products, so P stays exactly symmetric.
//
//  * VALU kernel  : plain LDS-tiled reference implementation (any T).
//  * MFMA f32     : v_mfma_f32_32x32x2_f32, 128x128 tile / 4 waves, each wave a
//                   32 x 128 slab = four 32x32 accumulators that share the A
//                   operand.  Column c of sub-tile t is tile column 4c+t, so a
//                   lane's four accumulators of one row are 16 contiguous
//                   bytes of P (dwordx4 loads/stores) and its four B operands
//                   are one ds_read_b128.
//  * MFMA f64     : v_mfma_f64_16x16x4_f64, 128x64 tile / 4 waves, each wave
//                   32 x 64 = 2x4 accumulators; column c of sub-tile t is tile
//                   column 32(t>>1) + 2c + (t&1)  (16-byte accesses again).
#include "ekf_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// --------------------------------------------------------------------------
// VALU reference kernel: 64x64 tile, 256 threads, 4x4 per thread
// --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ekf_cov_update_valu(EkfFrame fr) {
    __shared__ T sa[16][64];
    __shared__ T sb[16][64];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const T* __restrict__ wp = static_cast<const T*>(fr.wpanel);
    T acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = (T)0;
    for (int kc = 0; kc < fr.kpad; kc += 16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = tid + 256 * t, r = e >> 6, cc = e & 63;
            sa[r][cc] = wp[(int64_t)(kc + r) * fr.ldw + i0 + cc];
            sb[r][cc] = wp[(int64_t)(kc + r) * fr.ldw + j0 + cc];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            T a[4], b[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) a[x] = -sa[r][ty * 4 + x];
#pragma unroll
            for (int y = 0; y < 4; ++y) b[y] = sb[r][tx * 4 + y];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_fma(a[x], b[y], acc[x][y]);
        }
        __syncthreads();
    }
    T* __restrict__ P = static_cast<T*>(fr.cov);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int row = i0 + ty * 4 + x;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int col = j0 + tx * 4 + y;
            T v = P[(int64_t)row * fr.ld + col];
            if (row == col) v += (T)ekf_qdiag(row, fr.dims, fr.nz);
            P[(int64_t)row * fr.ld + col] = v + acc[x][y];
        }
    }
}

// --------------------------------------------------------------------------
// MFMA f32: 128x128 tile, K chunks of 32 through LDS
// --------------------------------------------------------------------------
#define F32_KC 32
__global__ __launch_bounds__(256, 2) void ekf_cov_update_mfma_f32(EkfFrame fr) {
    __shared__ __attribute__((aligned(16))) float sA[F32_KC][128];
    __shared__ __attribute__((aligned(16))) float sB[F32_KC][128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int i0 = blockIdx.y * 128, j0 = blockIdx.x * 128;
    const float* __restrict__ wp = static_cast<const float*>(fr.wpanel);
    float* __restrict__ P = static_cast<float*>(fr.cov);
    const int64_t ld = fr.ld, ldw = fr.ldw;

    // P slab of this wave: rows i0+32*wave+rowmap(reg,lhi), 16 B per lane per row
    float4 pt[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = i0 + 32 * wave + (reg & 3) + 8 * (reg >> 2) + 4 * lhi;
        pt[reg] = *reinterpret_cast<const float4*>(P + (int64_t)row * ld + j0 + 4 * l31);
    }
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    for (int kc = 0; kc < fr.kpad; kc += F32_KC) {
        // stage W[kc:kc+32][i0:i0+128] and [j0:j0+128]; 4 float4 per thread per panel
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = tid + 256 * t, r = e >> 5, c4 = e & 31;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (kc + r < fr.kpad) {
                va = *reinterpret_cast<const float4*>(wp + (int64_t)(kc + r) * ldw + i0 + 4 * c4);
                vb = *reinterpret_cast<const float4*>(wp + (int64_t)(kc + r) * ldw + j0 + 4 * c4);
            }
            *reinterpret_cast<float4*>(&sA[r][4 * c4]) = va;
            *reinterpret_cast<float4*>(&sB[r][4 * c4]) = vb;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < F32_KC / 2; ++kk) {
            // A[i = l31][k = lhi] = -W[k][i0 + 32 wave + l31];  B[k = lhi][j = l31] = W[k][j0 + 4 l31 + t]
            const float a = -sA[2 * kk + lhi][32 * wave + l31];
            const float4 b = *reinterpret_cast<const float4*>(&sB[2 * kk + lhi][4 * l31]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.w, acc[3], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = i0 + 32 * wave + (reg & 3) + 8 * (reg >> 2) + 4 * lhi;
        const int col = j0 + 4 * l31;
        float4 v = pt[reg];
        if (row >= col && row < col + 4) {
            const float q = (float)ekf_qdiag(row, fr.dims, fr.nz);
            if (row == col) v.x += q;
            else if (row == col + 1) v.y += q;
            else if (row == col + 2) v.z += q;
            else v.w += q;
        }
        v.x += acc[0][reg];
        v.y += acc[1][reg];
        v.z += acc[2][reg];
        v.w += acc[3][reg];
        *reinterpret_cast<float4*>(P + (int64_t)row * ld + col) = v;
    }
}

// --------------------------------------------------------------------------
// MFMA f64: 128 (rows) x 64 (cols) tile, K chunks of 16 through LDS
// --------------------------------------------------------------------------
#define F64_KC 16
__global__ __launch_bounds__(256, 2) void ekf_cov_update_mfma_f64(EkfFrame fr) {
    __shared__ __attribute__((aligned(16))) double sA[F64_KC][128];
    __shared__ __attribute__((aligned(16))) double sB[F64_KC][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int i0 = blockIdx.y * 128, j0 = blockIdx.x * 64;
    const double* __restrict__ wp = static_cast<const double*>(fr.wpanel);
    double* __restrict__ P = static_cast<double*>(fr.cov);
    const int64_t ld = fr.ld, ldw = fr.ldw;

    double2 pt[2][4][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = i0 + 32 * wave + 16 * rb + lg + 4 * reg;
#pragma unroll
            for (int p = 0; p < 2; ++p)
                pt[rb][reg][p] = *reinterpret_cast<const double2*>(
                    P + (int64_t)row * ld + j0 + 32 * p + 2 * l15);
        }
    f64x4 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rb][t][r] = 0.0;

    for (int kc = 0; kc < fr.kpad; kc += F64_KC) {   // kpad is a multiple of 16
#pragma unroll
        for (int t = 0; t < 4; ++t) {                // sA: 16 x 128 doubles = 1024 double2
            const int e = tid + 256 * t, r = e >> 6, c2 = e & 63;
            *reinterpret_cast<double2*>(&sA[r][2 * c2]) =
                *reinterpret_cast<const double2*>(wp + (int64_t)(kc + r) * ldw + i0 + 2 * c2);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {                // sB: 16 x 64 doubles = 512 double2
            const int e = tid + 256 * t, r = e >> 5, c2 = e & 31;
            *reinterpret_cast<double2*>(&sB[r][2 * c2]) =
                *reinterpret_cast<const double2*>(wp + (int64_t)(kc + r) * ldw + j0 + 2 * c2);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < F64_KC / 4; ++ks) {
            const int kr = 4 * ks + lg;
            // A[i = l15][k = lg] = -W[k][i0 + 32 wave + 16 rb + l15]
            const double a0 = -sA[kr][32 * wave + l15];
            const double a1 = -sA[kr][32 * wave + 16 + l15];
            // B[k = lg][j = l15] = W[k][j0 + 32 (t>>1) + 2 l15 + (t&1)]
            const double2 b01 = *reinterpret_cast<const double2*>(&sB[kr][2 * l15]);
            const double2 b23 = *reinterpret_cast<const double2*>(&sB[kr][32 + 2 * l15]);
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b01.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b01.y, acc[0][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b23.x, acc[0][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b23.y, acc[0][3], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b01.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b01.y, acc[1][1], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b23.x, acc[1][2], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b23.y, acc[1][3], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = i0 + 32 * wave + 16 * rb + lg + 4 * reg;   // f64 C/D: row = (lane>>4) + 4 reg
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int col = j0 + 32 * p + 2 * l15;
                double2 v = pt[rb][reg][p];
                if (row == col) v.x += ekf_qdiag(row, fr.dims, fr.nz);
                else if (row == col + 1) v.y += ekf_qdiag(row, fr.dims, fr.nz);
                v.x += acc[rb][2 * p][reg];
                v.y += acc[rb][2 * p + 1][reg];
                *reinterpret_cast<double2*>(P + (int64_t)row * ld + col) = v;
            }
        }
}

template <>
void ekf_launch_cov_update<float>(const EkfFrame& fr, int variant, hipStream_t s) {
    if (variant == 1) {
        hipLaunchKernelGGL(ekf_cov_update_valu<float>, dim3(fr.ncols / 64, fr.ncols / 64), dim3(256),
                           0, s, fr);
    } else {
        hipLaunchKernelGGL(ekf_cov_update_mfma_f32, dim3(fr.ncols / 128, fr.ncols / 128), dim3(256),
                           0, s, fr);
    }
}
template <>
void ekf_launch_cov_update<double>(const EkfFrame& fr, int variant, hipStream_t s) {
    if (variant == 1) {
        hipLaunchKernelGGL(ekf_cov_update_valu<double>, dim3(fr.ncols / 64, fr.ncols / 64),
                           dim3(256), 0, s, fr);
    } else {
        hipLaunchKernelGGL(ekf_cov_update_mfma_f64, dim3(fr.ncols / 64, fr.ncols / 128), dim3(256),
                           0, s, fr);
    }
}
