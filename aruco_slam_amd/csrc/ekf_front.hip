// Fused "front" kernel of the EKF update (gfx950): measurement model, A = H (P+Q),
// S = Hs (P+Q) Hs^T + R, blocked Cholesky, W = L^-1 A, dx = W^T y and the state injection in ONE
// launch (reference: extended_kalman_filter.py:107-152; ekf_with_rotations.py:115-177).
//
// Workgroup roles by blockIdx.x (a workgroup only ever waits for LOWER-indexed workgroups, which
// the dispatcher starts first, so the waits cannot deadlock whatever the residency):
//   [0, nS)        one 16x16 block of S each, straight from P (nS = nb (nb+1) / 2)
//   nS             the factorisation: streams the S blocks into LDS as they appear, runs the
//                  pivot chain, publishes -L / Dinv / y block column by block column
//   (nS, nS+nch]   one chunk of 64 columns each: A chunk into LDS, then the right-looking blocked
//                  forward substitution in registers, consuming block column q of the factor as soon
//                  as it is published; W chunk, dx chunk; the LAST chunk to finish injects dx into
//                  the state (every reader of the old state has passed by then)
// Exchange between workgroups: ekf_solve_device.h (agent-scope relaxed accesses + sentinel values;
// no flags, no fences, no cache flushes).  The factor buffers are double-buffered across frames
// and re-armed with sentinels by the S-block workgroups of the following frame; the S exchange has
// a single consumer, which re-arms what it has read.
//
// Arithmetic and its order are those of the stand-alone gather / solve / panel kernels
// (ekf_small_kernels.hip): results are bitwise identical (tests/test_hip_parity.py).
#include "ekf_solve_device.h"

#define FR_T 512
#define FR_ALD 66          // row stride (doubles) of the A chunk in LDS

typedef double pf64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ekf_poll_sleep() { __builtin_amdgcn_s_sleep(4); }

// ---------------------------------------------------------------------------------------------
// measurement model of every detection into LDS (one thread per detection), shared by the S-block
// and the chunk role.  `publish`: the workgroup that also leaves jac / resid / lmcol in memory.
// ---------------------------------------------------------------------------------------------
template <int MODEL>
__device__ __forceinline__ void fr_measure(const EkfFrame& fr, const double* cam, const int* lmc, double* hs,
                                           int tid, bool publish) {
    constexpr int RD = EkfModel<MODEL>::RD, LMD = EkfModel<MODEL>::LMD, JC = EkfModel<MODEL>::JC;
    if (tid < fr.m) {
        const int c0 = lmc[tid];
        double lm[LMD], h[RD], J[RD][JC];
        for (int d = 0; d < LMD; ++d) lm[d] = fr.state[c0 + d];
        ekf_measure_model<MODEL>(cam, lm, h, J);
        for (int d = 0; d < RD; ++d)
            for (int a = 0; a < JC; ++a) hs[(RD * tid + d) * JC + a] = J[d][a];
        if (publish) {
            for (int d = 0; d < RD; ++d) {
                for (int a = 0; a < JC; ++a) fr.jac[(size_t)(RD * tid + d) * EKF_JLD + a] = J[d][a];
                const double rv = fr.z[RD * tid + d] - h[d];          // additive residual (:140)
                fr.resid[RD * tid + d] = rv;
                ekf_stc(fr.xr + RD * tid + d, rv);
            }
            fr.lmcol[tid] = c0;
        }
    }
    if (publish)
        for (int r = fr.k + tid; r < fr.kpad; r += FR_T) ekf_stc(fr.xr + r, 0.0);
}

// ---------------------------------------------------------------------------------------------
// role: one block of S
// ---------------------------------------------------------------------------------------------
template <typename T, int MODEL>
__device__ __forceinline__ void fr_role_sblock(const EkfFrame& fr, int sb, int nS, double* sm) {
    constexpr int RD = EkfModel<MODEL>::RD, LMD = EkfModel<MODEL>::LMD, JC = EkfModel<MODEL>::JC;
    constexpr int NSLOT = EKF_CAM + LMD * EkfModel<MODEL>::NDET16;
    double* hs = sm;                                    // [k][JC]
    double* us = sm + fr.k * JC;                        // [NSLOT][16]
    int* lmc = reinterpret_cast<int*>(us + NSLOT * 16);
    const int tid = threadIdx.x, m = fr.m, k = fr.k;
    const T* __restrict__ P = static_cast<const T*>(fr.cov);
    const T* __restrict__ prow = static_cast<const T*>(fr.prow);
    const int64_t ld = fr.ld;
    double cam[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) cam[a] = fr.state[a];
    if (tid < m) lmc[tid] = EKF_CAM + LMD * fr.idx[tid];
    __syncthreads();
    fr_measure<MODEL>(fr, cam, lmc, hs, tid, sb == 0);
    __syncthreads();
    int bi = 0, bj = sb;
    while (bj > bi) { bj -= bi + 1; ++bi; }
    const int j0 = (16 * bi) / RD;
    for (int e = tid; e < NSLOT * 16; e += FR_T) {
        const int slot = e >> 4, c2 = e & 15, r2 = 16 * bj + c2;
        const int j = j0 + (slot - EKF_CAM) / LMD, d = (slot - EKF_CAM) % LMD;
        double acc = 0.0;
        if (r2 < k && (slot < EKF_CAM || j < m)) {
            const int rho = (slot < EKF_CAM) ? slot : lmc[j] + d;
            const T* prw = prow ? prow + (int64_t)((slot < EKF_CAM) ? slot : EKF_CAM + LMD * j + d) * fr.ldw
                                : P + (int64_t)rho * ld;
            const double* h2 = hs + r2 * JC;
            const int c20 = lmc[r2 / RD];
            T pv[JC];
#pragma unroll
            for (int b = 0; b < EKF_CAM; ++b) pv[b] = prw[b];
#pragma unroll
            for (int b = 0; b < LMD; ++b) pv[EKF_CAM + b] = prw[c20 + b];
#pragma unroll
            for (int b = 0; b < JC; ++b) {
                const int col = (b < EKF_CAM) ? b : c20 + (b - EKF_CAM);
                const double pq = (double)pv[b] + ((col == rho) ? ekf_qdiag(rho, fr.dims, fr.nz) : 0.0);
                acc += pq * h2[b];
            }
        }
        us[e] = acc;
    }
    __syncthreads();
    if (tid < 256) {
        const int i = tid >> 4, c2 = tid & 15, r1 = 16 * bi + i, r2 = 16 * bj + c2;
        double v;
        if (r1 >= k || r2 >= k) {
            v = (r1 == r2) ? 1.0 : 0.0;
        } else if (r2 > r1) {
            v = 0.0;
        } else {
            const double* h1 = hs + r1 * JC;
            const int s1 = EKF_CAM + LMD * (r1 / RD - j0);
            double acc = (r1 == r2) ? fr.nz.r_unc : 0.0;
#pragma unroll
            for (int a = 0; a < EKF_CAM; ++a) acc += h1[a] * us[a * 16 + c2];
#pragma unroll
            for (int d = 0; d < LMD; ++d) acc += h1[EKF_CAM + d] * us[(s1 + d) * 16 + c2];
            v = acc;
        }
        ekf_stc(fr.xs + ((size_t)bj * fr.sblk_rows + r1) * 16 + c2, v);
    }
    // re-arm the factor exchange buffer of the NEXT frame (the one frame t-1 used)
    {
        const int64_t lo = (int64_t)fr.xl_len * sb / nS, hi = (int64_t)fr.xl_len * (sb + 1) / nS;
        const double sent = ekf_sent();
        for (int64_t e = lo + tid; e < hi; e += FR_T) ekf_stc(fr.xl_next + e, sent);
    }
}

// ---------------------------------------------------------------------------------------------
// role: factorisation (the stand-alone solve kernel with exchange-buffer input / output)
// ---------------------------------------------------------------------------------------------
struct FrBlockSrc { double* p; bool live; };
__device__ __forceinline__ FrBlockSrc fr_block_src(const EkfFrame& fr, int nb, int i, int tc, int lane) {
    const int rr = lane >> 2, cg = lane & 3;
    FrBlockSrc s;
    if (i < nb) {
        s.p = fr.xs + ((size_t)tc * fr.sblk_rows + EKF_RB * i + rr) * 16 + 4 * cg;
        s.live = true;
    } else {
        s.p = fr.xr + EKF_RB * tc + 4 * cg;
        s.live = rr == 0;
    }
    return s;
}
__device__ __forceinline__ sf64x4 fr_block_load(const FrBlockSrc& s) {
    sf64x4 v;
#pragma unroll
    for (int x = 0; x < 4; ++x) v[x] = ekf_ldc(s.p + x);
    return v;
}
__device__ __forceinline__ bool fr_block_pending(const sf64x4& v) {
    const bool p = ekf_is_sent(v[0]) || ekf_is_sent(v[1]) || ekf_is_sent(v[2]) || ekf_is_sent(v[3]);
    return __any(p);
}
// wait until the block is there, then re-arm it (single consumer)
__device__ __forceinline__ sf64x4 fr_block_take(const FrBlockSrc& s, sf64x4 v, int& spin_fail) {
    int it = 0;
    while (fr_block_pending(v)) {
        if (++it > EKF_SPIN_MAX) { spin_fail = 1; break; }
        ekf_poll_sleep();
        v = fr_block_load(s);
    }
    if (s.live) {
        const double sent = ekf_sent();
#pragma unroll
        for (int x = 0; x < 4; ++x) ekf_stc(s.p + x, sent);
    }
    return v;
}

__device__ __forceinline__ void fr_role_factor(const EkfFrame& fr, int RS, double* v_sm) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int kp = fr.kpad, nb = kp / EKF_RB, rows = kp + 1;
    double* ring = v_sm;                                         // [RS][rows][SV_CLD]
    const bool all_resident = RS >= nb;
    auto colbuf = [&](int col) { return ring + (size_t)(col % RS) * rows * SV_CLD; };
    constexpr int NW = FR_T / 64;
    double* __restrict__ xlop = fr.xl;
    double* __restrict__ xdop = fr.xl + fr.xl_dop;
    double* __restrict__ xy = fr.xl + fr.xl_y;
    int nstamp = 0, spin_fail = 0;
#define EKF_STAMP() do { if (fr.stamps && tid == 0) fr.stamps[nstamp] = clock64(); ++nstamp; } while (0)
    EKF_STAMP();
    constexpr int PB = 6;
    const int ncol0 = all_resident ? nb : (nb > 1 ? 2 : 1);
    int ntot = 0;
    for (int tc = 0; tc < ncol0; ++tc) ntot += nb - tc + 1;
    sf64x4 pv[PB];
    FrBlockSrc ps[PB];
    int pi[PB], ptc[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        int u = wave + NW * j, tc = 0;
        const bool has = u < ntot;
        u = has ? u : 0;
        while (u >= nb - tc + 1) { u -= nb - tc + 1; ++tc; }
        pi[j] = has ? tc + u : -1;
        ptc[j] = tc;
        ps[j] = fr_block_src(fr, nb, tc + u, tc, lane);
        pv[j] = fr_block_load(ps[j]);
    }
    EKF_STAMP();
#pragma unroll
    for (int j = 0; j < PB; ++j)
        if (pi[j] >= 0) {
            pv[j] = fr_block_take(ps[j], pv[j], spin_fail);
            sv_put_block(pv[j], colbuf(ptc[j]), kp, nb, pi[j], lane);
        }
    for (int u0 = wave + NW * PB; u0 < ntot; u0 += NW) {
        int u = u0, tc = 0;
        while (u >= nb - tc + 1) { u -= nb - tc + 1; ++tc; }
        const FrBlockSrc s = fr_block_src(fr, nb, tc + u, tc, lane);
        sv_put_block(fr_block_take(s, fr_block_load(s), spin_fail), colbuf(tc), kp, nb, tc + u, lane);
    }
    __syncthreads();
    EKF_STAMP();

    int bad = 0;
    for (int b = 0; b < nb; ++b) {
        const int cb = EKF_RB * b;
        double* cur = colbuf(b);
        if (b > 0) {
            for (int i = b + wave; i <= nb; i += NW) {
                SvAcc a;
                sv_acc_load(a, cur, kp, i, c, g);
                sv_term_lds(a, colbuf(b - 1), kp, i, b, c, g);
                sv_acc_store(a, cur, kp, i, c, g);
            }
            __syncthreads();
        }
        EKF_STAMP();
        const int nrows = kp - cb + 1;
        const int nrw = (nrows > 64) ? (nrows - 64 + 47) / 48 : 0;
        if (wave <= nrw + 1) {
            const bool idw = (wave == nrw + 1);
            int row;
            if (wave == 0 || lane < EKF_RB) row = cb + lane;
            else if (idw) row = -1;
            else row = cb + 64 + 48 * (wave - 1) + (lane - EKF_RB);
            const bool ident = idw && lane >= EKF_RB && lane < 2 * EKF_RB;
            const bool live = !idw && (wave == 0 || lane >= EKF_RB) && row <= kp;
            const double* src = cur + (size_t)min(max(row, 0), kp) * SV_CLD;
            double a[EKF_RB];
#pragma unroll
            for (int x = 0; x < EKF_RB; ++x) a[x] = (row < 0) ? ((x == lane - EKF_RB) ? 1.0 : 0.0) : src[x];
#pragma unroll
            for (int j = 0; j < EKF_RB; ++j) {
                const double d = ekf_readlane_f64(a[j], j);
                bad |= !(d > 0.0);
                const double y = ekf_rsqrt_f64(d);
                const double lj = a[j] * y;
                a[j] = lj;
#pragma unroll
                for (int x = j + 1; x < EKF_RB; ++x) {
                    const double lx = ekf_readlane_f64(lj, x);
                    a[x] = __builtin_fma(-lj, lx, a[x]);
                }
            }
            if (live) {
                if (row == kp) {                         // residual row: y of this block
#pragma unroll
                    for (int x = 0; x < EKF_RB; ++x) ekf_stc(xy + cb + x, a[x]);
                } else if (row >= cb + EKF_RB) {         // -L block (row/16, b) in MFMA operand order
                    const size_t base = sv_lop_index(row >> 4, b) + (row & 15);
#pragma unroll
                    for (int x = 0; x < EKF_RB; ++x) ekf_stc(xlop + base + (x >> 2) * 64 + 16 * (x & 3), -a[x]);
                }
                double* dst = cur + (size_t)row * SV_CLD;
#pragma unroll
                for (int x = 0; x < EKF_RB; ++x) dst[x] = a[x];
                if (fr.wdbg && row < kp) {               // dense L for tests only
#pragma unroll
                    for (int x = 0; x < EKF_RB; ++x) fr.lmat[(size_t)row * fr.ldl + cb + x] = (cb + x <= row) ? a[x] : 0.0;
                }
            }
            if (ident) {                                 // a[x] = Dinv_b[x][i], i = lane - 16
                const int i = lane - EKF_RB;
#pragma unroll
                for (int x = 0; x < EKF_RB; ++x) ekf_stc(xdop + (size_t)(b * 4 + (i >> 2)) * 64 + x + 16 * (i & 3), a[x]);
            }
        } else {
            const int fw = wave - (nrw + 2), nfw = NW - (nrw + 2);
            if (b >= 1 && b + 1 < nb)
                for (int i = b + 1 + fw; i <= nb; i += nfw) {
                    SvAcc a;
                    sv_acc_load(a, colbuf(b + 1), kp, i, c, g);
                    sv_term_lds(a, colbuf(b - 1), kp, i, b + 1, c, g);
                    sv_acc_store(a, colbuf(b + 1), kp, i, c, g);
                }
            const int tc = b + 2;
            if (tc < nb)
                for (int i = tc + fw; i <= nb; i += nfw) {
                    if (!all_resident) {
                        const FrBlockSrc s = fr_block_src(fr, nb, i, tc, lane);
                        sv_put_block(fr_block_take(s, fr_block_load(s), spin_fail), colbuf(tc), kp, nb, i, lane);
                    }
                    if (b >= 1) {
                        SvAcc a;
                        sv_acc_load(a, colbuf(tc), kp, i, c, g);
                        int q = 0;
                        for (; q < b && !(q + RS >= nb || b <= q + RS - 3); ++q) {}
                        sv_terms_glb<true>(a, xlop, xy, nb, i, tc, q, g, lane);
                        for (; q < b; ++q) sv_term_lds(a, colbuf(q), kp, i, tc, c, g, (q & 1) != 0);
                        sv_acc_store(a, colbuf(tc), kp, i, c, g);
                    }
                }
        }
        __syncthreads();
        EKF_STAMP();
    }
    if ((bad | (spin_fail << 2)) && lane == 0) atomicOr(fr.status, bad | (spin_fail << 2));
    EKF_STAMP();
#undef EKF_STAMP
}

// ---------------------------------------------------------------------------------------------
// role: one chunk of 64 columns
// ---------------------------------------------------------------------------------------------
// right-looking blocked forward substitution on 16 columns, one wave, everything in registers.
// t[b] starts as A_b and ends as W_b; per finished block column q of the factor:
//   W_q = Dinv_q t[q] ;  t[i] += (-L_iq) W_q  for i > q        (same fma sequence per t[i] as the
// left-looking stand-alone panel kernel: q ascending)
template <typename T, int NB>
__device__ __forceinline__ void fr_panel(const EkfFrame& fr, const double* a_lds, int wv, int col0, int lane,
                                         int& spin_fail) {
    const int j = lane & 15, g = lane >> 4;
    const double* __restrict__ xlop = fr.xl;
    const double* __restrict__ xdop = fr.xl + fr.xl_dop;
    const double* __restrict__ xy = fr.xl + fr.xl_y;
    pf64x4 t[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[b][r] = a_lds[(16 * b + g + 4 * r) * FR_ALD + 16 * wv + j];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        double dq[4];
        double lq[NB][4];
        int it = 0;
        for (;;) {
            bool pend = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dq[r] = ekf_ldc(xdop + (size_t)(q * 4 + r) * 64 + lane);
                pend |= ekf_is_sent(dq[r]);
            }
#pragma unroll
            for (int i = q + 1; i < NB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    lq[i - q - 1][r] = ekf_ldc(xlop + sv_lop_index(i, q) + r * 64 + lane);
                    pend |= ekf_is_sent(lq[i - q - 1][r]);
                }
            if (!__any(pend)) break;
            if (++it > EKF_SPIN_MAX) { spin_fail = 1; break; }
            ekf_poll_sleep();
        }
        pf64x4 wq = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) wq = __builtin_amdgcn_mfma_f64_16x16x4f64(dq[r], t[q][r], wq, 0, 0, 0);
        t[q] = wq;
#pragma unroll
        for (int i = q + 1; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                t[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[i - q - 1][r], wq[r], t[i], 0, 0, 0);
    }
    // y = L^-1 (z - h): complete once the last block column is
    double yv[NB][4];
    {
        int it = 0;
        for (;;) {
            bool pend = false;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    yv[b][r] = ekf_ldc(xy + 16 * b + g + 4 * r);
                    pend |= ekf_is_sent(yv[b][r]);
                }
            if (!__any(pend)) break;
            if (++it > EKF_SPIN_MAX) { spin_fail = 1; break; }
            ekf_poll_sleep();
        }
    }
    T* __restrict__ wp = static_cast<T*>(fr.wpanel);
    double part = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * b + g + 4 * r;
            const double v = t[b][r];
            part += v * yv[b][r];
            wp[(int64_t)row * fr.ldw + col0 + j] = (T)v;
            if (fr.wdbg) fr.wdbg[(int64_t)row * fr.ldw + col0 + j] = v;
        }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (g == 0) ekf_stc(fr.dxvec + col0 + j, part);
}

template <typename T, int NU, int MODEL>
__device__ __forceinline__ void fr_role_chunk(const EkfFrame& fr, int chunk, double* sm) {
    constexpr int RD = EkfModel<MODEL>::RD, LMD = EkfModel<MODEL>::LMD, JC = EkfModel<MODEL>::JC;
    constexpr int NWV = FR_T / 64;
    double* hs = sm;                                    // [k][JC]
    double* a_lds = sm + fr.k * JC;                     // [kpad][FR_ALD]
    int* lmc = reinterpret_cast<int*>(a_lds + (size_t)fr.kpad * FR_ALD);
    int* flag = lmc + 64;
    const int tid = threadIdx.x, m = fr.m;
    const int chunk0 = chunk * 64, cl = tid & 63, c = chunk0 + cl, g = tid >> 6;
    const T* __restrict__ P = static_cast<const T*>(fr.cov);
    const T* __restrict__ prow = static_cast<const T*>(fr.prow);
    const int64_t ld = fr.ld;
    T pcr[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) pcr[a] = prow ? prow[(int64_t)a * fr.ldw + c] : P[a * ld + c];
    double cam[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) cam[a] = fr.state[a];
    if (tid < m) lmc[tid] = EKF_CAM + LMD * fr.idx[tid];
    __syncthreads();
    T plr[NU][LMD];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int ju = min(g + NWV * u, m - 1);
        const int c0 = lmc[ju];
#pragma unroll
        for (int d = 0; d < LMD; ++d)
            plr[u][d] = prow ? prow[(int64_t)(EKF_CAM + LMD * ju + d) * fr.ldw + c] : P[(int64_t)(c0 + d) * ld + c];
    }
    fr_measure<MODEL>(fr, cam, lmc, hs, tid, false);
    __syncthreads();
    double pc[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) pc[a] = (double)pcr[a] + ((a == c) ? ekf_qdiag(a, fr.dims, fr.nz) : 0.0);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = g + NWV * u;
        if (j < m) {
            const int c0 = lmc[j];
            double pl[LMD];
#pragma unroll
            for (int d = 0; d < LMD; ++d) pl[d] = (double)plr[u][d] + ((c0 + d == c) ? fr.nz.q_lm : 0.0);
#pragma unroll
            for (int d = 0; d < RD; ++d) {
                const int r = RD * j + d;
                const double* hr = hs + r * JC;
                double acc = 0.0;
#pragma unroll
                for (int a = 0; a < EKF_CAM; ++a) acc += hr[a] * pc[a];
#pragma unroll
                for (int e = 0; e < LMD; ++e) acc += hr[10 + e] * pl[e];
                a_lds[r * FR_ALD + cl] = acc;
                if (fr.wdbg) fr.amat[(int64_t)r * fr.lda + c] = acc;
            }
        }
    }
    for (int r = fr.k + g; r < fr.kpad; r += NWV) a_lds[r * FR_ALD + cl] = 0.0;
    __syncthreads();
    int spin_fail = 0;
    if (g < 4) {
        const int lane = tid & 63, col0 = chunk0 + 16 * g;
        switch (fr.kpad / EKF_RB) {
#define FR_CASE(NB) case NB: fr_panel<T, NB>(fr, a_lds, g, col0, lane, spin_fail); break;
            FR_CASE(1) FR_CASE(2) FR_CASE(3) FR_CASE(4) FR_CASE(5) FR_CASE(6)
            FR_CASE(7) FR_CASE(8) FR_CASE(9) FR_CASE(10) FR_CASE(11)
            default: fr_panel<T, 12>(fr, a_lds, g, col0, lane, spin_fail); break;
#undef FR_CASE
        }
        if (spin_fail && (tid & 63) == 0) atomicOr(fr.status, 4);
    }
    // dx of this chunk has to be in memory before the chunk is counted as done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned long long old =
            __hip_atomic_fetch_add(fr.done_ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = (old + 1 == fr.done_target);
    }
    __syncthreads();
    if (!flag[0]) return;
    // ---- last chunk: state injection (every workgroup has read the old state by now)
    if (MODEL == 0) {
        // extended_kalman_filter.py:133-152: xyz and every landmark additive, dx[3:7] dropped,
        // quaternion from the error state, error state reset
        for (int col = EKF_CAM + tid; col < fr.dims; col += FR_T) fr.state[col] = fr.state[col] + ekf_ldc(fr.dxvec + col);
        if (tid == 0) {
            double x[3];
            for (int i = 0; i < 3; ++i) {
                x[i] = fr.state[i] + ekf_ldc(fr.dxvec + i);
                fr.state[i] = x[i];
            }
            double q[4] = {fr.state[3], fr.state[4], fr.state[5], fr.state[6]};
            const double err[3] = {ekf_ldc(fr.dxvec + 7), ekf_ldc(fr.dxvec + 8), ekf_ldc(fr.dxvec + 9)};
            ekf_quat_inject(q, err, fr.quat_mode);
            for (int i = 0; i < 4; ++i) fr.state[3 + i] = q[i];
            for (int i = 0; i < 3; ++i) fr.state[7 + i] = 0.0;
            if (fr.traj_row) {
                for (int i = 0; i < 3; ++i) fr.traj_row[i] = x[i];
                for (int i = 0; i < 4; ++i) fr.traj_row[3 + i] = q[i];
            }
        }
    } else {
        // ekf_with_rotations.py:142-177: camera and every landmark: xyz additive, quaternion
        // multiplicative (scalar first); the landmarks' error states are never written
        for (int i = tid; i <= fr.n_lm; i += FR_T) {
            const int c0 = (i == 0) ? 0 : EKF_CAM + 10 * (i - 1);
            double* st = fr.state + c0;
            const double* dx = fr.dxvec + c0;
            double q[4] = {st[3], st[4], st[5], st[6]};
            const double err[3] = {ekf_ldc(dx + 7), ekf_ldc(dx + 8), ekf_ldc(dx + 9)};
            ekf_quat_inject(q, err, 1);
            const double x0 = st[0] + ekf_ldc(dx + 0), x1 = st[1] + ekf_ldc(dx + 1), x2 = st[2] + ekf_ldc(dx + 2);
            st[0] = x0; st[1] = x1; st[2] = x2;
            for (int e = 0; e < 4; ++e) st[3 + e] = q[e];
            if (i == 0) {
                for (int e = 0; e < 3; ++e) st[7 + e] = 0.0;
                if (fr.traj_row) {
                    fr.traj_row[0] = x0; fr.traj_row[1] = x1; fr.traj_row[2] = x2;
                    for (int e = 0; e < 4; ++e) fr.traj_row[3 + e] = q[e];
                }
            }
        }
    }
}

template <typename T, int NU, int MODEL>
__global__ __launch_bounds__(FR_T) void ekf_front_kernel(EkfFrame fr, int RS) {
    extern __shared__ __attribute__((aligned(16))) double fr_sm[];
    const int nb = fr.kpad / EKF_RB, nS = nb * (nb + 1) / 2;
    const int bx = blockIdx.x;
    if (bx < nS) fr_role_sblock<T, MODEL>(fr, bx, nS, fr_sm);
    else if (bx == nS) fr_role_factor(fr, RS, fr_sm);
    else fr_role_chunk<T, NU, MODEL>(fr, bx - nS - 1, fr_sm);
}

template <typename T, int NU, int MODEL>
static void ekf_front_go(const EkfFrame& fr, hipStream_t s) {
    constexpr int LMD = EkfModel<MODEL>::LMD, JC = EkfModel<MODEL>::JC;
    constexpr int NSLOT = EKF_CAM + LMD * EkfModel<MODEL>::NDET16;
    static bool once = false;
    if (!once) {   // > 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ekf_front_kernel<T, NU, MODEL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    const int nb = fr.kpad / EKF_RB, nS = nb * (nb + 1) / 2;
    const size_t lds_s = ((size_t)fr.k * JC + NSLOT * 16) * 8 + 64 * 4 + 16;
    const size_t lds_c = ((size_t)fr.k * JC + (size_t)fr.kpad * FR_ALD) * 8 + 64 * 4 + 16;
    const size_t lds_f = (size_t)ekf_solve_stream_lds_bytes(fr.kpad, fr.k);
    size_t lds = lds_s > lds_c ? lds_s : lds_c;
    if (lds_f > lds) lds = lds_f;
    hipLaunchKernelGGL((ekf_front_kernel<T, NU, MODEL>), dim3(nS + 1 + fr.ncols / 64), dim3(FR_T), lds, s, fr,
                       ekf_solve_stream_ring(fr.kpad, fr.k));
}

template <typename T>
void ekf_launch_front(const EkfFrame& fr, hipStream_t s) {
    if (fr.model == 1) return ekf_front_go<T, 4, 1>(fr, s);      // m <= 27
    if (fr.m <= 32) return ekf_front_go<T, 4, 0>(fr, s);
    return ekf_front_go<T, 8, 0>(fr, s);
}
template void ekf_launch_front<float>(const EkfFrame&, hipStream_t);
template void ekf_launch_front<double>(const EkfFrame&, hipStream_t);
