// (implementation header: instantiated for float in ekf_front.hip and for double in ekf_front_f64.hip,
// two translation units so that they compile side by side)
// Fused "front" kernel of the EKF update (gfx950): measurement model, A = H (P+Q),
// S = Hs (P+Q) Hs^T + R, blocked Cholesky, W = L^-1 A, dx = W^T y and the state injection in ONE
// launch (reference: extended_kalman_filter.py:107-152; ekf_with_rotations.py:115-177).
//
// Workgroup roles by blockIdx.x (a workgroup only ever waits for LOWER-indexed workgroups, which
// the dispatcher starts first, so the waits cannot deadlock whatever the residency):
//   0              measurement model: Jacobian rows / residual for the other roles
//   [1, nS]        one 16x16 block of S each, straight from P (nS = nb (nb+1) / 2)
//   nS + 1         the factorisation: takes the S blocks into registers as they appear, runs the
//                  pivot chain, publishes -L / Dinv / y block column by block column
//   > nS + 1       one chunk of 64 columns each: A chunk into LDS, then the right-looking blocked
//                  forward substitution in registers, consuming block column q of the factor as soon
//                  as it is published (requested one step ahead); W chunk, dx chunk, state injection
//                  for its own columns (the only readers of the old state are workgroups 0 .. nS, which
//                  the factorisation has waited for).  EKF_Rotations: the LAST chunk to finish injects
//                  (every landmark carries a quaternion that straddles chunks).
// Pipelined sequence mode (ekf_api.hip: ekf_observe_sequence_device): the launch runs BESIDE the covariance
// update of the previous frame, on the previous frame's input P: the S-block and chunk roles complete the
// entries of the current P they read from (P_prev, W_prev) themselves, with the covariance update's own
// per-element instruction sequence (same bits); the chunks leave the support columns of W the NEXT frame
// will need for that in a compact copy (W_sup).
// Exchange between workgroups: ekf_solve_device.h (agent-scope relaxed accesses: write-through
// stores, coherent loads, sentinel values; no flags, no global fences, no cache flushes).  The
// protocol needs exactly one hardware property, per-location coherence of agent-scope atomics:
//   * every word a consumer uses is checked against the sentinel (the single polled word is only a
//     hint for WHEN to fetch the bulk; a word that has not landed yet reads as the sentinel and is
//     fetched again);
//   * a buffer is armed with sentinels by ANOTHER launch: everything that travels inside a launch
//     (S blocks, residual, -L / Dinv / y, Jacobian rows, tags) lives in one buffer per fused-frame
//     parity, and the buffer of parity p is re-armed by the S-block workgroups of the following fused
//     frame -- one kernel boundary after its last reader, one before its next writer.  No role
//     re-arms what it has just read (round 1 did that for the S blocks);
//   * producers publish the fused-frame number next to the data (per S block, residual, Jacobian,
//     factor block column); a consumer that finds another frame's number next to data it has just
//     accepted raises a status bit: stale words that are NOT sentinels cannot pass silently.
// Rules learnt the hard way: a polled line is written by ONE full-line store;
// one wave per workgroup polls; write-through stores stay off the pivot chain's waves; a wait for
// a load also waits for every older store; LDS hand-overs need workgroup-scope fences, a volatile
// flag alone is not enough; code that runs once per launch runs from a cold instruction cache.
//
// Arithmetic and its order are those of the stand-alone gather / solve / panel kernels
// (ekf_small_kernels.hip): results are bitwise identical (tests/test_hip_parity.py).
#pragma once
#include "ekf_solve_big.h"

#define FR_T 512
#define FR_ALD 66          // row stride (doubles) of the A chunk in LDS
#ifndef FR_DOT_RING
#define FR_DOT_RING 2      // operand chunks (8 k-pairs) in flight, S-block role's support dots (pipelined mode)
#endif
// doubles of the chunk role's A region (pipelined mode: before A is built the region holds the staged operands of
// the support-row product, W_sup [96][128] and W [96][64], and its result [128][64], all f32: 104 KB)
#define FR_KS 96            // rows of W staged at a time
// (f64 covariance: 48 rows of W at a time and 64 slots per round: [48][64] + [48][64] + [64][64] doubles, 80 KB)
#define FR_KS64 48
__host__ __device__ inline size_t fr_chunk_a_len(int kpad, bool fix, int elem) {
    const size_t a = (size_t)kpad * FR_ALD;
    const size_t st = elem == 4 ? (size_t)(FR_KS * 128 + FR_KS * 64 + 128 * 64) / 2 : (size_t)(2 * FR_KS64 * 64 + 64 * 64);
    return (fix && a < st) ? st : a;
}
// NB >= 2: the four waves that do not substitute fetch whole block columns of the factor into LDS slots instead
// (fr_panel_stage), see fr_panel.
#define FR_STAGED_MIN_NB 2
#define FR_SLOTS 4

typedef double pf64x4 __attribute__((ext_vector_type(4)));
// Stores of W / W_sup (read by later launches only): write-through, like the covariance update's (ekf_cov_update.hip) --
// plain stores leave 1.2 MB of dirty lines for the end of the launch, where the next front kernel waits for them
// (us per frame pipelined / serial at n=1024, m=32: plain 24.4 / 36.1, write-through 23.9 / 35.6).
// FR_W_STORE_MODE (experiments): 0 plain, 2 write-through
#ifndef FR_W_STORE_MODE
#define FR_W_STORE_MODE 2
#endif
template <typename T>
__device__ __forceinline__ void fr_w_store(T* p, T v) {
    if (FR_W_STORE_MODE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
typedef float fr_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void ekf_poll_sleep() { __builtin_amdgcn_s_sleep(4); }
// every exchange read is a coherent load (`coherent` kept for readability at the call sites: first
// attempt vs retry)
__device__ __forceinline__ double ekf_ldt(const double* p, bool /*coherent*/) { return ekf_ldc(p); }

// ---------------------------------------------------------------------------------------------
// measurement model of every detection into LDS (one thread per detection), shared by the measurement
// and the S-block role.  `publish`: also leave jac / resid / lmcol in memory (ekf_debug_fetch) and
// the residual in LDS for the publication.
// ---------------------------------------------------------------------------------------------
template <int MODEL>
__device__ __forceinline__ void fr_measure(const EkfFrame& fr, const double* cam, const int* lmc, double* hs,
                                           double* rsd, int tid, bool publish) {
    constexpr int RD = EkfModel<MODEL>::RD, LMD = EkfModel<MODEL>::LMD, JC = EkfModel<MODEL>::JC;
    if (tid < fr.m) {
        const int c0 = lmc[tid];
        double lm[LMD], h[RD];
        for (int d = 0; d < LMD; ++d) lm[d] = fr.state[c0 + d];
        // the Jacobian rows are built in place in LDS (7 x 20 doubles would not fit the registers)
        double (*J)[JC] = reinterpret_cast<double (*)[JC]>(hs + (size_t)RD * tid * JC);
        ekf_measure_model<MODEL>(cam, lm, h, J);
        if (publish) {
            for (int d = 0; d < RD; ++d) {
                for (int a = 0; a < JC; ++a) fr.jac[(size_t)(RD * tid + d) * EKF_JLD + a] = J[d][a];
                const double rv = fr.z[RD * tid + d] - h[d];          // additive residual (:140)
                fr.resid[RD * tid + d] = rv;
                rsd[RD * tid + d] = rv;
            }
            fr.lmcol[tid] = c0;
        }
    }
    if (publish)
        for (int r = fr.k + tid; r < fr.kpad; r += FR_T) rsd[r] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// role: one block of S
// ---------------------------------------------------------------------------------------------
template <typename T, int MODEL, int NB>
__device__ __forceinline__ void fr_role_sblock(const EkfFrame& fr, int sb, int nS, double* sm) {
    constexpr int RD = EkfModel<MODEL>::RD, LMD = EkfModel<MODEL>::LMD, JC = EkfModel<MODEL>::JC;
    constexpr int NSLOT = EKF_CAM + LMD * EkfModel<MODEL>::NDET16;
    double* hs = sm;                                    // [k][JC]
    double* us = sm + fr.k * JC;                        // [NSLOT][16]
    int* lmc = reinterpret_cast<int*>(us + NSLOT * 16);
    double* rsd = us + NSLOT * 16 + 32;                 // [kpad] z - h (workgroup 0 only)
    const int tid = threadIdx.x, m = fr.m, k = fr.k;
    const T* __restrict__ P = static_cast<const T*>(fr.cov);
    const int64_t ld = fr.ld;
    double cam[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) cam[a] = fr.state[a];
    if (fr.stamps && sb == 0 && tid == 0) fr.stamps[60] = wall_clock64();
    int bi = 0, bj = sb;
    while (bj > bi) { bj -= bi + 1; ++bi; }
    const int j0 = (16 * bi) / RD, jc0 = (16 * bj) / RD;
    // Pipelined sequence mode: `cov` is P of the PREVIOUS frame (the covariance update that produces the current P
    // runs beside this launch).  The entries this block needs are completed on the fly,
    //     P_cur[r][c] = (P_prev[r][c] + Q[r == c]) + sum_k fma(-W[k][r], W[k][c]),     k ascending from zero,
    // the per-element instruction sequence of the covariance update (ekf_cov_update.hip): the same bits.  Both r and
    // c are support rows of this frame, so the W columns come from the compact copy W_sup[k][slot] that the previous
    // front kernel left behind: row slots {0..9} + [10 + LMD j0, ..), column slots {0..9} + [10 + LMD jc0, ..).
    // The sums run on the matrix cores (v_mfma_f32_32x32x2_f32 over k ascending, as in the covariance update), one
    // wave per 32 x 32 tile of [row slots] x [column slots] -- the LAST waves of the workgroup, beside the index ->
    // P round trips and the measurement model of the others; their operand loads do not depend on the indices.
    // (f64 covariance: the same with v_mfma_f64_16x16x4_f64 on 16 x 16 tiles, the f64 update's sequence.)
    const bool fix = fr.wsup_prev != nullptr;
    constexpr int NT16 = (NSLOT + 15) / 16, DLD16 = 16 * NT16, DLD64 = DLD16;
    double* dots64 = rsd + fr.kpad;                           // [16 NT16][16 NT16] (f64 covariance)
    float* dots = reinterpret_cast<float*>(rsd + fr.kpad);    // [16 NT16][16 NT16] (f32 covariance)
    const int dlane = tid & 63, dc = dlane & 15, dg = dlane >> 4, dtile = 7 - (tid >> 6);
    // f32: 16 x 16 tiles of v_mfma_f32_16x16x4_f32 -- four fused multiply-adds per element and instruction over k
    // ascending, the same chain as the covariance update's v_mfma_f32_32x32x2_f32 pairs (tools/probes/mfma_order_probe.hip:
    // bit for bit on random data of mixed magnitudes) -- one tile per wave, the last waves first: a quarter of the
    // matrix time of one 32 x 32 tile on one wave (48 x 64 cycles at k = 96, on the path of every S block).
    // Operand A: lane (i = c, k = g) = -W_sup[4 s + g][row slot], B: lane (k = g, j = c); up to 24 k-steps in flight.
    constexpr int KS4 = 4 * NB, KB = KS4 < 24 ? KS4 : 24;
    const bool dotw = sizeof(T) == 4 && fix && dtile < NT16 * NT16;
    const float* __restrict__ wsd = static_cast<const float*>(fr.wsup_prev);
    float oa[KB], ob[KB];
    auto dot_operands = [&](int tile, int s0) {
        const int smax = fr.wsup_ld - 1;
        const int rs = 16 * (tile / NT16) + dc, cs = 16 * (tile % NT16) + dc;
        const int gr = (rs < EKF_CAM) ? rs : min(EKF_CAM + LMD * j0 + (rs - EKF_CAM), smax);
        const int gc = (cs < EKF_CAM) ? cs : min(EKF_CAM + LMD * jc0 + (cs - EKF_CAM), smax);
#pragma unroll
        for (int u = 0; u < KB; ++u)
            if (s0 + u < KS4) {
                const float* kb = wsd + (int64_t)(4 * (s0 + u) + dg) * fr.wsup_ld;
                oa[u] = kb[gr];
                ob[u] = kb[gc];
            }
    };
    if (dotw) dot_operands(dtile, 0);      // (they do not depend on the indices: in flight beside the index -> P round trips)
    if (tid < m) lmc[tid] = ekf_lm_column(fr, LMD, tid, false);
    __syncthreads();
    // this thread's entry of U (slot, c2): its P values are requested before the measurement model
    // is evaluated, so the two dependent memory round trips overlap
    constexpr int NE = (NSLOT * 16 + FR_T - 1) / FR_T;  // entries of U per thread (1 or 2)
    int urho[NE], uc20[NE], ur2[NE];
    bool uact[NE];
    T pv[NE][JC];
#pragma unroll
    for (int n = 0; n < NE; ++n) {
        const int ue = tid + FR_T * n;
        const int uslot = ue >> 4, uc2 = ue & 15;
        ur2[n] = 16 * bj + uc2;
        const int uj = j0 + (uslot - EKF_CAM) / LMD, ud = (uslot - EKF_CAM) % LMD;
        uact[n] = ue < NSLOT * 16 && ur2[n] < k && (uslot < EKF_CAM || uj < m);
        urho[n] = 0;
        uc20[n] = 0;
        if (uact[n]) {
            urho[n] = (uslot < EKF_CAM) ? uslot : lmc[uj] + ud;
            const T* prw = P + (int64_t)urho[n] * ld;
            uc20[n] = lmc[ur2[n] / RD];
#pragma unroll
            for (int b = 0; b < EKF_CAM; ++b) pv[n][b] = prw[b];
#pragma unroll
            for (int b = 0; b < LMD; ++b) pv[n][EKF_CAM + b] = prw[uc20[n] + b];
        }
    }
    if (dotw) {
        typedef float fr_f32x4 __attribute__((ext_vector_type(4)));
        for (int tile = dtile; tile < NT16 * NT16; tile += 8) {
            fr_f32x4 dacc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s0 = 0; s0 < KS4; s0 += KB) {
                if (tile != dtile || s0 != 0) dot_operands(tile, s0);
#pragma unroll
                for (int u = 0; u < KB; ++u)
                    if (s0 + u < KS4) dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(-oa[u], ob[u], dacc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)      // C / D layout: column c, row 4 g + r
                dots[(16 * (tile / NT16) + 4 * dg + r) * DLD16 + 16 * (tile % NT16) + dc] = dacc[r];
        }
    }
    if constexpr (sizeof(T) == 8) {
        if (fix) {      // f64: 16 x 16 tiles, the last waves first; A[i = c][k = g] = -W_sup[4 s + g][row slot], B[k = g][j = c]
            const double* __restrict__ w64 = static_cast<const double*>(fr.wsup_prev);
            const int smax = fr.wsup_ld - 1;
            for (int tile = dtile; tile < NT16 * NT16; tile += 8) {
                const int rs = 16 * (tile / NT16) + dc, cs = 16 * (tile % NT16) + dc;
                const int gr = (rs < EKF_CAM) ? rs : min(EKF_CAM + LMD * j0 + (rs - EKF_CAM), smax);
                const int gc = (cs < EKF_CAM) ? cs : min(EKF_CAM + LMD * jc0 + (cs - EKF_CAM), smax);
                const double* wa = w64 + (int64_t)dg * fr.wsup_ld + gr;
                const double* wb = w64 + (int64_t)dg * fr.wsup_ld + gc;
                pf64x4 dacc = {0.0, 0.0, 0.0, 0.0};
                for (int ks = 0; ks < (fr.kpad >> 2); ks += 4) {
                    double a[4], b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a[u] = wa[(int64_t)(4 * (ks + u)) * fr.wsup_ld];
                        b[u] = wb[(int64_t)(4 * (ks + u)) * fr.wsup_ld];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[u], b[u], dacc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)      // C / D layout: column c, row g + 4 r
                    dots64[(16 * (tile / NT16) + dg + 4 * r) * DLD64 + 16 * (tile % NT16) + dc] = dacc[r];
            }
        }
    }
    fr_measure<MODEL>(fr, cam, lmc, hs, rsd, tid, false);
    __syncthreads();
    if (fix) {
#pragma unroll
        for (int n = 0; n < NE; ++n) {
            if (uact[n]) {
                const int uslot = (tid + FR_T * n) >> 4;
                const int cdet = EKF_CAM + LMD * (ur2[n] / RD - jc0);
#pragma unroll
                for (int b = 0; b < JC; ++b) {
                    const int col = (b < EKF_CAM) ? b : uc20[n] + (b - EKF_CAM);
                    const int cs = (b < EKF_CAM) ? b : cdet + (b - EKF_CAM);
                    if constexpr (sizeof(T) == 4) {
                        const float q = (col == urho[n]) ? (float)ekf_qdiag(urho[n], fr.dims, fr.nz) : 0.0f;
                        pv[n][b] = (T)(((float)pv[n][b] + q) + dots[uslot * DLD16 + cs]);
                    } else {      // (the f64 update: v = P; if (diagonal) v += Q; v += acc)
                        double v = (double)pv[n][b];
                        if (col == urho[n]) v += ekf_qdiag(urho[n], fr.dims, fr.nz);
                        pv[n][b] = (T)(v + dots64[uslot * DLD64 + cs]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NE; ++n) {
        const int ue = tid + FR_T * n;
        if (ue < NSLOT * 16) {
            double acc = 0.0;
            if (uact[n]) {
                const double* h2 = hs + ur2[n] * JC;
#pragma unroll
                for (int b = 0; b < JC; ++b) {
                    const int col = (b < EKF_CAM) ? b : uc20[n] + (b - EKF_CAM);
                    const double pq = (double)pv[n][b] + ((col == urho[n]) ? ekf_qdiag(urho[n], fr.dims, fr.nz) : 0.0);
                    acc = __builtin_fma(pq, h2[b], acc);
                }
            }
            us[ue] = acc;
        }
    }
    __syncthreads();
    double v = 0.0;
    if (tid < 256) {
        const int i = tid >> 4, c2 = tid & 15, r1 = 16 * bi + i, r2 = 16 * bj + c2;
        if (r1 >= k || r2 >= k) {
            v = (r1 == r2) ? 1.0 : 0.0;
        } else if (r2 > r1) {
            v = 0.0;                                   // strict upper part of a diagonal block: mirrored on emission
        } else {
            const double* h1 = hs + r1 * JC;
            const int s1 = EKF_CAM + LMD * (r1 / RD - j0);
            double acc = (r1 == r2) ? fr.nz.r_unc : 0.0;
#pragma unroll
            for (int a = 0; a < EKF_CAM; ++a) acc = __builtin_fma(h1[a], us[a * 16 + c2], acc);
#pragma unroll
            for (int d = 0; d < LMD; ++d) acc = __builtin_fma(h1[EKF_CAM + d], us[(s1 + d) * 16 + c2], acc);
            v = acc;
        }
    }
    __syncthreads();                                   // `us` is free: it becomes the block's tile
    if (tid < 256) us[(tid >> 4) * 17 + (tid & 15)] = v;
    __syncthreads();
    if (tid < 64) {                                    // OP memory order, two 1 KB write-through stores
        sv_sblock_emit<true>(fr.xs + sv_blk_index(bi, bj), us, bi == bj, tid);
        if (tid == 0) ekf_stc(fr.xs_tag + 16 * bj + bi, fr.seqno);
    }
    if (fr.stamps && sb == nS - 1 && tid == 0) fr.stamps[61] = wall_clock64();

    // re-arm the WHOLE exchange buffer of the next fused frame (the one the previous fused frame used:
    // S blocks, residual, factor, Jacobian, tags); its readers finished at least one kernel boundary ago
    {
        const int64_t lo = (int64_t)fr.xl_len * sb / nS, hi = (int64_t)fr.xl_len * (sb + 1) / nS;
        const double sent = ekf_sent();
        for (int64_t e = lo + tid; e < hi; e += FR_T) ekf_stc(fr.xl_next + e, sent);
    }
}

// ---------------------------------------------------------------------------------------------
// role: measurement model for the other roles (workgroup 0): Jacobian rows for the chunks, residual
// for the factorisation, plus the copies that ekf_debug_fetch reads.  Its own workgroup, so that no
// S block is delayed by the publication.
// ---------------------------------------------------------------------------------------------
template <int MODEL>
__device__ __forceinline__ void fr_role_measure(const EkfFrame& fr, double* sm) {
    constexpr int LMD = EkfModel<MODEL>::LMD, JC = EkfModel<MODEL>::JC;
    double* hs = sm;                                    // [k][JC]
    double* rsd = sm + fr.k * JC;                       // [kpad]
    int* lmc = reinterpret_cast<int*>(rsd + fr.kpad);
    const int tid = threadIdx.x;
    // (role-level stamps: the start of each of the last 16 launches, by frame number)
    if (fr.stamps && !fr.stamps_heavy && tid == 0) fr.stamps[16 + ((long long)fr.seqno & 15)] = wall_clock64();
    // pipelined sequence mode: this launch has started, i.e. everything before it on its stream is complete
    if (fr.la_signal && tid == 0)
        __hip_atomic_store(fr.la_sync, fr.la_signal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    double cam[EKF_CAM];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) cam[a] = fr.state[a];
    if (tid < fr.m) lmc[tid] = ekf_lm_column(fr, LMD, tid, true);      // (the role that reports a bad index)
    __syncthreads();
    fr_measure<MODEL>(fr, cam, lmc, hs, rsd, tid, true);
    __syncthreads();
    // whole cache lines per store instruction (a line that is polled must not be written piecemeal)
    double* __restrict__ xj = fr.xl + fr.xl_jac;
    const int nel = fr.k * JC;
    for (int e = tid; e < nel; e += FR_T) ekf_stc(xj + e, hs[e]);
    for (int e = tid; e < fr.kpad; e += FR_T) ekf_stc(fr.xr + e, rsd[e]);
    if (tid == 0) {
        ekf_stc(fr.xl + fr.xl_tag, fr.seqno);
        ekf_stc(fr.xl + fr.xl_tag + 16, fr.seqno);         // residual
    }
    if (fr.stamps && tid == 0) fr.stamps[55] = wall_clock64();
    // pipelined sequence mode: keep this launch open until the covariance update of the previous frame (other
    // stream) is complete -- whatever follows this launch on its stream may then read P
    if (fr.la_gate && tid == 0) {
        int it = 0;
        while (__hip_atomic_load(fr.la_sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < fr.la_gate) {
            if (++it > (1 << 22)) { ekf_raise(fr, EKF_ST_GATE_TIMEOUT); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// role: factorisation = the shared register-resident Cholesky (ekf_solve_device.h: sv_factor) with the
// exchange buffer as input and output
// ---------------------------------------------------------------------------------------------
// IO policy.  Input: the wave's S blocks (OP memory order, 2 KB each: four 8-byte coherent loads per lane,
// all blocks of the wave in flight together.  16-byte loads would halve the instruction count, but the two ways
// to get them are both out: inline asm (the compiler copied destination registers between issue and wait:
// tools/asm_load_hazards.py) and volatile accesses (`global_load_dwordx4 sc0 sc1`, but LLVM puts an
// `s_waitcnt vmcnt(0)` behind every volatile load: one round trip per load, 1.1 us per block) and, for the wave that owns the residual
// row, z - h.  Polling touches ONE word per row (the last word of the row's first block: the S-block
// workgroups all finish at about the same time); correctness does not depend on it: every word is
// checked against the sentinel after the bulk has arrived and fetched again until it has landed.
// Output: Dinv_b (`dop` order), -L blocks (`lop` order), y, the frame tags -- write-through stores,
// straight from the registers (both orders ARE the OP layout).
__device__ __forceinline__ int fr_tag_stale(double tag, double seqno) { return !ekf_is_sent(tag) && tag != seqno; }
struct SvIoFused {
    const EkfFrame& fr;
    int spin_fail, stale;
    __device__ __forceinline__ bool settled(const sf64x4& v) const {
        return !__any(ekf_is_sent(v[0]) || ekf_is_sent(v[1]) || ekf_is_sent(v[2]) || ekf_is_sent(v[3]));
    }
    __device__ __forceinline__ void settle_block(sf64x4& v, int i, int tc, int lane) {
        const double* base = fr.xs + sv_blk_index(i, tc);
        int it = 0;
        while (!settled(v)) {
            if (++it > EKF_SPIN_MAX) { spin_fail = 1; break; }
            ekf_poll_sleep();
            v = sf64x4{ekf_ldc(base + 2 * lane), ekf_ldc(base + 2 * lane + 1), ekf_ldc(base + 128 + 2 * lane),
                       ekf_ldc(base + 129 + 2 * lane)};
        }
    }
    __device__ __forceinline__ sf64x4 fetch_resid(int tc, int g) {
        const double* base = fr.xr + EKF_RB * tc + g;
        sf64x4 v = {ekf_ldc(base), ekf_ldc(base + 4), ekf_ldc(base + 8), ekf_ldc(base + 12)};
        int it = 0;
        while (!settled(v)) {
            if (++it > EKF_SPIN_MAX) { spin_fail = 1; break; }
            ekf_poll_sleep();
            v = sf64x4{ekf_ldc(base), ekf_ldc(base + 4), ekf_ldc(base + 8), ekf_ldc(base + 12)};
        }
        return v;
    }
    // Block columns [C0, C1) of the wave's rows into registers (and, with C0 == 0, the whole residual row if the
    // wave owns it).  Tags and bulk, all in flight together; then the per-word sentinel check, block by block, fetching
    // a block again until every word of it has landed.  Branch-free issue: block indices are clamped
    // instead of guarded, so that the loads of a round sit in one basic block and all go out before the first
    // result is needed (guarded loads were waited for one by one: ~1.2 us per block).
    template <int NB, int N0, int N1, int C0, int C1>
    __device__ __forceinline__ void load_cols(sf64x4 (&z0)[N0], sf64x4 (&z1)[N1], int i0, int i1, bool has0, bool has1,
                                              int lane, int g) {
        const bool blk0 = has0 && i0 < NB, blk1 = has1 && i1 < NB;       // S-block rows (not the residual)
        const bool res0 = C0 == 0 && has0 && i0 == NB, res1 = C0 == 0 && has1 && i1 == NB;
        const int r0 = blk0 ? i0 : 0, r1 = blk1 ? i1 : r0;
        // No separate polling round: tags and bulk go out at once and every block is fetched again until all its
        // words have landed (settle_block).  Polling one word per block first cost a memory round trip between "the
        // block is there" and "the block is in registers", on the path to the first chain.
        double t0[N0], t1[N1];
#pragma unroll
        for (int tc = C0; tc < N0 && tc < C1; ++tc) t0[tc] = ekf_ldc(fr.xs_tag + 16 * min(tc, r0) + r0);
#pragma unroll
        for (int tc = C0; tc < N1 && tc < C1; ++tc) t1[tc] = ekf_ldc(fr.xs_tag + 16 * min(tc, r1) + r1);
        // (straight into z: entries beyond the row's last block receive a clamped duplicate that nobody reads; with
        // C0 == 0 the residual row's entries are overwritten below)
#pragma unroll
        for (int tc = C0; tc < N0 && tc < C1; ++tc) {
            const double* p0 = fr.xs + sv_blk_index(r0, min(tc, r0)) + 2 * lane;
            if (!(has0 && i0 == NB)) z0[tc] = sf64x4{ekf_ldc(p0), ekf_ldc(p0 + 1), ekf_ldc(p0 + 128), ekf_ldc(p0 + 129)};
        }
#pragma unroll
        for (int tc = C0; tc < N1 && tc < C1; ++tc) {
            const double* p0 = fr.xs + sv_blk_index(r1, min(tc, r1)) + 2 * lane;
            if (!(has1 && i1 == NB)) z1[tc] = sf64x4{ekf_ldc(p0), ekf_ldc(p0 + 1), ekf_ldc(p0 + 128), ekf_ldc(p0 + 129)};
        }
        // (every tag is consumed here, also those of the clamped duplicates -- they are tags of real blocks of this
        // frame: a load whose result is never used stays "pending" for the compiler, which then waits for it, and for
        // every younger store, wherever the register is reused)
#pragma unroll
        for (int tc = C0; tc < N0 && tc < C1; ++tc) {
            stale |= fr_tag_stale(t0[tc], fr.seqno);
            if (blk0 && tc <= i0) settle_block(z0[tc], i0, tc, lane);
        }
#pragma unroll
        for (int tc = C0; tc < N1 && tc < C1; ++tc) {
            stale |= fr_tag_stale(t1[tc], fr.seqno);
            if (blk1 && tc <= i1) settle_block(z1[tc], i1, tc, lane);
        }
        // the residual row (one wave): z - h, replicated in all 16 rows of its "block".  All loads first, THEN the
        // sentinel checks (one round trip)
        if (res0) {
#pragma unroll
            for (int tc = 0; tc < N0; ++tc)
                if (tc < NB) {
                    const double* base = fr.xr + EKF_RB * tc + g;
                    z0[tc] = sf64x4{ekf_ldc(base), ekf_ldc(base + 4), ekf_ldc(base + 8), ekf_ldc(base + 12)};
                }
#pragma unroll
            for (int tc = 0; tc < N0; ++tc)
                if (tc < NB && !settled(z0[tc])) z0[tc] = fetch_resid(tc, g);
            stale |= fr_tag_stale(ekf_ldc(fr.xl + fr.xl_tag + 16), fr.seqno);
        }
        if (res1) {
#pragma unroll
            for (int tc = 0; tc < N1; ++tc)
                if (tc < NB) {
                    const double* base = fr.xr + EKF_RB * tc + g;
                    z1[tc] = sf64x4{ekf_ldc(base), ekf_ldc(base + 4), ekf_ldc(base + 8), ekf_ldc(base + 12)};
                }
#pragma unroll
            for (int tc = 0; tc < N1; ++tc)
                if (tc < NB && !settled(z1[tc])) z1[tc] = fetch_resid(tc, g);
            stale |= fr_tag_stale(ekf_ldc(fr.xl + fr.xl_tag + 16), fr.seqno);
        }
    }
    // (ekf_solve_big.h) one complete S block: all four words requested, then fetched again until every word has landed; its tag
    __device__ __forceinline__ sf64x4 fetch_block(int i, int tc, int lane) {
        const double* p0 = fr.xs + sv_blk_index(i, tc) + 2 * lane;
        const double tag = ekf_ldc(fr.xs_tag + 16 * tc + i);
        sf64x4 v = {ekf_ldc(p0), ekf_ldc(p0 + 1), ekf_ldc(p0 + 128), ekf_ldc(p0 + 129)};
        stale |= fr_tag_stale(tag, fr.seqno);
        settle_block(v, i, tc, lane);
        return v;
    }
    // a finished block of the factor read back from the exchange: -L_iq in `lop` order (the caller has seen the publisher's
    // "column q is in memory"; coherent loads)
    __device__ __forceinline__ sf64x4 hist_block(int i, int q, int lane) const {
        const double* base = fr.xl + sv_lop_index(i, q) + lane;
        return sf64x4{ekf_ldc(base), ekf_ldc(base + 64), ekf_ldc(base + 128), ekf_ldc(base + 192)};
    }
    __device__ __forceinline__ void put_dinv(int b, const sf64x4& xop, int lane) {
        double* xdop = fr.xl + fr.xl_dop;
#pragma unroll
        for (int r = 0; r < 4; ++r) ekf_stc(xdop + (size_t)(b * 4 + r) * 64 + lane, xop[r]);     // (the chunks poll the last word)
        if (lane == 0) ekf_stc(fr.xl + fr.xl_tag + 1 + b, fr.seqno);
    }
    __device__ __forceinline__ void put_l(int i, int b, const sf64x4& y, int lane) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ekf_stc(fr.xl + sv_lop_index(i, b) + r * 64 + lane, -y[r]);
    }
    __device__ __forceinline__ void put_y(int b, double yv, int c, bool active) {
        if (active) ekf_stc(fr.xl + fr.xl_y + EKF_RB * b + c, yv);        // one instruction: a whole cache line
    }
};

template <int NB>
__device__ __forceinline__ void fr_role_factor(const EkfFrame& fr, double* v_sm) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (fr.stamps && tid == 0) fr.stamps[62] = wall_clock64();
    SvIoFused io{fr, 0, 0};
    int bad = 0, badcol = 0;
    sv_factor_any<NB>(fr, io, v_sm, bad, badcol);
    if ((bad | io.spin_fail | io.stale) && lane == 0) {
        ekf_raise(fr, (bad ? EKF_ST_NOT_SPD : 0) | (io.spin_fail ? EKF_ST_TIMEOUT : 0) | (io.stale ? EKF_ST_STALE_S : 0));
        if (bad) {      // diagnostics: which wave saw it, and the first block column
            atomicOr(fr.status + 1, 1 << wave);
            atomicCAS(fr.status + 2, 0, badcol);
        }
    }
    if (fr.stamps && tid == 0) fr.stamps[63] = wall_clock64();
    // the S-block tags once more, now that every one of them has landed (the copy that travelled with the bulk may have
    // been read before its producer had written it: it then counted as "no objection")
    if (tid < 256) {
        const int tc = tid >> 4, i = tid & 15;
        if (tc <= i && i < NB && fr_tag_stale(ekf_ldc(fr.xs_tag + 16 * tc + i), fr.seqno)) ekf_raise(fr, EKF_ST_STALE_S);
    }
}

// ---------------------------------------------------------------------------------------------
// role: one chunk of 64 columns
// ---------------------------------------------------------------------------------------------
// right-looking blocked forward substitution on 16 columns, one wave, everything in registers.
// t[b] starts as A_b and ends as W_b; per finished block column q of the factor:
//   W_q = Dinv_q t[q] ;  t[i] += (-L_iq) W_q  for i > q ;  dx += W_q^T y_q
// (same fma sequence per t[i] and for dx as the left-looking stand-alone panel kernel).
// Only wave 0 of the workgroup polls memory for block column q (4 loads per round, so the
// factorisation's stores are not stuck behind a storm of polls); the other waves watch an LDS word.
// What a wave of the substitution requests BEFORE the A chunk is built (the requests are in flight during that
// build): the first blocks of -L of block column 0, (wave 0) Dinv_0 / y_0, and which of the next frame's support
// slots its column feeds.  A word that has not been published yet arrives as the sentinel and is fetched again.
template <int NB>
struct FrPre {
    static constexpr bool STAGED = NB >= FR_STAGED_MIN_NB;
    static constexpr int SLOT = NB * 256 + 16;      // doubles per staged block column: Dinv | -L blocks below the diagonal | y
    static constexpr int LG = (NB >= 11) ? 4 : 6;
    double lqa[LG][4], dqn[4], yqn[4];
    unsigned long long smask;       // next-frame detections whose landmark owns this lane's column (duplicates possible)
    int sdim;
    double st_old, q_old[4];        // EKF model: the old state of this lane's column / the old camera quaternion (wave 0)
};
template <int NB, int MODEL>
__device__ __forceinline__ void fr_panel_pre(const EkfFrame& fr, FrPre<NB>& pre, const unsigned long long* smask_l, int wv, int col0, int lane) {
    const int j = lane & 15, g = lane >> 4;
    const double* __restrict__ xlop = fr.xl;
    if constexpr (!FrPre<NB>::STAGED) {
#pragma unroll
        for (int i = 1; i < NB && i < 1 + FrPre<NB>::LG; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) pre.lqa[i - 1][r] = ekf_ldc(xlop + sv_lop_index(i, 0) + r * 64 + lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pre.dqn[r] = (wv == 0) ? ekf_ldc(fr.xl + fr.xl_dop + (size_t)r * 64 + lane) : 0.0;
            pre.yqn[r] = (wv == 0) ? ekf_ldc(fr.xl + fr.xl_y + g + 4 * r) : 0.0;
        }
    }
    const int mycol = col0 + j;
    // the old state of this wave's columns (nobody writes it before the injection at the end of fr_panel).  Requested here,
    // before the A chunk is built: requested in fr_panel it was waited for at that function's first barrier -- a memory
    // round trip between "A in LDS" and the first step of the substitution
    pre.st_old = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) pre.q_old[i] = 0.0;
    if (MODEL == 0) {
        if (mycol < fr.dims) pre.st_old = fr.state[mycol];
        if (col0 == 0)
            for (int i = 0; i < 4; ++i) pre.q_old[i] = fr.state[3 + i];
    }
    pre.smask = fr.wsup ? smask_l[16 * wv + j] : 0ull;
    pre.sdim = (mycol >= EKF_CAM) ? (mycol - EKF_CAM) % EkfModel<MODEL>::LMD : 0;
}

// Staged substitution (NB >= FR_STAGED_MIN_NB), the waves 4 .. 7 of a chunk workgroup: wave w fetches the block columns
// q = w, w + 4, w + 8 .. of the factor -- Dinv_q and y_q, then the -L_iq below them: four columns in flight per
// workgroup -- polls until every word has been published (sentinel), and puts the column into slot
// q % FR_SLOTS of the A-chunk region of LDS (free once the substituting waves hold their columns of A in registers: the
// barrier).  flags[s] = columns whose Dinv / y have been placed in slot s so far, flags[8 + s] = whose -L blocks have,
// flags[4 + s] = reads of slot s acknowledged (4 per column).
// Before: every substituting wave fetched its own copy of -L in groups of four blocks, a dependent round trip per group
// (22 per chunk at NB = 12: with everything published the twelve steps took ~60 us; staged: 13 us after the first column's
// round trip.  Pipelined n=1024 m=64: 82.4 -> 71.7 us per frame; EKF_Rotations n=400 m=27: 8.43k -> 9.1k updates/s).
template <int NB>
__device__ __forceinline__ void fr_panel_stage(const EkfFrame& fr, double* slots, volatile ekf_lds_int* flags, int w, int lane) {
    constexpr int SLOT = FrPre<NB>::SLOT;
    // The LAST block column (Dinv and y, nothing below) is on the path to the end of the launch: published -> seen by a
    // poll -> last substitution step -> W / dx / state.  One wave polling with one request in flight sees it a memory round
    // trip and a half after it was published on average (2.2 us per round trip beside the previous frame's covariance
    // update); the loaders that have no column left poll it as well, out of phase, and whoever has it first places it.
    constexpr bool HELP = NB >= 2 && NB <= 8;
    const double* __restrict__ xlop = fr.xl;
    const double* __restrict__ xdop = fr.xl + fr.xl_dop;
    const double* __restrict__ xy = fr.xl + fr.xl_y;
    __syncthreads();                                   // the A chunk has been taken into registers
    int spin_fail = 0;
    for (int q = w; q < NB; q += 4) {
        const int cnt = NB - 1 - q;                    // blocks below the diagonal
        const int sl = q % FR_SLOTS;
        double* slot = slots + sl * SLOT;
        // One attempt at the whole column (a chunk that is behind the factorisation finds it: one round trip).  A column
        // that is not there yet is waited for QUIETLY: first on LDS until the previous column has arrived in this workgroup,
        // then on ONE line (y_q, the last thing the factorisation publishes of a column), so that at most one wave per
        // chunk polls memory.  (Every chunk polling Dinv and y -- 17 lines, 4 waves each -- from the moment it was ready:
        // the factorisation's publication stores queued behind the polls, 5 - 7 us per block column instead of 2 at
        // n=4096.)
        double dv[4], yv = 0.0;
        double v[NB - 1][4];
        auto load_dy = [&]() {
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[r] = ekf_ldc(xdop + (size_t)(q * 4 + r) * 64 + lane);
            yv = ekf_ldc(xy + 16 * q + (lane & 15));
        };
        auto pend_dy = [&]() {
            bool pend = ekf_is_sent(yv);
#pragma unroll
            for (int r = 0; r < 4; ++r) pend = pend || ekf_is_sent(dv[r]);
            return __any(pend) != 0;
        };
        auto fetch = [&]() {
#pragma unroll
            for (int i = 0; i < NB - 1; ++i) {
                // (beyond the column's end: the Dinv block once more -- no branch between the loads, always a published word)
                const double* __restrict__ src = (i < cnt) ? xlop + sv_lop_index(q + 1 + i, q) : xdop + (size_t)q * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i][r] = ekf_ldc(src + r * 64 + lane);
            }
        };
        load_dy();
        fetch();
        if (pend_dy()) {
            if (q > 0) {
                int it = 0;
                while (flags[(q - 1) % FR_SLOTS] < (q - 1) / FR_SLOTS + 1) {
                    if (++it > 64 * EKF_SPIN_MAX) { spin_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            if (NB >= 9) {      // (many chunks: one line first; few: Dinv and y at once, a round trip less between "published" and "used")
                for (int it = 0; ekf_is_sent(ekf_ldc(xy + 16 * q + (lane & 15))); ++it) {
                    if (it > EKF_SPIN_MAX) { spin_fail = 1; break; }
                    ekf_poll_sleep();
                }
            }
            for (int it = 0;; ++it) {
                load_dy();
                if (!pend_dy()) break;
                if (it > EKF_SPIN_MAX) { spin_fail = 1; break; }
                ekf_poll_sleep();
            }
            fetch();
        }
        if (HELP && q == NB - 1) {                     // (the last column: whoever has it first places it, see below)
            int won = 0;
            if (lane == 0) won = __hip_atomic_fetch_add(const_cast<ekf_lds_int*>(flags) + 12, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;
            if (!__builtin_amdgcn_readfirstlane(won)) break;
        }
        if (q >= FR_SLOTS) {                           // the slot's previous column has been read by all four waves
            int it = 0;
            while (flags[4 + sl] < 4 * (q / FR_SLOTS)) {
                if (++it > 64 * EKF_SPIN_MAX) { spin_fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) slot[r * 64 + lane] = dv[r];
        if (lane < 16) slot[NB * 256 + lane] = yv;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) flags[sl] = q / FR_SLOTS + 1;
        for (int it = 0;; ++it) {
            bool pend = false;
#pragma unroll
            for (int i = 0; i < NB - 1; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) pend = pend || ekf_is_sent(v[i][r]);
            if (!__any(pend)) break;
            if (it > EKF_SPIN_MAX) { spin_fail = 1; break; }
            ekf_poll_sleep();
            fetch();
        }
#pragma unroll
        for (int i = 0; i < NB - 1; ++i)
            if (i < cnt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) slot[(1 + i) * 256 + r * 64 + lane] = v[i][r];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) flags[8 + sl] = q / FR_SLOTS + 1;
    }
    if (HELP && w != (NB - 1) % 4) {
        constexpr int q = NB - 1, sl = q % FR_SLOTS;
        double* slot = slots + sl * SLOT;
        for (int d = (((NB - 1) % 4 - w) & 3); d > 0; --d) __builtin_amdgcn_s_sleep(24);      // (out of phase with the column's own loader)
        for (int it = 0; flags[sl] < q / FR_SLOTS + 1; ++it) {
            double dv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[r] = ekf_ldc(xdop + (size_t)(q * 4 + r) * 64 + lane);
            const double yv = ekf_ldc(xy + 16 * q + (lane & 15));
            bool pend = ekf_is_sent(yv);
#pragma unroll
            for (int r = 0; r < 4; ++r) pend = pend || ekf_is_sent(dv[r]);
            if (!__any(pend)) {
                int won = 0;
                if (lane == 0) won = __hip_atomic_fetch_add(const_cast<ekf_lds_int*>(flags) + 12, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;
                if (__builtin_amdgcn_readfirstlane(won)) {
                    if (q >= FR_SLOTS) {
                        int it2 = 0;
                        while (flags[4 + sl] < 4 * (q / FR_SLOTS)) {
                            if (++it2 > 64 * EKF_SPIN_MAX) { spin_fail = 1; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) slot[r * 64 + lane] = dv[r];
                    if (lane < 16) slot[NB * 256 + lane] = yv;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) {
                        flags[sl] = q / FR_SLOTS + 1;
                        flags[8 + sl] = q / FR_SLOTS + 1;
                    }
                }
                break;
            }
            if (it > EKF_SPIN_MAX) { spin_fail = 1; break; }
            ekf_poll_sleep();
        }
    }
    if (spin_fail && lane == 0) ekf_raise(fr, EKF_ST_TIMEOUT);
}

template <typename T, int NB, int MODEL>
__device__ __forceinline__ void fr_panel(const EkfFrame& fr, const double* a_lds, volatile ekf_lds_int* sync, double* pshare,
                                         int wv, int col0, int lane, int& spin_fail, FrPre<NB>& pre) {
    const int j = lane & 15, g = lane >> 4;
    const double* __restrict__ xlop = fr.xl;
    const double* __restrict__ xdop = fr.xl + fr.xl_dop;
    const double* __restrict__ xy = fr.xl + fr.xl_y;
    long long* stp = (fr.stamps && col0 == 0 && lane == 0) ? fr.stamps + 34 : nullptr;
    T* __restrict__ wp = static_cast<T*>(fr.wpanel);
    // pipelined sequence mode: the columns of W that the NEXT frame's front kernel needs for its support rows
    // (camera + every next-frame detection's landmark) are also written compactly, W_sup[k][slot], so that it
    // stages them with 16-byte loads instead of gathering 96 x 106 scattered words.
    // smask: next-frame detections whose landmark owns this lane's column (duplicates possible); cslot: camera slot
    T* __restrict__ wsup = static_cast<T*>(fr.wsup);
    const unsigned long long smask = pre.smask;
    const int sdim = pre.sdim;
    const int mycol = col0 + j;
    const int cslot = (wsup && mycol < EKF_CAM) ? mycol : -1;
    pf64x4 t[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[b][r] = a_lds[(16 * b + g + 4 * r) * FR_ALD + 16 * wv + j];
    // the old state of this wave's columns (nobody writes it before the injection below)
    const int scol = col0 + j;
    const double st_old = pre.st_old;
    const double q_old[4] = {pre.q_old[0], pre.q_old[1], pre.q_old[2], pre.q_old[3]};
    const double q_rn = (MODEL == 0 && col0 == 0) ? ekf_quat_rnorm(q_old) : 0.0;      // (here: not on the tail of the launch)
    double part = 0.0;
    // Prefetch (a chunk that is BEHIND the factorisation -- every chunk in the pipelined sequence mode, whose
    // prologue is longer -- finds everything published already; fetching Dinv / y, then -L, then computing cost two
    // dependent memory round trips per block column, 2.2 - 2.8 us): the first LG blocks of -L of block column q + 1
    // and (wave 0) its Dinv / y are requested during step q.  A word that has not been published yet reads as the
    // sentinel: such a column goes through the polling path exactly as before.
    constexpr int LG = FrPre<NB>::LG;
    constexpr bool AHEAD = NB <= 8;                 // two register sets for -L
    if constexpr (FrPre<NB>::STAGED) {
        // block columns come through LDS (fr_panel_stage); `pshare` holds the slot flags
        constexpr int SLOT = FrPre<NB>::SLOT;
        volatile ekf_lds_int* flags = ekf_lds_flags(pshare);
        __syncthreads();                               // every wave has its columns of A: the region becomes the slots
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const double* __restrict__ slot = a_lds + (q % FR_SLOTS) * SLOT;
            int it = 0;
            while (flags[q % FR_SLOTS] < q / FR_SLOTS + 1) {
                if (++it > 64 * EKF_SPIN_MAX) { spin_fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            double dq[4], yq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dq[r] = slot[r * 64 + lane];
                yq[r] = slot[NB * 256 + g + 4 * r];
            }
            pf64x4 wq = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) wq = __builtin_amdgcn_mfma_f64_16x16x4f64(dq[r], t[q][r], wq, 0, 0, 0);
            t[q] = wq;
            if (q + 1 < NB) {
                it = 0;
                while (flags[8 + q % FR_SLOTS] < q / FR_SLOTS + 1) {
                    if (++it > 64 * EKF_SPIN_MAX) { spin_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
#pragma unroll
            for (int i = q + 1; i < NB; ++i) {
                double lq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) lq[r] = slot[(i - q) * 256 + r * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) t[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[r], wq[r], t[i], 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the reads before the acknowledgement
            if (lane == 0) __hip_atomic_fetch_add(const_cast<ekf_lds_int*>(flags) + 4 + q % FR_SLOTS, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int r = 0; r < 4; ++r) {                   // W rows of block q are final
                const int row = 16 * q + g + 4 * r;
                fr_w_store(wp + (int64_t)row * fr.ldw + col0 + j, (T)wq[r]);
                if (fr.wdbg) fr.wdbg[(int64_t)row * fr.ldw + col0 + j] = wq[r];
                if (cslot >= 0) fr_w_store(wsup + (int64_t)row * fr.wsup_ld + cslot, (T)wq[r]);
                for (unsigned long long mm = smask; mm; mm &= mm - 1) {
                    constexpr int LMD = EkfModel<MODEL>::LMD;
                    const int jj = __builtin_ctzll(mm);
                    fr_w_store(wsup + (int64_t)row * fr.wsup_ld + EKF_CAM + LMD * jj + sdim, (T)wq[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) part = __builtin_fma(wq[r], yq[r], part);
            if (stp) stp[q] = wall_clock64();
        }
    } else {
    auto& lqa = pre.lqa;
    auto& dqn = pre.dqn;
    auto& yqn = pre.yqn;
    double lqb[AHEAD ? LG : 1][4];
    auto issue_l = [&](auto& lq, int q) {
#pragma unroll
        for (int i = q + 1; i < NB && i < q + 1 + LG; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) lq[i - q - 1][r] = ekf_ldc(xlop + sv_lop_index(i, q) + r * 64 + lane);
    };
    auto issue_d = [&](int q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dqn[r] = ekf_ldc(xdop + (size_t)(q * 4 + r) * 64 + lane);
            yqn[r] = ekf_ldc(xy + 16 * q + g + 4 * r);
        }
    };
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        double dq[4], yq[4];
        if (!AHEAD && q > 0) issue_l(lqa, q);
        if (wv == 0) {
            // sync[0] = block columns shared so far, sync[1] = reads of the shared slot acknowledged
            bool pend = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dq[r] = dqn[r];
                yq[r] = yqn[r];
                pend = pend || ekf_is_sent(dq[r]) || ekf_is_sent(yq[r]);
            }
            if (__any(pend)) {
                // (polled on the data itself: a poll of one word first, then the bulk, then the tag were three dependent
                // memory round trips between "published" and "used" whenever the chunk waits for the factorisation --
                // the last block columns, i.e. the tail of the launch)
                for (int it = 0;; ++it) {
                    pend = false;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dq[r] = ekf_ldc(xdop + (size_t)(q * 4 + r) * 64 + lane);
                        yq[r] = ekf_ldc(xy + 16 * q + g + 4 * r);
                        pend = pend || ekf_is_sent(dq[r]) || ekf_is_sent(yq[r]);
                    }
                    if (!__any(pend)) break;
                    if (it > EKF_SPIN_MAX) { spin_fail = 1; break; }
                    ekf_poll_sleep();
                }
            }
            int it = 0;
            while (sync[1] < 3 * q) {                  // the slot's previous content has been read
                if (++it > 64 * EKF_SPIN_MAX) { spin_fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pshare[r * 64 + lane] = dq[r];
                pshare[(4 + r) * 64 + lane] = yq[r];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the slot before the word that announces it
            if (lane == 0) sync[0] = q + 1;
            if (q + 1 < NB) issue_d(q + 1);
        } else {
            int it = 0;
            while (sync[0] < q + 1) {
                if (++it > 64 * EKF_SPIN_MAX) { spin_fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dq[r] = pshare[r * 64 + lane];
                yq[r] = pshare[(4 + r) * 64 + lane];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the reads before the acknowledgement
            if (lane == 0) __hip_atomic_fetch_add(const_cast<ekf_lds_int*>(sync) + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (AHEAD && q + 1 < NB) {                   // (before this step's W stores: a wait for a load also waits for every older store)
            if ((q & 1) == 0) issue_l(lqb, q + 1);
            else issue_l(lqa, q + 1);
        }
        pf64x4 wq = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) wq = __builtin_amdgcn_mfma_f64_16x16x4f64(dq[r], t[q][r], wq, 0, 0, 0);
        t[q] = wq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                   // W rows of block q are final
            const int row = 16 * q + g + 4 * r;
            fr_w_store(wp + (int64_t)row * fr.ldw + col0 + j, (T)wq[r]);
            if (fr.wdbg) fr.wdbg[(int64_t)row * fr.ldw + col0 + j] = wq[r];
            if (cslot >= 0) fr_w_store(wsup + (int64_t)row * fr.wsup_ld + cslot, (T)wq[r]);
            for (unsigned long long mm = smask; mm; mm &= mm - 1) {
                constexpr int LMD = EkfModel<MODEL>::LMD;
                const int jj = __builtin_ctzll(mm);
                fr_w_store(wsup + (int64_t)row * fr.wsup_ld + EKF_CAM + LMD * jj + sdim, (T)wq[r]);
            }
        }
        // t[i] += (-L_iq) W_q for i > q, at most LG blocks of -L in registers at a time (the whole
        // kernel has to stay clear of register spills); the first LG were requested a step ago
        auto apply = [&](auto& lq, int i0, bool fresh) {
            int it = 0;
            for (;; fresh = true) {      // normally one pass
                bool pend = false;
#pragma unroll
                for (int i = i0; i < NB && i < i0 + LG; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (fresh) lq[i - i0][r] = ekf_ldc(xlop + sv_lop_index(i, q) + r * 64 + lane);
                        pend = pend || ekf_is_sent(lq[i - i0][r]);
                    }
                if (!__any(pend)) break;
                if (++it > EKF_SPIN_MAX) { spin_fail = 1; break; }
                ekf_poll_sleep();
            }
#pragma unroll
            for (int i = i0; i < NB && i < i0 + LG; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    t[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[i - i0][r], wq[r], t[i], 0, 0, 0);
        };
        if (q + 1 < NB) {
            if (AHEAD && (q & 1)) apply(lqb, q + 1, false);
            else apply(lqa, q + 1, false);
        }
#pragma unroll
        for (int i0 = q + 1 + LG; i0 < NB; i0 += LG) {
            double lq[LG][4];
            apply(lq, i0, true);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part = __builtin_fma(wq[r], yq[r], part);
        if (stp) stp[q] = wall_clock64();
    }
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);                    // dx[col0 + j] in every lane group
    const int col = col0 + j;
    if (MODEL == 1) {               // every landmark has a quaternion: injected by the last chunk
        if (g == 0) ekf_stc(fr.dxvec + col, part);
    } else {
        // extended_kalman_filter.py:133-152.  Nobody reads the state any more: its only readers
        // in this launch are the S-block workgroups, which the factorisation has waited for.
        double nv = 0.0;
        if (g == 0 && (col < 3 || (col >= EKF_CAM && col < fr.dims))) {
            nv = st_old + part;
            fr.state[col] = nv;
            if (fr.state_host) fr.state_host[col] = nv;
        }
        if (col0 == 0) {
            const double e0 = __shfl(part, 7), e1 = __shfl(part, 8), e2 = __shfl(part, 9);
            const double x0 = __shfl(nv, 0), x1 = __shfl(nv, 1), x2 = __shfl(nv, 2);
            if (lane == 0) {
                double qv[4] = {q_old[0], q_old[1], q_old[2], q_old[3]};
                const double err[3] = {e0, e1, e2};
                ekf_quat_inject_n(qv, q_rn, err, fr.quat_mode);
                for (int i = 0; i < 4; ++i) fr.state[3 + i] = qv[i];
                for (int i = 0; i < 3; ++i) fr.state[7 + i] = 0.0;   // :152
                if (fr.state_host) {
                    for (int i = 0; i < 4; ++i) fr.state_host[3 + i] = qv[i];
                    for (int i = 0; i < 3; ++i) fr.state_host[7 + i] = 0.0;
                }
                if (fr.traj_row) {
                    fr.traj_row[0] = x0; fr.traj_row[1] = x1; fr.traj_row[2] = x2;
                    for (int i = 0; i < 4; ++i) fr.traj_row[3 + i] = qv[i];
                }
            }
        }
    }
    if (stp) stp[NB] = wall_clock64();
    // the frame tags of the factor's block columns, all at once and after everything that depended on them has been
    // issued (a status bit is all they can raise; checked inside the steps they were a round trip per block column)
    if (wv == 0 && lane < NB && fr_tag_stale(ekf_ldc(fr.xl + fr.xl_tag + 1 + lane), fr.seqno)) ekf_raise(fr, EKF_ST_STALE_COL);
    if (wv == 0 && lane == NB && fr_tag_stale(ekf_ldc(fr.xl + fr.xl_tag), fr.seqno)) ekf_raise(fr, EKF_ST_STALE_JAC);
}

#ifdef FR_CHUNK_DIAG      // (diagnostic builds: prologue stamps of chunk 0, role-level stamp mode, slots 41 .. 52)
#define FR_DIAG(i) do { if (fr.stamps && !fr.stamps_heavy && chunk == 0 && tid == 0) fr.stamps[41 + (i)] = wall_clock64(); } while (0)
#else
#define FR_DIAG(i) do { } while (0)
#endif
template <typename T, int NU, int MODEL, int NB>
__device__ __forceinline__ void fr_role_chunk(const EkfFrame& fr, int chunk, double* sm) {
    constexpr int RD = EkfModel<MODEL>::RD, LMD = EkfModel<MODEL>::LMD, JC = EkfModel<MODEL>::JC;
    constexpr int NWV = FR_T / 64;
    double* hs = sm;                                    // [k][JC]
    double* a_lds = sm + fr.k * JC;                     // [kpad][FR_ALD]; before A is built: the support-row tiles (pipelined mode)
    const bool fix = fr.wprev != nullptr;
    const size_t a_len = fr_chunk_a_len(fr.kpad, fix, (int)sizeof(T));
    int* lmc = reinterpret_cast<int*>(a_lds + a_len);
    int* flag = lmc + 64;                               // [0] last-chunk flag, [1..2] panel sync words
    double* pshare = reinterpret_cast<double*>(lmc + 72);   // [8][64] Dinv operands and y of one block column
    unsigned long long* smask_l = reinterpret_cast<unsigned long long*>(pshare + 8 * 64);      // [64] per column of the chunk
    int* nidx = reinterpret_cast<int*>(smask_l + 64);       // [64] the next frame's landmark indices
    const int tid = threadIdx.x, m = fr.m;
    const int chunk0 = chunk * 64, cl = tid & 63, c = chunk0 + cl, g = tid >> 6;
    const T* __restrict__ P = static_cast<const T*>(fr.cov);
    const int64_t ld = fr.ld;
    if (fr.stamps && chunk == 0 && tid == 0) fr.stamps[32] = wall_clock64();
    // Jacobian rows: published by the measurement workgroup (this role never reads the state, so the
    // injection at the end of the launch cannot race with a chunk that starts late)
    // Two halves: the requests go out early (pipelined mode: before the matrix phase of the support rows, which does not
    // need them) and the values are taken, checked against the sentinel (fetched again until they have landed) and put
    // into LDS when the A chunk is about to be built.  Polled on the data itself, tag with the bulk (checked once more at
    // the end of the role: fr_panel).
    constexpr int NJ = (16 * NB * JC + FR_T - 1) / FR_T;
    double jv[NJ], jtag = 0.0;
    const double* __restrict__ xj = fr.xl + fr.xl_jac;
    const int nel = fr.k * JC;
    auto jac_issue = [&]() {
#pragma unroll
        for (int n = 0; n < NJ; ++n) jv[n] = ekf_ldc(xj + min(tid + FR_T * n, nel - 1));
        if (tid == 0) jtag = ekf_ldc(fr.xl + fr.xl_tag);
    };
    auto jac_finish = [&]() {
        int spin = 0;
#pragma unroll
        for (int n = 0; n < NJ; ++n) {
            const int e = tid + FR_T * n;
            if (e < nel) {
                double v = jv[n];
                int it = 0;
                while (ekf_is_sent(v)) {
                    if (++it > EKF_SPIN_MAX) { spin = 1; break; }
                    ekf_poll_sleep();
                    v = ekf_ldc(xj + e);
                }
                hs[e] = v;
            }
        }
        if (spin) ekf_raise(fr, EKF_ST_TIMEOUT);
        if (fr.stamps && fr.stamps_heavy && chunk == 0 && tid == 0 && NB <= 6) fr.stamps[44] = wall_clock64();
        if (tid == 0 && fr_tag_stale(jtag, fr.seqno)) ekf_raise(fr, EKF_ST_STALE_JAC);
    };
    T pcr[EKF_CAM];
    T plr[NU][LMD];
    if (tid < m) lmc[tid] = ekf_lm_column(fr, LMD, tid, false);
    if (fr.wsup && tid >= 64 && tid < 64 + fr.next_m) {     // (the next frame's indices, for the support-column copy of W)
        int ni = fr.next_idx[tid - 64];
        nidx[tid - 64] = ((unsigned)ni >= (unsigned)fr.n_lm) ? 0 : ni;
    }
    if (tid == 0) { flag[1] = 0; flag[2] = 0; }
    if (FrPre<NB>::STAGED && tid < 16) ekf_lds_flags(pshare)[tid] = 0;      // (staged substitution: slot flags)
    __syncthreads();
    if (fr.wsup && tid < 64) {      // next-frame detections whose landmark owns column c (duplicates possible)
        unsigned long long mk = 0ull;
        if (c >= EKF_CAM && c < fr.dims) {
            const int li = (c - EKF_CAM) / LMD;
            for (int jj = 0; jj < fr.next_m; ++jj) mk |= (unsigned long long)(nidx[jj] == li) << jj;
        }
        smask_l[tid] = mk;
    }
    FR_DIAG(0);
    if (!fix) {
        // (camera rows and landmark rows together: requested before the index round trip, the ten camera-row values
        // were spilled one by one, each load waited for -- hipcc)
#pragma unroll
        for (int a = 0; a < EKF_CAM; ++a) pcr[a] = P[a * ld + c];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int ju = min(g + NWV * u, m - 1);
            const int c0 = lmc[ju];
#pragma unroll
            for (int d = 0; d < LMD; ++d) plr[u][d] = P[(int64_t)(c0 + d) * ld + c];
        }
    } else if constexpr (sizeof(T) == 4) {
        // Pipelined sequence mode: `cov` is the PREVIOUS frame's P; the support rows of the current one (10 + LMD m
        // slots x this chunk's 64 columns) are completed on the matrix cores:
        //     (P_prev + Q) + sum_k fma(-W_sup[k][slot], W[k][col]),   v_mfma_f32_32x32x2_f32 over k in ascending order
        // -- per element the instruction sequence of ekf_cov_update_mfma_f32 (for an element above the diagonal that
        // kernel runs the mirrored product: the same bits).  128 slots per round: 8 tiles of 32 x 32, one per wave.
        // The operands are staged through LDS by the whole workgroup, 96 rows of W at a time, 16 bytes per lane and all
        // loads in flight together: ONE memory round trip per stage (a first version fed the MFMAs from a register
        // ring straight from the L2: 6.3 us per tile beside the covariance update of the previous frame, 1.3 us of
        // them matrix work).  The result goes through LDS too (the region becomes the A chunk afterwards); every
        // thread takes the rows of its detections from there.
        typedef float fr_f4 __attribute__((ext_vector_type(4)));
        float* sA = reinterpret_cast<float*>(a_lds);      // [FR_KS][128]  W_sup rows, this round's slots
        float* sB = sA + FR_KS * 128;                     // [FR_KS][64]   W rows, this chunk's columns
        float* pl = sB + FR_KS * 64;                      // [128][64]     support rows of the current P
        const float* __restrict__ wsa = static_cast<const float*>(fr.wsup_prev);
        const float* __restrict__ wpb = static_cast<const float*>(fr.wprev);
        const float* __restrict__ Pf = static_cast<const float*>(fr.cov);
        const int nslots = EKF_CAM + LMD * m;
        const int lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
        const bool stp_c = fr.stamps && fr.stamps_heavy && chunk == 0 && tid == 0 && NB <= 6;
        if (stp_c) fr.stamps[41] = wall_clock64();
        for (int round = 0; 128 * round < nslots; ++round) {
            const int tile_l = g >> 1, tile_i = 4 * round + tile_l, jl = 32 * (g & 1);
            const bool tile_ok = 32 * tile_i < nslots;
            fr_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            float pt[16];
            unsigned dmask = 0;
#pragma unroll
            for (int kb = 0; kb < NB; kb += FR_KS / 16) {
                constexpr int NA = FR_KS * 32 / FR_T, NBL = FR_KS * 16 / FR_T;      // 16-byte loads per thread
                const int rows = min(FR_KS, fr.kpad - 16 * kb);
                fr_f4 va[NA], vb[NBL];
#pragma unroll
                for (int n = 0; n < NA; ++n) {
                    const int e = tid + FR_T * n, row = min(e >> 5, rows - 1), c4 = e & 31;
                    const int col = min(128 * round + 4 * c4, fr.wsup_ld - 4);
                    va[n] = *reinterpret_cast<const fr_f4*>(wsa + (int64_t)(16 * kb + row) * fr.wsup_ld + col);
                }
#pragma unroll
                for (int n = 0; n < NBL; ++n) {
                    const int e = tid + FR_T * n, row = min(e >> 4, rows - 1), c4 = e & 15;
                    vb[n] = *reinterpret_cast<const fr_f4*>(wpb + (int64_t)(16 * kb + row) * fr.ldw + chunk0 + 4 * c4);
                }
                if (kb == 0) {
                    // C / D layout: register reg <-> tile row (reg & 3) + 8 (reg >> 2) + 4 lhi, column l31
                    // (unconditional -- a tile beyond the last slot reads row 0: inside a branch the wait for the stage's
                    // operands, which are older, became a wait for these scattered rows as well: 7 us per round at n=4096)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int slot = 32 * tile_i + (reg & 3) + 8 * (reg >> 2) + 4 * lhi;
                        const int sj = min(max((slot - EKF_CAM) / LMD, 0), m - 1);
                        const int prw = (slot < EKF_CAM) ? slot : (slot < nslots ? lmc[sj] + (slot - EKF_CAM) % LMD : 0);
                        pt[reg] = Pf[(int64_t)prw * ld + chunk0 + jl + l31];
                        dmask |= (unsigned)(prw == chunk0 + jl + l31) << reg;      // (diagonal entries: they get Q)
                    }
                }
                if (kb > 0) __syncthreads();                   // the previous stage has been consumed
#pragma unroll
                for (int n = 0; n < NA; ++n) *reinterpret_cast<fr_f4*>(sA + 4 * (tid + FR_T * n)) = va[n];
#pragma unroll
                for (int n = 0; n < NBL; ++n) *reinterpret_cast<fr_f4*>(sB + 4 * (tid + FR_T * n)) = vb[n];
                __syncthreads();
                FR_DIAG(1 + 4 * round + 2 * (kb / (FR_KS / 16)));
                if (kb == 0 && round == 0) jac_issue();      // (in flight during the matrix phase)
                if (stp_c && round == 0 && kb == 0) fr.stamps[45] = wall_clock64();
                if (tile_ok) {
#pragma unroll
                    for (int c = kb; c < NB && c < kb + FR_KS / 16; ++c) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int kr = 16 * (c - kb) + 2 * u + lhi;      // A: lane (i = l31, k = lhi); B: lane (k = lhi, j = l31)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-sA[kr * 128 + 32 * tile_l + l31], sB[kr * 64 + jl + l31], acc, 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);      // (the scheduler would hoist every LDS read of the stage: 16 registers per chunk)
                    }
                }
                FR_DIAG(2 + 4 * round + 2 * (kb / (FR_KS / 16)));
            }
            if (stp_c && round == 0) fr.stamps[46] = wall_clock64();
            if (tile_ok) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {      // (a diagonal entry's row is this lane's column)
                    const int rl = (reg & 3) + 8 * (reg >> 2) + 4 * lhi;
                    const float q = ((dmask >> reg) & 1u) ? (float)ekf_qdiag(chunk0 + jl + l31, fr.dims, fr.nz) : 0.0f;
                    pl[(32 * tile_l + rl) * 64 + jl + l31] = (pt[reg] + q) + acc[reg];
                }
            }
            if (stp_c && round == 0) fr.stamps[42] = wall_clock64();
            __syncthreads();
            if (stp_c && round == 0) fr.stamps[43] = wall_clock64();
            if (round == 0) {
#pragma unroll
                for (int a = 0; a < EKF_CAM; ++a) pcr[a] = (T)pl[a * 64 + cl];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int ju = min(g + NWV * u, m - 1);
#pragma unroll
                for (int d = 0; d < LMD; ++d) {
                    const int slot = EKF_CAM + LMD * ju + d;
                    if ((slot >> 7) == round) plr[u][d] = (T)pl[(slot & 127) * 64 + cl];
                }
            }
            __syncthreads();
        }
    } else {
        // f64 covariance: the same on v_mfma_f64_16x16x4_f64 (the f64 update's per-element sequence: v = P; on the
        // diagonal v += Q; v += acc), 64 slots per round: wave g computes the 16-slot tile row g / 2 for the
        // column tiles 2 (g & 1) and 2 (g & 1) + 1; W staged 48 rows at a time.
        typedef double fr_d2 __attribute__((ext_vector_type(2)));
        double* sA = a_lds;                           // [FR_KS64][64]
        double* sB = sA + FR_KS64 * 64;               // [FR_KS64][64]
        double* pl = sB + FR_KS64 * 64;               // [64][64]
        const double* __restrict__ wsa = static_cast<const double*>(fr.wsup_prev);
        const double* __restrict__ wpb = static_cast<const double*>(fr.wprev);
        const double* __restrict__ Pd = static_cast<const double*>(fr.cov);
        const int nslots = EKF_CAM + LMD * m;
        const int lane = tid & 63, lc = lane & 15, lg = lane >> 4;
        const int trow = g >> 1, ct0 = 2 * (g & 1);
        for (int round = 0; 64 * round < nslots; ++round) {
            const bool tile_ok = 64 * round + 16 * trow < nslots;
            pf64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            double pt[2][4];
            int prw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int slot = 64 * round + 16 * trow + lg + 4 * r;
                const int sj = min(max((slot - EKF_CAM) / LMD, 0), m - 1);
                prw[r] = (slot < EKF_CAM) ? slot : (slot < nslots ? lmc[sj] + (slot - EKF_CAM) % LMD : 0);
            }
            for (int kb0 = 0; kb0 < fr.kpad; kb0 += FR_KS64) {
                constexpr int NL = FR_KS64 * 32 / FR_T;         // 16-byte loads per thread and operand
                const int rows = min(FR_KS64, fr.kpad - kb0);
                fr_d2 va[NL], vb[NL];
#pragma unroll
                for (int n = 0; n < NL; ++n) {
                    const int e = tid + FR_T * n, row = min(e >> 5, rows - 1), c2 = e & 31;
                    const int col = min(64 * round + 2 * c2, fr.wsup_ld - 2);
                    va[n] = *reinterpret_cast<const fr_d2*>(wsa + (int64_t)(kb0 + row) * fr.wsup_ld + col);
                    vb[n] = *reinterpret_cast<const fr_d2*>(wpb + (int64_t)(kb0 + row) * fr.ldw + chunk0 + 2 * c2);
                }
                if (kb0 == 0 && tile_ok) {
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) pt[t2][r] = Pd[(int64_t)prw[r] * ld + chunk0 + 16 * (ct0 + t2) + lc];
                }
                if (kb0 > 0) __syncthreads();
#pragma unroll
                for (int n = 0; n < NL; ++n) {
                    *reinterpret_cast<fr_d2*>(sA + 2 * (tid + FR_T * n)) = va[n];
                    *reinterpret_cast<fr_d2*>(sB + 2 * (tid + FR_T * n)) = vb[n];
                }
                __syncthreads();
                if (kb0 == 0 && round == 0) jac_issue();
                if (tile_ok) {
                    for (int s4 = 0; s4 < (rows >> 2); ++s4) {      // A[i = c][k = g], B[k = g][j = c]
                        const double a = -sA[(4 * s4 + lg) * 64 + 16 * trow + lc];
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sB[(4 * s4 + lg) * 64 + 16 * ct0 + lc], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sB[(4 * s4 + lg) * 64 + 16 * ct0 + 16 + lc], acc1, 0, 0, 0);
                    }
                }
            }
            if (tile_ok) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {      // C / D layout: column c, row g + 4 r
                        const int colg = chunk0 + 16 * (ct0 + t2) + lc;
                        double v = pt[t2][r];
                        if (prw[r] == colg) v += ekf_qdiag(prw[r], fr.dims, fr.nz);
                        v += t2 ? acc1[r] : acc0[r];
                        pl[(16 * trow + lg + 4 * r) * 64 + 16 * (ct0 + t2) + lc] = v;
                    }
            }
            __syncthreads();
            if (round == 0) {
#pragma unroll
                for (int a = 0; a < EKF_CAM; ++a) pcr[a] = (T)pl[a * 64 + cl];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int ju = min(g + NWV * u, m - 1);
#pragma unroll
                for (int d = 0; d < LMD; ++d) {
                    const int slot = EKF_CAM + LMD * ju + d;
                    if ((slot >> 6) == round) plr[u][d] = (T)pl[(slot & 63) * 64 + cl];
                }
            }
            __syncthreads();
        }
    }
    // (pipelined mode: requested above, taken only now -- the support rows depend on nothing this launch computes, and
    // fetched "while the first stage is in flight" the Jacobian made the matrix phase wait ~3 us for the measurement
    // workgroup; fetched here as a whole it was 1.6 us of polling and bulk on the way to the A chunk)
    FR_DIAG(10);
    if (!fix) jac_issue();
    jac_finish();
    __syncthreads();
    FR_DIAG(11);
    FrPre<NB> pre;
    if (fr.stamps && fr.stamps_heavy && chunk == 0 && tid == 0 && NB <= 6) fr.stamps[14] = wall_clock64();
    // (the support rows become doubles BEFORE the substitution's first requests go out: they may come straight from
    // memory, and a wait for them placed after those requests would wait for the requests too)
    double pc[EKF_CAM], pld[NU][LMD];
#pragma unroll
    for (int a = 0; a < EKF_CAM; ++a) pc[a] = (double)pcr[a] + ((a == c) ? ekf_qdiag(a, fr.dims, fr.nz) : 0.0);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int c0 = lmc[min(g + NWV * u, m - 1)];
#pragma unroll
        for (int d = 0; d < LMD; ++d) pld[u][d] = (double)plr[u][d] + ((c0 + d == c) ? fr.nz.q_lm : 0.0);
    }
    asm volatile("" ::: "memory");
    if (g < 4) fr_panel_pre<NB, MODEL>(fr, pre, smask_l, g, chunk0 + 16 * g, tid & 63);
    if (fr.stamps && fr.stamps_heavy && chunk == 0 && tid == 0 && NB <= 6) fr.stamps[15] = wall_clock64();
    // (hs is only read and the A chunk only written here: without `restrict` every row's Jacobian reads waited for the
    // previous row's store)
    const double* __restrict__ hsr = hs;
    double* __restrict__ aw = a_lds;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = g + NWV * u;
        if (j < m) {
            const double (&pl)[LMD] = pld[u];
#pragma unroll
            for (int d = 0; d < RD; ++d) {
                const int r = RD * j + d;
                const double* __restrict__ hr = hsr + r * JC;
                double acc = 0.0;
#pragma unroll
                for (int a = 0; a < EKF_CAM; ++a) acc = __builtin_fma(hr[a], pc[a], acc);
#pragma unroll
                for (int e = 0; e < LMD; ++e) acc = __builtin_fma(hr[10 + e], pl[e], acc);
                aw[r * FR_ALD + cl] = acc;
                if (fr.wdbg) fr.amat[(int64_t)r * fr.lda + c] = acc;
            }
        }
    }
    for (int r = fr.k + g; r < fr.kpad; r += NWV) aw[r * FR_ALD + cl] = 0.0;
    __syncthreads();
    if (fr.stamps && chunk == 0 && tid == 0) fr.stamps[33] = wall_clock64();
    int spin_fail = 0;
    if (g < 4) {
        const int lane = tid & 63, col0 = chunk0 + 16 * g;
        fr_panel<T, NB, MODEL>(fr, a_lds, ekf_lds_flags(flag + 1), pshare, g, col0, lane, spin_fail, pre);
        if (spin_fail && (tid & 63) == 0) ekf_raise(fr, EKF_ST_TIMEOUT);
    } else if constexpr (FrPre<NB>::STAGED) {
        fr_panel_stage<NB>(fr, a_lds, ekf_lds_flags(pshare), g - 4, tid & 63);
    }
    if (MODEL == 0) return;
    // ---- EKF_Rotations: dx of this chunk has to be in memory before the chunk counts as done;
    // the last chunk injects (ekf_with_rotations.py:142-177): camera and every landmark: xyz
    // additive, quaternion multiplicative (scalar first); landmark error states are never written
    // dx travels through memory between chunks on different XCDs: the counter increment is a RELEASE
    // (this chunk's dx stores are performed at agent scope before it) and an ACQUIRE (the last chunk
    // sees every other chunk's dx), i.e. the memory model's own guarantee, not a timing assumption
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned long long old =
            __hip_atomic_fetch_add(fr.done_ctr, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = (old + 1 == fr.done_target);
    }
    __syncthreads();
    if (!flag[0]) return;
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
        if (fr.state_host) {      // (landmark error states are never written: zero in the mirror as on the device)
            double* sh = fr.state_host + c0;
            sh[0] = x0; sh[1] = x1; sh[2] = x2;
            for (int e = 0; e < 4; ++e) sh[3 + e] = q[e];
            if (i == 0)
                for (int e = 0; e < 3; ++e) sh[7 + e] = 0.0;
        }
        if (i == 0) {
            for (int e = 0; e < 3; ++e) st[7 + e] = 0.0;
            if (fr.traj_row) {
                fr.traj_row[0] = x0; fr.traj_row[1] = x1; fr.traj_row[2] = x2;
                for (int e = 0; e < 4; ++e) fr.traj_row[3 + e] = q[e];
            }
        }
    }
}

// One instantiation per number of 16-row blocks NB = kpad / 16: the forward substitution is unrolled
// over NB, and a kernel that carried all twelve variants spilled registers.
template <typename T, int NU, int MODEL, int NB>
__global__ __launch_bounds__(FR_T) void ekf_front_kernel(EkfFrame fr) {
    extern __shared__ __attribute__((aligned(16))) double fr_sm[];
    const int nb = fr.kpad / EKF_RB, nS = nb * (nb + 1) / 2;
    const int bx = blockIdx.x;
    if (bx == 0) fr_role_measure<MODEL>(fr, fr_sm);
    else if (bx <= nS) fr_role_sblock<T, MODEL, NB>(fr, bx - 1, nS, fr_sm);
    else if (bx == nS + 1) fr_role_factor<NB>(fr, fr_sm);
    else fr_role_chunk<T, NU, MODEL, NB>(fr, bx - nS - 2, fr_sm);
}

template <typename T, int NU, int MODEL, int NB>
static void ekf_front_go(const EkfFrame& fr, hipStream_t s) {
    constexpr int LMD = EkfModel<MODEL>::LMD, JC = EkfModel<MODEL>::JC;
    constexpr int NSLOT = EKF_CAM + LMD * EkfModel<MODEL>::NDET16;
    static bool once = false;
    if (!once) {   // > 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ekf_front_kernel<T, NU, MODEL, NB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    const int nb = fr.kpad / EKF_RB, nS = nb * (nb + 1) / 2;
    const bool fix = fr.wprev != nullptr;
    const size_t dots_b = sizeof(T) == 4 ? (size_t)(32 * ((NSLOT + 31) / 32)) * (32 * ((NSLOT + 31) / 32)) * 4
                                         : (size_t)(16 * ((NSLOT + 15) / 16)) * (16 * ((NSLOT + 15) / 16)) * 8;
    const size_t lds_s = ((size_t)fr.k * JC + NSLOT * 16 + 32 + fr.kpad) * 8 + 16 + (fix ? dots_b : 0);
    const size_t lds_c = ((size_t)fr.k * JC + fr_chunk_a_len(fr.kpad, fix, (int)sizeof(T))) * 8 + 72 * 4 + 8 * 64 * 8 + 64 * 8 + 64 * 4 + 16;
    const size_t lds_f = (size_t)sv_lds_doubles_any(NB) * 8;
    size_t lds = lds_s > lds_c ? lds_s : lds_c;
    if (lds_f > lds) lds = lds_f;
    if ((size_t)fr.lds_min > lds) lds = (size_t)fr.lds_min;
    hipLaunchKernelGGL((ekf_front_kernel<T, NU, MODEL, NB>), dim3(nS + 2 + fr.ncols / 64), dim3(FR_T), lds, s, fr);
}

template <typename T>
void ekf_launch_front(const EkfFrame& fr, hipStream_t s) {
    const int nb = fr.kpad / EKF_RB;
#define FR_GO(NU, MODEL, NB) case NB: return ekf_front_go<T, NU, MODEL, NB>(fr, s);
    if (fr.model == 1) {                 // EKF_Rotations: m <= 27, k = 7 m
        switch (nb) {
            FR_GO(4, 1, 1) FR_GO(4, 1, 2) FR_GO(4, 1, 3) FR_GO(4, 1, 4) FR_GO(4, 1, 5) FR_GO(4, 1, 6)
            FR_GO(4, 1, 7) FR_GO(4, 1, 8) FR_GO(4, 1, 9) FR_GO(4, 1, 10) FR_GO(4, 1, 11)
            default: return ekf_front_go<T, 4, 1, 12>(fr, s);
        }
    }
    if (fr.m <= 32) {                    // k = 3 m <= 96
        switch (nb) {
            FR_GO(4, 0, 1) FR_GO(4, 0, 2) FR_GO(4, 0, 3) FR_GO(4, 0, 4) FR_GO(4, 0, 5)
            default: return ekf_front_go<T, 4, 0, 6>(fr, s);
        }
    }
    switch (nb) {                        // m = 33 .. 64: k = 99 .. 192
        FR_GO(8, 0, 7) FR_GO(8, 0, 8) FR_GO(8, 0, 9) FR_GO(8, 0, 10) FR_GO(8, 0, 11)
        default: return ekf_front_go<T, 8, 0, 12>(fr, s);
    }
#undef FR_GO
}
