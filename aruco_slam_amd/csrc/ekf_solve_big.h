// Blocked Cholesky of S for MANY block columns (NB = 9 .. 24, k = 144 .. 384 rows), one workgroup of 8 waves.  Shared by the
// stand-alone solve kernel (ekf_small_kernels.hip) and the factorisation role of the fused front kernel (ekf_front_impl.h):
// same code, same bits; they differ only in the IO policy (ekf_solve_device.h: SvIoPlain, ekf_front_impl.h: SvIoFused).
//
// Reference: S^-1 = spsolve(S, I) and the products with it (extended_kalman_filter.py:124-131, ekf_with_rotations.py:
// 115-139), here S = L L^T, y = L^-1 (z - h).  The reference takes any number of detections per frame; EKF_Rotations with
// the 50 markers of DICT_5X5_50 in view has k = 350 rows (NB = 22).
//
// Why not sv_factor (ekf_solve_device.h).  That kernel keeps EVERY block of the trailing matrix in registers, one or two block
// rows per wave: 8 (8 + NB) registers per wave at NB >= 8 (13 - 99 spilled registers for NB = 9 .. 12, rounds 1 - 2), and
// at most 16 block rows at all.  The lower triangle of S is 2 KB x NB (NB + 1) / 2: 156 KB at NB = 12, 506 KB at NB = 22 --
// neither the registers nor the LDS of one CU hold it.  So here nothing but the CURRENT block columns live on chip:
//
//   * left-looking by block column with ONE column of look-ahead.  Block (i, c) of the factor is
//         L_ic = (S_ic - sum_{q < c} L_iq L_cq^T) X_c^T,      X_c = L_cc^-1,
//     the sums taken in ascending q (the order of the right-looking kernel).  A row keeps two accumulators: t(c) for the
//     column that is being finished and t(c + 1) that is being summed up meanwhile, 24 registers per row with its newest
//     block L_i,c-1;
//   * the finished columns -- the HISTORY -- are read back from where they are published anyway: the exchange buffer / `lop`
//     in memory (L2).  Total re-reads NB^3 / 6 blocks of 2 KB: 0.6 MB at NB = 12, 3.5 MB at NB = 22, beside the pivot chains.
//     A column may be read once the publishing wave has seen its stores complete (`colready`, an LDS word);
//   * roles.  Wave 0 runs every 16-pivot chain (sv_chain_shared) and nothing else; wave 4 -- its SIMD partner: an f64 MFMA
//     issued there would stall the chain's vector instructions (tools/probes/simd_share_probe.hip) -- does all the PUBLISHING
//     (global stores, from the LDS copies: no computing wave ever has a store in flight, so their loads never wait for one);
//     waves 1, 2, 3, 5, 6, 7 are the WORKERS: block row i belongs to worker i mod 6 (row NB is the residual z - h, replicated
//     in the 16 rows of its "block" so that it rides through the same instructions; its factor row is y);
//   * per block column b, ONE hardware barrier:
//         chain wave : diagonal block t_bb from LDS -> chain -> X_b into LDS                                  | barrier
//         workers    : [C] sums of column b + 1 (below)                                                         | barrier
//                      [A] L_ib = t_i(b) X_b^T for their rows i > b -> LDS (for the other rows and the publisher)
//                      [B] last term of column b + 1:  t_i(b+1) -= L_ib L_b+1,b^T;  row b + 1 hands its diagonal block to
//                          the chain wave, which starts chain b + 1 at once
//                      [C] column b + 2:  t_i(b+2) = S_i,b+2 - sum_{q < b} L_iq L_b+2,q^T (history from memory; row b + 2's
//                          blocks are staged in LDS once, each worker a sixth of them) - L_ib L_b+2,b^T
//         publisher  : X_b, the L_ib and y_b of column b from LDS to memory; colready[b] when the stores are complete.
//     Critical path per column: panel block (4 MFMAs) -> diagonal update (4 MFMAs) -> LDS -> chain (~3.3k cycles) -> LDS -- for
//     the first block columns.  From b ~ 3 on the workers' [C] is longer than a chain and sets the period (NB = 12,
//     tools/chunk_stamps.py svb: 2.9 us at b = 1 growing to 5.5 - 6.5 us from b = 5 on).  One worker at b = 6, first version
//     (a request, a flag wait and four dependent MFMAs per history term): [A] 0.5, [B] 0.5, S blocks of column b + 2 0.6, six
//     history terms 3.4, last term 0.4 us.  Now the requests of a row go out together (one round trip, ~0.9 us) and the terms
//     run on four accumulators, one per k-slice: six terms 1.8 us; n=1024 m=64 pipelined 71.3 -> 65.0 us per frame, n=4096
//     m=64 367 -> 362.  (Also tried: all flags and pivot blocks of a batch before its MFMAs -- 48 more registers, slower.)
#pragma once
#include "ekf_solve_device.h"
#include "ekf_solve_cw.h"

#define SVB_WORKERS 6
#ifdef SVB_DIAG      // (diagnostic builds, tools/chunk_stamps.py svb: per-column stamps of the publishing and the chain wave in
                     // fr.stamps[0 .. 47], of worker SVB_DIAG_W during block column SVB_DIAG_B in [48 .. 55])
#define SVB_STAMP(i) do { if (fr.stamps && lane == 0) fr.stamps[i] = wall_clock64(); } while (0)
#define SVB_WSTAMP(i) do { if (fr.stamps && lane == 0 && widx == SVB_DIAG_W && b == SVB_DIAG_B) fr.stamps[48 + (i)] = wall_clock64(); } while (0)
#else
#define SVB_STAMP(i) do { } while (0)
#define SVB_WSTAMP(i) do { } while (0)
#endif

__host__ __device__ constexpr int svb_lds_doubles(int nb) {
    // X (two slots) | L_ib of every row (NB + 1) | staged history of a pivot row (NB) | diagonal block | y (16 NB) | flags (3 NB + 8 ints)
    return 512 + (nb + 1) * 256 + nb * 256 + 256 + 16 * nb + (3 * nb + 8 + 1) / 2 + 8;
}

// wait until an LDS word has reached `want` (bounded; a publisher that never comes would be a bug, not a hang)
__device__ __forceinline__ void svb_wait(volatile ekf_lds_int* word, int want) {
    int it = 0;
    while (*word < want) {
        if (++it > (1 << 22)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void svb_post(volatile ekf_lds_int* word, int value, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) *word = value;
}

// IO policy (see SvIoPlain / SvIoFused): fetch_block(i, tc, lane) / fetch_resid(tc, g): a block of S / of the residual, complete
// (the fused policy waits for the S-block workgroups); hist_block(i, q, lane): -L_iq as published (`lop` order = OP layout);
// put_dinv / put_l / put_y as in sv_factor.
template <int NB, class IO>
__device__ __forceinline__ void sv_factor_big(const EkfFrame& fr, IO& io, double* lds, int& bad, int& badcol) {
    static_assert(NB >= 2 && NB <= 24, "block columns");
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NR = NB + 1;                                   // block rows, the residual row included
    constexpr int MR = (NR + SVB_WORKERS - 1) / SVB_WORKERS;     // rows per worker
    double* xbuf = lds;                                          // [2][256]  OP(X_b) in slot b & 1
    double* ybuf = lds + 512;                                    // [NR][256] OP(L_ib) of the current block column
    double* prow = ybuf + NR * 256;                              // [NB][256] -L_pq of the pivot row p = b + 2, q < b
    double* dblk = prow + NB * 256;                              // [256]     diagonal block for the chain wave
    double* yvec = dblk + 256;                                   // [16 NB]   y so far (the residual row's history)
    volatile ekf_lds_int* yflag = ekf_lds_flags(yvec + 16 * NB); // [NR] b + 1 once row i's block of column b is in ybuf
    volatile ekf_lds_int* pflag = yflag + NR;                    // [NB] p once block q of pivot row p is in prow
    volatile ekf_lds_int* colready = pflag + NB;                 // [NB] 1 once column q is in memory
    volatile ekf_lds_int* dready = colready + NB;                // b + 1 once the diagonal block of column b is in dblk
    for (int e = tid; e < 3 * NB + 2; e += SV_T) yflag[e] = 0;
    __syncthreads();
    const bool is_chain = wave == 0, is_pub = wave == 4;
    const int widx = wave < 4 ? wave - 1 : wave - 2;             // workers 0 .. 5 = waves 1, 2, 3, 5, 6, 7

    if (is_chain) {
        for (int b = 0; b < NB; ++b) {
            svb_wait(dready, b + 1);
            SVB_STAMP(24 + b);
            sf64x4 m = sv_lds_get(dblk, lane);
            sf64x4 xop;
            int badnow;
            if (fr.wdbg) {                                       // (tests: also the dense L_bb = L' D^1/2)
                SvChain s;
                s.m = m;
                badnow = sv_chain_t<true>(s, xop, c, g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = g + 4 * r;
                    const double sq = s.dr[r] * ekf_rsqrt_f64(s.dr[r]);
                    fr.lmat[(size_t)(EKF_RB * b + c) * fr.ldl + EKF_RB * b + col] = (c >= col) ? s.lp[r] * sq : 0.0;
                }
            } else {
                const SvChainRes res = sv_chain_shared(m, lane);
                xop = res.xop;
                badnow = res.bad;
            }
            if (badnow && !bad) badcol = 100 + b;
            bad |= badnow;
            sv_lds_put(xbuf + (b & 1) * 256, xop, lane);
            __syncthreads();                                     // X_b is there
        }
        return;
    }
    if (is_pub) {
        for (int b = 0; b < NB; ++b) {
            __syncthreads();
            io.put_dinv(b, sv_lds_get(xbuf + (b & 1) * 256, lane), lane);
            for (int i = b + 1; i <= NB; ++i) {
                svb_wait(yflag + i, b + 1);
                const sf64x4 yi = sv_lds_get(ybuf + i * 256, lane);
                if (i < NB) {
                    io.put_l(i, b, yi, lane);
                    if (fr.wdbg) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) fr.lmat[(size_t)(EKF_RB * i + c) * fr.ldl + EKF_RB * b + g + 4 * r] = yi[r];
                    }
                } else {
                    io.put_y(b, (c >> 2) == 0 ? yi[0] : (c >> 2) == 1 ? yi[1] : (c >> 2) == 2 ? yi[2] : yi[3], c, g == (c & 3));
                }
            }
            SVB_STAMP(12 + b);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the column is in memory: it may be read back as history
            SVB_STAMP(b);
            svb_post(colready + b, 1, lane);
        }
        return;
    }
    // ---- workers
    sf64x4 tcur[MR], tnxt[MR], ynew[MR];
    auto row_of = [&](int s) { return widx + SVB_WORKERS * s; };
    // S_i,col of this worker's rows i >= col (the residual row: z - h of that block), all requests first
    auto fetch_column = [&](sf64x4 (&dst)[MR], int col) {
#pragma unroll
        for (int s = 0; s < MR; ++s) {
            const int i = row_of(s);
            if (i >= col && i <= NB) dst[s] = (i == NB) ? io.fetch_resid(col, g) : io.fetch_block(i, col, lane);
        }
    };
    fetch_column(tcur, 0);
    if (NB > 1) fetch_column(tnxt, 1);
    if (widx == 0) {                                             // row 0: the first diagonal block
        sv_lds_put(dblk, tcur[0], lane);
        svb_post(dready, 1, lane);
    }
    for (int b = 0; b < NB; ++b) {
        SVB_WSTAMP(7);                                           // (the previous column's [C] is over)
        __syncthreads();                                         // X_b is there (and every reader of the previous column's LDS copies is done)
        SVB_WSTAMP(0);
        const sf64x4 xop = sv_lds_get(xbuf + (b & 1) * 256, lane);
        // [A] the panel: L_ib = t_i(b) X_b^T
#pragma unroll
        for (int s = 0; s < MR; ++s) {
            const int i = row_of(s);
            if (i > b && i <= NB) {
                ynew[s] = sv_mm(xop, tcur[s]);
                sv_lds_put(ybuf + i * 256, ynew[s], lane);
                if (i == NB && g == (c & 3))                     // y_b for the residual row's history
                    yvec[EKF_RB * b + c] = (c >> 2) == 0 ? ynew[s][0] : (c >> 2) == 1 ? ynew[s][1] : (c >> 2) == 2 ? ynew[s][2] : ynew[s][3];
                svb_post(yflag + i, b + 1, lane);
            }
        }
        SVB_WSTAMP(1);
        if (b + 1 >= NB) break;
        // [B] the last term of column b + 1; its diagonal block goes to the chain wave
#pragma unroll
        for (int s = 0; s < MR; ++s) {
            const int i = row_of(s);
            if (i >= b + 1 && i <= NB) {
                if (i == b + 1) {
                    sv_mm_sub(tnxt[s], ynew[s], ynew[s]);
                    sv_lds_put(dblk, tnxt[s], lane);
                    svb_post(dready, b + 2, lane);
                } else {
                    svb_wait(yflag + (b + 1), b + 1);
                    sv_mm_sub(tnxt[s], sv_lds_get(ybuf + (b + 1) * 256, lane), ynew[s]);
                }
                tcur[s] = tnxt[s];
            }
        }
        const int p = b + 2;                                     // [C] column p = b + 2, beside chain b + 1
        SVB_WSTAMP(2);
        if (p >= NB) continue;
        fetch_column(tnxt, p);
        SVB_WSTAMP(3);
        if (b > 0) {
            // History terms q < b: every request of a batch -- the worker's share of the pivot row, then up to HD blocks of a
            // row -- goes out before the first result is used (one memory round trip per batch; first version: one per term)
            svb_wait(colready + (b - 1), 1);                       // (the publisher posts the columns in order)
            SVB_WSTAMP(8);
            constexpr int PV = (NB - 2 + SVB_WORKERS - 1) / SVB_WORKERS;
            sf64x4 pv[PV];
#pragma unroll
            for (int n = 0; n < PV; ++n) pv[n] = io.hist_block(p, min(widx + SVB_WORKERS * n, b - 1), lane);
            bool staged = false;
            auto stage = [&]() {
#pragma unroll
                for (int n = 0; n < PV; ++n) {
                    const int q = widx + SVB_WORKERS * n;
                    if (q < b) {
                        sv_lds_put(prow + q * 256, pv[n], lane);
                        svb_post(pflag + q, p, lane);
                    }
                }
                staged = true;
            };
            constexpr int HD = 6;
#pragma unroll
            for (int s = 0; s < MR; ++s) {
                const int i = row_of(s);
                if (i < p || i > NB) continue;
                // Four accumulators, one per k-slice of the 16 x 16 x 16 product, summed at the end: with one accumulator every
                // term is four DEPENDENT f64 MFMAs (~200 cycles each with two workers per SIMD: 0.37 us per term measured), with
                // four the terms of a row are issue-bound.
                sf64x4 hacc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) hacc[r] = sf64x4{0.0, 0.0, 0.0, 0.0};
                auto term = [&](const sf64x4& pivb, const sf64x4& ownb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) hacc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pivb[r], ownb[r], hacc[r], 0, 0, 0);
                };
                if (i == NB) {                                     // the residual row: its history is y, in LDS
                    if (!staged) stage();
                    for (int q = 0; q < b; ++q) {
                        sf64x4 own;
#pragma unroll
                        for (int r = 0; r < 4; ++r) own[r] = -yvec[EKF_RB * q + g + 4 * r];
                        svb_wait(pflag + q, p);
                        term(sv_lds_get(prow + q * 256, lane), own);
                    }
                } else {
                    for (int q0 = 0; q0 < b; q0 += HD) {
                        sf64x4 own[HD];
#pragma unroll
                        for (int d = 0; d < HD; ++d) own[d] = io.hist_block(i, min(q0 + d, b - 1), lane);
                        if (!staged) { stage(); SVB_WSTAMP(9); }
#pragma unroll
                        for (int d = 0; d < HD; ++d) {
                            if (q0 + d < b) {
                                svb_wait(pflag + (q0 + d), p);
                                if (d == 0 && q0 == 0) SVB_WSTAMP(10);
                                term(sv_lds_get(prow + (q0 + d) * 256, lane), own[d]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) tnxt[s][r] += (hacc[0][r] + hacc[1][r]) + (hacc[2][r] + hacc[3][r]);
            }
            if (!staged) stage();
        }
        SVB_WSTAMP(4);
        SVB_WSTAMP(5);
        // ... and the term of the column that has just been finished (q = b), from the LDS copies
#pragma unroll
        for (int s = 0; s < MR; ++s) {
            const int i = row_of(s);
            if (i >= p && i <= NB) {
                if (i == p) sv_mm_sub(tnxt[s], ynew[s], ynew[s]);
                else {
                    svb_wait(yflag + p, b + 1);
                    sv_mm_sub(tnxt[s], sv_lds_get(ybuf + p * 256, lane), ynew[s]);
                }
            }
        }
        SVB_WSTAMP(6);
    }
}

// NB <= 6: one chain wave, no barriers (ekf_solve_cw.h); 7, 8: every block in registers, the chain on the row's owner
// (sv_factor); beyond: the streamed version above
template <int NB, class IO>
__device__ __forceinline__ void sv_factor_any(const EkfFrame& fr, IO& io, double* lds, int& bad, int& badcol) {
    if constexpr (NB <= 6) sv_factor_cw<NB>(fr, io, lds, bad, badcol);
    else if constexpr (NB <= 8) sv_factor<NB>(fr, io, lds, bad, badcol);
    else sv_factor_big<NB>(fr, io, lds, bad, badcol);
}
__host__ __device__ constexpr int sv_lds_doubles_any(int nb) {
    return nb <= 6 ? svc_lds_doubles(nb) : nb <= 8 ? sv_lds_doubles(nb) : svb_lds_doubles(nb);
}
