// Device helpers of the blocked Cholesky shared by the stand-alone solve kernel
// (ekf_small_kernels.hip) and the fused front kernel (ekf_front_impl.h).  gfx950 only.
#pragma once
#include "ekf_kernels.h"

// Cross-workgroup exchange inside ONE launch (fused front kernel): data goes through agent-scope
// relaxed atomics (global_load/store ... sc1: coherent across the 8 XCD L2s without any cache
// flush) and readiness is carried by the data itself: an exchange buffer holds EKF_SENT (a NaN
// payload no computation produces) until the producer overwrites it.
#define EKF_SENT_BITS 0xFFFBADC0FFFBADC0ull
#define EKF_SENT_WORD 0xFFFBADC0u
#define EKF_SPIN_MAX 400000
__device__ __forceinline__ double ekf_sent() { return __longlong_as_double((long long)EKF_SENT_BITS); }
__device__ __forceinline__ bool ekf_is_sent(double v) { return __double_as_longlong(v) == (long long)EKF_SENT_BITS; }
__device__ __forceinline__ double ekf_ldc(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ekf_stc(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH> __device__ __forceinline__ double ekf_ldx(const double* p) { return COH ? ekf_ldc(p) : *p; }

typedef double sf64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double ekf_readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double ekf_rsqrt_f64(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = __builtin_fma(-h * y, y, 0.5);
        y = __builtin_fma(y, e, y);
    }
    return y;
}

// --------------------------------------------------------------------------
// solve, column-streaming blocked Cholesky (kpad <= 192).  8 waves, one workgroup.
//   * S block columns live in LDS: all of them (kpad <= 128) or a ring; a finished block column b is
//     emitted in MFMA-operand order (-L, `lop`) -- the layout the panel kernel reads anyway -- and, in
//     ring mode, is read back from there by the later left-looking updates;
//   * per block column b:
//       wave 0      : diagonal block (b,b): last left-looking term, then the 16-pivot chain as 16
//                     rank-1 MFMA updates in registers (sv_diag_chain: no cross-lane traffic at all),
//                     carrying I_16 along -> X = L_bb^-1 (= Dinv_b), handed to the others through LDS
//       other waves : meanwhile the last left-looking term of the blocks below the diagonal, the
//                     pre-updates of block columns b+1 and b+2, the emission of block column b-1
//       barrier
//       all waves   : panel  L_ib = S'_ib X^T  (4 MFMAs per 16x16 block; the residual rides along as
//                     row kp: its factor row is y = L^-1 (z - h))
//       barrier
//   * S itself comes finished from the gather launch (`sblk`) / the S-block workgroups (fused kernel).
// Round 1 ran the chain with lane = row on every row of the block column (v_readlane broadcasts,
// 15 - j updates per pivot): ~310 cycles per pivot, ~5k cycles per block column.  The chain below is
// latency-bound at ~150 cycles per pivot, and only the diagonal block is on it.
// --------------------------------------------------------------------------
#define SV_T 512
#define SV_CLD 18

__device__ __forceinline__ size_t sv_lop_index(int i, int q) { return (size_t)(i * (i - 1) / 2 + q) * 256; }

// ---------------------------------------------------------------------------
// C layout of a 16 x 16 block in one wave (v_mfma_f64_16x16x4 accumulator): lane = 16 g + c,
// register r <-> element [g + 4 r][c].
// ---------------------------------------------------------------------------
// value of lane J of each 16-lane row, in every lane of that row (DPP row_newbcast: stays in the VALU)
template <int J> __device__ __forceinline__ double sv_row_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + J, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + J, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Cholesky of a 16 x 16 SPD block M (bitwise symmetric, C layout) as 16 rank-1 updates:
//   pivot j:  d = M[j][j];  y = 1/sqrt(d);  l_j = M[:, j] y;  M -= l_j l_j^T;
//             x_j = T[j, :] y;  T -= l_j x_j      (T starts as I_16; the rows x_j are X = L^-1)
// Row j of the symmetric M -- register j/4 in the 16 lanes of group g = j%4 -- IS column j, and a
// vector that lives in ONE 16-lane group with zeros in the other three is a valid A operand
// (lane (i, k)) and B operand (lane (k, j')) of v_mfma_f64_16x16x4 for the rank-1 product: no
// cross-lane movement except the broadcast of d inside the group.  Entries of rows / columns that
// are already eliminated turn into rounding residue; they only ever touch eliminated rows again.
// Out: x (C layout: X[g + 4 r][c], exact zeros above the diagonal), lt (debug: lt[r] = L[c][g + 4 r],
// meaningful for c >= g + 4 r).  Returns != 0 if a pivot was not positive.
struct SvDiag { sf64x4 m, t, x, lt; };
template <int J>
__device__ __forceinline__ void sv_pivot(SvDiag& s, int g, int& bad) {
    constexpr int R = J >> 2, G0 = J & 3;
    const double d = sv_row_bcast<J>(s.m[R]);
    const bool mine = (g == G0);
    bad |= (mine && !(d > 0.0)) ? 1 : 0;
    const double y = ekf_rsqrt_f64(d);
    const double ym = mine ? y : 0.0;
    const double op = s.m[R] * ym;                 // L[c][J]        (group G0)
    const double xo = s.t[R] * ym;                 // X[J][c]        (group G0)
    s.x[R] = mine ? xo : s.x[R];
    s.lt[R] = mine ? op : s.lt[R];
    const double nop = -op;
    s.m = __builtin_amdgcn_mfma_f64_16x16x4f64(nop, op, s.m, 0, 0, 0);
    s.t = __builtin_amdgcn_mfma_f64_16x16x4f64(nop, xo, s.t, 0, 0, 0);
}
__device__ __forceinline__ int sv_diag_chain(SvDiag& s, int c, int g) {
    int bad = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s.t[r] = (g + 4 * r == c) ? 1.0 : 0.0;
        s.x[r] = 0.0;
        s.lt[r] = 0.0;
    }
    sv_pivot<0>(s, g, bad);  sv_pivot<1>(s, g, bad);  sv_pivot<2>(s, g, bad);  sv_pivot<3>(s, g, bad);
    sv_pivot<4>(s, g, bad);  sv_pivot<5>(s, g, bad);  sv_pivot<6>(s, g, bad);  sv_pivot<7>(s, g, bad);
    sv_pivot<8>(s, g, bad);  sv_pivot<9>(s, g, bad);  sv_pivot<10>(s, g, bad); sv_pivot<11>(s, g, bad);
    sv_pivot<12>(s, g, bad); sv_pivot<13>(s, g, bad); sv_pivot<14>(s, g, bad); sv_pivot<15>(s, g, bad);
    return __any(bad);
}
// diagonal block (tc, tc) of a column buffer as a FULL symmetric matrix in C layout (the buffer's
// lower triangle is authoritative: the S producer leaves zeros above the diagonal)
__device__ __forceinline__ sf64x4 sv_diag_load(const double* buf, int tc, int c, int g) {
    sf64x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = g + 4 * r;
        v[r] = buf[(size_t)(EKF_RB * tc + max(row, c)) * SV_CLD + min(row, c)];
    }
    return v;
}
// the q term of the diagonal block (tc, tc): M -= L_tq L_tq^T, operands from column buffer q
__device__ __forceinline__ sf64x4 sv_diag_term(sf64x4 m, const double* qbuf, int tc, int c, int g) {
    const double* br = qbuf + (size_t)(EKF_RB * tc + c) * SV_CLD + g;
    double bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = br[4 * r];
#pragma unroll
    for (int r = 0; r < 4; ++r) m = __builtin_amdgcn_mfma_f64_16x16x4f64(-bv[r], bv[r], m, 0, 0, 0);
    return m;
}
// X (C layout) -> LDS scratch [16][17], row-major X[j][k]
__device__ __forceinline__ void sv_x_store(const sf64x4& x, double* dscr, int c, int g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dscr[(g + 4 * r) * 17 + c] = x[r];
}
// panel: row block i (> tc) of block column tc in `cur` becomes L_i,tc = S'_i,tc X^T, in place
// (k order of the product: g + 4 r, the `dop` operand order).  Row block nb is the residual row.
__device__ __forceinline__ void sv_panel_block(double* cur, const double* dscr, int kp, int i, int c, int g) {
    const double* ar = cur + (size_t)min(EKF_RB * i + c, kp) * SV_CLD + g;      // residual block: rows alias kp
    const double* br = dscr + c * 17 + g;
    double av[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { av[r] = ar[4 * r]; bv[r] = br[4 * r]; }
    sf64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv[r], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = EKF_RB * i + g + 4 * r;
        if (row <= kp) cur[(size_t)row * SV_CLD + c] = acc[r];
    }
}

// C-in / C-out of row block i of the column buffer `tgt` (block column tc), one wave
struct SvAcc { sf64x4 t, t2; };
__device__ __forceinline__ void sv_acc_load(SvAcc& a, const double* tgt, int kp, int i, int c, int g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a.t[r] = tgt[min(EKF_RB * i + g + 4 * r, kp) * SV_CLD + c];
        a.t2[r] = 0.0;
    }
}
__device__ __forceinline__ void sv_acc_store(const SvAcc& a, double* tgt, int kp, int i, int c, int g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = EKF_RB * i + g + 4 * r;
        if (row <= kp) tgt[row * SV_CLD + c] = a.t[r] + a.t2[r];
    }
}
// one q term, operands from the LDS buffer of block column q:  -= L_iq L_tq^T
__device__ __forceinline__ void sv_term_lds(SvAcc& a, const double* qbuf, int kp, int i, int tc, int c, int g,
                                            bool second = false) {
    const double* ar = qbuf + (size_t)min(EKF_RB * i + c, kp) * SV_CLD + g;   // residual block: rows alias kp
    const double* br = qbuf + (size_t)(EKF_RB * tc + c) * SV_CLD + g;
    double av[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { av[r] = ar[4 * r]; bv[r] = br[4 * r]; }
    if (second) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[r], bv[r], a.t2, 0, 0, 0);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.t = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[r], bv[r], a.t, 0, 0, 0);
    }
}
// q in [0, q1), operands from global: lop (-L blocks) and, for the residual block, y.
// COH (fused front kernel, ring mode): the operands are this workgroup's own publication in the
// exchange buffer, read back with coherent loads; every word is checked against the sentinel and
// fetched again until it has landed (returns 1 if a bounded wait ran out).
template <bool COH = false>
__device__ __forceinline__ int sv_terms_glb(SvAcc& a, const double* __restrict__ lop,
                                            const double* __restrict__ yv, int nb, int i, int tc, int q1,
                                            int g, int lane) {
    if (q1 <= 0) return 0;
    int fail = 0;
    // operands of term q+1 are in flight while the MFMAs of term q run
    double av[4], bv[4], an[4], bn[4];
    auto fetch = [&](int q, double (&x)[4], double (&y)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x[r] = (i < nb) ? ekf_ldx<COH>(lop + sv_lop_index(i, q) + r * 64 + lane)
                            : -ekf_ldx<COH>(yv + EKF_RB * q + g + 4 * r);
            y[r] = -ekf_ldx<COH>(lop + sv_lop_index(tc, q) + r * 64 + lane);
        }
    };
    auto settle = [&](int q, double (&x)[4], double (&y)[4]) {
        if (!COH) return;
        int it = 0;
        for (;;) {
            bool pend = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) pend = pend || ekf_is_sent(x[r]) || ekf_is_sent(-x[r]) || ekf_is_sent(y[r]) || ekf_is_sent(-y[r]);
            if (!__any(pend)) break;
            if (++it > EKF_SPIN_MAX) { fail = 1; break; }
            __builtin_amdgcn_s_sleep(4);
            fetch(q, x, y);
        }
    };
    fetch(0, av, bv);
    for (int q = 0; q < q1; ++q) {
        const int qn = min(q + 1, q1 - 1);
        fetch(qn, an, bn);
        settle(q, av, bv);
        if (q & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a.t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv[r], a.t2, 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) a.t = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv[r], a.t, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { av[r] = an[r]; bv[r] = bn[r]; }
    }
    return fail;
}

// S entries of row block i of block column tc (rows 16i.., columns 16tc..): one wave, lane =
// (row rr = lane & 15, column group cg = lane >> 4 -> 4 columns)
// One 16x16 block (row block i, block column tc) of S as the gather launch left it in `sblk`
// ([block column][row][16], 2 KB contiguous per block): lane = (row rr = lane >> 2, 4 columns).
// Row block nb is the residual row: z - h of the block column's 16 rows.
__device__ __forceinline__ sf64x4 sv_fetch_block(const EkfFrame& fr, int kp, int nb, int i, int tc, int lane) {
    const int rr = lane >> 2, cg = lane & 3;
    sf64x4 v;
    if (i < nb) {
        const double* src = fr.sblk + ((size_t)tc * fr.sblk_rows + EKF_RB * i + rr) * 16 + 4 * cg;
#pragma unroll
        for (int x = 0; x < 4; ++x) v[x] = src[x];
    } else {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int r2 = EKF_RB * tc + 4 * cg + x;
            const double rv = fr.resid[min(r2, fr.k - 1)];
            v[x] = (r2 < fr.k) ? rv : 0.0;
        }
    }
    return v;
}
__device__ __forceinline__ void sv_put_block(const sf64x4& v, double* tgt, int kp, int nb, int i, int lane) {
    const int rr = lane >> 2, cg = lane & 3;
    if (i == nb && rr != 0) return;
    const int row = (i < nb) ? EKF_RB * i + rr : kp;
#pragma unroll
    for (int x = 0; x < 4; ++x) tgt[row * SV_CLD + 4 * cg + x] = v[x];
}

