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

// Flag words in LDS that waves poll.  The accesses have to be volatile -- and a volatile access through a GENERIC
// pointer stays a flat access (the address-space inference leaves volatile accesses alone): `flat_load_dword ... sc0
// sc1` + `s_waitcnt vmcnt(0)`, so every poll also waited for every global store the wave had in flight (the
// write-through publication stores of the factorisation, the W rows of the substitution: 0.5 - 1 us per poll).
// Through an LDS-qualified pointer the same access is a ds_read_b32 that only counts in lgkmcnt.
typedef __attribute__((address_space(3))) int ekf_lds_int;
__device__ __forceinline__ volatile ekf_lds_int* ekf_lds_flags(void* p) { return (volatile ekf_lds_int*)p; }

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

// ============================================================================================
// Blocked Cholesky of S (k <= 192) by ONE workgroup of 8 waves, everything in registers.
// Shared by the stand-alone solve kernel (ekf_small_kernels.hip) and the factorisation role of the
// fused front kernel (ekf_front_impl.h): same code, same bits; they differ only in the IO policy.
//
// OP layout of a 16 x 16 block B in one wave: lane l = 16 g + c holds B[c][g + 4 r] in register r
// (r = 0..3).  It is at once
//     * the A operand of v_mfma_f64_16x16x4 for the k-chunk r          (lane (i = c, k = g)),
//     * the B operand of the same instruction for B^T                  (lane (k = g, j = c)),
//     * the accumulator (C) layout of B^T                              (lane (col = c, row = g + 4 r)),
// so a block that is kept as "OP layout of itself" can be multiplied, accumulated and handed to the
// consumers (`lop` / `dop` ARE this layout) without a single cross-lane move or LDS transpose:
//     panel   OP(L_ib)   = sum_r mfma(OP(X)[r], OP(S'_ib)[r])                 (L_ib = S'_ib X^T)
//     update  OP(S'_itc) -= sum_r mfma(OP(L_tc,b)[r], OP(L_ib)[r])            (S'_itc -= L_ib L_tc,b^T)
// In memory (exchange buffer / `sblk`): word (r >> 1) 128 + 2 l + (r & 1) of the block's 256 words:
// two 16-byte accesses per lane, each instruction covers 1 KB of whole cache lines.
//
// Ownership: wave w keeps row blocks w and w + 8 (row block NB is the residual row: z - h rides
// along, its factor row is y = L^-1 (z - h)) with all their blocks (i, tc), tc <= i, in registers.
// Per block column b (ONE barrier; everything else is ordered by per-row flags in LDS):
//   phase 1  owner of row b : chain on the finished diagonal block -> X = L_bb^-1 (OP layout = `dop`
//                             order) into LDS, published;   everyone: the trailing updates of step
//                             b - 1 that are not urgent (block columns >= b + 1)
//   barrier
//   phase 2  every row i > b: L_ib by 4 MFMAs, into LDS (the other rows need it as an operand; flag)
//                             and published (`lop` order = the registers as they are)
//   phase 3  every row i > b: the urgent updates, block columns b + 1 and b + 2 (so that the next chain
//                             and the next panel start from finished blocks, and the SIMD partner of
//                             the next chain has nothing left to do).  Row b + 1 -- the next chain --
//                             needs nothing but its own block and goes straight on.
// The chain itself (sv_chain): LDL^T by 16 rank-1 updates in the vector ALU, operands moved by DPP row
// broadcasts and cross-row swaps only (see there).  The square roots are taken once at the end.
// (Round 1: lane = row, v_readlane broadcasts, 15 - j updates per pivot: ~310 cycles per pivot on
// every row of the block column, plus an LDS round trip per left-looking term: ~6k cycles per block
// column at k = 96.)
// ============================================================================================
#define SV_T 512
#define SV_NW (SV_T / 64)

__device__ __forceinline__ size_t sv_lop_index(int i, int q) { return (size_t)(i * (i - 1) / 2 + q) * 256; }
__device__ __forceinline__ size_t sv_blk_index(int i, int tc) { return (size_t)(i * (i + 1) / 2 + tc) * 256; }
__host__ __device__ constexpr int sv_lds_doubles(int nb) { return 512 + (nb + 1) * 256 + 64; }     // X (two slots) | L_ib of every row | flags

typedef double sv_d2 __attribute__((ext_vector_type(2)));
// 1 / d for a wave-uniform d: v_rcp_f64 (about 2^-23) and one cubic step (error ~ e^3)
__device__ __forceinline__ double sv_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r0, 1.0);
    const double p = __builtin_fma(e, e, e);
    return __builtin_fma(r0, p, r0);
}

// value of lane J of each 16-lane row, in every lane of that row (one v_mov_b64_dpp row_newbcast)
template <int J> __device__ __forceinline__ double sv_row_bcast(double v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, false);
}
// the values of 16-lane group G0, in all four groups (gfx950 cross-row swaps; semantics checked by
// tools/probes/permlane_probe.hip:  permlane16_swap(D, S) -> D' = [D.r0 S.r0 D.r2 S.r2], S' = [D.r1 S.r1 D.r3 S.r3];
// permlane32_swap(D, S) -> D' = [D.lo32 S.lo32], S' = [D.hi32 S.hi32])
template <int G0> __device__ __forceinline__ int sv_group_bcast32(int v) {
    auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const int x = (G0 & 1) ? p[1] : p[0];
    auto q = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (G0 & 2) ? q[1] : q[0];
}
template <int G0> __device__ __forceinline__ double sv_group_bcast(double v) {
    return __hiloint2double(sv_group_bcast32<G0>(__double2hiint(v)), sv_group_bcast32<G0>(__double2loint(v)));
}

// The 16-pivot chain: LDL^T of one 16 x 16 block by rank-1 updates, entirely in the vector ALU of one
// wave (on MI355X an f64 MFMA costs as many cycles as the vector FMAs it replaces -- 64 cycles per
// 16x16x4 -- so a rank-1 update through the matrix pipe would waste three quarters of it).
// C layout: lane (c, g), register rr <-> M[g + 4 rr][c].  Pivot J (R = J / 4, G0 = J % 4):
//     M[i][c] -= M[i][J] * M[J][c] / d_J
//   * column J by row index, M[g + 4 rr][J], sits in lane (J, g): a row broadcast (DPP) of register rr;
//   * row J by column index, M[J][c], sits in register R of group G0: a group broadcast (two swaps per dword);
//   * U = (L'^-1)^T is carried along:  U[i][c] -= U[i][J] * L'[c][J]  needs the same two kinds of operand
//     and no other; at the end U in C layout IS L'^-1 in OP layout, i.e. `dop` order: no transpose.
// Only columns c > J are touched (exact zeros above the diagonal of L'^-1, row J of it stays intact).
// the values of 16-lane group G0 in all four groups through the LDS crossbar (ds_bpermute: two instructions
// to issue, ~80 cycles until the data is back -- time the chain spends on other instructions; the swap
// version above costs the same ~75 cycles, but as ISSUE time)
template <int G0> __device__ __forceinline__ double sv_group_bcast_async(double v, int c) {
    const int addr = 4 * (16 * G0 + c);
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}

// The 16-pivot chain: LDL^T of one 16 x 16 block by rank-1 updates, entirely in the vector ALU of one
// wave (on MI355X an f64 MFMA costs as many cycles as the vector FMAs it replaces -- 64 cycles per
// 16x16x4 -- so a rank-1 update through the matrix pipe would waste three quarters of it).
// C layout: lane (c, g), register rr <-> M[g + 4 rr][c].  Pivot J (R = J / 4, G0 = J % 4):
//     M[i][c] -= M[i][J] * M[J][c] / d_J          for c > J
//   * column J by row index, M[g + 4 rr][J], sits in lane (J, g): a row broadcast of register rr, folded
//     into the FMA (v_fmac_f64 with a DPP source, row_newbcast);
//   * row J by column index, M[J][c], sits in register R of group G0 and is needed in all four groups: a
//     cross-group broadcast.  It is taken ONE PIVOT AHEAD (row J + 1 before update J, corrected by one FMA
//     with the scalar M[J+1][J]), so its latency overlaps the FMAs of pivot J;
//   * U = (L'^-1)^T is carried along:  U[i][c] -= U[i][J] * L'[c][J]  needs the same two kinds of operand
//     and no other; at the end U in C layout IS L'^-1 in OP layout, i.e. `dop` order: no transpose.
// Only columns c > J are touched (exact zeros above the diagonal of L'^-1, row J of it stays intact).
// Measured (tools/probes/chain_probe.hip, one wave): see DESIGN.md section 9.
struct SvChain {
    sf64x4 m;        // the block (C layout) on entry; destroyed
    sf64x4 lp;       // debug: OP layout of L' (unit lower)
    sf64x4 dr;       // debug: pivots by g + 4 rr
    int dlo, dhi;    // pivot d_J in lane J (the 16 lanes of row group 0), as two dwords (v_writelane)
    int bad;
};
template <int J, bool DBG>
__device__ __forceinline__ void sv_pivot(SvChain& s, sf64x4& u, double& rowv, int c, int g) {
    constexpr int R = J >> 2, G0 = J & 3;
    constexpr int J1 = (J + 1) & 15, R1 = J1 >> 2, G1 = J1 & 3;
    double nxt = 0.0, a1 = 0.0;
    if (J < 15) {                                                    // row J + 1 as it is BEFORE update J
        nxt = sv_group_bcast_async<G1>(s.m[R1], c);
        a1 = ekf_readlane_f64(s.m[R1], 16 * G1 + J);                 // M[J + 1][J]
    }
    const double d = ekf_readlane_f64(rowv, J);                      // M[J][J]
    const double r = sv_rcp(d);
    // (the chain is bound by its instruction count: the pivot goes into lane J with two v_writelane instead of a
    // compare and four moves / selects, and its sign is checked once at the end, for all 16 at a time)
    asm("v_writelane_b32 %0, %1, %2" : "+v"(s.dlo) : "s"(__double2loint(d)), "n"(J));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(s.dhi) : "s"(__double2hiint(d)), "n"(J));
    const double lc = rowv * r;                                      // L'[c][J] (c >= J; 1 on the diagonal)
    double nl = (c > J) ? -lc : 0.0;                                 // strictly below the pivot: row / column J themselves stay
    if (DBG) {
        s.lp[R] = (g == G0) ? lc : s.lp[R];
        s.dr[R] = (g == G0) ? d : s.dr[R];
    }
    if (J == 15) return;
    // x += row_bcast<J>(x) * nl in ONE instruction each (v_fmac_f64 takes a DPP source; hipcc emits a copy, a
    // v_mov_b64_dpp and a v_fma_f64 instead).  s_nop: the data hazards of DPP reads (a VGPR written by the
    // vector ALU needs two wait states before a DPP instruction reads it) are not tracked across inline asm.
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %0, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %2, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %3, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %4, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %5, %5, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %6, %6, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %7, %7, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(s.m[0]), "+v"(s.m[1]), "+v"(s.m[2]), "+v"(s.m[3]), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]),
                   "+v"(nl)       // (formally an output too: the row update below -- the first use of the broadcast that was
                                  // requested above -- must stay BEHIND the eight FMAs, or hipcc hoists it, and the wait for
                                  // the LDS crossbar with it, in front of them: then only the reciprocal overlaps that latency)
                 : "n"(J));
    rowv = __builtin_fma(a1, nl, nxt);                               // M[J + 1][c] after update J (c > J)
}
// In: s.m.  Out: xop = OP layout of X = L^-1 (`dop` order), s.dcol (and s.lp, s.dr with DBG); returns != 0
// if a pivot was not positive.
template <bool DBG>
__device__ __forceinline__ int sv_chain_t(SvChain& s, sf64x4& xop, int c, int g) {
    s.bad = 0;
    s.dlo = 0;
    s.dhi = 0x3ff00000;
    sf64x4 u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        u[r] = (g + 4 * r == c) ? 1.0 : 0.0;
        s.lp[r] = 0.0;
        s.dr[r] = 1.0;
    }
    double rowv = sv_group_bcast<0>(s.m[0]);                         // row 0
    sv_pivot<0, DBG>(s, u, rowv, c, g);   sv_pivot<1, DBG>(s, u, rowv, c, g);   sv_pivot<2, DBG>(s, u, rowv, c, g);
    sv_pivot<3, DBG>(s, u, rowv, c, g);   sv_pivot<4, DBG>(s, u, rowv, c, g);   sv_pivot<5, DBG>(s, u, rowv, c, g);
    sv_pivot<6, DBG>(s, u, rowv, c, g);   sv_pivot<7, DBG>(s, u, rowv, c, g);   sv_pivot<8, DBG>(s, u, rowv, c, g);
    sv_pivot<9, DBG>(s, u, rowv, c, g);   sv_pivot<10, DBG>(s, u, rowv, c, g);  sv_pivot<11, DBG>(s, u, rowv, c, g);
    sv_pivot<12, DBG>(s, u, rowv, c, g);  sv_pivot<13, DBG>(s, u, rowv, c, g);  sv_pivot<14, DBG>(s, u, rowv, c, g);
    sv_pivot<15, DBG>(s, u, rowv, c, g);
    // X = D^-1/2 L'^-1: row c of the OP layout is scaled by 1 / sqrt(d_c)
    const double dcol = sv_group_bcast<0>(__hiloint2double(s.dhi, s.dlo));      // d_c in every lane with lane & 15 == c
    s.bad = !(dcol > 0.0);
    const double rs = ekf_rsqrt_f64(dcol);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) xop[rr] = u[rr] * rs;
    return __any(s.bad);
}
// ONE copy of the chain's code per kernel (a real call, arguments and results in registers): the front kernel runs
// every piece of its straight-line code once per launch, i.e. from a cold instruction cache (120 - 150 KB of code at
// k = 96, a 64 KB cache shared by two CUs), and the chain is the serial part everything else waits for.  Inlined, each
// of the NB chains was a fresh 4.4 KB of code.  As a function the first call warms the cache for the others -- and
// sv_factor makes that first call on an idle wave while the S blocks are still on their way.  (The callee starts with
// `s_waitcnt vmcnt(0)`, as every function does: free, as long as the calling wave has no store in flight -- see PUB.)
struct SvChainRes {
    sf64x4 xop;
    int bad;
};
__device__ inline __attribute__((noinline)) SvChainRes sv_chain_shared(sf64x4 m, int lane) {
    SvChain s;
    s.m = m;
    SvChainRes res;
    res.bad = sv_chain_t<false>(s, res.xop, lane & 15, lane >> 4);
    return res;
}

// One S block from its LDS tile (row-major [16][17]; on a diagonal block the lower triangle is
// authoritative and is mirrored, so that the block arrives bitwise symmetric) into OP memory order.
// One wave; every store instruction covers 1 KB of whole cache lines.  COH: write-through (sc1) stores
// for the in-launch exchange of the fused front kernel.
template <bool COH>
__device__ __forceinline__ void sv_sblock_emit(double* dst, const double* tile, bool diag, int lane) {
    const int c = lane & 15, g = lane >> 4;
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = g + 4 * r;
        v[r] = (diag && col > c) ? tile[col * 17 + c] : tile[c * 17 + col];
    }
    const sv_d2 a = {v[0], v[1]}, b = {v[2], v[3]};
    double* p0 = dst + 2 * lane;
    double* p1 = dst + 128 + 2 * lane;
    if (COH) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p0), "v"(a) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p1), "v"(b) : "memory");
    } else {
        *reinterpret_cast<sv_d2*>(p0) = a;
        *reinterpret_cast<sv_d2*>(p1) = b;
    }
}

// block operand from / to LDS (word r 64 + lane: conflict-free)
__device__ __forceinline__ sf64x4 sv_lds_get(const double* p, int lane) {
    return sf64x4{p[lane], p[64 + lane], p[128 + lane], p[192 + lane]};
}
__device__ __forceinline__ void sv_lds_put(double* p, const sf64x4& v, int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r * 64 + lane] = v[r];
}
__device__ __forceinline__ sf64x4 sv_mm(const sf64x4& a, const sf64x4& b) {          // OP(a) x OP(b)^T ...
    sf64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], b[r], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ void sv_mm_sub(sf64x4& acc, const sf64x4& a, const sf64x4& b) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[r], b[r], acc, 0, 0, 0);
}

// IO policy of the stand-alone solve kernel: S blocks and the residual from the gather launch
// (plain loads), the factor to lop / dop / yvec / dinv (plain stores).  The fused front kernel has
// its own policy (exchange buffer, sentinels, tags) in ekf_front_impl.h.
struct SvIoPlain {
    const EkfFrame& fr;
    __device__ __forceinline__ sf64x4 load_block(int i, int tc, int lane) const {
        const double* base = fr.sblk + sv_blk_index(i, tc);
        const sv_d2 a = *reinterpret_cast<const sv_d2*>(base + 2 * lane);
        const sv_d2 b = *reinterpret_cast<const sv_d2*>(base + 128 + 2 * lane);
        return sf64x4{a[0], a[1], b[0], b[1]};
    }
    __device__ __forceinline__ sf64x4 load_resid(int tc, int g) const {
        sf64x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = EKF_RB * tc + g + 4 * r;
            const double rv = fr.resid[min(row, fr.k - 1)];
            v[r] = (row < fr.k) ? rv : 0.0;
        }
        return v;
    }
    // block columns [C0, C1) of the wave's rows i0 / i1 (block (i, tc) exists for tc <= min(i, NB - 1); row NB is the
    // residual, replicated in all 16 rows of its "block" so that it rides through the same instructions: all of it
    // comes with C0 == 0)
    template <int NB, int N0, int N1, int C0, int C1>
    __device__ __forceinline__ void load_cols(sf64x4 (&z0)[N0], sf64x4 (&z1)[N1], int i0, int i1, bool has0, bool has1,
                                              int lane, int g) {
#pragma unroll
        for (int tc = 0; tc < N0; ++tc) {
            if (has0 && i0 == NB && C0 == 0 && tc < NB) z0[tc] = load_resid(tc, g);
            else if (has0 && i0 < NB && tc >= C0 && tc < C1 && tc <= i0) z0[tc] = load_block(i0, tc, lane);
        }
#pragma unroll
        for (int tc = 0; tc < N1; ++tc) {
            if (has1 && i1 == NB && C0 == 0 && tc < NB) z1[tc] = load_resid(tc, g);
            else if (has1 && i1 < NB && tc >= C0 && tc < C1 && tc <= i1) z1[tc] = load_block(i1, tc, lane);
        }
    }
    // (ekf_solve_big.h) complete blocks of S / the residual, and a finished block of the factor read back: -L_iq in `lop` order
    __device__ __forceinline__ sf64x4 fetch_block(int i, int tc, int lane) const { return load_block(i, tc, lane); }
    __device__ __forceinline__ sf64x4 fetch_resid(int tc, int g) const { return load_resid(tc, g); }
    __device__ __forceinline__ sf64x4 hist_block(int i, int q, int lane) const {
        const double* base = fr.lop + sv_lop_index(i, q) + lane;
        return sf64x4{ekf_ldc(base), ekf_ldc(base + 64), ekf_ldc(base + 128), ekf_ldc(base + 192)};
    }
    __device__ __forceinline__ void put_dinv(int b, const sf64x4& xop, int lane) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            fr.dop[(size_t)(b * 4 + r) * 64 + lane] = xop[r];
            fr.dinv[(size_t)(EKF_RB * b + (lane & 15)) * EKF_RB + (lane >> 4) + 4 * r] = xop[r];
        }
    }
    __device__ __forceinline__ void put_l(int i, int b, const sf64x4& y, int lane) {
#pragma unroll
        for (int r = 0; r < 4; ++r) fr.lop[sv_lop_index(i, b) + r * 64 + lane] = -y[r];
    }
    __device__ __forceinline__ void put_y(int b, double yv, int c, bool active) {
        if (active) fr.yvec[EKF_RB * b + c] = yv;
    }
};

// The factorisation.  `lds`: sv_lds_doubles(NB) doubles.  Every wave of the workgroup calls it.
template <int NB, class IO>
__device__ __forceinline__ void sv_factor(const EkfFrame& fr, IO& io, double* lds, int& bad, int& badcol) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    constexpr int N0 = NB < 8 ? NB : 8;             // blocks a row i <= 7 can have
    constexpr bool TWO = NB >= 8;                   // rows 8 .. NB exist: second row per wave
    constexpr int N1 = TWO ? NB : 1;
    double* xbuf0 = lds;                            // [2][256]       OP(X) of block column b in slot b & 1
    double* ybuf = lds + 512;                       // [NB + 1][256]  OP(L_ib) of the current block column
    // yflag[i] = b + 1 once row i's L_ib of block column b is in ybuf (the rows publish at slightly different
    // times; a hardware barrier there would make the next chain wait for the slowest panel)
    volatile ekf_lds_int* yflag = ekf_lds_flags(lds + 512 + (NB + 1) * 256);
    if (tid <= NB) yflag[tid] = 0;
    __syncthreads();
    auto wait_row = [&](int row, int b) {           // row's block of column b is in ybuf
        int it = 0;
        while (yflag[row] < b + 1) {
            if (++it > (1 << 22)) break;            // (cannot happen: the publisher never waits for this wave)
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto post_row = [&](int row, int b) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) yflag[row] = b + 1;
    };
    // Row blocks 2 s and 2 s + 1 live on the two waves of SIMD s (waves w and w + 4 of a workgroup share a
    // SIMD, and an f64 MFMA issued by one of them stalls the other's f64 vector instructions: a partner busy
    // with trailing updates slowed the chain 2.5x, tools/probes/simd_share_probe.hip).  With this pairing
    // the partner of the wave that runs chain b is either finished (b odd: row b - 1) or has nothing left to
    // do (b even: row b + 1, whose blocks are all covered by the urgent updates below).
    const int i0 = 2 * (wave & 3) + (wave >> 2), i1 = i0 + 8;
    const bool has0 = i0 <= NB, has1 = TWO && i1 <= NB;
    // Up to NB = 6 the last wave owns no row: it does ALL the publishing (Dinv, -L, y: from the LDS copies the other
    // waves leave anyway), so that no wave that computes ever has a global store in flight.  A wait for ANYTHING that
    // counts in vmcnt -- and hipcc places conservative `s_waitcnt vmcnt(0)` wherever a register with a formally
    // pending load is reused, e.g. at the top of the next chain -- also waits for every older store of the wave, and
    // the acknowledgement of a write-through store takes 1 - 2k cycles beside the covariance update's traffic: with
    // every row publishing its own block, that was the larger part of the hand-over between two chains.
    constexpr bool PUB = NB <= 6;
    const bool is_pub = PUB && wave == SV_NW - 1;
    sf64x4 z0[N0], z1[N1];
    // optional time stamps (debug): [0] start, [1] blocks in registers, [2 + 2 b] chain phase of block column b
    // over (barrier), [3 + 2 b] panel + urgent update over; [47 + 2 b], [48 + 2 b] the chain alone (b < 4)
    long long* stp = (fr.stamps && fr.stamps_heavy && lane == 0) ? fr.stamps : nullptr;
    // Inside the loop the stamps stay in registers and are stored at the very end: a stamp that is stored at once is a
    // global store, and the wave that took it later waits for its acknowledgement -- it perturbs what it measures.
    constexpr int NTS = NB <= 6 ? NB : 1;
    long long ts_cs[NTS], ts_ce[NTS], ts_bar[NTS], ts_pan[NTS], ts_ph3[NTS];
#pragma unroll
    for (int q = 0; q < NTS; ++q) ts_cs[q] = ts_ce[q] = ts_bar[q] = ts_pan[q] = ts_ph3[q] = 0;
    if (stp && wave == 0) stp[0] = clock64();
    // (light stamps: taken by the publishing wave, whose stores delay nobody)
    long long* stl = (fr.stamps && !fr.stamps_heavy && is_pub && lane == 0) ? fr.stamps : nullptr;
    if (stl) stl[0] = clock64();
    if (wave == SV_NW - 1) {      // warm the instruction cache (per CU) with the chain's code: identity block, result unused
        sf64x4 idm;
#pragma unroll
        for (int r = 0; r < 4; ++r) idm[r] = (g + 4 * r == c) ? 1.0 : 0.0;
        const SvChainRes warm = sv_chain_shared(idm, lane);
        asm volatile("" ::"v"(warm.xop[0]), "v"(warm.bad));
    }
    // block column 0 first (all the first chain and the first panel need); the other block columns come in
    // beside the first chain
    io.template load_cols<NB, N0, N1, 0, 1>(z0, z1, i0, i1, has0, has1, lane, g);
    if (stp && wave == 0) stp[1] = clock64();
    if (stp && NB <= 10) stp[24 + wave] = clock64();            // (each wave: its blocks are in registers)
    sf64x4 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};      // OP(L_ib) of the wave's rows, column b
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int owner = ((b & 7) >> 1) + 4 * (b & 1);
        // ---- phase 1
        if (wave == owner) {
            SvChain s;
            s.m = (b < 8) ? z0[b < N0 ? b : 0] : z1[b < N1 ? b : 0];
            sf64x4 xop;
            if (stp && b < NTS) ts_cs[b < NTS ? b : 0] = clock64();
            int badnow;
            if (fr.wdbg) badnow = sv_chain_t<true>(s, xop, c, g);      // (tests: also L' and the pivots)
            else {
                const SvChainRes res = sv_chain_shared(s.m, lane);
                xop = res.xop;
                badnow = res.bad;
            }
            if (stp && b < NTS) ts_ce[b < NTS ? b : 0] = clock64();
            if (badnow && !bad) badcol = 100 + b;
            bad |= badnow;
            sv_lds_put(xbuf0 + (b & 1) * 256, xop, lane);
            if (!PUB) io.put_dinv(b, xop, lane);
            if (fr.wdbg) {                            // dense L for tests only: L_bb = L' D^1/2
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = g + 4 * r;
                    const double sq = s.dr[r] * ekf_rsqrt_f64(s.dr[r]);
                    fr.lmat[(size_t)(EKF_RB * b + c) * fr.ldl + EKF_RB * b + col] = (c >= col) ? s.lp[r] * sq : 0.0;
                }
            }
        }
        if (b == 0) io.template load_cols<NB, N0, N1, 1, NB>(z0, z1, i0, i1, has0, has1, lane, g);
        if (b >= 1) {
            // the trailing updates of step b - 1 that were not urgent: block columns b + 2 .. of the wave's rows
#pragma unroll
            for (int tc = b + 2; tc < N0; ++tc)
                if (has0 && tc <= min(i0, NB - 1)) {
                    wait_row(tc, b - 1);
                    sv_mm_sub(z0[tc], sv_lds_get(ybuf + tc * 256, lane), y0);
                }
#pragma unroll
            for (int tc = b + 2; tc < N1; ++tc)
                if (has1 && tc <= min(i1, NB - 1)) {
                    wait_row(tc, b - 1);
                    sv_mm_sub(z1[tc], sv_lds_get(ybuf + tc * 256, lane), y1);
                }
        }
        __syncthreads();
        if (stp && b < NTS) ts_bar[b < NTS ? b : 0] = clock64();
        if (stl) stl[2 + 2 * b] = clock64();
        // ---- phase 2: panel
        {
            const sf64x4 xop = sv_lds_get(xbuf0 + (b & 1) * 256, lane);
            if (has0 && i0 > b) {
                y0 = sv_mm(xop, z0[b < N0 ? b : 0]);
                sv_lds_put(ybuf + i0 * 256, y0, lane);
                post_row(i0, b);
            }
            if (has1 && i1 > b) {
                y1 = sv_mm(xop, z1[b < N1 ? b : 0]);
                sv_lds_put(ybuf + i1 * 256, y1, lane);
                post_row(i1, b);
            }
        }
        if (stp && b < NTS) ts_pan[b < NTS ? b : 0] = clock64();
        // publication / emission: -L blocks in `lop` order (the registers as they are), y of the residual row
        if (!PUB) {
            if (has0 && i0 > b) {
                if (i0 < NB) io.put_l(i0, b, y0, lane);
                else io.put_y(b, (c >> 2) == 0 ? y0[0] : (c >> 2) == 1 ? y0[1] : (c >> 2) == 2 ? y0[2] : y0[3], c, g == (c & 3));
            }
            if (has1 && i1 > b) {
                if (i1 < NB) io.put_l(i1, b, y1, lane);
                else io.put_y(b, (c >> 2) == 0 ? y1[0] : (c >> 2) == 1 ? y1[1] : (c >> 2) == 2 ? y1[2] : y1[3], c, g == (c & 3));
            }
        } else if (is_pub) {
            // (X_b stays in its slot until chain b + 2 is over and L_ib in its slot until the panel of step b + 1: both
            // come after the next barrier, which this wave only reaches when it has read them)
            io.put_dinv(b, sv_lds_get(xbuf0 + (b & 1) * 256, lane), lane);
#pragma unroll
            for (int i = b + 1; i <= NB; ++i) {
                wait_row(i, b);
                const sf64x4 yi = sv_lds_get(ybuf + i * 256, lane);
                if (i < NB) io.put_l(i, b, yi, lane);
                else io.put_y(b, (c >> 2) == 0 ? yi[0] : (c >> 2) == 1 ? yi[1] : (c >> 2) == 2 ? yi[2] : yi[3], c, g == (c & 3));
            }
        }
        if (fr.wdbg) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (has0 && i0 > b && i0 < NB) fr.lmat[(size_t)(EKF_RB * i0 + c) * fr.ldl + EKF_RB * b + g + 4 * r] = y0[r];
                if (has1 && i1 > b && i1 < NB) fr.lmat[(size_t)(EKF_RB * i1 + c) * fr.ldl + EKF_RB * b + g + 4 * r] = y1[r];
            }
        }
        // ---- phase 3: the urgent updates, block columns b + 1 and b + 2 (the rest waits for the next chain to
        // run beside it).  Row b + 1 -- the next chain -- needs only its own L_(b+1)b and goes straight on; the
        // other rows wait for the one or two blocks they need, not for every panel.
#pragma unroll
        for (int du = 1; du <= 2; ++du) {
            const int tc = b + du;
            if (tc < NB) {
                if (has0 && i0 > b && tc < N0 && tc <= min(i0, NB - 1)) {
                    if (i0 == tc) sv_mm_sub(z0[tc < N0 ? tc : 0], y0, y0);
                    else {
                        wait_row(tc, b);
                        sv_mm_sub(z0[tc < N0 ? tc : 0], sv_lds_get(ybuf + tc * 256, lane), y0);
                    }
                }
                if (has1 && i1 > b && tc <= min(i1, NB - 1)) {
                    if (i1 == tc) sv_mm_sub(z1[tc < N1 ? tc : 0], y1, y1);
                    else {
                        wait_row(tc, b);
                        sv_mm_sub(z1[tc < N1 ? tc : 0], sv_lds_get(ybuf + tc * 256, lane), y1);
                    }
                }
            }
        }
        if (stp && b < NTS) ts_ph3[b < NTS ? b : 0] = clock64();
    }
    if (stp) {      // [2 + 2 b] barrier, [3 + 2 b] end of step (wave 0); chain start / end (its owner); [16 + b] panel posted (row b + 1)
#pragma unroll
        for (int b = 0; b < NTS; ++b) {
            if (wave == 0) {
                stp[2 + 2 * b] = ts_bar[b];
                stp[3 + 2 * b] = ts_ph3[b];
            }
            if (wave == ((b & 7) >> 1) + 4 * (b & 1)) {
                stp[(b < 4 ? 47 : 48) + 2 * b] = ts_cs[b];
                stp[(b < 4 ? 48 : 49) + 2 * b] = ts_ce[b];
            }
            if (has0 && i0 == b + 1) stp[16 + b] = ts_pan[b];
        }
    }
}
