// Blocked Cholesky of S for FEW block columns (NB <= 6, k <= 96 rows: the headline shapes), one workgroup of 8 waves, with ONE
// wave running every pivot chain.  Same arithmetic as sv_factor (ekf_solve_device.h) -- every block receives the same MFMA
// sequences in the same order, so the factor has the same bits -- but a different choreography:
//
//   sv_factor      the chain of block column b runs on the wave that owns row b.  Between two chains: LDS put of X_b -> hardware
//                  barrier (every wave has to arrive) -> the owner of row b + 1 gets X_b from LDS -> its panel (4 dependent f64
//                  MFMAs) -> its diagonal update (4 more) -> call into the chain.  1.3 - 2.2k cycles, of which the eight MFMAs
//                  and one LDS hop are ~0.9k: the rest is the barrier and the SIMD partner's 12 MFMAs (the panel and urgent
//                  updates of ITS row) issued into the same matrix pipe; the partner also slows the chain itself (3.3 - 3.6k
//                  cycles in the kernel, 2.64k alone on a CU).
//   here           wave 0 runs all chains and, between two of them, the panel and the diagonal update of row b + 1 ITSELF, from
//                  X_b in its own registers: row b + 1's owner has left P = S'_(b+1)b and D = S'_(b+1)(b+1), complete up to block
//                  column b - 1, in LDS while chain b was running.  No barrier anywhere (flags only), no foreign MFMA on the
//                  chain's SIMD: its partner, wave 4, does all the publishing (LDS -> memory) and nothing else.  Waves 1, 2, 3,
//                  5, 6, 7 are the workers: row i belongs to worker i mod 6 (row NB is the residual z - h, replicated in the 16
//                  rows of its "block"; its factor row is y), every block of the row in registers, right-looking updates.
//
// Nothing in LDS is ever overwritten -- X_b of every block column, L_ib of every row and column, the hand-over blocks of every
// row have their own place (123 KB at NB = 6) -- so that the only synchronisation is read-after-write:
//   xready      = b + 1 once X_b is in xbuf[b]                         (chain wave -> workers, publisher)
//   yflag[i]    = b + 1 once L_ib is in ybuf[i][b]                     (row i's owner, or the chain wave for i = b + 1)
//   dready[i]   = 1 once P / D of row i are in dep[i]                  (row i's owner -> chain wave)
// Critical path per block column: chain -> (P, D from LDS) -> 4 MFMAs -> 4 MFMAs -> chain.
#pragma once
#include <type_traits>
#include "ekf_solve_device.h"

#define SVC_WORKERS 6

__host__ __device__ constexpr int svc_lds_doubles(int nb) {
    // X_b [nb] | L_ib [nb + 1][nb] | hand-over blocks P, D [nb][2] | flags (2 nb + 2 ints)
    return nb * 256 + (nb + 1) * nb * 256 + nb * 512 + (2 * nb + 2 + 1) / 2 + 8;
}

// (bounded: a poster that never comes would be a bug, not a hang -- the caller then reports the factorisation as failed)
__device__ __forceinline__ bool svc_wait_ok(volatile ekf_lds_int* word, int want) {
    int it = 0;
    bool ok = true;
    while (*word < want) {
        if (++it > (1 << 22)) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}
__device__ __forceinline__ void svc_post(volatile ekf_lds_int* word, int value, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) *word = value;
}
// The chain wave's posts: no wait for the block it has just written.  The LDS unit executes the ds instructions of ONE
// wave in the order they were issued, so the flag write cannot overtake the data writes in front of it (a release fence
// would be an `s_waitcnt lgkmcnt(0)`: ~100 cycles twice per block column on the critical path); the compiler barrier
// keeps the program order, the readers acquire as usual.
__device__ __forceinline__ void svc_post_in_order(volatile ekf_lds_int* word, int value, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) *word = value;
    asm volatile("" ::: "memory");
}

// f(integral_constant<int, B>) for B = B0 .. N - 1 while it returns true (block column indices have to be compile-time
// constants: they index register arrays)
template <int B0, int N, class F>
__device__ __forceinline__ void svc_static_for(F&& f) {
    if constexpr (B0 < N) {
        if (f(std::integral_constant<int, B0>{})) svc_static_for<B0 + 1, N>(f);
    }
}

template <int NB, class IO>
__device__ __forceinline__ void sv_factor_cw(const EkfFrame& fr, IO& io, double* lds, int& bad, int& badcol) {
    static_assert(NB >= 1 && NB <= 6, "block columns");
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* xbuf = lds;                                          // [NB][256]          OP(X_b)
    double* ybuf = xbuf + NB * 256;                              // [NB + 1][NB][256]  OP(L_ib)
    double* dep = ybuf + (NB + 1) * NB * 256;                    // [NB][2][256]       P = S'_i(i-1), D = S'_ii of row i
    volatile ekf_lds_int* xready = ekf_lds_flags(dep + NB * 512);
    volatile ekf_lds_int* yflag = xready + 1;                    // [NB + 1]
    volatile ekf_lds_int* dready = yflag + (NB + 1);             // [NB]
    for (int e = tid; e < 2 * NB + 2; e += SV_T) xready[e] = 0;
    __syncthreads();
    const bool is_chain = wave == 0, is_pub = wave == 4;
    // a wait that ran out: the factorisation counts as failed (status bit through the caller), block column code 200 + flag
    auto svc_wait = [&](volatile ekf_lds_int* word, int want) {
        if (!svc_wait_ok(word, want)) {
            if (!bad) badcol = 200 + (int)(word - xready);
            bad = 1;
        }
    };
    auto yb = [&](int i, int b) { return ybuf + ((size_t)i * NB + b) * 256; };
    // (light stamps, taken by the publishing wave, whose stores delay nobody: [0] start, [2 + 2 b] X_b seen)
    long long* stl = (fr.stamps && !fr.stamps_heavy && is_pub && lane == 0) ? fr.stamps : nullptr;
    if (stl) stl[0] = clock64();

    if (is_chain) {
        {   // warm the instruction cache with the chain's code while the S blocks are on their way: identity block, result unused
            sf64x4 idm;
#pragma unroll
            for (int r = 0; r < 4; ++r) idm[r] = (g + 4 * r == c) ? 1.0 : 0.0;
            const SvChainRes warm = sv_chain_shared(idm, lane);
            asm volatile("" ::"v"(warm.xop[0]), "v"(warm.bad));
        }
        svc_wait(dready, 1);
        sf64x4 d = sv_lds_get(dep + 256, lane);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            sf64x4 xop;
            int badnow;
            if (fr.wdbg) {                                       // (tests: also the dense L_bb = L' D^1/2)
                SvChain s;
                s.m = d;
                badnow = sv_chain_t<true>(s, xop, c, g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = g + 4 * r;
                    const double sq = s.dr[r] * ekf_rsqrt_f64(s.dr[r]);
                    fr.lmat[(size_t)(EKF_RB * b + c) * fr.ldl + EKF_RB * b + col] = (c >= col) ? s.lp[r] * sq : 0.0;
                }
            } else {
                const SvChainRes res = sv_chain_shared(d, lane);
                xop = res.xop;
                badnow = res.bad;
            }
            if (badnow && !bad) badcol = 100 + b;
            bad |= badnow;
            sf64x4 p = {0.0, 0.0, 0.0, 0.0};
            if (b + 1 < NB) {                                    // (row b + 1's blocks: requested before X_b goes out, used after)
                svc_wait(dready + (b + 1), 1);
                p = sv_lds_get(dep + (b + 1) * 512, lane);
                d = sv_lds_get(dep + (b + 1) * 512 + 256, lane);
            }
            sv_lds_put(xbuf + b * 256, xop, lane);
            svc_post_in_order(xready, b + 1, lane);
            if (b + 1 < NB) {
                // row b + 1: its panel and its diagonal update here, straight into the next chain
                const sf64x4 y = sv_mm(xop, p);
                sv_mm_sub(d, y, y);
                sv_lds_put(yb(b + 1, b), y, lane);
                svc_post_in_order(yflag + (b + 1), b + 1, lane);
                if (fr.wdbg) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) fr.lmat[(size_t)(EKF_RB * (b + 1) + c) * fr.ldl + EKF_RB * b + g + 4 * r] = y[r];
                }
            }
        }
        return;
    }
    if (is_pub) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            svc_wait(xready, b + 1);
            if (stl) stl[2 + 2 * b] = clock64();
            io.put_dinv(b, sv_lds_get(xbuf + b * 256, lane), lane);
#pragma unroll
            for (int i = b + 1; i <= NB; ++i) {
                svc_wait(yflag + i, b + 1);
                const sf64x4 yi = sv_lds_get(yb(i, b), lane);
                if (i < NB) io.put_l(i, b, yi, lane);
                else io.put_y(b, (c >> 2) == 0 ? yi[0] : (c >> 2) == 1 ? yi[1] : (c >> 2) == 2 ? yi[2] : yi[3], c, g == (c & 3));
            }
        }
        return;
    }
    // ---- workers: rows i0 = widx and i1 = widx + 6 (the latter exists for worker 0 at NB = 6 only: the residual row)
    const int widx = wave < 4 ? wave - 1 : wave - 2;             // workers 0 .. 5 = waves 1, 2, 3, 5, 6, 7
    const int i0 = widx, i1 = widx + SVC_WORKERS;
    const bool has0 = i0 <= NB, has1 = i1 <= NB;
    sf64x4 z0[NB], z1[NB];
    // block column 0 first (all the first chain and the first hand-over need); the others come in beside the first chain
    io.template load_cols<NB, NB, NB, 0, 1>(z0, z1, i0, i1, has0, has1, lane, g);
    if (has0 && i0 == 0) {                                       // row 0: the first diagonal block
        sv_lds_put(dep + 256, z0[0], lane);
        svc_post(dready, 1, lane);
    }
    if (NB > 1) io.template load_cols<NB, NB, NB, 1, NB>(z0, z1, i0, i1, has0, has1, lane, g);
    if (NB > 1 && has0 && i0 == 1) {                             // row 1: nothing to wait for
        sv_lds_put(dep + 512, z0[0], lane);
        sv_lds_put(dep + 512 + 256, z0[1], lane);
        svc_post(dready + 1, 1, lane);
    }
    // one row through block column b: its panel (unless the chain wave does it), then the right-looking updates of its blocks
    // b + 1 .. ; a row that is next but one in line for the chain (i = b + 2) hands P and D over as soon as they are complete
    auto row_step = [&](sf64x4 (&z)[NB], int i, auto bc, const sf64x4& xop) {
        constexpr int b = decltype(bc)::value;
        if (i <= b) return;
        if (i == b + 1 && i < NB) return;                        // (panel and diagonal update: the chain wave; nothing else is left of this row)
        const sf64x4 y = sv_mm(xop, z[b]);
        sv_lds_put(yb(i, b), y, lane);
        svc_post(yflag + i, b + 1, lane);
        if (fr.wdbg && i < NB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) fr.lmat[(size_t)(EKF_RB * i + c) * fr.ldl + EKF_RB * b + g + 4 * r] = y[r];
        }
        svc_static_for<b + 1, NB>([&](auto tcc) {
            constexpr int tc = decltype(tcc)::value;
            if (tc > i) return false;
            if (tc == i) sv_mm_sub(z[tc], y, y);
            else {
                svc_wait(yflag + tc, b + 1);
                sv_mm_sub(z[tc], sv_lds_get(yb(tc, b), lane), y);
            }
            if constexpr (tc == b + 2) {
                if (i == tc) {                                   // next but one in line for the chain: P and D are complete
                    sv_lds_put(dep + i * 512, z[b + 1], lane);
                    sv_lds_put(dep + i * 512 + 256, z[b + 2], lane);
                    svc_post(dready + i, 1, lane);
                }
            }
            return true;
        });
    };
    svc_static_for<0, NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        if (!((has0 && i0 > b) || (has1 && i1 > b))) return false;      // nothing left for this worker
        svc_wait(xready, b + 1);
        const sf64x4 xop = sv_lds_get(xbuf + b * 256, lane);
        if (has0) row_step(z0, i0, bc, xop);
        if (has1) row_step(z1, i1, bc, xop);
        return true;
    });
}
