// Covariance update  P <- P + Q - W^T W  for LARGE problems (n = 4096: N = 12298, k = 192, f32): one workgroup per
// 128 x 128 macro tile of the lower triangle.
//
// Reference: P <- (I - K H) P after P += Q (extended_kalman_filter.py:95-105, 155-156).
//
// Why a second kernel.  The wave-per-32x32-tile kernel (ekf_cov_update.hip) pulls both MFMA operands of every tile
// straight from the W panel: 49 KB of W per 393 kflop.  At n = 1024 the panel (1.2 MB) lives in every XCD's L2 and the
// kernel is bound by the P traffic; at n = 4096, k = 192 the panel is 9.5 MB -- more than an XCD's 4 MB L2 -- and it is
// fetched again from the Infinity Cache by every tile row (measured, profiles/r02_c5: 1.89 GB of fabric traffic per
// launch against the 0.91 GB the update needs, 330 us, 53 % of the f32 matrix peak).  The problem is MFMA-bound there
// (N^2 k = 29 GF: 185 us at 157 TF), so the operands have to reach the matrix cores without that traffic:
//   * 128 x 128 tile per workgroup, 4 waves as 2 x 2, each wave a 2 x 2 register tile of v_mfma_f32_32x32x2_f32
//     accumulators: an element of W that is staged once feeds 128 columns;
//   * W goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers), 32 rows of k at a time for the
//     tile's row block and its column block, double buffered, two chunks in flight; the k-major panel rows ARE the
//     lane-linear image the DMA writes, and MFMA operand reads are conflict-free ds_read_b32;
//   * the launch order keeps an XCD inside one 8 x 8 SUPER-TILE of macro tiles at a time (blocks b, b + 8, ... share an
//     XCD: host-built table, longest-processing-time balanced over the 8 XCDs): the 2 x 1024 panel columns a super-tile
//     needs (1.5 MB) stay in that XCD's L2 and are fetched once per 64 tiles.  Placement is a speed matter only;
//   * P is read once (lower triangle, nontemporal: it must not push W out of the L2), in the accumulator layout, in
//     four batches spread over the first chunks; the tile leaves through LDS: D to (I,J) and D^T to (J,I) as whole rows,
//     16-byte write-through stores.
// Arithmetic: per element the k-ascending fma chain from zero and v = (P + q) + acc, exactly the sequence of
// ekf_cov_update_mfma_f32 / ekf_cov_update_valu: bitwise the same results (tests/test_hip_parity.py).
#include <hip/hip_ext.h>
#include "ekf_kernels.h"

typedef float cm_f32x16 __attribute__((ext_vector_type(16)));
typedef float cm_f4 __attribute__((ext_vector_type(4)));

#ifndef CM_NW
#define CM_NW 8                         // waves per workgroup (4 or 8)
#endif
#define CM_T (64 * CM_NW)
#define CM_LD 132                       // floats per row of the epilogue image (528 B: b128 accesses stay conflict-free)
#define CM_KC 16                        // rows of W per chunk
#define CM_NBUF 5                       // chunk buffers ([A: 16 x 128 | B: 16 x 128] floats = 16 KB each): four chunks in flight
#define CM_LDS_BYTES (CM_NBUF * 2 * CM_KC * 128 * 4)   // 80 KB (>= the 67,584 B of the epilogue image): two workgroups fill a CU's 160 KB

#ifndef CM_D_DIRECT
#define CM_D_DIRECT 0                   // 1: D straight from the accumulator registers (dword stores); 0: through LDS (16-byte stores)
#endif
#ifndef CM_STORE_MODE
#define CM_STORE_MODE 2                 // 0 plain, 1 nontemporal, 2 write-through (sc1)
#endif
// timing ablations (tools/cov_macro_ab.sh; never set in the product build): bit 0 no MFMAs, bit 1 no P loads,
// bit 2 no global stores, bit 3 no W staging
#ifndef CM_ABLATE
#define CM_ABLATE 0
#endif
// Two workgroups share a CU (one wave of each per SIMD) and the matrix pipe.  Started together they run in lockstep:
// both stage, both multiply (each at half rate), both write their tile -- and the pipe idles during every epilogue
// (measured: 51k cycles of loop + 4k before + 15k after it per pair of tiles, 70 % busy).  The second workgroup of the
// FIRST round (wave slot 1 of its SIMDs) therefore starts one multiply phase late; its successors inherit the offset
// (a slot is refilled when its workgroup ends), so from then on one workgroup's loads and stores run beside the
// other's MFMAs.  Placement and slot numbers are a speed matter only.
#ifndef CM_MID
#define CM_MID 0                        // k-pair of a chunk in front of which the workgroup's barrier for the NEXT chunk sits
#endif
#ifndef CM_STAGGER
#define CM_STAGGER 24000                // cycles (one tile's 384 MFMAs per wave: 24.6k)
#endif

__device__ __forceinline__ void cm_store16(float* p, const cm_f4& v) {
    if (CM_ABLATE & 4) { asm volatile("" ::"v"(p), "v"(v)); return; }
    if (CM_STORE_MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if (CM_STORE_MODE == 1) __builtin_nontemporal_store(v, reinterpret_cast<cm_f4*>(p));
    else *reinterpret_cast<cm_f4*>(p) = v;
}
__device__ __forceinline__ void cm_store4(float* p, float v) {
    if (CM_ABLATE & 4) { asm volatile("" ::"v"(p), "v"(v)); return; }
    if (CM_STORE_MODE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (CM_STORE_MODE == 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// one 16-byte LDS-DMA per lane: 64 lanes x 16 B land at lds_wave_base + 16 lane
__device__ __forceinline__ void cm_dma16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

#include <type_traits>
template <int I, int N, class F>
__device__ __forceinline__ void cm_static_for(F&& f) {      // f(integral_constant<int, I>) for I in [I, N): indices usable in constant expressions
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cm_static_for<I + 1, N>(f);
    }
}
// Vector-memory operations complete in order.  Issue order of a wave: chunks 0 .. depth-1, then at the middle of every chunk i:
// chunk i + depth (dpw LDS-DMAs), P batch i (16 loads, i < npb).  Chunk `target` has landed once all but the operations
// issued behind it are done; `steps` = mid-chunk events that have issued theirs so far.
constexpr int cm_younger(int target, int steps, int nch, int depth, int dpw, int npb) {
    int n = 0;
    bool seen = false;
    for (int ch = 0; ch < depth && ch < nch; ++ch) {
        if (seen) n += dpw;
        if (ch == target) seen = true;
    }
    for (int i = 0; i < steps; ++i) {
        if (i + depth < nch) {
            if (seen) n += dpw;
            if (i + depth == target) seen = true;
        }
        if (i < npb && seen) n += 16;
    }
    return n;
}
template <int N> __device__ __forceinline__ void cm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KB = kpad / 16: the chunk loop is unrolled completely (<= 12 chunks), so that every wait count, P batch and buffer is static.
// NW waves per workgroup as 2 x NW/2: a wave owns 64 rows x (256 / NW) columns of the tile, i.e. 2 x TJ accumulators of
// v_mfma_f32_32x32x2_f32.  NW = 8 (two waves of the workgroup per SIMD): a workgroup that is alone in its multiply phase
// -- its CU partner staging, waiting or writing its tile -- still keeps the matrix pipe busy; a single wave per SIMD reaches
// 76 % of the issue rate at best (measured with the operands held in registers: 84 cycles per MFMA instead of 64).
template <int KB, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 2, NW / 2)))
void ekf_cov_update_macro_f32(EkfFrame fr, const uint32_t* __restrict__ tiles) {
    extern __shared__ __attribute__((aligned(16))) float cm_sm[];
    constexpr int NT = 64 * NW, WC = NW / 2, CW = 128 / WC, TJ = CW / 32;      // wave columns, columns per wave, 32-wide sub-tiles per wave
    constexpr int DPW = 16 / NW;                   // LDS-DMAs per wave and chunk (A and B together)
    constexpr int NPB = 2 * TJ;                    // P batches (16 loads each) per wave
    constexpr int DEPTH = CM_NBUF - 1;             // chunks in flight
    static_assert(NW == 8, "wait counts below are worked out for 8 waves");
    const uint32_t tl = tiles[blockIdx.x];
    if (tl == 0xFFFFFFFFu) return;
    if ((CM_ABLATE & 16) && fr.dims > 0) return;          // (dispatch cost alone)
    const int I = (int)(tl >> 16), J = (int)(tl & 0xFFFFu);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int i0 = 128 * I, j0 = 128 * J;
    const float* __restrict__ wp = static_cast<const float*>(fr.wpanel);
    const float* __restrict__ P = static_cast<const float*>(fr.cov);
    float* Pout = fr.cov_out ? static_cast<float*>(fr.cov_out) : static_cast<float*>(fr.cov);
    const int ld = (int)fr.ld, ldw = (int)fr.ldw;
    constexpr int KPAD = 16 * KB, NCH = KB;       // chunks of 16 rows

    if (CM_STAGGER > 0 && blockIdx.x < 512 && ((__builtin_amdgcn_s_getreg((31 << 11) | 4) / (NW / 4)) & 1)) {      // HW_ID[3:0]: wave slot
        const long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < CM_STAGGER) __builtin_amdgcn_s_sleep(32);
    }
#ifdef CM_STAMPS
    const long long ts0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
    long long ts1 = 0;
#endif
    // chunk c -> buffer c % 5: [A: 16 x 128 | B: 16 x 128] floats.  Wave w stages rows 2 w, 2 w + 1 of both (lanes 0-31: row r,
    // lanes 32-63: row r + 1; the k-major panel rows ARE the lane-linear image the DMA writes).
    const unsigned wlane = (unsigned)(lhi * ldw + 4 * l31);
    auto stage = [&](int c) {
        float* buf = cm_sm + (c % CM_NBUF) * (2 * CM_KC * 128);
        const int rloc = 2 * wave;
        const float* src = wp + (int64_t)(CM_KC * c + rloc) * ldw;
        if (CM_ABLATE & 8) return;
        cm_dma16(src + i0 + wlane, buf + rloc * 128);
        cm_dma16(src + j0 + wlane, buf + CM_KC * 128 + rloc * 128);
    };
    cm_f32x16 acc[2][TJ];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float pt[2][TJ][16];
    // P sub-tile (ti, tj) of this wave in the accumulator layout: register reg <-> row (reg & 3) + 8 (reg >> 2) + 4 lhi
    // (scalar row base + one 32-bit lane offset per load: no 64-bit address registers)
    const unsigned plane = (unsigned)(4 * lhi * ld + l31);
    auto load_p = [&](int batch) {
        const int ti = batch / TJ, tj = batch % TJ;
        const float* base = P + (int64_t)(i0 + 64 * wr + 32 * ti) * ld + j0 + CW * wc + 32 * tj;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            pt[ti][tj][reg] = (CM_ABLATE & 2) ? 0.0f : __builtin_nontemporal_load(base + (int64_t)((reg & 3) + 8 * (reg >> 2)) * ld + plane);
    };
#pragma unroll
    for (int c = 0; c < DEPTH && c < NCH; ++c) stage(c);
    cm_wait_vm<cm_younger(0, 0, NCH, DEPTH, DPW, NPB)>();
    __builtin_amdgcn_s_barrier();                    // chunk 0 is in LDS for every wave
#ifdef CM_STAMPS
    ts1 = __builtin_amdgcn_s_memtime();
#endif
    // One stream of k-pairs over all chunks.  Operands of k-pair kp + 1 are requested BEFORE the MFMAs of k-pair kp are issued
    // (a wave's MFMAs issue in order, 64 cycles apart: requested behind them, as the compiler schedules a plain loop, the LDS
    // round trip of every k-pair lies open).  The workgroup's barrier for chunk c + 1 ("landed for every wave"; it also says
    // that every wave is done with chunk c - 1, whose buffer the next LDS-DMAs overwrite) sits in the MIDDLE of chunk c: at
    // the chunk boundary itself nothing waits, the operand requests run on into the next buffer.
    // A: lane (i = l31, k = lhi), B: lane (k = lhi, j = l31)
    constexpr int NPAIR = CM_KC / 2, NKP = NPAIR * NCH, MID = CM_MID;
    auto pa = [&](int kp) { return cm_sm + ((kp / NPAIR) % CM_NBUF) * (2 * CM_KC * 128) + (2 * (kp % NPAIR) + lhi) * 128 + 64 * wr + l31; };
    auto pb = [&](int kp) { return cm_sm + ((kp / NPAIR) % CM_NBUF) * (2 * CM_KC * 128) + CM_KC * 128 + (2 * (kp % NPAIR) + lhi) * 128 + CW * wc + l31; };
    float ca[2], cb[TJ];
    ca[0] = pa(0)[0]; ca[1] = pa(0)[32];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) cb[tj] = pb(0)[32 * tj];
    cm_static_for<0, NCH>([&](auto c_) {
    constexpr int c = decltype(c_)::value;
#pragma unroll
    for (int u = 0; u < NPAIR; ++u) {
        const int kp = NPAIR * c + u;
        if (u == MID) {
            if constexpr (c + 1 < NCH) {
                static_assert(cm_younger(c + 1, c, NCH, DEPTH, DPW, NPB) < 64, "vmcnt has 6 bits");
                cm_wait_vm<cm_younger(c + 1, c, NCH, DEPTH, DPW, NPB)>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (c + DEPTH < NCH) stage(c + DEPTH);       // into the buffer of chunk c - 1
            }
            if (c < NPB) load_p(c);
        }
        float na[2] = {0.0f, 0.0f}, nb[TJ];
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) nb[tj] = 0.0f;
        // (the wait for THIS k-pair's operands -- requested a whole MFMA group ago -- goes in front of the next requests:
        // behind them the compiler's `s_waitcnt lgkmcnt(0)` would wait for those too)
        if (TJ == 2) asm volatile("" : "+v"(ca[0]), "+v"(ca[1]), "+v"(cb[0]), "+v"(cb[TJ - 1]));
        else asm volatile("" : "+v"(ca[0]), "+v"(ca[1]), "+v"(cb[0]));
        if ((CM_ABLATE & 32) && kp > 0) {      // (matrix pipe alone: no LDS reads)
            na[0] = ca[0]; na[1] = ca[1];
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) nb[tj] = cb[tj];
        } else if (kp + 1 < NKP) {
            na[0] = pa(kp + 1)[0]; na[1] = pa(kp + 1)[32];
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) nb[tj] = pb(kp + 1)[32 * tj];
        }
        __builtin_amdgcn_sched_barrier(0);
        const float a0 = -ca[0], a1 = -ca[1];
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
            if (CM_ABLATE & 1) {
                acc[0][tj][kp & 15] += a0 * cb[tj]; acc[1][tj][kp & 15] += a1 * cb[tj];
            } else {
                acc[0][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, cb[tj], acc[0][tj], 0, 0, 0);
                acc[1][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, cb[tj], acc[1][tj], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        ca[0] = na[0]; ca[1] = na[1];
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) cb[tj] = nb[tj];
    }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // every wave has read the last chunk: the buffers become the epilogue image
#pragma unroll
    for (int c = NCH; c < NPB; ++c) load_p(c);      // (small k: the batches no step has issued)
#ifdef CM_STAMPS
    const long long ts2 = __builtin_amdgcn_s_memtime();
#endif
    // (the P tile is used from here on: without this the compiler starts the additions below -- `pt + 0` where the
    // sub-tile cannot be diagonal -- inside the chunk loop and waits there, vmcnt(0), for the batch it has just requested)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
            asm volatile("" : "+v"(pt[ti][tj][0]), "+v"(pt[ti][tj][1]), "+v"(pt[ti][tj][2]), "+v"(pt[ti][tj][3]), "+v"(pt[ti][tj][4]),
                         "+v"(pt[ti][tj][5]), "+v"(pt[ti][tj][6]), "+v"(pt[ti][tj][7]), "+v"(pt[ti][tj][8]), "+v"(pt[ti][tj][9]),
                         "+v"(pt[ti][tj][10]), "+v"(pt[ti][tj][11]), "+v"(pt[ti][tj][12]), "+v"(pt[ti][tj][13]), "+v"(pt[ti][tj][14]),
                         "+v"(pt[ti][tj][15]));
    // v = (P + q) + acc: q on the diagonal of the matrix only (the expression of the other covariance kernels)
    const bool dtile = I == J;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
            const bool dsub = dtile && 64 * wr + 32 * ti == CW * wc + 32 * tj;
            const float qlane = dsub ? (float)ekf_qdiag(i0 + 64 * wr + 32 * ti + l31, fr.dims, fr.nz) : 0.0f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rl = (reg & 3) + 8 * (reg >> 2) + 4 * lhi;
                acc[ti][tj][reg] = (pt[ti][tj][reg] + ((rl == l31) ? qlane : 0.0f)) + acc[ti][tj][reg];
            }
        }
    float* img = cm_sm;                              // [128][CM_LD]: free since the barrier behind the loop
    constexpr int RPP = NT / 32, NPASS = 128 / RPP;  // rows per pass of the workgroup (32 lanes x 16 B per row), passes
    const int prow = tid >> 5, pcol = 4 * (tid & 31);
#if CM_D_DIRECT
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
            float* ob = Pout + (int64_t)(i0 + 64 * wr + 32 * ti) * ld + j0 + CW * wc + 32 * tj;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) cm_store4(ob + (int64_t)((reg & 3) + 8 * (reg >> 2)) * ld + plane, acc[ti][tj][reg]);
        }
#else
    // image of the tile, [row][column]: one ds_write_b32 per register (lanes = 32 consecutive columns)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                img[(64 * wr + 32 * ti + (reg & 3) + 8 * (reg >> 2) + 4 * lhi) * CM_LD + CW * wc + 32 * tj + l31] = acc[ti][tj][reg];
    __syncthreads();
    {
        cm_f4 rowv[NPASS];
#pragma unroll
        for (int it = 0; it < NPASS; ++it) rowv[it] = *reinterpret_cast<const cm_f4*>(img + (RPP * it + prow) * CM_LD + pcol);
        float* ob = Pout + (int64_t)i0 * ld + j0;
#pragma unroll
        for (int it = 0; it < NPASS; ++it) cm_store16(ob + (int64_t)(RPP * it) * ld + (unsigned)(prow * ld + pcol), rowv[it]);
    }
    if (!dtile) __syncthreads();                     // (uniform: I, J are the workgroup's)
#endif
    if (!dtile) {
    // image of the transposed tile, [column][row]: a lane holds four consecutive rows of one column: ds_write_b128
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const cm_f4 v = {acc[ti][tj][4 * rg], acc[ti][tj][4 * rg + 1], acc[ti][tj][4 * rg + 2], acc[ti][tj][4 * rg + 3]};
                *reinterpret_cast<cm_f4*>(img + (CW * wc + 32 * tj + l31) * CM_LD + 64 * wr + 32 * ti + 8 * rg + 4 * lhi) = v;
            }
    __syncthreads();
    {
        cm_f4 rowv[NPASS];
#pragma unroll
        for (int it = 0; it < NPASS; ++it) rowv[it] = *reinterpret_cast<const cm_f4*>(img + (RPP * it + prow) * CM_LD + pcol);
        float* ob = Pout + (int64_t)j0 * ld + i0;
#pragma unroll
        for (int it = 0; it < NPASS; ++it) cm_store16(ob + (int64_t)(RPP * it) * ld + (unsigned)(prow * ld + pcol), rowv[it]);
    }
    }
#ifdef CM_STAMPS
    if (fr.cov_stats && tid == 0) {
        const long long ts3 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
        // [0] start, [1] end (100 MHz ticks), [2] HW_ID, [3] XCC_ID, [4] start -> chunk 0 landed, [5] chunk loop, [6] epilogue (cycles), [7] tile
        double* tlrow = fr.amat + (int64_t)(blockIdx.x / 1500) * fr.lda + 8 * (blockIdx.x % 1500);
        tlrow[0] = (double)tr0;
        tlrow[1] = (double)tr1;
        tlrow[2] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        tlrow[3] = (double)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15);
        tlrow[4] = (double)(ts1 - ts0);
        tlrow[5] = (double)(ts2 - ts1);
        tlrow[6] = (double)(ts3 - ts2);
        tlrow[7] = (double)tl;
    }
#endif
}

// Host side: the launch order.  Block b runs on XCD b % 8 (observed placement; speed only), so the table deals whole
// super-tiles (S x S macro tiles) to the 8 XCDs, largest first to the least loaded (LPT), and lists each XCD's tiles at
// b = xcd, xcd + 8, ...  Entries beyond an XCD's list are 0xFFFFFFFF (the block exits).
int ekf_cov_macro_table(int T, int S, uint32_t* out, int capacity) {
    struct Sup { int si, sj, count; };
    const int NS = (T + S - 1) / S;
    Sup sup[4096];
    int nsup = 0;
    for (int si = 0; si < NS; ++si)
        for (int sj = 0; sj <= si; ++sj) {
            int cnt = 0;
            for (int I = si * S; I < T && I < (si + 1) * S; ++I)
                for (int J = sj * S; J < T && J < (sj + 1) * S && J <= I; ++J) ++cnt;
            if (cnt > 0 && nsup < 4096) sup[nsup++] = Sup{si, sj, cnt};
        }
    // stable insertion sort by count, descending (a few hundred entries at most)
    for (int a = 1; a < nsup; ++a) {
        const Sup x = sup[a];
        int b = a - 1;
        while (b >= 0 && sup[b].count < x.count) { sup[b + 1] = sup[b]; --b; }
        sup[b + 1] = x;
    }
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int owner[4096];
    for (int a = 0; a < nsup; ++a) {
        int x = 0;
        for (int q = 1; q < 8; ++q)
            if (load[q] < load[x]) x = q;
        owner[a] = x;
        load[x] += sup[a].count;
    }
    int maxlen = 0;
    for (int q = 0; q < 8; ++q) maxlen = load[q] > maxlen ? load[q] : maxlen;
    const int grid = 8 * maxlen;
    if (!out) return grid;
    if (grid > capacity) return -1;
    for (int b = 0; b < grid; ++b) out[b] = 0xFFFFFFFFu;
    int pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < nsup; ++a) {
        const int x = owner[a];
        for (int I = sup[a].si * S; I < T && I < (sup[a].si + 1) * S; ++I)
            for (int J = sup[a].sj * S; J < T && J < (sup[a].sj + 1) * S && J <= I; ++J)
                out[8 * (pos[x]++) + x] = ((uint32_t)I << 16) | (uint32_t)J;
    }
    return grid;
}

template <int KB>
static void cm_go(const EkfFrame& fr, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    static bool once = false;
    if (!once) {   // > 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_cov_update_macro_f32<KB, CM_NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, CM_LDS_BYTES);
        once = true;
    }
    if (e0) hipExtLaunchKernelGGL((ekf_cov_update_macro_f32<KB, CM_NW>), dim3(fr.cov_grid), dim3(CM_T), CM_LDS_BYTES, s, e0, e1, 0, fr, fr.cov_tiles);
    else hipLaunchKernelGGL((ekf_cov_update_macro_f32<KB, CM_NW>), dim3(fr.cov_grid), dim3(CM_T), CM_LDS_BYTES, s, fr, fr.cov_tiles);
}

void ekf_launch_cov_update_macro(const EkfFrame& fr, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    switch (fr.kpad / 16) {
#ifdef CM_ONLY_KB
        case CM_ONLY_KB: return cm_go<CM_ONLY_KB>(fr, s, e0, e1);
        default: return;
#else
#define CM_CASE(KB) case KB: return cm_go<KB>(fr, s, e0, e1);
        CM_CASE(1) CM_CASE(2) CM_CASE(3) CM_CASE(4) CM_CASE(5) CM_CASE(6) CM_CASE(7) CM_CASE(8) CM_CASE(9) CM_CASE(10)
        CM_CASE(11) CM_CASE(12) CM_CASE(13) CM_CASE(14) CM_CASE(15) CM_CASE(16) CM_CASE(17) CM_CASE(18) CM_CASE(19)
        CM_CASE(20) CM_CASE(21) CM_CASE(22) CM_CASE(23) CM_CASE(24)
#undef CM_CASE
        default: return;      // (check_config keeps kpad <= 384)
#endif
    }
}
