// Host-visible launch interface of the EKF kernels (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ekf_device.h"

// status word bits (sticky until ekf_reset; decoded by sync_and_check in ekf_api.hip)
#define EKF_ST_NOT_SPD 1          // a pivot of the innovation covariance was not positive
#define EKF_ST_TIMEOUT 4          // a bounded exchange wait inside the fused front kernel ran out
#define EKF_ST_STALE_JAC 16       // a chunk accepted Jacobian rows that carry another frame's tag
#define EKF_ST_STALE_COL 32       // a chunk accepted a factor block column that carries another frame's tag
#define EKF_ST_STALE_S 64         // the factorisation accepted an S block / residual that carries another frame's tag
#define EKF_ST_BAD_INDEX 128      // a landmark index outside [0, n_lm) reached a kernel (clamped to 0 there)
#define EKF_ST_GATE_TIMEOUT 256   // pipelined sequence mode: a device-side wait for the other stream ran out

// Everything one frame's kernels need; passed by value.
struct EkfFrame {
    void* cov;             // P, f32 or f64, row-major, leading dim ld
    int64_t ld;
    double* state;         // [cap]
    int32_t model;         // 0 = EKF, 1 = EKF_Rotations (see EkfModel in ekf_device.h)
    int32_t dims;          // N = lmd n + 10
    int32_t ncols;         // N rounded up to 128 (<= ld): columns the panel covers
    int32_t m;             // detections this frame
    int32_t k;             // 3 m
    int32_t kpad;          // k rounded up to EKF_RB
    const int32_t* idx;    // [m] landmark indices (device)
    const double* z;       // [m,3] measured positions, camera frame (device)
    double* jac;           // [kmax, EKF_JLD] Jacobian rows (13 used)
    double* resid;         // [kmax] z - h
    int32_t* lmcol;        // [mmax] first state column of each detection
    double* amat;          // A = H (P+Q), [kmax, lda] f64
    int64_t lda;
    double* sblk;          // S = Hs (P+Q)[supp,supp] Hs^T + R in 16x16 blocks (lower block triangle), block (i, tc) at
                           // (i (i + 1) / 2 + tc) * 256 in OP memory order (ekf_solve_device.h); diagonal blocks symmetric
    int32_t sblk_rows;     // (unused)
    double* lmat;          // Cholesky factor L of S, [kmax, ldl] f64 (lower)
    int32_t ldl;
    double* dinv;          // inverse of the 16x16 diagonal blocks of L, [kmax/16,16,16]
    // the same factor pre-arranged as v_mfma_f64_16x16x4 A operands (one
    // coalesced 512-byte read per MFMA in the panel kernel):
    //   lop[((b(b-1)/2 + q) 4 + r) 64 + lane] = -L[16b + (lane&15)][16q + (lane>>4) + 4r], q < b
    //   dop[(4 b + r) 64 + lane]             = Dinv_b[lane&15][(lane>>4) + 4r]
    double* lop;
    double* dop;
    double* yvec;          // L^-1 (z - h), [kmax]
    void* wpanel;          // W = L^-1 A, [kmax, ldw], cov dtype, k-major
    int64_t ldw;
    double* wdbg;          // optional f64 copy of W for tests (may be null)
    int32_t* status;       // [0] sticky error bits (EKF_ST_*), [1..2] diagnostics of the first non-SPD block column
    double* traj_row;      // optional: state[0:7] after the update
    double* dxvec;         // model 1: dx = W^T y for every state dimension (input of the injection kernel)
    long long* stamps;     // optional: s_memtime stamps of the solve kernel's phases (diagnostics)
    int32_t stamps_heavy;  // 0: only the stamps at the start / end of the roles (a stamp is a global store: the wave that takes it
                           // later waits for its acknowledgement), 1: also inside the factorisation and the chunk prologue
    // Pipelined sequence mode (ekf_api.hip: ekf_observe_sequence_device).  The covariance lives in TWO buffers: the
    // update of frame t reads `cov` (P_t) and writes `cov_out` (P_{t+1}); the front kernel of frame t+1 runs beside it
    // and takes the entries of P_{t+1} it needs -- support rows only -- from P_t and W_t on the fly:
    //     P_{t+1}[r][c] = (P_t[r][c] + Q[r == c]) + sum_k fma(-W_t[k][r], W_t[k][c])     (k ascending, from zero)
    // which is, instruction for instruction, what the covariance update computes for that element (bitwise equal).
    // For such a front kernel `cov` is P_t (the PREVIOUS frame's input), `wprev` the previous frame's W panel and
    // `wsup_prev` the compact copy of its support columns, W_t[:, row(slot)], slot = 10 + lmd j + d for detection
    // j of THIS frame (written by the previous front kernel's chunks, which know this frame's indices: next_idx).
    const int32_t* next_idx;   // [next_m] landmark indices of the next frame (device); null: none
    int32_t next_m;
    void* cov_out;             // covariance update: destination (null = in place)
    const void* wprev;         // front kernel: W panel of the previous frame [kpad][ldw] (null = `cov` is current)
    const void* wsup_prev;     // front kernel: [kpad][wsup_ld], see above
    // device-side cross-stream ordering of the pipelined sequence mode (ekf_api.hip: run_pipelined):
    // la_sync[0] = "front kernel of frame n has started" counter, la_sync[1] = "covariance update of frame n is
    // complete" counter.  A front kernel stores la_signal into [0] when it starts (0 = no) and does not
    // finish before [1] >= la_gate (0 = no gate).
    unsigned long long* la_sync;
    unsigned long long la_signal, la_gate;
    int32_t lds_min;           // front kernel: claim at least this much LDS (keeps other kernels' workgroups off its CUs)
    void* wsup;                // pipelined mode, written by the chunks: W[:, support rows of the NEXT frame], [kpad][wsup_ld] in cov dtype (null: none)
    int32_t wsup_ld;
    EkfNoise nz;
    int32_t quat_mode;
    // fused front kernel (ekf_front_impl.h): exchange buffers between its workgroups.  ONE buffer per
    // fused-frame parity holds everything that travels between workgroups inside a launch:
    //   [-L operands | Dinv operands | y | Jacobian rows [k][JC] | tags | S blocks | residual | S-block tags]
    // Every word is sentinel-armed (EKF_SENT_BITS) until its producer overwrites it; the buffer of
    // parity p is re-armed during the NEXT fused frame (parity p^1) by that frame's S-block
    // workgroups, i.e. at least one kernel boundary after its last reader and one before its next
    // writer.  No role ever re-arms what it has just read.
    double* xl;                // this frame's exchange buffer
    double* xl_next;           // the other buffer, re-armed during this frame for the next fused frame
    int32_t xl_dop, xl_y, xl_jac;   // offsets (doubles) of the Dinv operands, y and the Jacobian rows inside xl
    int32_t xl_len;            // doubles per buffer
    double* xs;                // = xl + offset: S blocks, layout of sblk (OP memory order)
    double* xr;                // = xl + offset: [kmax] z - h (0 for rows k..kpad-1)
    double* xs_tag;            // = xl + offset: frame tag of S block (row block i, block column tc) at [16 tc + i]
    int32_t n_lm;              // landmarks in the state (index validation; model 1 injection)
    unsigned long long* done_ctr;      // chunks finished since reset (device)
    unsigned long long done_target;    // value of done_ctr once this frame's last chunk is done
    int32_t xl_tag;                    // frame tags inside xl ([0] Jacobian, [1 + q] block column q, [16] residual): integrity check
    double seqno;                      // this frame's tag (fused frames since reset, from 1)
    // Per-frame host boundary (ekf_observe + a state getter every frame): the injection code of the fused front kernel
    // also writes the new state into pinned HOST memory, and whoever raises a status bit also sets a word there, so
    // that the getter is a wait for the front kernel's event and a memcpy -- no device-to-host copies (null: off).
    double* state_host;
    int32_t* status_host;
    // Large problems (f32 covariance): launch order of the 128 x 128 macro-tile covariance update (ekf_cov_macro.hip):
    // tile (I << 16 | J) of block b, 0xFFFFFFFF = none; null: the wave-per-tile kernel (ekf_cov_update.hip)
    const uint32_t* cov_tiles;
    int32_t cov_grid;
    unsigned long long* cov_stats;     // diagnostics of the macro-tile kernel (builds with CM_STAMPS only; else unused)
};

// raise sticky status bits (and tell the host mirror, if there is one, that the status word is no longer zero)
__device__ __forceinline__ void ekf_raise(const EkfFrame& fr, int bits) {
    atomicOr(fr.status, bits);
    if (fr.status_host) __hip_atomic_store(fr.status_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// First state column of detection j.  Indices that arrive through the device-pointer entry points
// cannot be checked on the host: an index outside [0, n_lm) is clamped to 0 (so that nothing is read
// or written out of bounds) and, where `flag` is set, recorded in the status word -> EKF_ERR_INVALID
// at the next synchronising call.
__device__ __forceinline__ int ekf_lm_column(const EkfFrame& fr, int lmd, int j, bool flag) {
    int i = fr.idx[j];
    if ((unsigned)i >= (unsigned)fr.n_lm) {
        if (flag) ekf_raise(fr, EKF_ST_BAD_INDEX);
        i = 0;
    }
    return EKF_CAM + lmd * i;
}

// fused gather + solve + panel (+ injection); see ekf_front_impl.h
template <typename T> void ekf_launch_front(const EkfFrame& fr, hipStream_t s);
template <typename T> void ekf_launch_gather(const EkfFrame& fr, hipStream_t s);
void ekf_launch_solve(const EkfFrame& fr, hipStream_t s);
template <typename T> void ekf_launch_panel(const EkfFrame& fr, hipStream_t s);
// P <- P + Q - W^T W.  variant: 1 = VALU reference kernel, 2 = MFMA kernel.
// e0 / e1 (optional): events that receive the kernel's own start / stop time stamps (hipExtLaunchKernelGGL)
template <typename T> void ekf_launch_cov_update(const EkfFrame& fr, int variant, hipStream_t s,
                                                 hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// f32, large problems: one workgroup per 128 x 128 macro tile, launch order fr.cov_tiles / fr.cov_grid (ekf_cov_macro.hip)
void ekf_launch_cov_update_macro(const EkfFrame& fr, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// launch order for T x T macro tiles (lower triangle), super-tiles of S x S dealt to the 8 XCDs: returns the grid size
// (out == nullptr: only that), -1 if it exceeds `capacity` entries
int ekf_cov_macro_table(int T, int S, uint32_t* out, int capacity);
// pipelined sequence mode: one-wave kernels that order the two streams on the device
void ekf_launch_gate(unsigned long long* counter, unsigned long long target, int32_t* status, hipStream_t s,
                     int max_polls = 1 << 22);
void ekf_launch_signal(unsigned long long* counter, unsigned long long value, hipStream_t s);

template <typename T>
void ekf_launch_add_markers(void* cov, int64_t ld, double* state, int32_t dims,
                            const double* xyz_dev, const double* unc_dev, double default_unc,
                            int32_t count, hipStream_t s);
void ekf_launch_inject_rot(const EkfFrame& fr, int n_lm, hipStream_t s);
template <typename T>
void ekf_launch_add_markers_rot(void* cov, int64_t ld, double* state, int32_t dims, const double* pose6_dev,
                                const double* unc_dev, double default_unc, int32_t count, hipStream_t s);
template <typename T>
void ekf_launch_cov_diag(const void* cov, int64_t ld, double* out_dev, int32_t count, hipStream_t s);

// Detection -> pose front end (ekf_pose_ippe.hip): pinhole camera + Brown-Conrady distortion k1 k2 p1 p2 k3 k4 k5 k6
struct EkfCamera {
    double fx, fy, cx, cy;
    double k[8];
};
// one [tvec | rvec] per marker from its four pixel corners [count][4][2] (IPPE for a square of side marker_size)
void ekf_launch_ippe_square(const double* corners_dev, int count, double marker_size, const EkfCamera& cam,
                            double* poses_dev, hipStream_t s);
