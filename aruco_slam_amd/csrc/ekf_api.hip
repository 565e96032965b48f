// C ABI of the EKF-SLAM HIP library (see include/ekf_slam_hip.h).
// Host side only: argument checking, workspace carving, launch sequencing.
#include "../../include/ekf_slam_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ekf_kernels.h"

namespace {

thread_local std::string g_err;

// Pipelined sequence mode orders its two streams with device-side gates that spin.  HIP multiplexes streams onto a small
// pool of hardware queues: with two handles pipelining at once, handle X's gate can sit in the queue in front of the
// launch handle Y's gate waits for, and vice versa -- both poll budgets run out.  So at most ONE handle of the
// process has gates in flight; another handle that asks for the pipelined mode meanwhile runs its call in serial order
// (same results, bit for bit) and says so (ekf_last_sequence_mode).
std::atomic<ekf_filter*> g_pipelining{nullptr};

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(EKF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

inline int64_t round_up(int64_t v, int64_t q) { return (v + q - 1) / q * q; }
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

constexpr int kStageSlots = 64;   // pinned host ring for ekf_observe
#ifndef EKF_MACRO_SUPER
#define EKF_MACRO_SUPER 8
#endif
constexpr int kMacroSuper = EKF_MACRO_SUPER;    // macro-tile covariance update: super-tiles of 8 x 8 macro tiles per XCD (ekf_cov_macro.hip)
constexpr int kMacroMinTiles = 3000;   // ... chosen from this many 128 x 128 tiles of the lower triangle (n >= 3300 or so; measured: n=2048 1225 tiles 98 vs 80 us for the wave-per-tile kernel, n=4096 4753 tiles 290 vs 323)
constexpr int kTimedKernels = 4;
constexpr int kEventPool = 2048;  // frames of timing events kept before folding

struct Layout {
    int64_t cap;      // padded state dimension (multiple of 128)
    int rd, lmd;      // rows per detection, landmark dims (model)
    int kmax;         // rd * max_visible rounded up to 16
    size_t elem;      // sizeof(cov element)
    size_t off_jac, off_resid, off_y, off_lmcol, off_amat, off_sblk, off_lmat, off_dinv, off_lop, off_dop, off_wpanel, off_wpanel2, off_cov2, off_wdbg,
        off_idx, off_z, off_status, off_stamps, off_covstats, off_dx, off_diag, off_xyz, off_unc,
        off_xl, off_done, off_sync, off_wsup, total;
    int wsup_ld;      // row length of the compact support-column copy of W (pipelined sequence mode; two copies, by frame parity)
    bool has_cov2;
    size_t off_tiles; // launch order of the macro-tile covariance update (f32, large problems)
    int tiles_cap;    // entries
    // fused front kernel: one exchange buffer per fused-frame parity (offsets / length in doubles)
    size_t xl_len, xl_dop, xl_y, xl_jac, xl_tag, xl_xs, xl_xr, xl_stag;
};

Layout make_layout(const ekf_config& c) {
    Layout L{};
    L.rd = c.model == EKF_MODEL_ROTATIONS ? 7 : 3;
    L.lmd = c.model == EKF_MODEL_ROTATIONS ? 10 : 3;
    L.cap = round_up((int64_t)L.lmd * c.max_landmarks + EKF_CAM, 128);
    L.kmax = (int)round_up(L.rd * c.max_visible, EKF_RB);
    L.elem = c.cov_dtype == EKF_COV_F32 ? 4 : 8;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align256(bytes); return at; };
    L.off_jac = take((size_t)L.kmax * EKF_JLD * 8);
    L.off_resid = take((size_t)L.kmax * 8);
    L.off_y = take((size_t)L.kmax * 8);
    L.off_lmcol = take((size_t)c.max_visible * 4);
    L.off_amat = take((size_t)L.kmax * L.cap * 8);
    L.off_sblk = take((size_t)(L.kmax / EKF_RB) * (L.kmax / EKF_RB + 1) / 2 * 256 * 8);
    L.off_lmat = take((size_t)L.kmax * L.kmax * 8);
    L.off_dinv = take((size_t)L.kmax * EKF_RB * 8);
    {
        const size_t nb = (size_t)L.kmax / EKF_RB;
        L.off_lop = take(nb * (nb - 1) / 2 * 256 * 8 + 256);
        L.off_dop = take(nb * 256 * 8);
    }
    L.off_wpanel = take((size_t)L.kmax * L.cap * L.elem);
    L.off_wpanel2 = take((size_t)L.kmax * L.cap * L.elem);
    // pipelined sequence mode (MFMA update, fused front kernel): the second covariance buffer
    // (P_t is read, P_{t+1} written elsewhere, so that the next front kernel can read P_t beside the update)
    L.has_cov2 = c.cov_kernel != EKF_COVK_VALU && (c.flags & 5) == 0;
    L.off_cov2 = L.has_cov2 ? take((size_t)L.cap * L.cap * L.elem) : 0;
    L.off_wdbg = take((size_t)L.kmax * L.cap * 8);
    L.off_idx = take((size_t)c.max_visible * 4);
    L.off_z = take((size_t)c.max_visible * 7 * 8);
    L.off_status = take(256);
    L.off_stamps = take(64 * 8);
    L.off_covstats = take(32 * 8);
    L.off_dx = take((size_t)L.cap * 8);
    L.off_diag = take((size_t)L.cap * 8);
    L.off_xyz = take((size_t)256 * 6 * 8);
    L.off_unc = take((size_t)256 * 10 * 8);
    {
        const size_t nb = (size_t)L.kmax / EKF_RB;
        L.xl_dop = nb * (nb - 1) / 2 * 256 + 64;
        L.xl_y = L.xl_dop + nb * 256;
        L.xl_jac = L.xl_y + L.kmax;
        L.xl_tag = L.xl_jac + (size_t)L.kmax * EKF_JLD;
        L.xl_xs = L.xl_tag + 32;                          // S blocks, layout of sblk (block (i, tc) at (i (i + 1) / 2 + tc) * 256)
        L.xl_xr = L.xl_xs + nb * (nb + 1) / 2 * 256;      // residual
        L.xl_stag = L.xl_xr + (size_t)round_up(L.kmax, 16);   // S-block tags [16 tc + i]
        L.xl_len = L.xl_stag + 256;
        L.off_xl = take(2 * L.xl_len * 8);
        L.off_done = take(256);
        L.off_sync = take(256);
        L.wsup_ld = (int)round_up(EKF_CAM + (int64_t)L.lmd * c.max_visible, 32);
        L.off_wsup = take((size_t)2 * L.kmax * L.wsup_ld * L.elem);
    }
    {
        // (the grid is 8 x the longest per-XCD list: the tile count plus, at worst, one super-tile per XCD)
        const int tmax = (int)(L.cap / 128);
        L.tiles_cap = (L.elem == 4 && c.cov_kernel != EKF_COVK_VALU && c.cov_kernel != EKF_COVK_MFMA_TILE)
                          ? tmax * (tmax + 1) / 2 + 8 * kMacroSuper * kMacroSuper : 0;
        L.off_tiles = take((size_t)L.tiles_cap * 4);
    }
    L.total = o;
    return L;
}

int check_config(const ekf_config* c) {
    if (!c) return fail(EKF_ERR_INVALID, "config is NULL");
    if (c->max_landmarks < 1) return fail(EKF_ERR_INVALID, "max_landmarks must be >= 1");
    if (c->model != EKF_MODEL_EKF && c->model != EKF_MODEL_ROTATIONS)
        return fail(EKF_ERR_INVALID, "unknown model");
    if (c->max_visible < 1 || c->max_visible > (c->model == EKF_MODEL_ROTATIONS ? 50 : 64))
        return fail(EKF_ERR_INVALID, "max_visible must be in 1..64 (1..50 for EKF_MODEL_ROTATIONS)");
    if (c->cov_dtype != EKF_COV_F64 && c->cov_dtype != EKF_COV_F32)
        return fail(EKF_ERR_INVALID, "cov_dtype must be EKF_COV_F64 or EKF_COV_F32");
    if (c->quat_mode != EKF_QUAT_AS_WRITTEN && c->quat_mode != EKF_QUAT_SCALAR_FIRST)
        return fail(EKF_ERR_INVALID, "unknown quat_mode");
    if (c->cov_kernel < EKF_COVK_AUTO || c->cov_kernel > EKF_COVK_MFMA_MACRO)
        return fail(EKF_ERR_INVALID, "unknown cov_kernel");
    if (c->cov_kernel == EKF_COVK_MFMA_MACRO && c->cov_dtype != EKF_COV_F32)
        return fail(EKF_ERR_INVALID, "EKF_COVK_MFMA_MACRO exists for the f32 covariance only");
    if (!(c->r_uncertainty > 0.0)) return fail(EKF_ERR_INVALID, "r_uncertainty must be > 0");
    return EKF_OK;
}

}  // namespace

struct ekf_filter {
    ekf_config cfg{};
    Layout lay{};
    hipStream_t stream = nullptr;
    hipStream_t big = nullptr;          // internal stream: big covariance update in sequence mode
    hipEvent_t ev_small[2] = {}, ev_big[2] = {};
    hipEvent_t ev_front = nullptr;      // recorded behind the front part of the last per-frame observe: the state is final there
    bool front_pending = false;         // ... and nothing that changes the state has been enqueued since
    // host mirror of the state (readback + 256) and of "the status word is not zero" (readback + 128), written by the fused
    // front kernel of a per-frame observe: a state getter is then an event wait and a memcpy
    bool mirror_fresh = false;          // the last frame wrote the mirror
    bool mirror_trust = false;          // entries no frame writes (EKF_Rotations: landmark error states) agree with the device
    bool status_clean = false;          // the device status word was zero when it was last read
    int device = 0;
    void* cov = nullptr;
    int64_t ld = 0;
    double* state = nullptr;
    char* ws = nullptr;
    bool bound = false, is_reset = false;
    int n_lm = 0;
    int last_m = 0;
    bool debug_w = false;
    bool debug_stamps = false;
    bool debug_stamps_light = false;
    uint64_t fseq = 0;         // FUSED frames enqueued since reset: parity of the exchange buffer, frame tag
    uint64_t done_total = 0;   // column chunks of fused frames enqueued since reset
    uint64_t la_base = 0;      // frames that went through the pipelined sequence mode since reset (device counters)
    int la_ok = -1;            // pipelined mode usable (-1: not probed yet; 0: the two streams share a hardware queue)
    int seq_mode = EKF_SEQ_NONE;   // what the last ekf_observe_sequence_device call did (ekf_last_sequence_mode)
    // macro-tile covariance update: launch-order table in the workspace, rebuilt when the tile count changes
    int tiles_T = -1, tiles_grid = 0;
    uint32_t* tiles_host = nullptr;     // pinned staging copy
    // pinned staging ring for host-pointer observes
    char* pinned = nullptr;
    size_t slot_bytes = 0;
    int slot = 0;
    hipEvent_t slot_done[kStageSlots] = {};
    char* readback = nullptr;           // pinned: [status words (256 B) | state (cap doubles)], one sync per getter
    // kernel timing
    bool timing = false;
    bool timing_cov_only = false;
    std::vector<hipEvent_t> ev;   // (kTimedKernels + 1) per frame
    int ev_frames = 0;
    double t_sum_us[kTimedKernels] = {};
    int64_t t_cnt[kTimedKernels] = {};

    int dims() const { return lay.lmd * n_lm + EKF_CAM; }
    template <typename P> P* at(size_t off) const { return reinterpret_cast<P*>(ws + off); }
};

namespace {

int fold_timing(ekf_filter* f) {
    if (f->ev_frames == 0) return EKF_OK;
    HIP_TRY(hipStreamSynchronize(f->stream));
    for (int fr = 0; fr < f->ev_frames; ++fr) {
        for (int kq = f->timing_cov_only ? kTimedKernels - 1 : 0; kq < kTimedKernels; ++kq) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, f->ev[fr * (kTimedKernels + 1) + kq],
                                        f->ev[fr * (kTimedKernels + 1) + kq + 1]));
            f->t_sum_us[kq] += 1e3 * ms;
            f->t_cnt[kq] += 1;
        }
    }
    f->ev_frames = 0;
    return EKF_OK;
}

EkfFrame make_frame(ekf_filter* f, const int32_t* idx_dev, const double* z_dev, int m,
                    double* traj_row) {
    EkfFrame fr{};
    const Layout& L = f->lay;
    fr.cov = f->cov;
    fr.ld = f->ld;
    fr.state = f->state;
    fr.model = f->cfg.model;
    fr.dims = f->dims();
    fr.ncols = (int)round_up(fr.dims, 128);
    fr.m = m;
    fr.k = L.rd * m;
    fr.kpad = (int)round_up(fr.k, EKF_RB);
    fr.idx = idx_dev;
    fr.z = z_dev;
    fr.jac = f->at<double>(L.off_jac);
    fr.resid = f->at<double>(L.off_resid);
    fr.lmcol = f->at<int32_t>(L.off_lmcol);
    fr.amat = f->at<double>(L.off_amat);
    fr.lda = L.cap;
    fr.sblk = f->at<double>(L.off_sblk);
    fr.sblk_rows = L.kmax;
    fr.lmat = f->at<double>(L.off_lmat);
    fr.ldl = L.kmax;
    fr.dinv = f->at<double>(L.off_dinv);
    fr.lop = f->at<double>(L.off_lop);
    fr.dop = f->at<double>(L.off_dop);
    fr.yvec = f->at<double>(L.off_y);
    fr.wpanel = f->at<void>(L.off_wpanel);
    fr.ldw = L.cap;
    fr.wdbg = f->debug_w ? f->at<double>(L.off_wdbg) : nullptr;
    fr.status = f->at<int32_t>(L.off_status);
    fr.traj_row = traj_row;
    fr.dxvec = f->at<double>(L.off_dx);
    fr.stamps = (f->debug_w || f->debug_stamps) ? f->at<long long>(L.off_stamps) : nullptr;
    fr.stamps_heavy = f->debug_stamps_light ? 0 : 1;
    fr.nz = EkfNoise{f->cfg.q_cam, f->cfg.q_err, f->cfg.q_lm, f->cfg.r_uncertainty};
    fr.quat_mode = f->cfg.quat_mode;
    fr.n_lm = f->n_lm;
    fr.state_host = nullptr;
    fr.status_host = nullptr;
    fr.cov_tiles = nullptr;
    fr.cov_grid = 0;
    fr.cov_stats = f->debug_stamps ? f->at<unsigned long long>(L.off_covstats) : nullptr;
    if (f->tiles_T > 0 && f->tiles_T == (fr.dims + 127) / 128) {
        fr.cov_tiles = f->at<uint32_t>(L.off_tiles);
        fr.cov_grid = f->tiles_grid;
    }
    return fr;
}

// Fused front kernel or the three stage kernels?  Fused unless the caller opted out (flags bit 2).
// Beyond k = 192 rows (EKF_Rotations with more than 27 markers in view) the frame runs through the stage kernels: the fused
// kernel's chunk and S-block roles are laid out for at most twelve block columns.
bool use_front_kernel(const ekf_filter* f, const EkfFrame& fr) { return (f->cfg.flags & 4) == 0 && fr.kpad <= 192; }

// Exchange buffers of a FUSED frame.  Parity, frame tag and the chunk counter advance with the fused
// frames only: a stage-kernel frame neither uses nor re-arms the exchange, so a filter may alternate
// between the two paths (flags can differ per handle, not per frame, today -- but the bookkeeping does
// not depend on that).
void bind_exchange(ekf_filter* f, EkfFrame& fr) {
    const Layout& L = f->lay;
    double* base = f->at<double>(L.off_xl);
    fr.xl = base + (f->fseq & 1) * L.xl_len;
    fr.xl_next = base + ((f->fseq + 1) & 1) * L.xl_len;
    fr.xl_dop = (int32_t)L.xl_dop;
    fr.xl_y = (int32_t)L.xl_y;
    fr.xl_jac = (int32_t)L.xl_jac;
    fr.xl_tag = (int32_t)L.xl_tag;
    fr.xl_len = (int32_t)L.xl_len;
    fr.xs = fr.xl + L.xl_xs;
    fr.xr = fr.xl + L.xl_xr;
    fr.xs_tag = fr.xl + L.xl_stag;
    fr.seqno = (double)(f->fseq + 1);
    fr.done_ctr = f->at<unsigned long long>(L.off_done);
    f->done_total += (uint64_t)(fr.ncols / 64);
    fr.done_target = f->done_total;
    f->fseq++;
}

// Macro-tile covariance update (f32, large problems: ekf_cov_macro.hip): chosen from kMacroMinTiles tiles of 128 x 128
// (or forced); its launch-order table depends on the tile count only and is rebuilt -- synchronously: this happens when
// landmarks are added, a few dozen times in the life of a filter -- whenever that changes.
int ensure_tiles(ekf_filter* f) {
    const Layout& L = f->lay;
    const int T = (f->dims() + 127) / 128;
    const bool want = L.tiles_cap > 0 &&
                      (f->cfg.cov_kernel == EKF_COVK_MFMA_MACRO || T * (T + 1) / 2 >= kMacroMinTiles);
    if (!want) {
        f->tiles_T = -1;
        return EKF_OK;
    }
    if (f->tiles_T == T) return EKF_OK;
    HIP_TRY(hipStreamSynchronize(f->stream));
    HIP_TRY(hipStreamSynchronize(f->big));
    const int grid = ekf_cov_macro_table(T, kMacroSuper, f->tiles_host, L.tiles_cap);
    if (grid <= 0) return fail(EKF_ERR_STATE, "macro-tile launch table does not fit the workspace");
    HIP_TRY(hipMemcpy(f->at<uint32_t>(L.off_tiles), f->tiles_host, (size_t)grid * 4, hipMemcpyHostToDevice));
    f->tiles_T = T;
    f->tiles_grid = grid;
    return EKF_OK;
}

// predict + update for one frame: the front kernel (or gather / solve / panel) and the covariance update
// `mirror`: the frame also leaves its state in the pinned host mirror (per-frame entry points: a state getter usually follows;
// frames of a sequence call do not -- no getter can run between them, and the mirror is 8 N bytes over PCIe per frame)
int enqueue_frame(ekf_filter* f, const int32_t* idx_dev, const double* z_dev, int m,
                  double* traj_row, bool mirror) {
    int trc = ensure_tiles(f);
    if (trc) return trc;
    EkfFrame fr = make_frame(f, idx_dev, z_dev, m, traj_row);
    const bool f32 = f->cfg.cov_dtype == EKF_COV_F32;
    const int variant = f->cfg.cov_kernel == EKF_COVK_VALU ? 1 : 2;
    hipEvent_t* ev = nullptr;
    if (f->timing) {
        if (f->ev_frames == kEventPool) {
            int rc = fold_timing(f);
            if (rc) return rc;
        }
        ev = &f->ev[f->ev_frames * (kTimedKernels + 1)];
        f->ev_frames++;
    }
    hipEvent_t* ev_all = (ev && !f->timing_cov_only) ? ev : nullptr;
    if (ev_all) HIP_TRY(hipEventRecord(ev[0], f->stream));
    f->mirror_fresh = false;
    if (use_front_kernel(f, fr)) {
        // one launch: timing slot 0 = the whole front kernel, slots 1 and 2 stay empty
        bind_exchange(f, fr);
        if (!f->timing && mirror) {
            fr.state_host = reinterpret_cast<double*>(f->readback + 256);
            fr.status_host = reinterpret_cast<int32_t*>(f->readback + 128);
            f->mirror_fresh = true;
        }
        if (f32) ekf_launch_front<float>(fr, f->stream); else ekf_launch_front<double>(fr, f->stream);
        if (ev_all) {
            HIP_TRY(hipEventRecord(ev[1], f->stream));
            HIP_TRY(hipEventRecord(ev[2], f->stream));
        }
    } else {
        if (f32) ekf_launch_gather<float>(fr, f->stream); else ekf_launch_gather<double>(fr, f->stream);
        if (ev_all) HIP_TRY(hipEventRecord(ev[1], f->stream));
        ekf_launch_solve(fr, f->stream);
        if (ev_all) HIP_TRY(hipEventRecord(ev[2], f->stream));
        if (f32) ekf_launch_panel<float>(fr, f->stream); else ekf_launch_panel<double>(fr, f->stream);
        if (fr.model == 1) ekf_launch_inject_rot(fr, f->n_lm, f->stream);
    }
    // The state (and the status word) are final here: a state getter that follows waits for this event only and
    // reads back beside the covariance update (sync_and_check) -- what BaseFilter.process_frame does every frame.
    f->front_pending = false;
    if (!f->timing && hipEventRecord(f->ev_front, f->stream) == hipSuccess) f->front_pending = true;
    // the covariance update is timed by its own start / stop time stamps (what rocprofv3 reports), not by
    // events recorded around the launch (those also hold ~4 us of dispatch gap)
    if (f32) ekf_launch_cov_update<float>(fr, variant, f->stream, ev ? ev[3] : nullptr, ev ? ev[4] : nullptr);
    else ekf_launch_cov_update<double>(fr, variant, f->stream, ev ? ev[3] : nullptr, ev ? ev[4] : nullptr);
    HIP_TRY(hipGetLastError());
    f->last_m = m;
    return EKF_OK;
}

int check_ready(ekf_filter* f) {
    if (!f) return fail(EKF_ERR_INVALID, "filter handle is NULL");
    if (!f->bound) return fail(EKF_ERR_STATE, "ekf_bind_buffers has not been called");
    if (!f->is_reset) return fail(EKF_ERR_STATE, "ekf_reset has not been called");
    HIP_TRY(hipSetDevice(f->device));
    return EKF_OK;
}

// `state_count` > 0: the first state_count doubles of the state come back with the same synchronisation
// (f->readback + 256): one stream sync per getter instead of a sync and two blocking copies
int sync_and_check(ekf_filter* f, int state_count = 0) {
    if (state_count > 0 && f->front_pending) {
        // State getter right after a per-frame observe: the state and the status word were final when the front part
        // of that frame ended; the covariance update behind it goes on while the host reads back (internal stream: idle
        // outside ekf_observe_sequence_device, ordered here by the host having seen the event) and prepares the next frame.
        HIP_TRY(hipEventSynchronize(f->ev_front));
        // ... and if the fused front kernel has written the state into the pinned mirror and nobody has raised a status
        // bit (their stores to host memory are complete with the kernel), there is nothing to copy at all
        if (f->mirror_fresh && f->mirror_trust && f->status_clean &&
            *reinterpret_cast<volatile int32_t*>(f->readback + 128) == 0)
            return EKF_OK;
        HIP_TRY(hipMemcpyAsync(f->readback, f->at<int32_t>(f->lay.off_status), 32, hipMemcpyDeviceToHost, f->big));
        HIP_TRY(hipMemcpyAsync(f->readback + 256, f->state, (size_t)state_count * 8, hipMemcpyDeviceToHost, f->big));
        HIP_TRY(hipStreamSynchronize(f->big));
    } else {
        HIP_TRY(hipMemcpyAsync(f->readback, f->at<int32_t>(f->lay.off_status), 32, hipMemcpyDeviceToHost, f->stream));
        if (state_count > 0)
            HIP_TRY(hipMemcpyAsync(f->readback + 256, f->state, (size_t)state_count * 8, hipMemcpyDeviceToHost, f->stream));
        HIP_TRY(hipStreamSynchronize(f->stream));
        ekf_filter* me = f;      // (everything this handle has enqueued is complete: its gates are gone)
        (void)g_pipelining.compare_exchange_strong(me, nullptr);
    }
    const int32_t st = reinterpret_cast<const int32_t*>(f->readback)[0];
    f->status_clean = (st == 0);
    if (state_count == f->dims()) f->mirror_trust = true;      // (a full copy has just refreshed the mirror)
    if (st != 0) {
        // (sticky until ekf_reset: after any of these the filter state is not trustworthy)
        if (st & EKF_ST_BAD_INDEX)
            return fail(EKF_ERR_INVALID, "landmark index out of range in a device-resident detection array "
                                         "(clamped on the device; the filter state is no longer meaningful, reset it)");
        if (st & (EKF_ST_STALE_JAC | EKF_ST_STALE_COL | EKF_ST_STALE_S))
            return fail(EKF_ERR_NUMERIC, "internal: exchange data of another frame accepted in the front kernel (status " +
                                             std::to_string(st) + ")");
        if (st & EKF_ST_GATE_TIMEOUT)
            return fail(EKF_ERR_NUMERIC, "internal: a device-side wait between the two streams of the pipelined sequence mode timed out");
        if (st & EKF_ST_TIMEOUT)   // a bounded wait inside the fused front kernel ran out (should never happen)
            return fail(EKF_ERR_NUMERIC, "internal: exchange wait timed out in the front kernel (status " +
                                             std::to_string(st) + ")");
        const int32_t* info = reinterpret_cast<const int32_t*>(f->readback);
        if (info[2] >= 200)      // (ekf_solve_cw.h: a bounded wait on one of the factorisation's LDS flag words ran out; should never happen)
            return fail(EKF_ERR_NUMERIC, "internal: a wait inside the factorisation workgroup timed out (flag word " +
                                             std::to_string(info[2] - 200) + ")");
        return fail(EKF_ERR_NUMERIC, "innovation covariance S was not positive definite (block column " +
                                         std::to_string(info[2] - 100) + ", wave mask " + std::to_string(info[1]) + ")");
    }
    return EKF_OK;
}

}  // namespace

extern "C" {

const char* ekf_last_error_string(void) { return g_err.c_str(); }

int ekf_default_config(ekf_config* cfg) {
    if (!cfg) return fail(EKF_ERR_INVALID, "config is NULL");
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->max_landmarks = 50;   // DICT_5X5_50, base_filter.py:81-82
    cfg->max_visible = 50;
    cfg->cov_dtype = EKF_COV_F64;
    cfg->quat_mode = EKF_QUAT_AS_WRITTEN;
    cfg->cov_kernel = EKF_COVK_AUTO;
    cfg->model = EKF_MODEL_EKF;
    cfg->initial_camera_uncertainty = 0.1;
    cfg->initial_landmark_uncertainty = 0.7;
    cfg->r_uncertainty = 0.9;
    cfg->q_cam = 0.3;
    cfg->q_err = 0.5;
    cfg->q_lm = 0.01;
    cfg->stream = nullptr;
    return EKF_OK;
}

int ekf_query_sizes(const ekf_config* cfg, int64_t* ld, size_t* cov_bytes, size_t* state_bytes,
                    size_t* workspace_bytes) {
    int rc = check_config(cfg);
    if (rc) return rc;
    Layout L = make_layout(*cfg);
    if (ld) *ld = L.cap;
    if (cov_bytes) *cov_bytes = (size_t)L.cap * L.cap * L.elem;
    if (state_bytes) *state_bytes = (size_t)L.cap * 8;
    if (workspace_bytes) *workspace_bytes = L.total;
    return EKF_OK;
}

int ekf_create(const ekf_config* cfg, ekf_filter** out) {
    int rc = check_config(cfg);
    if (rc) return rc;
    if (!out) return fail(EKF_ERR_INVALID, "out is NULL");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(EKF_ERR_HIP, "no HIP device visible");
    ekf_filter* f = new ekf_filter();
    if (hipGetDevice(&f->device) != hipSuccess) f->device = 0;   // the caller's current device
    f->cfg = *cfg;
    f->lay = make_layout(*cfg);
    f->stream = static_cast<hipStream_t>(cfg->stream);
    f->slot_bytes = align256((size_t)cfg->max_visible * 4) + align256((size_t)cfg->max_visible * 56);
    if (f->slot_bytes < align256(256 * 48) + align256(256 * 80)) f->slot_bytes = align256(256 * 48) + align256(256 * 80);
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&f->pinned), f->slot_bytes * kStageSlots,
                                 hipHostMallocDefault);
    if (e != hipSuccess) {
        delete f;
        return fail(EKF_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    e = hipHostMalloc(reinterpret_cast<void**>(&f->readback), 256 + (size_t)f->lay.cap * 8, hipHostMallocDefault);
    if (e != hipSuccess) {
        ekf_destroy(f);
        return fail(EKF_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    if (f->lay.tiles_cap > 0) {
        e = hipHostMalloc(reinterpret_cast<void**>(&f->tiles_host), (size_t)f->lay.tiles_cap * 4, hipHostMallocDefault);
        if (e != hipSuccess) {
            ekf_destroy(f);
            return fail(EKF_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        }
    }
    e = hipStreamCreateWithFlags(&f->big, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_front, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&f->ev_small[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_big[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        ekf_destroy(f);
        return fail(EKF_ERR_HIP, std::string("stream/event create: ") + hipGetErrorString(e));
    }
    for (int i = 0; i < kStageSlots; ++i) {
        e = hipEventCreateWithFlags(&f->slot_done[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            ekf_destroy(f);
            return fail(EKF_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
        }
    }
    *out = f;
    return EKF_OK;
}

int ekf_destroy(ekf_filter* f) {
    if (!f) return EKF_OK;
    (void)hipStreamSynchronize(f->stream);
    if (f->big) {
        (void)hipStreamSynchronize(f->big);
        (void)hipStreamDestroy(f->big);
    }
    {
        ekf_filter* me = f;
        (void)g_pipelining.compare_exchange_strong(me, nullptr);
    }
    for (int i = 0; i < 2; ++i) {
        if (f->ev_small[i]) (void)hipEventDestroy(f->ev_small[i]);
        if (f->ev_big[i]) (void)hipEventDestroy(f->ev_big[i]);
    }
    if (f->ev_front) (void)hipEventDestroy(f->ev_front);
    for (auto& e : f->ev) (void)hipEventDestroy(e);
    for (int i = 0; i < kStageSlots; ++i)
        if (f->slot_done[i]) (void)hipEventDestroy(f->slot_done[i]);
    if (f->pinned) (void)hipHostFree(f->pinned);
    if (f->readback) (void)hipHostFree(f->readback);
    if (f->tiles_host) (void)hipHostFree(f->tiles_host);
    delete f;
    return EKF_OK;
}

int ekf_bind_buffers(ekf_filter* f, void* cov_dev, int64_t ld, double* state_dev, void* workspace_dev,
                     size_t workspace_bytes) {
    if (!f) return fail(EKF_ERR_INVALID, "filter handle is NULL");
    if (!cov_dev || !state_dev || !workspace_dev) return fail(EKF_ERR_INVALID, "NULL device buffer");
    if (ld != f->lay.cap) return fail(EKF_ERR_INVALID, "ld must equal the value from ekf_query_sizes");
    if (workspace_bytes < f->lay.total) return fail(EKF_ERR_INVALID, "workspace too small");
    if ((reinterpret_cast<uintptr_t>(cov_dev) | reinterpret_cast<uintptr_t>(workspace_dev) |
         reinterpret_cast<uintptr_t>(state_dev)) & 0xFF)
        return fail(EKF_ERR_INVALID, "device buffers must be 256-byte aligned");
    f->cov = cov_dev;
    f->ld = ld;
    f->state = state_dev;
    f->ws = static_cast<char*>(workspace_dev);
    f->bound = true;
    f->is_reset = false;
    f->tiles_T = -1;
    return EKF_OK;
}

int ekf_grow(ekf_filter* f, int32_t new_max_landmarks, int32_t new_max_visible, void* cov_dev, int64_t ld, double* state_dev,
             void* workspace_dev, size_t workspace_bytes) {
    if (!f) return fail(EKF_ERR_INVALID, "filter handle is NULL");
    if (!f->bound || !f->is_reset) return fail(EKF_ERR_STATE, "ekf_grow needs a bound, reset filter");
    if (new_max_landmarks < f->cfg.max_landmarks || new_max_visible < f->cfg.max_visible)
        return fail(EKF_ERR_INVALID, "ekf_grow cannot shrink the capacity");
    if (!cov_dev || !state_dev || !workspace_dev) return fail(EKF_ERR_INVALID, "NULL device buffer");
    ekf_config ncfg = f->cfg;
    ncfg.max_landmarks = new_max_landmarks;
    ncfg.max_visible = new_max_visible;
    {
        int rc = check_config(&ncfg);
        if (rc) return rc;
    }
    const Layout nl = make_layout(ncfg);
    if (ld != nl.cap) return fail(EKF_ERR_INVALID, "ld must equal the value from ekf_query_sizes for the new capacity");
    if (workspace_bytes < nl.total) return fail(EKF_ERR_INVALID, "workspace too small for the new capacity");
    if ((reinterpret_cast<uintptr_t>(cov_dev) | reinterpret_cast<uintptr_t>(workspace_dev) |
         reinterpret_cast<uintptr_t>(state_dev)) & 0xFF)
        return fail(EKF_ERR_INVALID, "device buffers must be 256-byte aligned");
    if (cov_dev == f->cov || state_dev == f->state || workspace_dev == static_cast<void*>(f->ws))
        return fail(EKF_ERR_INVALID, "ekf_grow needs NEW buffers (the old ones are read)");
    HIP_TRY(hipSetDevice(f->device));
    HIP_TRY(hipStreamSynchronize(f->stream));
    HIP_TRY(hipStreamSynchronize(f->big));
    {
        ekf_filter* me = f;
        (void)g_pipelining.compare_exchange_strong(me, nullptr);
    }
    const Layout& ol = f->lay;
    char* nws = static_cast<char*>(workspace_dev);
    // pinned host buffers that are sized by the capacity
    size_t nslot = align256((size_t)ncfg.max_visible * 4) + align256((size_t)ncfg.max_visible * 56);
    if (nslot < align256(256 * 48) + align256(256 * 80)) nslot = align256(256 * 48) + align256(256 * 80);
    if (nslot != f->slot_bytes) {       // (every slot's event is complete: the stream has just been synchronised)
        char* npin = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&npin), nslot * kStageSlots, hipHostMallocDefault));
        (void)hipHostFree(f->pinned);
        f->pinned = npin;
        f->slot_bytes = nslot;
    }
    char* nread = nullptr;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&nread), 256 + (size_t)nl.cap * 8, hipHostMallocDefault));
    uint32_t* ntiles = nullptr;
    if (nl.tiles_cap > 0) {
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&ntiles), (size_t)nl.tiles_cap * 4, hipHostMallocDefault);
        if (e != hipSuccess) {
            (void)hipHostFree(nread);
            return fail(EKF_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        }
    }
    // new buffers: zero (capacity padding stays exactly zero), exchange buffers armed, then the old contents
    HIP_TRY(hipMemsetAsync(cov_dev, 0, (size_t)nl.cap * nl.cap * nl.elem, f->stream));
    HIP_TRY(hipMemsetAsync(state_dev, 0, (size_t)nl.cap * 8, f->stream));
    HIP_TRY(hipMemsetAsync(nws, 0, nl.total, f->stream));
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(nws + nl.off_xl), (int)0xFFFBADC0u, nl.xl_len * 4, f->stream));
    HIP_TRY(hipMemcpyAsync(state_dev, f->state, (size_t)ol.cap * 8, hipMemcpyDeviceToDevice, f->stream));
    HIP_TRY(hipMemcpy2DAsync(cov_dev, (size_t)nl.cap * nl.elem, f->cov, (size_t)ol.cap * ol.elem, (size_t)ol.cap * ol.elem,
                             (size_t)ol.cap, hipMemcpyDeviceToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(nws + nl.off_status, f->ws + ol.off_status, 32, hipMemcpyDeviceToDevice, f->stream));   // (sticky bits stay)
    HIP_TRY(hipStreamSynchronize(f->stream));
    (void)hipHostFree(f->readback);
    if (f->tiles_host) (void)hipHostFree(f->tiles_host);
    f->readback = nread;
    std::memset(f->readback, 0, 256 + (size_t)nl.cap * 8);
    f->tiles_host = ntiles;
    f->cfg = ncfg;
    f->lay = nl;
    f->cov = cov_dev;
    f->ld = ld;
    f->state = state_dev;
    f->ws = nws;
    // the workspace is new: everything that counts frames in it starts again (as after ekf_reset)
    f->fseq = 0;
    f->done_total = 0;
    f->la_base = 0;
    f->tiles_T = -1;
    f->front_pending = false;
    f->mirror_fresh = false;
    f->mirror_trust = false;
    f->status_clean = false;
    return EKF_OK;
}

int ekf_reset(ekf_filter* f, const double initial_camera_pose[10]) {
    if (!f) return fail(EKF_ERR_INVALID, "filter handle is NULL");
    if (!f->bound) return fail(EKF_ERR_STATE, "ekf_bind_buffers has not been called");
    if (!initial_camera_pose) return fail(EKF_ERR_INVALID, "initial pose is NULL");
    HIP_TRY(hipSetDevice(f->device));
    const Layout& L = f->lay;
    HIP_TRY(hipStreamSynchronize(f->stream));
    HIP_TRY(hipMemsetAsync(f->cov, 0, (size_t)L.cap * L.cap * L.elem, f->stream));
    HIP_TRY(hipMemsetAsync(f->state, 0, (size_t)L.cap * 8, f->stream));
    HIP_TRY(hipMemsetAsync(f->ws, 0, L.total, f->stream));
    // arm the exchange buffers of the fused front kernel (ekf_solve_device.h: EKF_SENT_BITS)
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(f->at<char>(L.off_xl)), (int)0xFFFBADC0u,
                              L.xl_len * 4, f->stream));
    f->fseq = 0;
    f->done_total = 0;
    f->la_base = 0;
    f->tiles_T = -1;               // (the workspace, the table with it, has just been cleared)
    f->front_pending = false;
    std::memset(f->readback, 0, 256 + (size_t)L.cap * 8);      // (the stream is idle: synchronised above)
    f->mirror_fresh = false;
    f->mirror_trust = true;        // state and mirror are both zero beyond what frames write
    f->status_clean = true;
    HIP_TRY(hipMemcpyAsync(f->state, initial_camera_pose, 10 * sizeof(double), hipMemcpyHostToDevice,
                           f->stream));
    // P = 0.1 I_10  (extended_kalman_filter.py:48)
    char diag[8 * EKF_CAM];
    for (int i = 0; i < EKF_CAM; ++i) {
        if (L.elem == 4) reinterpret_cast<float*>(diag)[i] = (float)f->cfg.initial_camera_uncertainty;
        else reinterpret_cast<double*>(diag)[i] = f->cfg.initial_camera_uncertainty;
    }
    HIP_TRY(hipMemcpy2DAsync(f->cov, (size_t)(L.cap + 1) * L.elem, diag, L.elem, L.elem, EKF_CAM,
                             hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipStreamSynchronize(f->stream));
    f->n_lm = 0;
    f->last_m = 0;
    f->is_reset = true;
    return EKF_OK;
}

int ekf_add_markers(ekf_filter* f, const double* cam_frame_xyz, const double* diag_uncertainty,
                    int32_t count) {
    int rc = check_ready(f);
    if (rc) return rc;
    if (count < 0 || (count > 0 && !cam_frame_xyz)) return fail(EKF_ERR_INVALID, "bad marker list");
    if (f->n_lm + count > f->cfg.max_landmarks)
        return fail(EKF_ERR_CAPACITY, "more landmarks than max_landmarks");
    const Layout& L = f->lay;
    f->front_pending = false;
    int done = 0;
    while (done < count) {
        const int chunk = std::min(256, count - done);
        char* slot = f->pinned + (size_t)f->slot * f->slot_bytes;
        HIP_TRY(hipEventSynchronize(f->slot_done[f->slot]));
        double* hx = reinterpret_cast<double*>(slot);
        double* hu = reinterpret_cast<double*>(slot + align256(256 * 48));
        const bool rot = f->cfg.model == EKF_MODEL_ROTATIONS;
        const size_t pb = rot ? 48 : 24, ub = rot ? 80 : 24;      // bytes per marker: pose / variances
        std::memcpy(hx, cam_frame_xyz + (pb / 8) * done, (size_t)chunk * pb);
        if (diag_uncertainty) std::memcpy(hu, diag_uncertainty + (ub / 8) * done, (size_t)chunk * ub);
        HIP_TRY(hipMemcpyAsync(f->at<double>(L.off_xyz), hx, (size_t)chunk * pb, hipMemcpyHostToDevice,
                               f->stream));
        if (diag_uncertainty)
            HIP_TRY(hipMemcpyAsync(f->at<double>(L.off_unc), hu, (size_t)chunk * ub,
                                   hipMemcpyHostToDevice, f->stream));
        const double* unc_dev = diag_uncertainty ? f->at<double>(L.off_unc) : nullptr;
        if (rot) {
            if (L.elem == 4)
                ekf_launch_add_markers_rot<float>(f->cov, f->ld, f->state, f->dims(), f->at<double>(L.off_xyz),
                                                  unc_dev, f->cfg.initial_landmark_uncertainty, chunk, f->stream);
            else
                ekf_launch_add_markers_rot<double>(f->cov, f->ld, f->state, f->dims(), f->at<double>(L.off_xyz),
                                                   unc_dev, f->cfg.initial_landmark_uncertainty, chunk, f->stream);
        } else if (L.elem == 4)
            ekf_launch_add_markers<float>(f->cov, f->ld, f->state, f->dims(), f->at<double>(L.off_xyz),
                                          unc_dev, f->cfg.initial_landmark_uncertainty, chunk, f->stream);
        else
            ekf_launch_add_markers<double>(f->cov, f->ld, f->state, f->dims(), f->at<double>(L.off_xyz),
                                           unc_dev, f->cfg.initial_landmark_uncertainty, chunk,
                                           f->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(f->slot_done[f->slot], f->stream));
        f->slot = (f->slot + 1) % kStageSlots;
        f->n_lm += chunk;
        done += chunk;
    }
    return EKF_OK;
}

int ekf_observe(ekf_filter* f, const int32_t* lm_index, const double* z, int32_t m) {
    int rc = check_ready(f);
    if (rc) return rc;
    if (m < 1) return fail(EKF_ERR_INVALID, "observe needs at least one detection");
    if (m > f->cfg.max_visible) return fail(EKF_ERR_CAPACITY, "more detections than max_visible");
    if (!lm_index || !z) return fail(EKF_ERR_INVALID, "NULL detections");
    for (int i = 0; i < m; ++i)
        if (lm_index[i] < 0 || lm_index[i] >= f->n_lm)
            return fail(EKF_ERR_INVALID, "landmark index out of range");
    const Layout& L = f->lay;
    char* slot = f->pinned + (size_t)f->slot * f->slot_bytes;
    HIP_TRY(hipEventSynchronize(f->slot_done[f->slot]));
    int32_t* hidx = reinterpret_cast<int32_t*>(slot);
    double* hz = reinterpret_cast<double*>(slot + align256((size_t)f->cfg.max_visible * 4));
    const size_t zb = (size_t)m * L.rd * 8;
    std::memcpy(hidx, lm_index, (size_t)m * 4);
    std::memcpy(hz, z, zb);
    // one copy: the pinned slot mirrors the device staging layout [indices, padded to 256 B | z]
    static_assert(sizeof(int32_t) == 4, "layout");
    if (L.off_z - L.off_idx != align256((size_t)f->cfg.max_visible * 4)) return fail(EKF_ERR_STATE, "staging layout");
    // The kernels read the frame's detections (128 + 768 bytes at m = 32) straight from the pinned slot: a host-to-device
    // copy in front of them costs more (API call + DMA start, ~8 us before the front kernel begins) than the PCIe reads
    // cost the kernel's first round trip.  The slot is not reused before this frame's event (64-slot ring).
    rc = enqueue_frame(f, hidx, hz, m, nullptr, true);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(f->slot_done[f->slot], f->stream));
    f->slot = (f->slot + 1) % kStageSlots;
    return EKF_OK;
}

int ekf_observe_device(ekf_filter* f, const int32_t* lm_index_dev, const double* z_dev, int32_t m) {
    int rc = check_ready(f);
    if (rc) return rc;
    if (m < 1) return fail(EKF_ERR_INVALID, "observe needs at least one detection");
    if (m > f->cfg.max_visible) return fail(EKF_ERR_CAPACITY, "more detections than max_visible");
    if (!lm_index_dev || !z_dev) return fail(EKF_ERR_INVALID, "NULL detections");
    if (f->n_lm < 1) return fail(EKF_ERR_STATE, "observe before any landmark was added");
    // the indices are device-resident: range-checked by the kernels (EKF_ERR_INVALID at the next sync)
    return enqueue_frame(f, lm_index_dev, z_dev, m, nullptr, true);
}

int ekf_observe_sequence_device(ekf_filter* f, const int32_t* lm_index_dev, const double* z_dev,
                                int32_t m, int32_t frames, double* trajectory_dev) {
    int rc = check_ready(f);
    if (rc) return rc;
    if (m < 1 || frames < 0) return fail(EKF_ERR_INVALID, "bad sequence shape");
    if (m > f->cfg.max_visible) return fail(EKF_ERR_CAPACITY, "more detections than max_visible");
    if (!lm_index_dev || !z_dev) return fail(EKF_ERR_INVALID, "NULL detections");
    if (f->n_lm < 1) return fail(EKF_ERR_STATE, "observe before any landmark was added");
    // Pipelined mode (F(t+1) beside C(t), see below): flags bit 1 forces it, bit 0 forbids it, otherwise it is chosen where it
    // was measured to win.  MFMA covariance update (f32 or f64) and the fused front kernel only.
    // tools/mode_select.py (profiles/r03_mode_select.txt: 2000 frames per call after a warm-up call, start-to-start of the
    // front kernels on the device clock, pipelined / serial in us per frame): n=32 m=3 11.1 / 16.1 - n=64 m=8 12.5 / 18.5 -
    // n=128 m=16 15.5 / 21.7 - C2 (n=256 m=16 f64) 15.9 / 21.5 - n=512 m=32 22.8 / 31.6 - n=1024 m=32 24.2 / 39.3: it wins at
    // every shape down to the smallest.  (Round 2's soak tool had it slower at C2: its timed region began with the handle's
    // very first pipelined call, which holds the one-time queue self-test.)  tools/pipeline_sweep.py, larger shapes:
    // n=1024 m=64 81.8 / 95.1 - n=2048 m=32 61.7 / 73.3 - n=2048 m=64 115.7 / 152.4 - n=4096 m=32 237 / 238 - n=4096 m=64
    // 398 / 371: there the front kernel's 273 workgroups hold every CU while they wait for the factorisation, and the update
    // cannot run beside them.
    const int dims_now = f->dims();
    const int kpad_now = (int)round_up(f->lay.rd * m, EKF_RB);
    // (measured, profiles/r03_mode_select.txt.  EKF model: pipelined wins everywhere except N > 9000 with k > 96, where the front
    // kernel's workgroups and the macro-tile update cannot share CUs.  EKF_Rotations: its chunks complete 10 + 10 m support rows
    // per frame in the pipelined form; from k = 91 (m = 13) on the serial order is faster: 28.2k vs 24.3k updates/s at n=100
    // m=13, 25.5k vs 16.4k at m=16, 15.7k vs 12.7k at n=200 m=21, 9.3k vs 9.1k at n=400 m=27; below, pipelined: 41.0k vs
    // 32.5k at n=100 m=9)
    const bool auto_on = !(dims_now > 9000 && kpad_now > 96) && !(f->cfg.model == EKF_MODEL_ROTATIONS && kpad_now > 64);
    const bool want = (f->cfg.flags & 2) != 0 || ((f->cfg.flags & 1) == 0 && auto_on);
    bool pipelined = want && f->lay.has_cov2 && !f->timing && frames >= 2 && (f->cfg.flags & 4) == 0 && kpad_now <= 192;
    if (pipelined && f->la_ok < 0) {
        // The device-side gates need the two streams on DIFFERENT hardware queues (HIP maps streams to a small pool
        // of queues): a gate that shares its queue with the launch it waits for would wait for ever.  Probe once: a
        // gate on the internal stream, the matching signal on the handle's stream, a short poll budget.
        // HIP deals its hardware queues to streams as they are created, so when the probe fails a FRESH internal stream usually
        // sits on another queue: up to eight are tried before the handle settles for the serial order.
        unsigned long long* probe = f->at<unsigned long long>(f->lay.off_sync) + 2;
        int32_t* pstat = f->at<int32_t>(f->lay.off_sync) + 8;
        f->la_ok = 0;
        for (int attempt = 0; attempt < 8 && f->la_ok == 0; ++attempt) {
            if (attempt > 0) {
                hipStream_t fresh = nullptr;
                HIP_TRY(hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
                HIP_TRY(hipStreamSynchronize(f->big));
                (void)hipStreamDestroy(f->big);
                f->big = fresh;
            }
            HIP_TRY(hipMemsetAsync(probe, 0, 64, f->stream));
            HIP_TRY(hipStreamSynchronize(f->stream));
            ekf_launch_gate(probe, 1ull, pstat, f->big, 1 << 14);
            ekf_launch_signal(probe, 1ull, f->stream);
            HIP_TRY(hipStreamSynchronize(f->big));
            HIP_TRY(hipStreamSynchronize(f->stream));
            int32_t ps = 0;
            HIP_TRY(hipMemcpy(&ps, pstat, 4, hipMemcpyDeviceToHost));
            f->la_ok = (ps == 0) ? 1 : 0;
        }
    }
    f->seq_mode = EKF_SEQ_SERIAL;
    if (pipelined && f->la_ok == 0) {
        pipelined = false;
        f->seq_mode = EKF_SEQ_SERIAL_ONE_QUEUE;
    }
    if (pipelined) {
        // one pipelining handle per process (see g_pipelining): take the token, or take it over from a handle whose streams
        // have drained, or run this call in serial order
        ekf_filter* owner = nullptr;
        if (!g_pipelining.compare_exchange_strong(owner, f) && owner != f) {
            bool idle = hipStreamQuery(owner->stream) == hipSuccess && hipStreamQuery(owner->big) == hipSuccess;
            (void)hipGetLastError();      // (hipErrorNotReady is not an error here)
            if (!(idle && g_pipelining.compare_exchange_strong(owner, f))) {
                pipelined = false;
                f->seq_mode = EKF_SEQ_SERIAL_OTHER_HANDLE;
            }
        }
    }
    if (!pipelined) {
        for (int t = 0; t < frames; ++t) {
            rc = enqueue_frame(f, lm_index_dev + (size_t)t * m, z_dev + (size_t)t * m * f->lay.rd, m,
                               trajectory_dev ? trajectory_dev + (size_t)t * 7 : nullptr, false);
            if (rc) return rc;
        }
        return EKF_OK;
    }
    // Pipelined sequence mode.  One frame is a serial chain on one P, but the front kernel F(t+1) needs of P_{t+1}
    // only its support rows (camera + the landmarks of frame t+1's detections), and those follow from P_t and W_t:
    //     P_{t+1}[r][c] = (P_t[r][c] + Q[r == c]) + sum_k fma(-W_t[k][r], W_t[k][c])
    // -- per element the instruction sequence of the covariance update, so the bits are the same.  The covariance
    // therefore ping-pongs between two buffers: C(t) reads buf[t & 1] (P_t) and writes buf[(t + 1) & 1]; F(t+1) runs
    // BESIDE C(t) on the other stream, reads P_t from buf[t & 1] and W_t, and completes the entries it needs itself
    // (S-block workgroups: from the compact support columns W_sup the chunks of F(t) left behind; chunk workgroups:
    // on the matrix cores).  Nothing on the critical path F(t) -> F(t+1) waits for a covariance update.
    //   stream A (the handle's):  F(0) - F(1) - F(2) - ... - F(last) - C(last)
    //   stream B (internal)    :  gate - C(0) - signal - gate - C(1) - signal ...
    // Edges between the streams are ordered on the device (an event pair costs ~13 us per edge):
    //   F(t) complete   -> C(t) may start (it reads W_t and overwrites the buffer F(t) read, P_{t-1}): F(t+1) stores
    //                      "t+1 started" when it starts (it follows F(t) on stream A); a one-wave gate kernel in front
    //                      of C(t) polls that counter (the last update of a call follows its front kernel on stream A);
    //   C(t-1) complete -> F(t+1) may read buf[t & 1] and overwrite W_{t-1}: a one-thread kernel behind C(t-1) bumps a
    //                      second counter; F(t) does not finish before it has seen it (its measurement workgroup polls
    //                      at its end), and F(t+1) follows F(t) on stream A.
    // Kernel boundaries on each stream give the memory ordering; the counters only carry "that launch is over".
    // Every wait is bounded.  The front kernel claims (almost) all LDS of its CUs while its grid is small, so the
    // covariance update's workgroups run on the other CUs instead of next to the pivot chain.
    const Layout& L = f->lay;
    rc = ensure_tiles(f);
    if (rc) return rc;
    void* wbuf[2] = {f->at<void>(L.off_wpanel), f->at<void>(L.off_wpanel2)};
    void* cbuf[2] = {f->cov, f->at<void>(L.off_cov2)};
    char* wsup0 = f->at<char>(L.off_wsup);
    void* wsup[2] = {wsup0, wsup0 + (size_t)L.kmax * L.wsup_ld * L.elem};
    unsigned long long* sync = f->at<unsigned long long>(L.off_sync);
    int32_t* status = f->at<int32_t>(L.off_status);
    const uint64_t base = f->la_base;
    const int nb_now = (int)round_up(L.rd * m, EKF_RB) / EKF_RB;
    const int grid_now = nb_now * (nb_now + 1) / 2 + 2 + (int)round_up(f->dims(), 128) / 64;
    const int la_lds = grid_now <= 100 ? 148 * 1024 : 0;
    // (stream B needs no edge from stream A at the start: its first launch is the gate in front of C(0), which waits
    // for "F(1) has started", i.e. for everything that is on stream A now and F(0); the previous call ended with
    // stream A waiting for stream B)
    for (int t = 0; t < frames; ++t) {
        const int par = t & 1;
        EkfFrame fr = make_frame(f, lm_index_dev + (size_t)t * m, z_dev + (size_t)t * m * L.rd, m,
                                 trajectory_dev ? trajectory_dev + (size_t)t * 7 : nullptr);
        fr.wpanel = wbuf[par];
        if (t > 0) {                                           // P_{t-1} + the completion from W_{t-1}
            fr.cov = cbuf[par ^ 1];
            fr.wprev = wbuf[par ^ 1];
            fr.wsup_prev = wsup[par ^ 1];
        }
        fr.wsup_ld = L.wsup_ld;
        fr.la_sync = sync;
        fr.la_signal = (t > 0) ? base + (uint64_t)t : 0;       // "F(t) has started": F(t-1) is complete
        fr.la_gate = (t > 0) ? base + (uint64_t)t : 0;         // C(t-1) complete before F(t) ends
        fr.lds_min = la_lds;
        if (t + 1 < frames) {                                  // the next frame's detections: its support columns of W_t
            fr.next_idx = lm_index_dev + (size_t)(t + 1) * m;
            fr.next_m = m;
            fr.wsup = wsup[par];
        }
        bind_exchange(f, fr);
        if (L.elem == 4) ekf_launch_front<float>(fr, f->stream); else ekf_launch_front<double>(fr, f->stream);
        EkfFrame cf = fr;
        cf.cov = cbuf[par];
        cf.cov_out = cbuf[par ^ 1];
        if (t + 1 < frames) {
            ekf_launch_gate(sync, base + (uint64_t)t + 1, status, f->big);
            if (L.elem == 4) ekf_launch_cov_update<float>(cf, 2, f->big); else ekf_launch_cov_update<double>(cf, 2, f->big);
            ekf_launch_signal(sync + 1, base + (uint64_t)t + 1, f->big);
        } else {
            // The LAST update of the call runs on the handle's stream, straight behind its front kernel: stream order says
            // that F(t) is over, and F(t) did not finish before it had seen C(t-1) complete (its end gate) -- no gate, no
            // signal, and nothing to join afterwards (4 device-side hops of ~1.2 us per call).  The counters keep the
            // values of frame t - 1; the next call's waits are for values beyond base + frames, which its own launches set.
            if (L.elem == 4) ekf_launch_cov_update<float>(cf, 2, f->stream); else ekf_launch_cov_update<double>(cf, 2, f->stream);
            // an odd number of frames leaves the covariance in the internal buffer: back into the caller's
            if (frames & 1)
                HIP_TRY(hipMemcpyAsync(f->cov, cbuf[1], (size_t)L.cap * L.cap * L.elem, hipMemcpyDeviceToDevice, f->stream));
        }
        HIP_TRY(hipGetLastError());
    }
    f->la_base = base + (uint64_t)frames;
    f->seq_mode = EKF_SEQ_PIPELINED;
    f->front_pending = false;
    f->status_clean = false;       // (gate kernels raise status bits without the host word)
    f->last_m = m;
    return EKF_OK;
}

// ---- detection -> pose front end (base_filter.py:92-171): stateless, no filter handle -------------------------
static int make_camera(const double camera_matrix[9], const double* dist_coeffs, int32_t n_dist, EkfCamera* cam) {
    if (!camera_matrix) return fail(EKF_ERR_INVALID, "camera matrix is NULL");
    if (n_dist < 0 || n_dist > 8 || (n_dist > 0 && !dist_coeffs))
        return fail(EKF_ERR_INVALID, "0..8 distortion coefficients (k1 k2 p1 p2 k3 k4 k5 k6) are supported");
    if (!(camera_matrix[0] > 0.0) || !(camera_matrix[4] > 0.0)) return fail(EKF_ERR_INVALID, "focal lengths must be > 0");
    cam->fx = camera_matrix[0];
    cam->fy = camera_matrix[4];
    cam->cx = camera_matrix[2];
    cam->cy = camera_matrix[5];
    for (int i = 0; i < 8; ++i) cam->k[i] = (i < n_dist) ? dist_coeffs[i] : 0.0;
    return EKF_OK;
}

int ekf_estimate_poses_device(const double* corners_dev, int32_t count, double marker_size, const double camera_matrix[9],
                              const double* dist_coeffs, int32_t n_dist, double* poses_dev, void* stream) {
    if (count < 0) return fail(EKF_ERR_INVALID, "negative marker count");
    if (count == 0) return EKF_OK;
    if (!corners_dev || !poses_dev) return fail(EKF_ERR_INVALID, "NULL device buffer");
    if (!(marker_size > 0.0)) return fail(EKF_ERR_INVALID, "marker_size must be > 0");
    EkfCamera cam;
    int rc = make_camera(camera_matrix, dist_coeffs, n_dist, &cam);
    if (rc) return rc;
    ekf_launch_ippe_square(corners_dev, count, marker_size, cam, poses_dev, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return EKF_OK;
}

int ekf_estimate_poses(const double* corners, int32_t count, double marker_size, const double camera_matrix[9],
                       const double* dist_coeffs, int32_t n_dist, double* poses, void* stream) {
    if (count < 0) return fail(EKF_ERR_INVALID, "negative marker count");
    if (count == 0) return EKF_OK;
    if (!corners || !poses) return fail(EKF_ERR_INVALID, "NULL host buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* dev = nullptr;
    HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&dev), (size_t)count * 14 * 8, s));
    hipError_t e = hipMemcpyAsync(dev, corners, (size_t)count * 8 * 8, hipMemcpyHostToDevice, s);
    int rc = EKF_OK;
    if (e == hipSuccess) {
        rc = ekf_estimate_poses_device(dev, count, marker_size, camera_matrix, dist_coeffs, n_dist, dev + (size_t)count * 8, s);
        if (rc == EKF_OK) e = hipMemcpyAsync(poses, dev + (size_t)count * 8, (size_t)count * 6 * 8, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFreeAsync(dev, s);
    if (rc) return rc;
    if (e != hipSuccess) return fail(EKF_ERR_HIP, std::string("ekf_estimate_poses: ") + hipGetErrorString(e));
    return EKF_OK;
}

int ekf_sync(ekf_filter* f) {
    int rc = check_ready(f);
    if (rc) return rc;
    return sync_and_check(f);
}

int ekf_num_landmarks(const ekf_filter* f) { return f ? f->n_lm : EKF_ERR_INVALID; }

int ekf_get_state(ekf_filter* f, double* out, int32_t count) {
    int rc = check_ready(f);
    if (rc) return rc;
    if (!out || count < 0 || count > f->dims()) return fail(EKF_ERR_INVALID, "bad state request");
    rc = sync_and_check(f, count);
    if (rc) return rc;
    std::memcpy(out, f->readback + 256, (size_t)count * 8);
    return EKF_OK;
}

int ekf_get_camera(ekf_filter* f, double out[10]) { return ekf_get_state(f, out, EKF_CAM); }

int ekf_get_cov_diag(ekf_filter* f, double* out, int32_t count) {
    int rc = check_ready(f);
    if (rc) return rc;
    if (!out || count < 0 || count > f->dims()) return fail(EKF_ERR_INVALID, "bad diag request");
    double* scratch = f->at<double>(f->lay.off_diag);
    if (f->lay.elem == 4) ekf_launch_cov_diag<float>(f->cov, f->ld, scratch, count, f->stream);
    else ekf_launch_cov_diag<double>(f->cov, f->ld, scratch, count, f->stream);
    HIP_TRY(hipGetLastError());
    rc = sync_and_check(f);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, scratch, (size_t)count * 8, hipMemcpyDeviceToHost));
    return EKF_OK;
}

int ekf_get_cov(ekf_filter* f, double* out, int32_t dims) {
    int rc = check_ready(f);
    if (rc) return rc;
    if (!out || dims != f->dims()) return fail(EKF_ERR_INVALID, "dims must equal the state dimension");
    rc = sync_and_check(f);
    if (rc) return rc;
    const size_t el = f->lay.elem;
    if (el == 8) {
        HIP_TRY(hipMemcpy2D(out, (size_t)dims * 8, f->cov, (size_t)f->ld * 8, (size_t)dims * 8, dims,
                            hipMemcpyDeviceToHost));
    } else {
        std::vector<float> tmp((size_t)dims * dims);
        HIP_TRY(hipMemcpy2D(tmp.data(), (size_t)dims * 4, f->cov, (size_t)f->ld * 4, (size_t)dims * 4,
                            dims, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); ++i) out[i] = (double)tmp[i];
    }
    return EKF_OK;
}

int ekf_set_state(ekf_filter* f, const double* state, int32_t num_landmarks) {
    int rc = check_ready(f);
    if (rc) return rc;
    if (!state || num_landmarks < 0) return fail(EKF_ERR_INVALID, "bad state");
    if (num_landmarks > f->cfg.max_landmarks)
        return fail(EKF_ERR_CAPACITY, "more landmarks than max_landmarks");
    f->front_pending = false;
    f->mirror_trust = false;
    HIP_TRY(hipStreamSynchronize(f->stream));
    HIP_TRY(hipMemset(f->state, 0, (size_t)f->lay.cap * 8));
    HIP_TRY(hipMemcpy(f->state, state, (size_t)(f->lay.lmd * num_landmarks + EKF_CAM) * 8,
                      hipMemcpyHostToDevice));
    f->n_lm = num_landmarks;
    return EKF_OK;
}

int ekf_set_cov(ekf_filter* f, const double* cov, int32_t dims) {
    int rc = check_ready(f);
    if (rc) return rc;
    if (!cov || dims != f->dims())
        return fail(EKF_ERR_INVALID, "dims must equal the state dimension (call ekf_set_state first)");
    const Layout& L = f->lay;
    HIP_TRY(hipStreamSynchronize(f->stream));
    HIP_TRY(hipMemset(f->cov, 0, (size_t)L.cap * L.cap * L.elem));
    // symmetrise on upload: the kernels keep P bitwise symmetric from then on
    if (L.elem == 8) {
        std::vector<double> tmp((size_t)dims * dims);
        for (int i = 0; i < dims; ++i)
            for (int j = 0; j < dims; ++j)
                tmp[(size_t)i * dims + j] = 0.5 * (cov[(size_t)i * dims + j] + cov[(size_t)j * dims + i]);
        HIP_TRY(hipMemcpy2D(f->cov, (size_t)f->ld * 8, tmp.data(), (size_t)dims * 8, (size_t)dims * 8,
                            dims, hipMemcpyHostToDevice));
    } else {
        std::vector<float> tmp((size_t)dims * dims);
        for (int i = 0; i < dims; ++i)
            for (int j = 0; j < dims; ++j)
                tmp[(size_t)i * dims + j] =
                    (float)(0.5 * (cov[(size_t)i * dims + j] + cov[(size_t)j * dims + i]));
        HIP_TRY(hipMemcpy2D(f->cov, (size_t)f->ld * 4, tmp.data(), (size_t)dims * 4, (size_t)dims * 4,
                            dims, hipMemcpyHostToDevice));
    }
    return EKF_OK;
}

int ekf_last_sequence_mode(const ekf_filter* f) { return f ? f->seq_mode : EKF_ERR_INVALID; }

int ekf_set_fused(ekf_filter* f, int32_t enable) {
    if (!f) return fail(EKF_ERR_INVALID, "filter handle is NULL");
    if (enable) f->cfg.flags &= ~4; else f->cfg.flags |= 4;
    return EKF_OK;
}

int ekf_set_kernel_timing(ekf_filter* f, int32_t enable) {
    if (!f) return fail(EKF_ERR_INVALID, "filter handle is NULL");
    HIP_TRY(hipStreamSynchronize(f->stream));
    if (enable && f->ev.empty()) {
        f->ev.resize((size_t)kEventPool * (kTimedKernels + 1));
        for (auto& e : f->ev) HIP_TRY(hipEventCreate(&e));
    }
    f->timing = enable != 0;
    f->timing_cov_only = enable == 2;
    f->ev_frames = 0;
    for (int i = 0; i < kTimedKernels; ++i) {
        f->t_sum_us[i] = 0.0;
        f->t_cnt[i] = 0;
    }
    return EKF_OK;
}

int ekf_get_kernel_timing(ekf_filter* f, int32_t which, double* mean_us, int64_t* launches) {
    if (!f) return fail(EKF_ERR_INVALID, "filter handle is NULL");
    if (which < 0 || which >= kTimedKernels) return fail(EKF_ERR_INVALID, "which must be 0..3");
    int rc = fold_timing(f);
    if (rc) return rc;
    if (mean_us) *mean_us = f->t_cnt[which] ? f->t_sum_us[which] / (double)f->t_cnt[which] : 0.0;
    if (launches) *launches = f->t_cnt[which];
    return EKF_OK;
}

int ekf_debug_fetch(ekf_filter* f, int32_t what, double* out, size_t count) {
    int rc = check_ready(f);
    if (rc) return rc;
    if (!out) return fail(EKF_ERR_INVALID, "out is NULL");
    if (what == -1) {           // enable the f64 copy of W in subsequent frames
        f->debug_w = true;
        return EKF_OK;
    }
    if (what == -2 || what == -3) {   // in-kernel time stamps only (no debug copies: the production code path); -3: role level only
        f->debug_stamps = true;
        f->debug_stamps_light = what == -3;
        return EKF_OK;
    }
    const Layout& L = f->lay;
    const int k = L.rd * f->last_m, kp = (int)round_up(k, EKF_RB), dims = f->dims();
    const int jc = f->cfg.model == EKF_MODEL_ROTATIONS ? 20 : EKF_JCOLS;
    HIP_TRY(hipStreamSynchronize(f->stream));
    switch (what) {
        case 0:
            if (count < (size_t)k * jc) return fail(EKF_ERR_INVALID, "out too small");
            HIP_TRY(hipMemcpy2D(out, (size_t)jc * 8, f->at<double>(L.off_jac), EKF_JLD * 8, (size_t)jc * 8,
                                k, hipMemcpyDeviceToHost));
            return EKF_OK;
        case 1:
            if (count < (size_t)k) return fail(EKF_ERR_INVALID, "out too small");
            HIP_TRY(hipMemcpy(out, f->at<double>(L.off_resid), (size_t)k * 8, hipMemcpyDeviceToHost));
            return EKF_OK;
        case 2:
            if (count < (size_t)kp * kp) return fail(EKF_ERR_INVALID, "out too small");
            HIP_TRY(hipMemcpy2D(out, (size_t)kp * 8, f->at<double>(L.off_lmat), (size_t)L.kmax * 8,
                                (size_t)kp * 8, kp, hipMemcpyDeviceToHost));
            return EKF_OK;
        case 3:
            if (!f->debug_w) return fail(EKF_ERR_STATE, "call ekf_debug_fetch(f,-1,..) first");
            if (count < (size_t)kp * dims) return fail(EKF_ERR_INVALID, "out too small");
            HIP_TRY(hipMemcpy2D(out, (size_t)dims * 8, f->at<double>(L.off_wdbg), (size_t)L.cap * 8,
                                (size_t)dims * 8, kp, hipMemcpyDeviceToHost));
            return EKF_OK;
        case 4:
            if (count < (size_t)k * dims) return fail(EKF_ERR_INVALID, "out too small");
            HIP_TRY(hipMemcpy2D(out, (size_t)dims * 8, f->at<double>(L.off_amat), (size_t)L.cap * 8,
                                (size_t)dims * 8, k, hipMemcpyDeviceToHost));
            return EKF_OK;
        case 5:
            if (count < 64) return fail(EKF_ERR_INVALID, "out too small");
            {
                long long st[64];
                HIP_TRY(hipMemcpy(st, f->at<long long>(L.off_stamps), sizeof(st), hipMemcpyDeviceToHost));
                for (int i = 0; i < 64; ++i) out[i] = (double)st[i];
            }
            return EKF_OK;
        case 6:      // counters of the macro-tile covariance update (diagnostic builds only), then cleared
            if (count < 32) return fail(EKF_ERR_INVALID, "out too small");
            {
                unsigned long long st[32];
                HIP_TRY(hipMemcpy(st, f->at<unsigned long long>(L.off_covstats), sizeof(st), hipMemcpyDeviceToHost));
                for (int i = 0; i < 32; ++i) out[i] = (double)st[i];
                HIP_TRY(hipMemset(f->at<unsigned long long>(L.off_covstats), 0, sizeof(st)));
            }
            return EKF_OK;
        default:
            return fail(EKF_ERR_INVALID, "unknown debug item");
    }
}

}  // extern "C"
