/*
 * ekf_slam_hip.h -- C ABI of the MI355X (gfx950) EKF-SLAM update path.
 *
 * Drop-in boundary for the reference filter back-end
 *   /root/reference/filters/extended_kalman_filter.py  (class EKF)
 * as it is driven by
 *   /root/reference/filters/base_filter.py:203-207  (observe, get_poses)
 *   /root/reference/filters/base_filter.py:214-247  (save_map getters)
 *
 * Plain C, no torch / C++ types.  Every function returns 0 (EKF_OK) or a
 * negative error code; ekf_last_error_string() describes the last failure on
 * the calling thread.  A filter handle has exactly one caller thread; all
 * device work is enqueued on the handle's HIP stream; observe calls return
 * before the GPU has finished, getters synchronise.
 *
 * Memory: the covariance, state and workspace live in DEVICE memory owned by
 * the caller (the Python shim passes torch tensor data_ptr()s).  The library
 * never allocates or frees them; they must outlive the handle.
 *
 * State layout (extended_kalman_filter.py:29-34,46-51):
 *   state  f64 [3 n + 10] = [x y z | qw qx qy qz | ex ey ez | l0 | l1 | ...]
 *   cov    f32 or f64, row-major, leading dimension `ld` (capacity-padded to a
 *          multiple of 128; rows/cols >= 3 n + 10 are kept exactly zero)
 */
#ifndef EKF_SLAM_HIP_H
#define EKF_SLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ekf_filter ekf_filter; /* opaque handle */

enum { EKF_COV_F64 = 0, EKF_COV_F32 = 1 };
/* camera quaternion injection (extended_kalman_filter.py:138-149):
 * AS_WRITTEN reproduces the reference exactly (scalar-first arrays passed to
 * SciPy's scalar-last from_quat); SCALAR_FIRST is the consistent convention
 * (what ekf_with_rotations.py:150-151 does). */
enum { EKF_QUAT_AS_WRITTEN = 0, EKF_QUAT_SCALAR_FIRST = 1 };
/* filter model: EKF = extended_kalman_filter.py (landmark xyz, 3 rows per detection);
 * EKF_ROTATIONS = ekf_with_rotations.py (landmark [xyz | quat | err], 10 dims, 7 rows per
 * detection [xyz_cl ; q_cl], consistent quaternion convention, q_cam default 0.2) */
enum { EKF_MODEL_EKF = 0, EKF_MODEL_ROTATIONS = 1 };
/* covariance-update kernel selection (0 = best available: as EKF_COVK_MFMA).  MFMA: the symmetric matrix-core kernels,
 * one wave per 32 x 32 tile, and for the f32 covariance from 3000 tiles of 128 x 128 (n >= 3300 or so) one workgroup per
 * 128 x 128 macro tile with LDS-staged operands; MFMA_TILE / MFMA_MACRO force one of the two (tests, measurements; MACRO:
 * f32 only).  All of them give the same bits as the VALU reference kernel. */
enum { EKF_COVK_AUTO = 0, EKF_COVK_VALU = 1, EKF_COVK_MFMA = 2, EKF_COVK_MFMA_TILE = 3, EKF_COVK_MFMA_MACRO = 4 };

enum {
    EKF_OK = 0,
    EKF_ERR_INVALID = -1,   /* bad argument */
    EKF_ERR_CAPACITY = -2,  /* more landmarks / observations than configured */
    EKF_ERR_HIP = -3,       /* a HIP runtime call failed */
    EKF_ERR_STATE = -4,     /* buffers not bound, filter not reset, ... */
    EKF_ERR_NUMERIC = -5    /* innovation covariance not positive definite, or an internal integrity check of the
                             * front kernel's in-launch exchange tripped */
};

typedef struct ekf_config {
    int32_t max_landmarks;  /* capacity n_max */
    int32_t max_visible;    /* max observations per frame (<= 64; <= 50 for EKF_MODEL_ROTATIONS: k = 7 m <= 350 rows; beyond k = 192
                             * the frame runs through the stage kernels, in serial order) */
    int32_t cov_dtype;      /* EKF_COV_F64 / EKF_COV_F32 */
    int32_t quat_mode;      /* EKF_QUAT_* */
    int32_t cov_kernel;     /* EKF_COVK_*: covariance-update kernel */
    int32_t model;          /* EKF_MODEL_* */
    int32_t reserved;
    int32_t flags;          /* bits 0-1: pipelined mode of ekf_observe_sequence_device (the front kernel of frame t+1 runs
                             * beside the covariance update of frame t; same results, bit for bit): 0 = chosen by size
                             * (MFMA covariance update only, either dtype; on except above 9000 state dimensions
                             * with more than 32 detections per frame), bit 0 = never, bit 1 = always.  Unless bit 0
                             * is set, a capable configuration adds a second covariance buffer to the workspace
                             * (ekf_query_sizes: + ld^2 elements).  One handle per process pipelines at a time
                             * (ekf_last_sequence_mode).
                             * bit 2: run gather / solve / panel as three separate launches instead of the fused front
                             * kernel (same results, bit for bit; no pipelined mode).  bit 3 is ignored. */
    /* noise constants, defaults = extended_kalman_filter.py:21-27 */
    double initial_camera_uncertainty;   /* 0.1  */
    double initial_landmark_uncertainty; /* 0.7  */
    double r_uncertainty;                /* 0.9  */
    double q_cam;                        /* 0.3  */
    double q_err;                        /* 0.5  */
    double q_lm;                         /* 0.01 */
    void *stream;                        /* hipStream_t, NULL = default */
} ekf_config;

/* Fill `cfg` with the reference's constants (extended_kalman_filter.py:19-34). */
int ekf_default_config(ekf_config *cfg);

/* Sizes of the caller-owned device buffers for this configuration. */
int ekf_query_sizes(const ekf_config *cfg, int64_t *ld, size_t *cov_bytes,
                    size_t *state_bytes, size_t *workspace_bytes);

/* EKF.__init__ (extended_kalman_filter.py:40-56) without the SymPy step. */
int ekf_create(const ekf_config *cfg, ekf_filter **out);
int ekf_destroy(ekf_filter *f);

/* Borrow caller-owned device memory (see ekf_query_sizes). */
int ekf_bind_buffers(ekf_filter *f, void *cov_dev, int64_t ld, double *state_dev,
                     void *workspace_dev, size_t workspace_bytes);

/* Capacity growth.  The reference appends landmarks without limit (extended_kalman_filter.py:274-290: hstack / block
 * matrix per add_marker); here the capacity is fixed by the buffers, and a filter that is about to exceed max_landmarks
 * moves into larger ones (likewise for more detections per frame than max_visible): the caller sizes new buffers with
 * ekf_query_sizes for the same configuration with the new max_landmarks / max_visible (neither may shrink), and the library copies state, covariance (device to device, capacity padding zero) and
 * the status word, re-arms its workspace and borrows the new buffers from then on; the old ones may be freed when the call
 * returns.  The filter continues bit for bit as one that had been created with the larger capacity. */
int ekf_grow(ekf_filter *f, int32_t new_max_landmarks, int32_t new_max_visible, void *cov_dev, int64_t ld,
             double *state_dev, void *workspace_dev, size_t workspace_bytes);

/* state = initial pose, P = 0.1 I_10, no landmarks
 * (extended_kalman_filter.py:46-51). */
int ekf_reset(ekf_filter *f, const double initial_camera_pose[10]);

/* EKF.add_marker for `count` new landmarks (extended_kalman_filter.py:239-290):
 * t_ml = R(q)^-1 * cam_frame_xyz + cam_xyz with the CURRENT camera state,
 * appended to the state; P grows by diag(diag_uncertainty) (or 0.7 I if NULL).
 * Host pointers: cam_frame_xyz [count,3], diag_uncertainty [count,3] or NULL.
 * EKF_MODEL_ROTATIONS (ekf_with_rotations.py:275-335): cam_frame_xyz is the full pose
 * [count,6] = [tvec | rvec] (rvec read as extrinsic xyz Euler angles, :307-310), the new
 * landmark is [t_ml | q_ml | 0 0 0], diag_uncertainty is [count,10].
 * The new landmarks get indices n .. n+count-1 (the marker-id -> index dict
 * stays on the Python side). */
int ekf_add_markers(ekf_filter *f, const double *cam_frame_xyz,
                    const double *diag_uncertainty, int32_t count);

/* EKF.predict + EKF.update for one frame (extended_kalman_filter.py:95-156):
 * lm_index [m] = landmark indices of the visible markers in `ids` order
 * (duplicates legal), z [m,3] = pose[0:3] of every detection (EKF_MODEL_ROTATIONS: z [m,7] =
 * [pose[0:3] | quaternion of from_euler("xyz", pose[3:6]), scalar first], :216-224).
 * ekf_observe takes host pointers (copied during the call; indices are range-checked at once);
 * ekf_observe_device takes device pointers that must stay valid until the
 * stream has consumed them.  Device-resident indices are range-checked BY THE KERNELS: an index
 * outside [0, num_landmarks) is clamped (nothing is read or written out of bounds) and reported as
 * EKF_ERR_INVALID by the next synchronising call (ekf_sync, any getter); the error is sticky until
 * ekf_reset, because the frames computed from clamped indices have already changed the filter. */
int ekf_observe(ekf_filter *f, const int32_t *lm_index, const double *z, int32_t m);
int ekf_observe_device(ekf_filter *f, const int32_t *lm_index_dev,
                       const double *z_dev, int32_t m);

/* `frames` consecutive observe() calls on device-resident detections
 * lm_index_dev [frames,m], z_dev [frames,m,3]; after every frame the camera
 * pose state[0:7] is appended to trajectory_dev [frames,7] (may be NULL).
 * Pipelined mode (see ekf_config.flags): the covariance update of frame t runs on an internal
 * second stream BESIDE frame t+1's front kernel.  The covariance ping-pongs between the caller's
 * buffer and a second one in the workspace (the update reads P_t and writes P_{t+1} elsewhere), and
 * the front kernel of frame t+1 completes the few rows of P_{t+1} it reads from P_t and W_t on the
 * fly, with the update's own per-element instruction sequence: results are bitwise those of
 * per-frame ekf_observe calls.  The two streams are ordered by one-wave gate kernels on the
 * device, every wait bounded.  The last update of a call runs on the handle's stream again, so after the
 * call that stream alone orders everything that follows, and the covariance is back in the caller's buffer. */
int ekf_observe_sequence_device(ekf_filter *f, const int32_t *lm_index_dev,
                                const double *z_dev, int32_t m, int32_t frames,
                                double *trajectory_dev);
/* What the last ekf_observe_sequence_device call of this handle did (results are the same bits in every mode):
 * PIPELINED; SERIAL (not asked for / not chosen for this size, fewer than 2 frames, kernel timing on, stage kernels);
 * SERIAL_ONE_QUEUE: asked for, but the handle's two streams share one hardware queue (found by a self-test on first use);
 * SERIAL_OTHER_HANDLE: asked for, but another handle of the process has a pipelined call in flight -- the device-side
 * gates of two handles could wait for each other across the shared hardware queues, so only one handle pipelines at a time. */
enum { EKF_SEQ_NONE = 0, EKF_SEQ_SERIAL = 1, EKF_SEQ_PIPELINED = 2, EKF_SEQ_SERIAL_ONE_QUEUE = 3, EKF_SEQ_SERIAL_OTHER_HANDLE = 4 };
int ekf_last_sequence_mode(const ekf_filter *f);

/* EKF.get_poses / get_lm_uncertainties (extended_kalman_filter.py:84-93).
 * Synchronise and copy to host.  ekf_get_camera / ekf_get_state directly after ekf_observe / ekf_observe_device wait
 * for the part of that frame that produces the state (and the status word) only -- the front kernel leaves the state in a
 * pinned host mirror, so nothing is copied -- and the covariance update of the frame may still be running when they return.  ekf_get_cov_diag / ekf_get_cov / ekf_sync wait for everything. */
int ekf_get_camera(ekf_filter *f, double out[10]);
int ekf_get_state(ekf_filter *f, double *out, int32_t count);
int ekf_get_cov_diag(ekf_filter *f, double *out, int32_t count);
/* Full covariance as host f64 [dims,dims] (tests, checkpointing). */
int ekf_get_cov(ekf_filter *f, double *out, int32_t dims);

/* Restore (state, P) from host f64 (map restore / teacher-forced tests).
 * cov is [dims,dims] with dims = 3*num_landmarks+10; it is symmetrised
 * ((P+P^T)/2) on upload. */
int ekf_set_state(ekf_filter *f, const double *state, int32_t num_landmarks);
int ekf_set_cov(ekf_filter *f, const double *cov, int32_t dims);

int ekf_num_landmarks(const ekf_filter *f);
int ekf_sync(ekf_filter *f);

/* Front part of the update (measurement model, S, Cholesky, W, dx, injection): enable != 0 = the
 * fused front kernel (default), 0 = the three stage kernels (gather / solve / panel).  Same results,
 * bit for bit; may be switched between any two frames (same as ekf_config.flags bit 2). */
int ekf_set_fused(ekf_filter *f, int32_t enable);

/* Per-kernel device timing with HIP events on the handle's stream.
 * which: 0 gather, 1 solve, 2 panel, 3 covariance update.
 * enable: 0 off, 1 all four kernels (5 events per frame), 2 covariance update only
 * (2 events per frame: least perturbation of the pipeline).
 * ekf_get_kernel_timing synchronises, returns the mean duration [us] and the
 * launch count since the last enable, then clears the accumulated events. */
int ekf_set_kernel_timing(ekf_filter *f, int32_t enable);
int ekf_get_kernel_timing(ekf_filter *f, int32_t which, double *mean_us, int64_t *launches);

/* Last frame's intermediates as host f64 (tests): what = 0 Jacobian blocks
 * [k,13], 1 residual [k], 2 Cholesky factor L [kpad,kpad], 3 whitened panel
 * W = L^-1 H P [kpad,dims], 4 A = H (P+Q) [k,dims].  `count` = capacity of out. */
int ekf_debug_fetch(ekf_filter *f, int32_t what, double *out, size_t count);

/* Detection -> pose front end, batched (replaces the per-marker loop of cv2.solvePnP(..., SOLVEPNP_IPPE_SQUARE) in
 * BaseFilter.estimate_pose_of_markers, filters/base_filter.py:92-171).  Stateless: no filter handle.
 * corners [count,4,2]: pixel coordinates of each marker's corners in the detector's order (top-left, top-right,
 * bottom-right, bottom-left = object points (-s/2, s/2), (s/2, s/2), (s/2, -s/2), (-s/2, -s/2), base_filter.py:113-121);
 * camera_matrix: row-major 3x3 (fx, cx, fy, cy are read); dist_coeffs: n_dist <= 8 values k1 k2 p1 p2 k3 k4 k5 k6;
 * poses [count,6] = [tvec | rvec] per marker, the layout observe() takes (base_filter.py:166-171).
 * The `_device` form enqueues one kernel on `stream` (device pointers); the host form copies, runs and synchronises. */
int ekf_estimate_poses_device(const double *corners_dev, int32_t count, double marker_size,
                              const double camera_matrix[9], const double *dist_coeffs, int32_t n_dist,
                              double *poses_dev, void *stream);
int ekf_estimate_poses(const double *corners, int32_t count, double marker_size,
                       const double camera_matrix[9], const double *dist_coeffs, int32_t n_dist,
                       double *poses, void *stream);

const char *ekf_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif /* EKF_SLAM_HIP_H */
