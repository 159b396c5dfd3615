/*
 * mocap_hip.h -- C-ABI of libmocap_hip.so, the MI355X (gfx950) implementation of MocapV2's per-frame hot path.
 *
 * Plain C, pointers and sizes only.  Every entry point returns 0 on success or a negative MOCAP_E_* code;
 * mocap_last_error() returns a thread-local description.  No C++ exception crosses the boundary.  The library
 * never owns caller memory: `*_dev` arguments are raw HIP device pointers supplied by the caller (e.g.
 * torch.Tensor.data_ptr()), `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 * enqueued on `stream` and is asynchronous unless stated otherwise.
 *
 * Each declaration cites the reference interface (RashmikaDushan/MocapV2) it stands in for.
 */
#ifndef MOCAP_HIP_H
#define MOCAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOCAP_ABI_VERSION 5
#define MOCAP_API __attribute__((visibility("default")))

enum {
    MOCAP_OK = 0,
    MOCAP_E_INVALID = -1,     /* bad argument */
    MOCAP_E_HIP = -2,         /* a HIP runtime call failed (no GPU, out of memory, ...) */
    MOCAP_E_UNSUPPORTED = -3, /* parameter outside what the kernels implement */
    MOCAP_E_STATE = -4        /* required set-up call missing */
};

/* per-image status written to out_count by the blob kernels when an internal capacity is exceeded */
enum {
    MOCAP_BLOB_E_CANDIDATES = -2, /* more than 1024 border start candidates in one image */
    MOCAP_BLOB_E_CONTOURS = -3,   /* more than 384 borders or 256 kept contours in one image */
    MOCAP_BLOB_E_STEPS = -4,      /* a border longer than the step limit */
    MOCAP_BLOB_E_DEPTH = -5,      /* a kept contour nested deeper than 8 levels */
    MOCAP_CORR_E_GROUPS = -2,     /* more than 16 candidates for one (root, camera), > max_groups groups for one root, or more
                                     groups in the whole time step than its share of the error scratch holds: max(2 * max_groups,
                                     8192) by default, raised with mocap_set_tuning(ctx, "corr_step_groups", n) */
    MOCAP_CORR_E_TRUNCATED = -3,  /* a camera holds more image points than the P that mocap_correspond was told to read */
    MOCAP_CORR_E_BLOB = -4        /* a camera's point count is negative: its blob stage reported MOCAP_BLOB_E_* */
};

typedef struct mocap_ctx* mocap_ctx_t;

/* The literals of reference lib/ImageOperations.py:19-20,28-30,50 as a parameter block (same defaults). */
typedef struct mocap_blob_params {
    int32_t ksize;        /* box blur window, 5   (ImageOperations.py:28)  -- only 5 is implemented */
    int32_t median;       /* median window, 5     (ImageOperations.py:30)  -- only 5 is implemented */
    double thresh;        /* 255*0.85             (ImageOperations.py:29) */
    double min_area;      /* 500                  (ImageOperations.py:50) */
    double min_circ;      /* 0.5                  (ImageOperations.py:50) */
} mocap_blob_params;

/* One border as seen by the contour kernel; used by the parity tests. */
typedef struct mocap_contour {
    int32_t key, is_hole, sx, sy, npts, steps;
    int64_t a00, a10, a01;
    double area, perimeter;
    int32_t kept, cx, cy, link, parent, order;
} mocap_contour;

MOCAP_API int mocap_abi_version(void);
MOCAP_API const char* mocap_last_error(void);

/* Context for one GPU and one image geometry.  n_slots = number of undistortion maps kept resident
 * (one per camera; the reference itself always uses camera 0's, lib/ImageOperations.py:37).
 * A context owns the per-batch scratch of the blob stage (bit mask, occupancy words, patches, contour workspace):
 * its mocap_blob_centroids / mocap_filter_mask / mocap_image_filter_u8 calls must be ordered on one stream.  For
 * several batches in flight use one context per stream. */
MOCAP_API int mocap_ctx_create(int device_id, int width, int height, int n_slots, mocap_ctx_t* out);
MOCAP_API int mocap_ctx_destroy(mocap_ctx_t ctx);
MOCAP_API int mocap_sync(mocap_ctx_t ctx, void* stream); /* hipStreamSynchronize */

MOCAP_API int mocap_set_blob_params(mocap_ctx_t ctx, const mocap_blob_params* p);

/* Performance switches of one context (the reference has none: its tunables are literals).  None of
 * them changes a result; they select between equivalent code paths (A/B measurements, tests of the alternative paths) or
 * size a scratch buffer.  mocap_ctx_create reads each of them ONCE from the environment (MOCAP_<NAME> in capitals; the hot
 * path never calls getenv), this call changes one for the context afterwards.  Names: skip_dark,
 * general_filter, dense_boxes, remap_pipeline, cluster, wide_quads_remap, wide_quads_identity, wide_bands, wide_fork,
 * box_prio, scan_prio, contour_prio, corr_prio, box_stage_bytes, box_timing, contour_timing, follow_timing, scan_wide, scan_hotmap, scan_serial, rows_staged, rows_stage_dw, wide_blocks_per_cu, contour_blocks_per_cu, mark_blocks_per_cu, contour_defer,
 * scan_blocks_per_cu, scan_slices, excess_base, probe_debug, contour_boxes, contours_split, corr_threads, corr_step_groups.  (rows, box_blocks_per_cu and base_sel
 * shape the context at creation: environment only.)  Must not race with a batch call on the same context. */
MOCAP_API int mocap_set_tuning(mocap_ctx_t ctx, const char* name, int value);

/* cv.undistort(img, K, dist) set-up (lib/ImageOperations.py:38): builds the quantised remap table of `slot`
 * on the device (synchronous).  identity_out (optional) receives 1 when the table is the identity. */
MOCAP_API int mocap_set_undistort(mocap_ctx_t ctx, int slot, const double K[9], const double dist[5], int* identity_out);

/* What mocap_set_undistort found out about a slot's table -- and with it, which road the slot's images take in
 * mocap_blob_centroids.  The sparse road (one streaming pass that proves most of a dark IR frame's mask zero, then the
 * filter on the marked tiles only) needs (a) a provable early-out: every 5x5 window of the undistorted image reads at most
 * 9 x 9 source pixels, and (b) the compact table: tap displacements within 11 bits.  A lens model outside either bound is
 * still filtered exactly, but every tile of every image goes through the dense row pipeline: the same results at about 7x
 * the time on a dark scene.  This call makes that visible instead of silent (sparse_path = 0). */
typedef struct mocap_undistort_info_t {
    int32_t identity;            /* the table is the identity (zero distortion) */
    int32_t compact_table;       /* displacements fit the box kernel's 4-byte table */
    int32_t early_out_provable;  /* the dark-tile bound holds for this table */
    int32_t max_source_weight;   /* largest total blend weight of one source pixel over all output pixels (1024 = one pixel; 0 = not provable) */
    int32_t sparse_path;         /* 1 = images of this slot take the sparse road (also needs the tuning switches at their defaults) */
} mocap_undistort_info_t;
MOCAP_API int mocap_undistort_info(mocap_ctx_t ctx, int slot, mocap_undistort_info_t* out);

/* camera_params + camera_poses of lib/Helpers.py (K_i, dist_i from jsons/camera-params-in.json :30-40,
 * R_i, t_i from get_extrinsics :282-291); n <= 32.  Synchronous host->device copy. */
MOCAP_API int mocap_set_cameras(mocap_ctx_t ctx, int n, const double* K /*[n][9]*/, const double* dist /*[n][5]*/,
                      const double* R /*[n][9]*/, const double* t /*[n][3]*/);
/* Fs of lib/Helpers.py:22-28: F[i-1] maps a camera-0 pixel to its epipolar line in camera i; n <= 31. */
MOCAP_API int mocap_set_fundamentals(mocap_ctx_t ctx, int n, const double* F /*[n][9]*/);

/* _find_dot over a batch (lib/ImageOperations.py:33-78 without the drawing calls).
 * frames_dev: n_images uint8 images of height x width, rows `pitch` bytes apart, images `image_stride` bytes
 * apart; image n uses undistort slot slot_base + n % cam_mod (frames laid out [time][camera]).
 * Image n writes its (cx, cy) pairs, in the reference's contour order, to out_xy_dev + n*xy_stride (int32
 * [max_blobs][2]) and its number of image points to out_count_dev[n*count_stride] (0 where the reference returns
 * [[None, None]]; values above max_blobs mean truncation; negative = MOCAP_BLOB_E_*).  The strides (in int32
 * elements) let both land in one fixed-size centroid record per image, ready for the all-gather. */
MOCAP_API int mocap_blob_centroids(mocap_ctx_t ctx, const void* frames_dev, int n_images, int cam_mod, int slot_base,
                         size_t image_stride, int pitch, int32_t* out_xy_dev, long xy_stride, int32_t* out_count_dev,
                         long count_stride, int max_blobs, void* stream);

/* The two halves of mocap_blob_centroids, exposed for tests and profiling.
 * mask_dev: [n_images][height][ceil(width/32)] uint32, bit b of word k = pixel 32k+b.  Must be zero-initialised
 * once by the caller (padding bits are never written). */
MOCAP_API int mocap_filter_mask(mocap_ctx_t ctx, const void* frames_dev, int n_images, int cam_mod, int slot_base,
                      size_t image_stride, int pitch, uint32_t* mask_dev, void* stream);
MOCAP_API int mocap_contours_from_mask(mocap_ctx_t ctx, const uint32_t* mask_dev, int n_images, int32_t* out_xy_dev,
                             long xy_stride, int32_t* out_count_dev, long count_stride, int max_blobs,
                             mocap_contour* dbg_dev /*[n_images][dbg_cap] or NULL*/, int32_t* dbg_count_dev,
                             int dbg_cap, void* stream);

/* image_filter_gpu (order 0: blur -> threshold -> median, lib/ImageOperations.py:23-31) and image_filter_cpu
 * (order 1: median -> threshold, :15-21) on one image; slot >= 0 applies that undistortion first (as _find_dot
 * does), slot < 0 filters the image as given.  dst receives the {0,255} image. */
MOCAP_API int mocap_image_filter_u8(mocap_ctx_t ctx, const void* src_dev, void* dst_dev, int spitch, int dpitch, int order,
                          int slot, void* stream);
/* cv.undistort through the table of `slot` (lib/ImageOperations.py:38) */
MOCAP_API int mocap_undistort_u8(mocap_ctx_t ctx, int slot, const void* src_dev, void* dst_dev, int spitch, int dpitch,
                       void* stream);
/* fast_cuda_blur(image, kernel_size) (lib/CudaOperations.py:24-41): uint8 in, uint8 out, any size */
MOCAP_API int mocap_box_blur_u8(mocap_ctx_t ctx, const void* src_dev, void* dst_dev, int height, int width, int spitch,
                      int dpitch, int ksize, void* stream);
/* The two pixel steps in front of _find_dot in the camera loop (RealtimeTracking_FLIR.py:103-104):
 * cv2.cvtColor(raw, cv2.COLOR_BAYER_GR2BGR) then cv2.cvtColor(., cv2.COLOR_BGR2GRAY), fused (no BGR image), for
 * n_images frames per launch.  pattern 0..3 = BG, GB, RG, GR (cv2.COLOR_BayerBG2BGR + pattern; the reference uses GR = 3);
 * gray_shift 14 = OpenCV's R2Y/G2Y/B2Y fixed point (4899, 9617, 1868, >> 14), 15 = its 15-bit set (9798, 19235, 3735).
 * Bilinear demosaic with rounded means; first/last row and column repeat their inner neighbours.  H, W >= 3.
 * Image i starts at bayer_dev + i * src_image_stride / gray_dev + i * dst_image_stride (bytes). */
MOCAP_API int mocap_bayer_gray_u8(mocap_ctx_t ctx, const void* bayer_dev, void* gray_dev, int n_images, int height, int width,
                        long spitch, long dpitch, size_t src_image_stride, size_t dst_image_stride, int pattern,
                        int gray_shift, void* stream);
/* mocap_blob_centroids on raw Bayer frames: the camera loop's cvtColor pair (RealtimeTracking_FLIR.py:103-104) followed by
 * _find_dot (:105) for a batch.  gray_frames_dev receives the gray frames (same pitch and image stride as the Bayer
 * frames); where the geometry allows (width a multiple of 16, height of 8, 16-byte aligned) the conversion is fused
 * with the early-out's streaming pass, so every frame byte is read once and the gray bytes are not read back to be
 * scanned.  Results equal mocap_bayer_gray_u8 followed by mocap_blob_centroids.  H, W >= 3. */
MOCAP_API int mocap_blob_centroids_bayer(mocap_ctx_t ctx, const void* bayer_frames_dev, void* gray_frames_dev, int n_images,
                               int cam_mod, int slot_base, size_t image_stride, int pitch, int pattern, int gray_shift,
                               int32_t* out_xy_dev, long xy_stride, int32_t* out_count_dev, long count_stride,
                               int max_blobs, void* stream);
/* fast_cuda_demosaic(bayer) (lib/CudaOperations.py:84-100): uint8[H][W] -> uint8[H][W][3] (B,G,R) */
MOCAP_API int mocap_demosaic_u8(mocap_ctx_t ctx, const void* bayer_dev, void* bgr_dev, int height, int width, int spitch,
                      void* stream);

/* find_point_correspondance_and_object_points for T time steps (lib/Helpers.py:178-280).
 * The up-to-P points of camera c at time step t start at pts_dev + t*pt_stride_t + c*pt_stride_c (strides in
 * scalars of the point type: int32, or float64 when pts_f64; a point is 2 scalars), their number (sentinel
 * already removed) is counts_dev[t*cnt_stride_t + c*cnt_stride_c].  A dense [T][C][P][2] array has strides
 * (C*P*2, P*2) and (C, 1); centroid records gathered from other GPUs are read in place through other strides.
 * Per time step and surviving camera-0 root o (in root order):
 *   root_xyz [T][P][3]  3-D point of the root's first group          (Helpers.py:272)
 *   root_err [T][P]     mean reprojection error over its groups      (Helpers.py:273)
 *   root_grp [T][P][C][2] the first group's image points             (Helpers.py:268)
 *   root_idx [T][P]     camera-0 index of the root
 *   order    [T][P]     argsort of root_err                          (Helpers.py:274)
 *   n_roots  [T]        number of surviving roots, or MOCAP_CORR_E_* (< 0): the time step has no result.  Nothing is
 *                       ever shortened silently: counts above P or below 0 fail the step (the reference has no
 *                       capacity limits, lib/Helpers.py:191,203-245)
 * The caller applies obj_count (Helpers.py:275-279).  Requires mocap_set_cameras + mocap_set_fundamentals. */
MOCAP_API int mocap_correspond(mocap_ctx_t ctx, const void* pts_dev, long pt_stride_t, long pt_stride_c,
                     const int32_t* counts_dev, long cnt_stride_t, long cnt_stride_c, int pts_f64, int T, int C,
                     int P, double cutoff, int max_groups, double* root_xyz_dev, double* root_err_dev,
                     double* root_grp_dev, int32_t* root_idx_dev, int32_t* order_dev, int32_t* n_roots_dev,
                     void* stream);

/* The scoring step of find_point_correspondance_and_object_points on its own (lib/Helpers.py:205-220), for one camera
 * pair: the epipolar line of every root point under Fs[f_index] (cv.computeCorrespondEpilines on the float32 point, :207;
 * line coefficients normalised in FP64, rounded to float32) and the distance of every candidate point of camera
 * f_index + 1 to it by the expression of :217, evaluated in FP64 -- the very device functions mocap_correspond scores with.
 * roots_dev [n_roots][2], cand_dev [n_cand][2] (int32, or float64 when pts_f64) -> dist_dev [n_roots][n_cand] float64;
 * lines_dev (optional, may be NULL) [n_roots][3] float32 = (a, b, c).  The caller applies the cutoff (< 10, :219).
 * Requires mocap_set_fundamentals. */
MOCAP_API int mocap_epipolar_scores(mocap_ctx_t ctx, const void* roots_dev, int n_roots, const void* cand_dev, int n_cand,
                                    int pts_f64, int f_index, double* dist_dev, float* lines_dev, void* stream);

/* triangulate_point(s) over N groups (lib/Helpers.py:43-99).  pts_dev [N][C][2] float64, valid_dev [N][C]
 * (0 = [None, None]).  compact_k != 0 reproduces the reference's indexing of the intrinsics by position after
 * the None entries are dropped (:59-61).  ok_dev[n] = 0 where the reference returns [None, None, None]. */
MOCAP_API int mocap_triangulate_batch(mocap_ctx_t ctx, const double* pts_dev, const uint8_t* valid_dev, int N, int C,
                            int compact_k, double* xyz_dev, int32_t* ok_dev, void* stream);
/* calculate_reprojection_error over N (group, object point) pairs (lib/Helpers.py:113-143);
 * ok_dev[n] = 0 where the reference returns None. */
MOCAP_API int mocap_reproject_batch(mocap_ctx_t ctx, const double* pts_dev, const uint8_t* valid_dev, const double* xyz_dev,
                          int N, int C, int compact_k, double* mse_dev, int32_t* ok_dev, void* stream);

/* bundle_adjustment's residual_function (lib/Helpers.py:161-167) with params_to_camera_poses (:145-156) for B parameter
 * vectors in ONE launch: camera 0 at the origin, cameras 1..C-1 from (rotation vector, t) sextuples
 * (Rotation.from_rotvec(.).as_matrix() on the device), every group triangulated from its C views (groups holding a
 * [None, None] are skipped, :93), reprojected into them, per-point MSE cast to float32 (:165); groups and object points are
 * paired positionally as the reference's zip does (:104).  The image points stay resident: pts_dev [N][C][2] float64 and
 * valid_dev [N][C] are device buffers the caller uploads once per problem; K and dist come from mocap_set_cameras.
 * params_host [B][6 (C - 1)], residuals_host [B][N] and counts_host [B] (residuals per vector, <= N) are HOST arrays: the
 * library hands them over through one pinned block the kernel reads and writes directly, so an evaluation costs one launch
 * and one stream wait.  Synchronous.  SciPy's least_squares stays the driver, as in the reference; a forward-difference
 * Jacobian is one call with B = 1 + 6 (C - 1). */
MOCAP_API int mocap_ba_residuals(mocap_ctx_t ctx, const double* params_host, int B, const double* pts_dev,
                                 const uint8_t* valid_dev, int N, int C, float* residuals_host, int32_t* counts_host,
                                 void* stream);

/* The path's one exchange step: with the cameras sharded over GPUs (one process per GPU), every rank
 * contributes the fixed-size centroid records of its images and receives all ranks' records, in rank order, before
 * correspondence -- one ncclAllGather (RCCL over xGMI) per batch.  The reference has no counterpart: its camera
 * threads hand their image points to the `track` thread through queue.Queue (RealtimeTracking_FLIR.py:107-113,
 * 180-183, 304-312); this is that hand-off across GPUs.  librccl is bound at run time (dlopen), so single-GPU users
 * do not need it.
 *   mocap_comm_unique_id  rank 0 obtains MOCAP_COMM_ID_BYTES opaque bytes (ncclGetUniqueId) and hands them to the
 *                         other ranks by any host-side means (file, socket, MPI, torch.distributed store ...);
 *   mocap_comm_available  0 when librccl can be loaded in this process (no communication: a local check every rank makes
 *                         BEFORE any rank enters the collective mocap_comm_init, so that all take the same road);
 *   mocap_comm_init       every rank, collectively: creates a communicator (ncclCommInitRank) on the context's device;
 *   mocap_comm_share      local: lets another context of the same rank and device (another batch in flight, with its own
 *                         HIP stream) use src's communicator -- ONE communicator per rank; the library orders the
 *                         all-gathers issued through it with an event chain, whatever streams they run on;
 *   mocap_allgather_centroids  asynchronous on `stream`: local_records_dev [ints_per_rank] int32 of every rank
 *                         -> gathered_dev [world][ints_per_rank] on every rank (the records mocap_blob_centroids
 *                         wrote; mocap_correspond then reads them in place through its strides);
 *   mocap_comm_destroy    drops the context's reference; the communicator goes with its last user (mocap_ctx_destroy
 *                         does this too). */
#define MOCAP_COMM_ID_BYTES 128
MOCAP_API int mocap_comm_unique_id(void* id_out /*[MOCAP_COMM_ID_BYTES]*/);
MOCAP_API int mocap_comm_available(void);
MOCAP_API int mocap_comm_init(mocap_ctx_t ctx, const void* id /*[MOCAP_COMM_ID_BYTES]*/, int rank, int world);
MOCAP_API int mocap_comm_share(mocap_ctx_t dst, mocap_ctx_t src);
MOCAP_API int mocap_comm_destroy(mocap_ctx_t ctx);
MOCAP_API int mocap_allgather_centroids(mocap_ctx_t ctx, const int32_t* local_records_dev, int32_t* gathered_dev,
                                        long ints_per_rank, void* stream);

/* Dark-tile early-out of the filter stage: one streaming kernel sums the excess max(0, p - 63) of every 8x8 cell of the
 * frames; a filter tile whose source region provably cannot produce a set mask bit (bound in DESIGN.md 4.1) is then
 * answered with zeros without reading its pixels again.  Results are identical either way; MOCAP_SKIP_DARK=0
 * disables it.  mocap_tile_stats: number of (strip, chunk) tiles of the most recent batch and how many of them were
 * resolved that way.  Synchronises the device. */
MOCAP_API int mocap_tile_stats(mocap_ctx_t ctx, uint64_t* tiles, uint64_t* skipped);
/* HIP-event timing of the kernels launched by mocap_blob_centroids / mocap_filter_mask / mocap_correspond, recorded
 * on their stream.  mocap_profile_read synchronises, returns accumulated milliseconds and launch counts and resets:
 * index 0 = box_filter_kernel (or the general filter_mask_kernel), 1 = contours_kernel, 2 = correspond_kernel,
 * 3 = bright_cells_kernel, 4 = settle_tiles_kernel. */
MOCAP_API int mocap_profile_enable(mocap_ctx_t ctx, int on);
MOCAP_API int mocap_profile_read(mocap_ctx_t ctx, double ms[5], int launches[5]);

#ifdef __cplusplus
}
#endif
#endif /* MOCAP_HIP_H */
