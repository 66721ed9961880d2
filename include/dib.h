/*
 * dib.h -- C ABI of libdib_hip.so: the MI355X (gfx950) implementation of detectInBlur's
 * data-parallel hot path (motion-blur augmentation feeding Faster R-CNN).
 *
 * The reference (mohammed-amr/detectInBlur) is pure Python; it has no FFI.  Each entry point
 * below replaces the body of one reference function and is what a ctypes binding inside the
 * reference would call.  Conventions:
 *   - plain pointers and sizes only; no torch / C++ types cross the boundary;
 *   - every `*_dev` pointer is device memory owned by the caller (e.g. a torch tensor's
 *     data_ptr()); host arrays are read during the call and may be freed on return;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); all work is
 *     enqueued on it, no entry point synchronises the host or allocates device memory;
 *   - return 0 on success, a negative DIB_E* code otherwise; dib_last_error() gives the text
 *     (thread-local).  No C++ exception crosses the boundary.
 */
#ifndef DIB_H_
#define DIB_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIB_ABI_VERSION 7 /* 7: no kernel traps any more: DIB_ETIMEOUT, dib_device_status ("Device status" below); dib_blur_step takes an optional caller workspace through dib_blur_step_ws and bounds its private buffers; 6: + dib_sparse_blur_normalized (the blur with the input transform's float + normalise + zero-padded batch as its store phase); dib_blur_step runs compaction + blur as ONE launch where the shapes allow it (same results, same signature); 5: + dib_blur_step / dib_blur_step_release (DIB_ECAPTURE, DIB_STEP_PSFS_COMPLETE), dib_normalize_resize_pad, dib_fold_bn_multi, dib_scale_rows_multi, dib_box_match / _encode_matched / _decode / _pool / _labels, dib_topk_levels, dib_det_candidates, dib_bias_act_transpose, the large LDS window; 4: + dib_bias_act_mask_nhwc, dib_relu_mask_backward, dib_add_relu_mask, dib_scatter_add_nhwc, dib_fpn_topdown_merge_nhwc, dib_stem_pool_forward / _backward, dib_post_ops, dib_jpeg_roundtrip; 2: tap-table buffers carry no scheduler trailer any more; 3: tables carry a second
                             per-tap offset array (sizes come from dib_tap_table_bytes as before)          */

/* error codes */
#define DIB_OK 0
#define DIB_EINVAL (-1)  /* bad argument (NULL pointer, unsupported K / dtype, ...)            */
#define DIB_ESHAPE (-2)  /* reference would raise: reflect padding needs H,W > 64 (or < 64)    */
#define DIB_EHIP (-3)    /* a HIP runtime call failed; text in dib_last_error()                */
#define DIB_ENOT128 (-4) /* expand_targets on a PSF that is not 128 wide (utils.py:369-370)    */
#define DIB_ECAPTURE (-5) /* dib_blur_step on a stream under graph capture without caller tables */
#define DIB_ETIMEOUT (-6) /* an EARLIER dib_blur_step on this device gave up waiting inside its launch: see "Device status" */

/* element types */
#define DIB_F16 0
#define DIB_F32 1

/* accumulation modes of the sparse blur */
#define DIB_ACC_BITEXACT 0 /* reference arithmetic: round after every multiply and every add,
                              taps in row-major order (blur_functions.py:66-67)               */
#define DIB_ACC_FP32 1     /* fp32 accumulate, one final rounding (fp16 images only)           */
#define DIB_ACC_FMA16 2    /* fp16 accumulate with a fused multiply-add: one rounding per tap
                              instead of two; half the arithmetic, not the reference's (fp16 only) */
#define DIB_ACC_FAST16 3   /* the tolerance mode built for speed: DIB_ACC_FMA16's arithmetic (one fused fp16 multiply-add per
                              pixel and tap) with the taps of every window REORDERED into vertical runs -- taps of one PSF column
                              in consecutive rows share their LDS reads -- so the result differs from DIB_ACC_FMA16's in the last
                              bits (same stated tolerance against the reference: 1e-2 absolute on images in [0, 1]).  fp16 images,
                              K = 128, standard window (DIB_EINVAL otherwise); a table without the groups (compacted without
                              DIB_COMPACT_VRUNS, or a PSF of more than 4,096 taps) is blurred in row-major order: DIB_ACC_FMA16's
                              result */
/* The LARGE LDS window of the default fp16 tiles (segments of up to 21 PSF rows x 64 columns instead of 13 x 25; 39.9 KB,
 * four workgroups per CU instead of eight): fewer window refills for PSFs that span many columns or rows -- full-exposure
 * trajectories at batch 1, where the grid leaves most workgroup slots empty anyway.  Same results, bit for bit.  A table is
 * compacted for ONE geometry: pass DIB_COMPACT_LARGE_WINDOW (or-ed into `normalize`) to dib_psf_compact* and DIB_WINDOW_LARGE
 * (or-ed into `acc_mode`) to the dib_sparse_blur calls that read those tables, DIB_STEP_LARGE_WINDOW to dib_blur_step.
 * fp16 images with DIB_ACC_BITEXACT / DIB_ACC_FMA16 only (DIB_EINVAL otherwise). */
#define DIB_COMPACT_LARGE_WINDOW 8
#define DIB_WINDOW_LARGE 0x100
/* or-ed into `normalize` of dib_psf_compact*: also write the table's vertical-run groups, which DIB_ACC_FAST16 reads (fp16 PSFs,
 * K = 128, standard window; ~1 us more per compaction launch).  Tables without them serve every other mode as before. */
#define DIB_COMPACT_VRUNS 16

int dib_abi_version(void);
const char *dib_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Device status.  No kernel of this library traps (a trap ends the process's GPU context, i.e. the training job the blur runs
 * inside: reference engine.py:101 calls blur_image_list from a DDP rank).  Two conditions can only be seen from inside a launch:
 *   - the single launch of dib_blur_step hands its tap tables from the grid's first workgroups to the blur's workgroups behind
 *     them through memory.  Forward progress rests on the hardware dispatching a grid's workgroups lowest index first -- observed
 *     on gfx950, promised by nobody.  A blur workgroup that has polled ~1 s (2^20 polls) without seeing its tables gives up;
 *   - a tap table compacted for the other LDS window geometry (DIB_COMPACT_LARGE_WINDOW without DIB_WINDOW_LARGE or vice versa).
 * In both cases the workgroup leaves its tile unwritten, stores a code into a word of pinned host memory the library owns (one
 * per device) and ends.  The host never waits for that word: every dib_blur_step / dib_sparse_blur* call LOOKS at it first, and
 * the first call after such a launch returns DIB_ETIMEOUT (hand-off) or DIB_EINVAL (geometry) with the text in
 * dib_last_error(), launches nothing and clears the word -- the caller re-issues its batch (and knows that an earlier batch on
 * the device came out incomplete).  After a DIB_ETIMEOUT the device stays on compaction + blur as two ordinary launches for the
 * rest of the process.  dib_device_status(clear) looks at the current device's word without launching anything (e.g. behind a
 * hipStreamSynchronize): DIB_OK, DIB_ETIMEOUT or DIB_EINVAL; clear != 0 also consumes it as a call above would.
 * ------------------------------------------------------------------------------------- */
int dib_device_status(int clear);

/* ---------------------------------------------------------------------------------------
 * Tap tables.  A tap table is the device-side compacted form of one K x K PSF: header,
 * CSR row pointers and the row-major list of non-zero taps {row, col, weight}.  It is produced
 * once per PSF by dib_psf_compact and consumed by the blur and by the box growth, replacing the
 * two `psf/psf.sum()` + `psf.nonzero()` passes of the reference
 * (models/blur_functions.py:63,98 and utils.py:372-374) and their host synchronisations.
 * Layout (int32 words): [0]=ntaps [1]=rmin [2]=rmax [3]=cmin [4]=cmax [5]=K [6]=sum bits
 * [7]=nsegs | rowptr[K+1] | pad to x4 | taps[K*K] as {uint32 (row<<8|col), uint32 weight bits}
 * | segments[K*K] as {first tap, end tap, r_first<<8|r_last, cmin<<8|cmax} (runs of consecutive
 * taps with a bounding box of at most 13 rows x 25 columns: the unit staged in LDS by the blur)
 * | ltaps[K*K+8] one word per tap: byte offset of its source word in the blur's LDS window (low
 * 16 bits) and the fp16 weight bits (high 16 bits) | ltaps_q[K*K+8] the same for the window layout
 * of the default 128-wide tiles (8-byte elements {P[k], P[k+32], P[k+64], P[k+96]}).  With
 * DIB_COMPACT_LARGE_WINDOW the segments are runs of at most 21 rows x 64 columns and ltaps_q holds the
 * large window's offsets; word [5] carries the geometry in bit 16.
 * | vgroups[K*K] (16-byte records, behind the offsets; with DIB_COMPACT_VRUNS, word [5] bit 17 set when valid): per segment, its
 * taps regrouped into vertical runs of at most 4 taps of one PSF column in consecutive rows -- record g of a segment whose taps are
 * [t0, t1) sits at index t0 + g: {LDS offset | taps - 1 << 16 of the NEXT group, w0 | w1 << 16, w2 | w3 << 16, taps - 1 (| own offset
 * << 2 | groups of the segment << 18 in the segment's first record)}; what the DIB_ACC_FAST16 tap loop walks.
 * ------------------------------------------------------------------------------------- */
size_t dib_tap_table_bytes(int K); /* bytes of ONE table; K is 128 or 256 */
/* bytes of the buffer dib_psf_compact fills for B PSFs: B tables, dib_tap_table_bytes(K) apart */
size_t dib_tap_tables_bytes(int K, int B);

/* psf_dev: [B][K][K] of `dtype`.  normalize != 0 divides by the PSF's sum first, in the PSF's
 * dtype, exactly like `psf_GPU / psf_GPU.sum()` (blur_functions.py:98, utils.py:372); 0 takes the
 * weights as they are (manual_blur's contract, blur_functions.py:13).  tables_dev
 * (dib_tap_tables_bytes(K, B) bytes) receives B tables, dib_tap_table_bytes(K) apart. */
int dib_psf_compact(const void *psf_dev, int dtype, int B, int K, int normalize,
                    void *tables_dev, void *stream);
/* same, for PSFs that live in B separate device buffers (host array of B device pointers, each
 * 16-byte aligned): the reference's `psfs_GPU` list needs no stacking copy */
int dib_psf_compact_list(const void *const *psf_ptrs, int dtype, int B, int K, int normalize,
                         void *tables_dev, void *stream);

/* ---------------------------------------------------------------------------------------
 * Sparse PSF (x) image correlation: models/blur_functions.py:11-69 (`manual_blur`, both canvas
 * branches, post-ops excluded) for a whole batch in one launch; with tables built by
 * dib_psf_compact(normalize=1) it is `blur_image_list` (blur_functions.py:92-100).
 * Host arrays of length B: in/out device pointers (C x H x W planar, `dtype`, contiguous;
 * out must not alias in), C, H, W and table_index (which table of tables_dev image i uses;
 * < 0 = leave the image untouched, i.e. blur_dict["blurring"] == False).  tables_dev is the buffer
 * dib_psf_compact filled for num_tables PSFs (read only: any number of dib_sparse_blur calls, on any
 * streams, may share it).  Padding mode follows
 * the reference: K = 256 -> replicate; K = 128 -> zero if H < 64 or W < 64, else reflect.
 * Returns DIB_ESHAPE where the reference raises (K = 128 and H or W == 64).
 * ------------------------------------------------------------------------------------- */
int dib_sparse_blur(const void *const *in_dev, void *const *out_dev, const int *C, const int *H,
                    const int *W, const int *table_index, int B, int dtype,
                    void *tables_dev, int num_tables, int K, int acc_mode, void *stream);

/* ---------------------------------------------------------------------------------------
 * One blur step = dib_psf_compact_list + dib_sparse_blur behind ONE call: `blur_image_list`
 * (models/blur_functions.py:92-100) for a batch.  The first seven arguments are dib_psf_compact_list's (the batch's
 * blurring PSFs, in table order), the next ten dib_sparse_blur's (table_index[i] indexes those PSFs).
 *   tables_dev == NULL (the normal case): the tap tables live in two buffers per (device, stream) owned by the library
 *     and used alternately -- the ONE exception to "no entry point allocates device memory": grown to the largest batch
 *     seen on that stream, freed by dib_blur_step_release().  By default both launches are ordinary stream-ordered launches
 *     (the PSFs may be the product of work queued on `stream` before the call, e.g. an asynchronous upload).
 *     DIB_STEP_PSFS_COMPLETE in `flags` states that the PSF buffers are already complete when the call is made (resident
 *     PSFs; the reference's own `torch.HalfTensor(psf).to(device)` of engine.py:84, a synchronous copy from pageable
 *     memory): the compaction is then launched without a barrier in front of it (hipExtAnyOrderLaunch) and overlaps the
 *     kernel queued before it on `stream` -- the previous step's blur; the blur of this step waits for it as for any earlier
 *     launch.
 *   tables_dev != NULL: dib_tap_tables_bytes(K, num_psfs) bytes owned by the caller; two ordinary launches.  Required
 *     while `stream` is being captured into a HIP graph (a replay must not touch the library's buffers): without it the
 *     call returns DIB_ECAPTURE and launches nothing.
 * Results are those of the two separate calls, bit for bit.  No host synchronisation (except when a batch outgrows the
 * stream's table buffers: that call waits for the stream once).
 * ------------------------------------------------------------------------------------- */
#define DIB_STEP_PSFS_COMPLETE 1
#define DIB_STEP_LARGE_WINDOW 2 /* compact for and blur with the large LDS window (see DIB_WINDOW_LARGE) */
int dib_blur_step(const void *const *psf_ptrs, int psf_dtype, int num_psfs, int K, int normalize,
                  const void *const *in_dev, void *const *out_dev, const int *C, const int *H, const int *W,
                  const int *table_index, int B, int dtype, int acc_mode, void *tables_dev, int flags,
                  void *stream);
/* dib_blur_step with its nine host arrays packed into two (for callers that pay per array they marshal, e.g. ctypes):
 * ptrs = psf_ptrs[num_psfs] | in_dev[B] | out_dev[B];  ints = C[B] | H[B] | W[B] | table_index[B]. */
int dib_blur_step_packed(const void *const *ptrs, const int *ints, int psf_dtype, int num_psfs, int K, int normalize, int B,
                         int dtype, int acc_mode, void *tables_dev, int flags, void *stream);
int dib_blur_step_release(void);
/* The library's own table buffers are bounded: at most 8 streams per device keep a pair; a ninth stream takes over the least
 * recently used one's (behind one device synchronisation: that stream's last step may still be running, or the stream may be
 * gone).  A process that cycles through short-lived streams therefore holds at most 16 buffers per device.
 *
 * dib_blur_step_ws: the same step on a workspace the CALLER owns -- no allocation, no library state, usable from any number of
 * streams (one workspace per stream in flight).  workspace_dev: dib_blur_step_workspace_bytes(K, num_psfs) bytes of device
 * memory, 256-byte aligned (hand-off words first, the tables behind them; size it for the largest batch).  ws_state: ONE host word
 * per workspace, in / out: 0 before the workspace's first use -- that call clears the hand-off words with one
 * hipMemsetAsync on `stream` --, updated by every call.  All launches are ordinary stream-ordered ones, so one buffer
 * suffices: step n + 1's compaction cannot start before step n's blur has read its tables.  While `stream` is being captured
 * into a graph the call makes two ordinary launches into the workspace's table area (a replay cannot advance ws_state).
 * Everything else as dib_blur_step with tables_dev == NULL; results bit-identical. */
size_t dib_blur_step_workspace_bytes(int K, int num_psfs);
int dib_blur_step_ws(const void *const *psf_ptrs, int psf_dtype, int num_psfs, int K, int normalize,
                     const void *const *in_dev, void *const *out_dev, const int *C, const int *H, const int *W,
                     const int *table_index, int B, int dtype, int acc_mode, void *workspace_dev, size_t workspace_bytes,
                     unsigned long long *ws_state, int flags, void *stream);

/* ---------------------------------------------------------------------------------------
 * Fused epilogue of the blur for the detector's input: `.float()` (engine.py:107-110), per-image
 * `(x - mean) / std` (models/net_transforms.py:112-121, :135-139) and the zero-padded batch
 * (net_transforms.py:238-247) in one pass, for batches whose images need no resize (scale factor 1).
 * in_dev: host array of B device pointers to 3 x H[i] x W[i] planar images of `dtype`; mean / std:
 * host arrays [B][3] (fp32: the values torch.as_tensor(row, dtype=float32) would hold);
 * out_dev: [B][3][Hp][Wp] fp32, planar (channels_last = 0) or stored channels-last, i.e. as
 * [B][Hp][Wp][3] (channels_last != 0); Hp >= H[i], Wp >= W[i]; pixels outside an image become 0.
 * Same arithmetic as the reference (fp32 subtract, IEEE divide): bit-identical results.
 * ------------------------------------------------------------------------------------- */
int dib_normalize_pad(const void *const *in_dev, int dtype, const int *H, const int *W, int B,
                      const float *mean, const float *std, float *out_dev, int Hp, int Wp,
                      int channels_last, void *stream);

/* The same epilogue for batches whose images need the detector's internal resize (every native-size COCO batch: the reference
 * blurs before the model resizes, engine.py:101 -> models/net_transforms.py:151-175): float conversion, per-image normalisation,
 * bilinear resize to Ho[i] x Wo[i] -- torch.nn.functional.interpolate(mode="bilinear", recompute_scale_factor=True,
 * align_corners=False) as ATen computes it on the GPU, the scale recomputed from the integer sizes -- and the zero-padded batch,
 * one launch.  Ho / Wo: host arrays, the output size of image i (the caller's int(H * scale), net_transforms.py:36-46,167);
 * an image with Ho == H and Wo == W is not interpolated (the reference skips the call at scale factor 1).  Everything else as
 * dib_normalize_pad.  Bit-identical to the unfused GPU path; within 2e-6 of the reference's CPU result. */
int dib_normalize_resize_pad(const void *const *in_dev, int dtype, const int *H, const int *W, const int *Ho, const int *Wo, int B,
                             const float *mean, const float *std, float *out_dev, int Hp, int Wp, int channels_last,
                             void *stream);

/* The blur WITH that epilogue as its store phase: dib_sparse_blur + dib_normalize_pad in one launch, for batches whose images
 * need no resize (the model's internal scale factor is 1: BASELINE's synthetic 800 x 1333; the reference blurs, engine.py:101,
 * converts to float, :107-110, and normalises + pads inside the model, net_transforms.py:112-121, :238-247 -- the blurred fp16
 * image never has to exist).  in_dev: B device pointers to 3 x H[i] x W[i] fp16 images, table_index[i] >= 0 the table of
 * tables_dev image i uses; slot (may be NULL = identity): the batch position image i goes to (the caller may hand the images over
 * heaviest PSF first); mean / std: host [B][3] in the order of in_dev; out_dev: [B][3][Hp][Wp] fp32 planar or channels-last as
 * dib_normalize_pad, padding pixels are written as 0 by the launch.  acc_mode: DIB_ACC_BITEXACT or DIB_ACC_FMA16.
 * Returns DIB_OK, a negative error, or 1 = "not served, nothing launched" (an image that is not blurred, more than 32 images, a
 * padded extent the image's own tiles do not cover: Hp > ceil(H / 32) * 32 or Wp > ceil(W / 128) * 128): the caller then takes
 * the two-launch path.  Bit-identical to dib_sparse_blur followed by dib_normalize_pad. */
int dib_sparse_blur_normalized(const void *const *in_dev, const int *H, const int *W, const int *table_index, const int *slot, int B,
                               void *tables_dev, int num_tables, int K, int acc_mode, const float *mean, const float *std,
                               float *out_dev, int Hp, int Wp, int channels_last, void *stream);

/* ---------------------------------------------------------------------------------------
 * Box growth and clamping: utils.py:360-392 (`expand_targets`, one image) and utils.py:395-434
 * (`fix_bounding_box_squeeze`).  boxes_dev: [N][4] float32 xyxy, updated in place.
 * ------------------------------------------------------------------------------------- */
int dib_expand_boxes(float *boxes_dev, int N, const void *table_dev, int H, int W, void *stream);
int dib_clamp_boxes(float *boxes_dev, int N, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------
 * PSF rasteriser: motion_blur/generate_PSF.py:31-83 (`PSF.fit`, single exposure fraction) and
 * :106-123 (`centerPSF`), then the centre crop of transforms.py:334-335, batched.
 * traj_dev: [B][iters] complex128 (re, im interleaved); fraction: host array [B].
 * canvas x canvas float64 accumulation in the reference's order (sample index ascending per
 * cell), so psf64_dev is bit-identical to the reference's float64 PSF.
 *   center != 0: roll the weighted centroid to the canvas centre;
 *   out_n: canvas, or 128 with canvas 256 for the [64:192] crop.
 * psf64_dev: [B][out_n][out_n] float64 (may be NULL); psf16_dev: same shape, float16 converted
 * as torch.HalfTensor(ndarray) does (float64 -> float32 -> float16; engine.py:84) (may be NULL).
 * workspace_dev: dib_psf_rasterize_workspace_bytes(B, iters, canvas) bytes.
 * ------------------------------------------------------------------------------------- */
size_t dib_psf_rasterize_workspace_bytes(int B, int iters, int canvas);
int dib_psf_rasterize(const double *traj_dev, int B, int iters, const double *fraction, int canvas,
                      int center, int out_n, double *psf64_dev, void *psf16_dev,
                      void *workspace_dev, void *stream);

/* ---------------------------------------------------------------------------------------
 * Detector ops the blurred batches feed (torchvision is not a dependency): RoIAlign as used by
 * torchvision.ops.MultiScaleRoIAlign(output_size=7, sampling_ratio=2) (reference
 * models/faster_rcnn.py:204-208) and NMS as used by RegionProposalNetwork / RoIHeads
 * (reference models/faster_rcnn.py:198-229).  fp32, NCHW.
 *   rois_dev: [K][5] = (batch index, x1, y1, x2, y2) in image coordinates.
 *   dib_roi_align_backward ACCUMULATES into grad_feat_dev (caller zero-fills it first).
 *   dib_nms: boxes [n][4] xyxy already sorted by descending score; keep_dev receives the kept
 *   indices (ascending = score order; entries past the count are 0), count_dev their number;
 *   workspace of dib_nms_workspace_bytes(n).
 * ------------------------------------------------------------------------------------- */
int dib_roi_align_forward(const float *feat_dev, const float *rois_dev, int K, int C, int H, int W,
                          float spatial_scale, int pooled, int sampling_ratio, int aligned,
                          float *out_dev, void *stream);
int dib_roi_align_backward(const float *grad_out_dev, const float *rois_dev, int K, int C, int H,
                           int W, float spatial_scale, int pooled, int sampling_ratio, int aligned,
                           float *grad_feat_dev, void *stream);
 /* Channels-last (NHWC) form over 1..4 pyramid levels in ONE launch: MultiScaleRoIAlign without the
 * per-level index_select / scatter (and their host syncs).  feat_dev / grad_feat_dev: host arrays of
 * n_levels device pointers to [N][H_l][W_l][C] fp32; H, W, scale: host arrays [n_levels];
 * level_dev: device int32 [K], the level of each RoI (may be NULL when n_levels == 1);
 * out / grad_out: [K][C][pooled][pooled] contiguous, as above.  pooled <= 7. */
int dib_roi_align_nhwc_forward(const float *const *feat_dev, const int *H, const int *W, const float *scale,
                               int n_levels, const float *rois_dev, const int *level_dev, int K, int C, int pooled,
                               int sampling_ratio, int aligned, float *out_dev, void *stream);
int dib_roi_align_nhwc_backward(const float *grad_out_dev, const int *H, const int *W, const float *scale,
                                int n_levels, const float *rois_dev, const int *level_dev, int K, int C, int pooled,
                                int sampling_ratio, int aligned, float *const *grad_feat_dev, void *stream);
size_t dib_nms_workspace_bytes(int n);
int dib_nms(const float *boxes_sorted_dev, int n, float iou_threshold, void *workspace_dev,
            long long *keep_dev, int *count_dev, void *stream);
/* B independent box sets of n boxes each in one launch pair (RPN: one set per image).  boxes [B][n][4],
 * keep [B][n] (entries past count[b] are 0), count [B]; valid_dev (may be NULL): [B][n] bytes, 0 = the
 * box takes no part (neither kept nor suppressing), which replaces a compaction + host sync in front
 * of the call; workspace: B * dib_nms_workspace_bytes(n). */
int dib_nms_batched(const float *boxes_sorted_dev, const unsigned char *valid_dev, int B, int n,
                    float iou_threshold, void *workspace_dev, long long *keep_dev, int *count_dev,
                    void *stream);

/* ---------------------------------------------------------------------------------------
 * Box bookkeeping of the training step (torchvision's box_iou + Matcher + BoxCoder as the reference's RPN / RoIHeads use
 * them, models/faster_rcnn.py:150-159, 198-229) for a whole batch per launch.  Ground truth is ragged: gt_cat_dev holds every
 * image's boxes back to back ([T][4] xyxy), gt_offset (HOST, N + 1 ints) where each image's boxes start; at most 32 images, at
 * most 256 boxes per image.  Candidates (anchors / proposals): cand_dev [N][M][4], or [M][4] shared by every image
 * (cand_shared != 0).  Same operations in the same order as the tensor expressions (IoU thresholds fall the same way).
 *   dib_box_match: match_dev [N][M] int64 = index (inside the image's list) of the ground truth of highest IoU, lowest index
 *     on ties; -1 where that IoU < low, -2 where low <= IoU < high; with allow_low_quality every candidate that realises some
 *     ground truth's best IoU keeps its index (best_dev: workspace of T unsigned).  Images without ground truth: -1.
 *   dib_box_encode_matched: targets_dev [N][M][4] = BoxCoder(weights).encode(gt[max(match, 0)], candidate) and / or
 *     matched_dev [N][M][4] = that ground-truth box (a zero box for images without ground truth); either may be NULL.
 *   dib_box_decode: out [R][4] = BoxCoder.decode(deltas [R][4], anchors [A][4]), row r against anchor r % A,
 *     dw / dh clamped at `clip` (log(1000 / 16)).
 * ------------------------------------------------------------------------------------- */
int dib_box_match(const float *gt_cat_dev, const int *gt_offset, int N, const float *cand_dev, int M, int cand_shared, float high,
                  float low, int allow_low_quality, unsigned *best_dev, long long *match_dev, void *stream);
int dib_box_encode_matched(const float *gt_cat_dev, const int *gt_offset, int N, const long long *match_dev, const float *cand_dev, int M,
                           int cand_shared, float wx, float wy, float ww, float wh, float *targets_dev, float *matched_dev, void *stream);
int dib_box_decode(const float *deltas_dev, const float *anchors_dev, long long R, int A, float wx, float wy, float ww, float wh,
                   float clip, float *out_dev, void *stream);
/* act(in + bias[c]) with the layout change between channels-last and planar in the same pass (inference: around the 3x3
 * convolutions that run through MIOpen's planar kernels).  to_planar = 1: in [N][HW][C] -> out [N][C][HW]; 0: in [N][C][HW] ->
 * out [N][HW][C].  Values identical to dib_bias_act_nhwc followed by a copy.  in_dev and out_dev must not alias. */
int dib_bias_act_transpose(const float *in_dev, const float *bias_dev, float *out_dev, int N, int C, long long HW, int to_planar, int relu,
                           void *stream);
/* Sorted top-k of every (image, level) row of scores in one launch (torchvision filter_proposals' per-level torch.topk + gather +
 * clip_boxes_to_image + the small-box test, reference models/faster_rcnn.py:198-207 sets the counts).  values_dev: [N][row_stride];
 * level l covers elements level_offset[l] .. level_offset[l + 1] of a row and yields its level_k[l] (<= K <= 2048) highest scores in
 * descending order -- equal scores in ascending index order, NaN first -- into out_scores_dev [N][L][K] (rows shorter than K end in
 * -inf), their level-relative indices into out_index_dev (optional).  With boxes_dev ([N][row_stride][4], same indexing): the
 * winners' boxes into out_boxes_dev [N][L][K][4], clipped to clip_wh_dev[n] = (width, height) when that is given, and
 * out_valid_dev [N][L][K] = score > -inf and both clipped sides >= min_size.  L <= 32.  (A long row is bound by the one compute unit
 * that sweeps it: callers split it into consecutive levels and merge the winners with a second call -- the order is the same.) */
int dib_topk_levels(const float *values_dev, long long row_stride, int N, const int *level_offset, const int *level_k, int L, int K,
                    const float *boxes_dev, const float *clip_wh_dev, float min_size, float *out_scores_dev, long long *out_index_dev,
                    float *out_boxes_dev, unsigned char *out_valid_dev, void *stream);
/* Detections of one image up to the NMS (torchvision RoIHeads.postprocess_detections; reference models/faster_rcnn.py:213-229): softmax
 * over the C <= 128 class logits of every RoI (ATen's operation order), per-class box decoding with BoxCoder weights (wx, wy, ww, wh)
 * and the log(1000 / 16) clip, clipping to the image, the score > score_thresh and side >= min_size tests.  logits_dev [R][C],
 * deltas_dev [R][4 C], rois_dev [R][4].  Class-major outputs without the background class: scores_cm_dev [C - 1][R] (-inf where the
 * candidate is dropped), boxes_cm_dev [C - 1][R][4]; stats_dev[0] = number of candidates kept, stats_dev[1] = bit pattern of the
 * largest coordinate among them (what batched_nms moves the classes apart by). */
int dib_det_candidates(const float *logits_dev, const float *deltas_dev, const float *rois_dev, int R, int C, float img_h, float img_w, float wx,
                       float wy, float ww, float wh, float clip, float score_thresh, float min_size, float *scores_cm_dev, float *boxes_cm_dev,
                       unsigned *stats_dev, void *stream);
/* RoI-head candidate pool: cands[n] = proposals[n] (P rows) ++ the ground truth of image n ++ [0, 0, 1, 1] rows up to P + Gpad
 * (torchvision RoIHeads.add_gt_proposals with a fixed shape).  cands_dev: [N][P + Gpad][4]. */
int dib_box_pool(const float *proposals_dev, int P, const float *gt_cat_dev, const int *gt_offset, int N, int Gpad, float *cands_dev, void *stream);
/* Class per pool row (RoIHeads.assign_targets_to_proposals): gt_labels[match] for match >= 0, 0 for -1, -1 for -2 and for padding
 * rows (proposal rows whose ok byte is 0 -- ok_dev may be null: all live --, ground-truth rows beyond the image's count).
 * match_dev / labels_dev: [N][M] int64, M >= P; gt_labels_cat_dev: int64, concatenated like the boxes. */
int dib_box_labels(const long long *match_dev, const long long *gt_labels_cat_dev, const int *gt_offset, int N, const unsigned char *ok_dev, int P,
                   int M, long long *labels_dev, void *stream);

/* ---------------------------------------------------------------------------------------
 * COCO box IoU: pycocotools' bbIou (reference cocoapi/common/maskApi.c:109-120), the inner loop of the
 * evaluation sweep's COCOeval.  dt_dev [m][4], gt_dev [n][4]: float64 (x, y, w, h); iscrowd_dev [n]
 * bytes or NULL; out_dev [n][m] float64 = intersection / union (union = detection area for crowd
 * ground truth), bit-identical to the C routine.
 * ------------------------------------------------------------------------------------- */
int dib_coco_box_iou(const double *dt_dev, const double *gt_dev, const unsigned char *iscrowd_dev, int m, int n,
                     double *out_dev, void *stream);

/* ---------------------------------------------------------------------------------------
 * Convolution epilogue of the ResNet-50 trunk with its frozen batch-norm folded into the weights
 * (reference models/faster_rcnn.py:367 builds the trunk with torchvision's FrozenBatchNorm2d):
 *   x = act(x + bias[c] (+ residual)), in place, channels-last fp32 (channel index fastest);
 * one pass instead of eager PyTorch's 2-4.  residual_dev may be NULL; relu != 0 applies ReLU.
 * ------------------------------------------------------------------------------------- */
int dib_bias_act_nhwc(float *x_dev, const float *bias_dev, const float *residual_dev, long long n_elems,
                      int C, int relu, void *stream);

/* The ReLU form of the above that also leaves what its backward pass needs of the result: mask_dev[n_elems / 4],
 * one byte per 4 consecutive elements, bit k = element 4 i + k is positive (C % 4 == 0, 16-byte aligned tensors).
 * dib_relu_mask_backward: grad_out = mask ? grad_in : 0 -- torch's threshold_backward(grad, y, 0) (the ReLU backward
 * behind every trunk convolution, reference models/faster_rcnn.py:367 / torchvision Bottleneck) at 8.25 instead of
 * 12 bytes per element; grad_out may alias grad_in. */
int dib_bias_act_mask_nhwc(float *x_dev, const float *bias_dev, const float *residual_dev, long long n_elems,
                           int C, unsigned char *mask_dev, void *stream);
int dib_relu_mask_backward(const float *grad_in_dev, const unsigned char *mask_dev, float *grad_out_dev,
                           long long n_elems, void *stream);
/* Gradient accumulation at a residual block's input (torchvision Bottleneck: the block input feeds conv1 AND the skip
 * connection, autograd adds the two gradients) fused with the ReLU backward of that input: a = (a + b), zeroed where the
 * mask bit is clear; mask_dev NULL = plain in-place accumulate. */
int dib_add_relu_mask(float *a_dev, const float *b_dev, const unsigned char *mask_dev, long long n_elems, void *stream);
/* The same accumulation for the first block of a ResNet stage, whose skip is a strided 1x1 convolution: the data gradient
 * of that convolution, computed densely on the strided pixels, is added in place into the data gradient of conv1:
 * a[n, ys * stride, xs * stride, :] += b[n, ys, xs, :] (channels-last fp32, C % 4 == 0) -- instead of a zero-filled
 * full-size gradient and a full-size add. */
int dib_scatter_add_nhwc(float *a_dev, const float *b_dev, int N, int H, int W, int Hs, int Ws, int C, int stride, void *stream);
/* FPN top-down merge (torchvision FeaturePyramidNetwork.forward: inner = lateral + interpolate(top, size, "nearest")) in one
 * in-place pass on the lateral convolution's output: x[n, h, w, :] += bias[:] + top[n, sh, sw, :], sh = min(int(floorf(h *
 * float(Ht) / H)), Ht - 1) (ATen's nearest index) -- instead of a bias add, an upsample that writes a full-size tensor and an
 * add that reads it back. */
int dib_fpn_topdown_merge_nhwc(float *x_dev, const float *bias_dev, const float *top_dev, int N, int H, int W, int Ht, int Wt, int C,
                               void *stream);
/* ResNet stem (torchvision resnet50: conv1 -> bn1 -> relu -> maxpool(3, stride 2, padding 1)) behind the folded convolution:
 * out[N, Ho, Wo, C] = max_pool2d(relu(x + bias)), Ho = (H - 1) / 2 + 1, in one pass; arg_dev: one unsigned short per 4 output
 * channels (4-bit window position of the winner, 15 = no gradient).  The backward pass turns grad_out + arg into the dense
 * gradient of x (max-pool backward and ReLU backward in one pass).  Channels-last fp32, C % 4 == 0, 16-byte aligned. */
int dib_stem_pool_forward(const float *x_dev, const float *bias_dev, float *out_dev, unsigned short *arg_dev, int N, int H, int W, int C,
                          void *stream);
int dib_stem_pool_backward(const float *grad_out_dev, const unsigned short *arg_dev, float *grad_in_dev, int N, int H, int W, int C,
                           void *stream);

/* Frozen batch-norm folds of n trunk convolutions in one launch per 32 (torchvision's FrozenBatchNorm2d behind every ResNet
 * convolution, reference models/faster_rcnn.py:367; training re-folds every step because the weights move):
 *   scale = bn_w * rsqrt(var + eps);  shift = bn_b - mean * scale;  wf[co][...] = w[co][...] * scale[co]
 * -- the same operations in the same order as the tensor expressions, so bit-identical to them.  Host arrays of n device
 * pointers / sizes; w[k] is Co[k] x inner[k] dense with the output channel outermost (contiguous or channels-last weights);
 * wf[k] has w[k]'s element order; scale[k], shift[k]: Co[k] floats.  dib_scale_rows_multi is the backward of the weight
 * product: dw[k][co][...] = g[k][co][...] * scale[k][co]. */
int dib_fold_bn_multi(const float *const *w, const float *const *bn_w, const float *const *bn_b, const float *const *mean,
                      const float *const *var, const int *Co, const int *inner, int n, float eps, float *const *wf,
                      float *const *scale, float *const *shift, void *stream);
int dib_scale_rows_multi(const float *const *g, const float *const *scale, const int *Co, const int *inner, int n,
                         float *const *dw, void *stream);

/* ---------------------------------------------------------------------------------------
 * Post-blur corruption chain of manual_blur (reference models/blur_functions.py:72-81) in one pass:
 *   noise_var > 0:    out = clamp(in + N(0, noise_var), 0, 1)                    (:72-74, --add_noise)
 *   block_scale > 0:  out = nearest-up(nearest-down(., scale_factor = block_scale), size = original)   (:76-81, --add_block)
 * applied in that order (the noise of a source pixel travels with it into every block copy).  The block path equals
 * torch's two interpolate calls bit for bit; the noise field comes from a counter-based generator keyed by `seed`
 * (same distribution and rounding steps as torch's expression, not torch's sample values).  C x H x W planes of
 * DIB_F16 / DIB_F32; out must not alias in.  The host draws (variance, coin flip, scale factor) stay with the caller, in
 * the reference's order on numpy's global stream.
 * ------------------------------------------------------------------------------------- */
int dib_post_ops(const void *in_dev, void *out_dev, int C, int H, int W, int dtype, double noise_var,
                 unsigned long long seed, double block_scale, void *stream);

/* JPEG round trip of one image in one launch (reference transforms.py:467-493 around models/jpeg/DiffJPEG.py:
 * reflect-pad to a multiple of 16, RGB -> YCbCr 4:2:0 -> 8x8 DCT -> quantise with table x factor -> back -> crop).
 * in_dev: 3 x H x W planes of DIB_F16 / DIB_F32 in [0, 1]; out_dev: 3 x H x W planes of fp16 (the reference returns
 * `.half()`); q_luma / q_chroma: HOST pointers to the 8 x 8 tables already multiplied by the quality factor, row-major
 * [u][v].  fp32 arithmetic as the reference's; the DCT sums run in this kernel's order: results within one
 * quantisation step of the reference's (tests/test_jpeg.py), as torch's own GPU path is. */
int dib_jpeg_roundtrip(const void *in_dev, void *out_dev, int H, int W, int dtype, const float *q_luma,
                       const float *q_chroma, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DIB_H_ */
