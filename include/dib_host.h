/*
 * dib_host.h -- C ABI of libdib_host.so: host-side (CPU) native code of the blur hot path.
 *
 * dib_trajectory_fit replaces the body of `Trajectory.fit` (reference
 * motion_blur/generate_trajectory.py:38-98), a 1999-iteration pure-Python loop that the
 * reference runs twice per image inside its DataLoader workers (transforms.py:316-317,
 * ~30 ms per fit).  It consumes numpy's LEGACY global random stream (MT19937 + polar
 * Box-Muller with a cached second deviate) draw for draw, so results are bit-identical to the
 * reference under the same `np.random.seed`: the caller passes `np.random.get_state()` in and
 * writes the updated state back with `np.random.set_state()`.
 */
#ifndef DIB_HOST_H_
#define DIB_HOST_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* numpy legacy RandomState, updated in place */
typedef struct dib_mt19937 {
  uint32_t key[624];
  int32_t pos;
  int32_t has_gauss;
  double gauss;
} dib_mt19937;

/* One `Trajectory(canvas, iters, max_len, expl).fit()`.
 * x_out: [iters][2] (re, im) = Trajectory.x;  unprocessed_out: same shape or NULL
 * (= Trajectory.unprocessedX);  stats_out[0] = tot_length, stats_out[1] = big_expl_count.
 * Returns 0, or -1 on bad arguments. */
int dib_trajectory_fit(dib_mt19937 *rng, int canvas, int iters, double max_len, double expl,
                       double *x_out, double *unprocessed_out, double *stats_out);

/* `PSF(canvas, trajectory, fractions).fit()` (reference motion_blur/generate_PSF.py:31-83):
 * traj: [iters][2]; fractions: [nfrac] (cumulative exposure windows, the reference's shared
 * accumulator semantics); psfs_out: [nfrac][canvas][canvas] float64.  Returns 0, -1 on bad
 * arguments, -2 if a splat falls outside the canvas (the reference raises IndexError). */
int dib_psf_fit(const double *traj, int iters, const double *fractions, int nfrac, int canvas,
                double *psfs_out);

/* `PSF.centerPSF()` (generate_PSF.py:106-123) in place on one canvas x canvas float64 PSF;
 * offsets_out[0..1] = (offsetX, offsetY).  np.sum's summation order is reproduced. */
int dib_psf_center(double *psf, int canvas, int *offsets_out);

/* draws for tests / the expl=None constructor path: np.random.uniform(0,1), np.random.randn() */
double dib_rng_uniform(dib_mt19937 *rng);
double dib_rng_gauss(dib_mt19937 *rng);

/* COCO matching of one (image, category): pycocotools COCOeval.evaluateImg (reference cocoapi/PythonAPI/pycocotools/cocoeval.py:
 * 235-310) for A area ranges x T IoU thresholds in one call.  ious: [D][G] float64, detections (already sorted by descending score
 * and cut to maxDets) x ground truth of the category; dt_area [D]; crowd [G] (nonzero = iscrowd / ignore), gt_area [G];
 * area_rng [A][2]; iou_thrs [T].  Outputs: dtm_out / dt_ig_out [A][T][D] bytes (detection matched / ignored), n_gt_out [A] =
 * ground truth counted in the range.  Returns 0, -1 on bad arguments, -3 out of memory. */
int dib_coco_match(const double *ious, int D, int G, const double *dt_area, const long long *crowd, const double *gt_area,
                   const double *area_rng, int A, const double *iou_thrs, int T, unsigned char *dtm_out,
                   unsigned char *dt_ig_out, int *n_gt_out);

/* One image, every category (COCOeval.evaluate's loop over catIds around evaluateImg, reference cocoeval.py:160-184): iou_all
 * [D][G] = every detection x every ground truth of the image.  Category cats[k]'s detections in stable descending-score order, at
 * most max_det of them, are det_order[cat_start[k] .. cat_start[k + 1]) (indices into the D detections); dtm_out / dt_ig_out
 * [A][T][D] hold their results at those positions; n_gt_out [K][A]; gt_count_out [K] = the category's ground truth in the image. */
int dib_coco_match_image(const double *iou_all, int D, int G, const long long *dt_label, const double *dt_score, const double *dt_area,
                         const long long *gt_label, const long long *gt_crowd, const double *gt_area, const long long *cats, int K,
                         int max_det, const double *area_rng, int A, const double *iou_thrs, int T, int *det_order, int *cat_start,
                         unsigned char *dtm_out, unsigned char *dt_ig_out, int *n_gt_out, int *gt_count_out);

/* COCOeval.accumulate for the records of one category (reference cocoeval.py:315-420): scores [n] = the images' detections laid
 * end to end, lens [nrec] detections per image, dtm / dt_ig [A][T][n] bytes, n_gt [A]; max_dets [M]; rec_thrs [R] ascending.
 * precision_out [A][M][T][R] and recall_out [A][M][T] are written where n_gt[a] > 0 (pre-fill with -1). */
int dib_coco_accumulate_cat(const double *scores, const int *lens, int nrec, const unsigned char *dtm, const unsigned char *dt_ig,
                            const int *n_gt, int A, int T, const int *max_dets, int M, const double *rec_thrs, int R,
                            double *precision_out, double *recall_out);

/* Segmentation masks of ConvertCocoPolysToMask (reference coco_utils.py:34-49 = pycocotools frPyObjects + decode + any over an
 * object's polygons; rasterisation as cocoapi/common/maskApi.c:162-218 rleFrPoly does it, decode as :43-47).  xy: k vertices
 * (x0, y0, x1, y1, ...) in pixel units; the polygon's pixels are OR-ed into mask [h][w] (row-major bytes, 0 / 1).
 * dib_mask_or_runs: an uncompressed RLE's run lengths (column-major, first run zeros).  Return 0, -1 bad arguments, -3 memory. */
int dib_mask_or_polygon(const double *xy, long k, long h, long w, unsigned char *mask);
int dib_mask_or_runs(const unsigned *runs, long n_runs, long h, long w, unsigned char *mask);

#ifdef __cplusplus
}
#endif
#endif
