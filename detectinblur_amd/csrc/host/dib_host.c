/* Host-side native code (see include/dib_host.h).  Plain C99 + libm; built with
 * -ffp-contract=off so every operation below rounds exactly once, as numpy's scalar math does. */
#include <complex.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

#include "../../../include/dib_host.h"

/* ---- MT19937 (Matsumoto & Nishimura) in numpy's legacy layout ---------------------------- */
static void mt_refill(dib_mt19937 *s) {
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAT = 0x9908b0dfu;
  uint32_t *k = s->key, y;
  int i;
  for (i = 0; i < 624 - 397; i++) {
    y = (k[i] & UPPER) | (k[i + 1] & LOWER);
    k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
  }
  for (; i < 623; i++) {
    y = (k[i] & UPPER) | (k[i + 1] & LOWER);
    k[i] = k[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
  }
  y = (k[623] & UPPER) | (k[0] & LOWER);
  k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
  s->pos = 0;
}

static inline uint32_t mt_next32(dib_mt19937 *s) {
  uint32_t y;
  if (s->pos >= 624) mt_refill(s);
  y = s->key[s->pos++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

/* 53-bit double in [0,1): numpy's mt19937_next_double */
double dib_rng_uniform(dib_mt19937 *s) {
  uint32_t a = mt_next32(s) >> 5, b = mt_next32(s) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* numpy's legacy_gauss: polar method, second deviate cached */
double dib_rng_gauss(dib_mt19937 *s) {
  if (s->has_gauss) {
    double t = s->gauss;
    s->has_gauss = 0;
    s->gauss = 0.0;
    return t;
  }
  double f, x1, x2, r2;
  do {
    x1 = 2.0 * dib_rng_uniform(s) - 1.0;
    x2 = 2.0 * dib_rng_uniform(s) - 1.0;
    r2 = x1 * x1 + x2 * x2;
  } while (r2 >= 1.0 || r2 == 0.0);
  f = sqrt(-2.0 * log(r2) / r2);
  s->gauss = f * x1;
  s->has_gauss = 1;
  return f * x2;
}

/* np.abs(complex128) as numpy >= 1.25 evaluates it: max * sqrt(fma(q, q, 1)), q = min / max */
static inline double cabs_np(double re, double im) {
  double a = fabs(re), b = fabs(im), mx = a >= b ? a : b, mn = a >= b ? b : a;
  if (mx == 0.0) return 0.0;
  double q = mn / mx;
  return mx * sqrt(fma(q, q, 1.0));
}

int dib_trajectory_fit(dib_mt19937 *rng, int canvas, int iters, double max_len, double expl, double *x_out,
                       double *unprocessed_out, double *stats_out) {
  if (!rng || !x_out || iters < 2) return -1;
  /* generate_trajectory.py:48-56 */
  const double centripetal = 0.7 * dib_rng_uniform(rng);
  const double prob_big_shake = 0.2 * dib_rng_uniform(rng);
  const double gaussian_shake = 10 * dib_rng_uniform(rng);
  const double init_angle = 360 * dib_rng_uniform(rng);
  const double rad = init_angle * (M_PI / 180.0);
  const double v0_im = sin(rad), v0_re = cos(rad);
  const double step = max_len / (double)(iters - 1);
  double v_re = (v0_re * max_len) / (double)(iters - 1), v_im = (v0_im * max_len) / (double)(iters - 1); /* :59 */
  if (expl > 0) { v_re = v0_re * expl; v_im = v0_im * expl; }                                          /* :61-62 */
  const double thr = prob_big_shake * expl;
  double xr = 0.0, xi = 0.0, tot = 0.0;
  long big = 0;
  x_out[0] = 0.0; x_out[1] = 0.0;
  for (int t = 0; t < iters - 1; ++t) {
    double nd_re = 0.0, nd_im = 0.0;
    if (dib_rng_uniform(rng) < thr) {                                                                /* :69 */
      double complex e = cexp((M_PI + (dib_rng_uniform(rng) - 0.5)) * I);                             /* :70 */
      double a_re = 2 * v_re, a_im = 2 * v_im, e_re = creal(e), e_im = cimag(e);
      nd_re = a_re * e_re - a_im * e_im;
      nd_im = a_re * e_im + a_im * e_re;
      big++;
    }
    const double g_re = dib_rng_gauss(rng), g_im = dib_rng_gauss(rng);                                /* :76 */
    const double in_re = gaussian_shake * g_re - centripetal * xr;
    const double in_im = gaussian_shake * g_im - centripetal * xi;
    const double dv_re = nd_re + (expl * in_re) * step, dv_im = nd_im + (expl * in_im) * step;        /* :75-77 */
    v_re = v_re + dv_re;                                                                               /* :79 */
    v_im = v_im + dv_im;
    const double scl = 1.0 / cabs_np(v_re, v_im); /* numpy complex / real = times the reciprocal (:80) */
    v_re = (v_re * scl) * step;
    v_im = (v_im * scl) * step;
    const double nr = xr + v_re, ni = xi + v_im;                                                       /* :81 */
    tot = tot + hypot(nr - xr, ni - xi);                                                               /* :82 */
    xr = nr; xi = ni;
    x_out[2 * (t + 1)] = xr;
    x_out[2 * (t + 1) + 1] = xi;
  }
  const double shift = canvas / 2.0;                                                                   /* :92 */
  for (int t = 0; t < iters; ++t) {
    if (unprocessed_out) { unprocessed_out[2 * t] = x_out[2 * t]; unprocessed_out[2 * t + 1] = x_out[2 * t + 1]; }
    x_out[2 * t] = x_out[2 * t] + shift;
    x_out[2 * t + 1] = x_out[2 * t + 1] + shift;
  }
  if (stats_out) { stats_out[0] = tot; stats_out[1] = (double)big; }
  return 0;
}

/* ---- PSF.fit on the host (generate_PSF.py:31-83) ------------------------------------------- */
static double sample_weight(int t, double frac, double prev, int iters) {
  const double fn = frac * iters, pn = prev * iters; /* :47-56 */
  if (fn >= t && pn < t - 1) return 1.0;
  if (fn >= t - 1 && pn < t - 1) return fn - (t - 1);
  if (fn >= t && pn < t) return t - pn;
  if (fn >= t - 1 && pn < t) return (frac - prev) * iters;
  return 0.0;
}

static inline double tri(double u) { double v = 1.0 - fabs(u); return v > 0.0 ? v : 0.0; }

int dib_psf_fit(const double *traj, int iters, const double *fractions, int nfrac, int canvas, double *psfs_out) {
  if (!traj || !fractions || !psfs_out || iters <= 0 || nfrac <= 0 || canvas < 3) return -1;
  const size_t n = (size_t)canvas * canvas;
  double *acc = psfs_out + (size_t)(nfrac - 1) * n; /* the shared accumulator lives in the last slot */
  for (size_t i = 0; i < n; ++i) acc[i] = 0.0;
  for (int j = 0; j < nfrac; ++j) {
    const double prev = j == 0 ? 0.0 : fractions[j - 1];
    for (int t = 0; t < iters; ++t) {
      const double w = sample_weight(t, fractions[j], prev, iters);
      const double re = traj[2 * t], im = traj[2 * t + 1];
      double fr = floor(re), fi = floor(im);
      int m2 = (int)(fr < 1 ? 1 : (fr > canvas - 1 ? canvas - 1 : fr)); /* :59 */
      int m1 = (int)(fi < 1 ? 1 : (fi > canvas - 1 ? canvas - 1 : fi)); /* :61 */
      if (m1 + 1 >= canvas || m2 + 1 >= canvas) return -2;              /* IndexError in the reference */
      acc[(size_t)m1 * canvas + m2] += w * (tri(re - m2) * tri(im - m1));                 /* :64-75 */
      acc[(size_t)m1 * canvas + m2 + 1] += w * (tri(re - (m2 + 1)) * tri(im - m1));
      acc[(size_t)(m1 + 1) * canvas + m2] += w * (tri(re - m2) * tri(im - (m1 + 1)));
      acc[(size_t)(m1 + 1) * canvas + m2 + 1] += w * (tri(re - (m2 + 1)) * tri(im - (m1 + 1)));
    }
    double *out = psfs_out + (size_t)j * n;
    if (j == nfrac - 1) {
      for (size_t i = 0; i < n; ++i) acc[i] = acc[i] / iters;           /* :77 (last: in place) */
    } else {
      for (size_t i = 0; i < n; ++i) out[i] = acc[i] / iters;
    }
  }
  return 0;
}

/* numpy's DOUBLE_pairwise_sum over contiguous data */
static double pairwise(const double *a, size_t n) {
  if (n < 8) {
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += a[i];
    return s;
  }
  if (n <= 128) {
    double r[8];
    size_t i;
    for (i = 0; i < 8; ++i) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  size_t n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise(a, n2) + pairwise(a + n2, n - n2);
}

/* np.sum of a contiguous float64 array: 8192-element iterator chunks, left to right */
static double numpy_sum(const double *a, size_t n) {
  double s = 0.0;
  for (size_t i = 0; i < n; i += 8192) s += pairwise(a + i, n - i < 8192 ? n - i : 8192);
  return s;
}

int dib_psf_center(double *psf, int canvas, int *offsets_out) {
  if (!psf || canvas <= 0) return -1;
  const size_t n = (size_t)canvas * canvas;
  const double total = numpy_sum(psf, n);                               /* :108 */
  double ax = 0.0, ay = 0.0;
  for (int r = 0; r < canvas; ++r)
    for (int c = 0; c < canvas; ++c) {
      const double v = psf[(size_t)r * canvas + c];
      if (v > 0) {                                                      /* :110-117 */
        const double w = v / total;
        ax += (double)c * w;
        ay += (double)r * w;
      }
    }
  const int ox = (int)(ax - canvas / 2.0), oy = (int)(ay - canvas / 2.0); /* :119-120 */
  if (offsets_out) { offsets_out[0] = ox; offsets_out[1] = oy; }
  if (ox == 0 && oy == 0) return 0;
  /* np.roll(-ox, axis=1) then np.roll(-oy, axis=0): out[r][c] = in[(r+oy) mod n][(c+ox) mod n] */
  double *tmp = (double *)__builtin_malloc(n * sizeof(double));
  if (!tmp) return -3;
  for (int r = 0; r < canvas; ++r) {
    int sr = ((r + oy) % canvas + canvas) % canvas;
    for (int c = 0; c < canvas; ++c) {
      int sc = ((c + ox) % canvas + canvas) % canvas;
      tmp[(size_t)r * canvas + c] = psf[(size_t)sr * canvas + sc];
    }
  }
  for (size_t i = 0; i < n; ++i) psf[i] = tmp[i];
  __builtin_free(tmp);
  return 0;
}

/* ---- COCO matching of one (image, category) for every area range and IoU threshold ---------------------------------------
 * pycocotools COCOeval.evaluateImg (reference cocoapi/PythonAPI/pycocotools/cocoeval.py:235-310), the loop nest the evaluation
 * spends its host time in (10 thresholds x detections x ground truth x 4 area ranges, in interpreted Python there).  Same
 * statements in the same order: ground truth sorted "counted first" by a stable sort of the ignore flag (:257), per threshold
 * and detection the best still-available ground truth at IoU >= min(thr, 1 - 1e-10) with the early stop at the first ignored
 * one once a counted one matched (:273-296), unmatched detections outside the area range ignored (:298-299).
 * Called from a worker thread of engine.evaluate through ctypes, which drops the interpreter lock for the call. */
int dib_coco_match(const double *ious, int D, int G, const double *dt_area, const long long *crowd, const double *gt_area,
                   const double *area_rng, int A, const double *iou_thrs, int T, unsigned char *dtm_out,
                   unsigned char *dt_ig_out, int *n_gt_out) {
  if (D < 0 || G < 0 || A <= 0 || T <= 0 || !area_rng || !iou_thrs || !dtm_out || !dt_ig_out || !n_gt_out) return -1;
  if ((D > 0 && !dt_area) || (G > 0 && (!crowd || !gt_area)) || (D > 0 && G > 0 && !ious)) return -1;
  int *order = (int *)__builtin_malloc((size_t)(G > 0 ? G : 1) * sizeof(int));
  unsigned char *ig = (unsigned char *)__builtin_malloc((size_t)(G > 0 ? G : 1) * 2);
  unsigned char *gtm = (unsigned char *)__builtin_malloc((size_t)(G > 0 ? G : 1));
  if (!order || !ig || !gtm) { __builtin_free(order); __builtin_free(ig); __builtin_free(gtm); return -3; }
  unsigned char *cr = ig + (G > 0 ? G : 1);
  for (int a = 0; a < A; ++a) {
    const double lo = area_rng[2 * a], hi = area_rng[2 * a + 1];
    /* stable partition: counted ground truth first (np.argsort(ignore, kind="mergesort")) */
    int n = 0, counted = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int g = 0; g < G; ++g) {
        const int ignored = (crowd[g] != 0) || (gt_area[g] < lo) || (gt_area[g] > hi);
        if (ignored == pass) { order[n] = g; ig[n] = (unsigned char)ignored; cr[n] = (unsigned char)(crowd[g] != 0); ++n; }
      }
    for (int g = 0; g < G; ++g) counted += !ig[g];
    n_gt_out[a] = counted;
    unsigned char *dtm = dtm_out + (size_t)a * T * D, *dt_ig = dt_ig_out + (size_t)a * T * D;
    for (int t = 0; t < T; ++t) {
      for (int g = 0; g < G; ++g) gtm[g] = 0;
      const double thr = iou_thrs[t] < 1 - 1e-10 ? iou_thrs[t] : 1 - 1e-10;
      for (int d = 0; d < D; ++d) {
        double best = thr;
        int m = -1;
        for (int k = 0; k < G; ++k) {
          if (gtm[k] && !cr[k]) continue;
          if (m > -1 && !ig[m] && ig[k]) break;
          const double v = ious[(size_t)d * G + order[k]];
          if (v < best) continue;
          best = v;
          m = k;
        }
        dtm[(size_t)t * D + d] = (unsigned char)(m != -1);
        dt_ig[(size_t)t * D + d] = (unsigned char)(m != -1 ? ig[m] : 0);
        if (m != -1) gtm[m] = 1;
      }
      for (int d = 0; d < D; ++d)
        if (!dtm[(size_t)t * D + d] && (dt_area[d] < lo || dt_area[d] > hi)) dt_ig[(size_t)t * D + d] = 1;
    }
  }
  __builtin_free(order); __builtin_free(ig); __builtin_free(gtm);
  return 0;
}

/* One image, every category: what CocoBoxEvaluator.update does per image (COCOeval.evaluate's loop over catIds calling
 * evaluateImg, reference cocoeval.py:160-165, with computeIoU's cut to the maxDets[-1] best detections of a category, :178-184).
 * iou_all: [D][G] (every detection x every ground truth of the image); labels / scores / areas as given.  For category k (cats[k]):
 * its detections in stable descending-score order, at most max_det of them, are det_order[cat_start[k] .. cat_start[k + 1]);
 * dtm_out / dt_ig_out [A][T][D] hold their matching results at those positions (positions past cat_start[K] are unused);
 * n_gt_out [K][A]; gt_count_out [K] = ground truth of the category (a category without detections but with ground truth still
 * yields a record). */
int dib_coco_match_image(const double *iou_all, int D, int G, const long long *dt_label, const double *dt_score, const double *dt_area,
                         const long long *gt_label, const long long *gt_crowd, const double *gt_area, const long long *cats, int K,
                         int max_det, const double *area_rng, int A, const double *iou_thrs, int T, int *det_order, int *cat_start,
                         unsigned char *dtm_out, unsigned char *dt_ig_out, int *n_gt_out, int *gt_count_out) {
  if (D < 0 || G < 0 || K < 0 || A <= 0 || T <= 0 || max_det < 0 || !cat_start || !n_gt_out || !gt_count_out) return -1;
  if ((D > 0 && (!dt_label || !dt_score || !dt_area || !det_order || !dtm_out || !dt_ig_out)) || (G > 0 && (!gt_label || !gt_crowd || !gt_area))) return -1;
  if (K > 0 && !cats) return -1;
  const size_t dn = (size_t)(D > 0 ? D : 1), gn = (size_t)(G > 0 ? G : 1);
  int *gi = (int *)__builtin_malloc(gn * sizeof(int));
  double *sub = (double *)__builtin_malloc(dn * gn * sizeof(double));
  double *da = (double *)__builtin_malloc(dn * sizeof(double)), *ga = (double *)__builtin_malloc(gn * sizeof(double));
  long long *cr = (long long *)__builtin_malloc(gn * sizeof(long long));
  unsigned char *m1 = (unsigned char *)__builtin_malloc((size_t)A * T * dn * 2);
  int rc = 0;
  if (!gi || !sub || !da || !ga || !cr || !m1) rc = -3;
  int pos = 0;
  for (int k = 0; k < K && rc == 0; ++k) {
    cat_start[k] = pos;
    int nd = 0, ng = 0;
    for (int d = 0; d < D; ++d)
      if (dt_label[d] == cats[k]) {                      /* stable insertion by descending score (np.argsort(-scores, "mergesort")) */
        int j = pos + nd;
        while (j > pos && dt_score[det_order[j - 1]] < dt_score[d]) { det_order[j] = det_order[j - 1]; --j; }
        det_order[j] = d;
        ++nd;
      }
    if (nd > max_det) nd = max_det;
    for (int g = 0; g < G; ++g)
      if (gt_label[g] == cats[k]) gi[ng++] = g;
    gt_count_out[k] = ng;
    for (int a = 0; a < A; ++a) n_gt_out[(size_t)k * A + a] = 0;
    if (nd == 0 && ng == 0) continue;
    for (int i = 0; i < nd; ++i) {
      const int d = det_order[pos + i];
      da[i] = dt_area[d];
      for (int j = 0; j < ng; ++j) sub[(size_t)i * ng + j] = iou_all[(size_t)d * G + gi[j]];
    }
    for (int j = 0; j < ng; ++j) { cr[j] = gt_crowd[gi[j]]; ga[j] = gt_area[gi[j]]; }
    unsigned char *m2 = m1 + (size_t)A * T * dn;
    rc = dib_coco_match(sub, nd, ng, da, cr, ga, area_rng, A, iou_thrs, T, m1, m2, n_gt_out + (size_t)k * A);
    if (rc != 0) break;
    for (int a = 0; a < A; ++a)
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < nd; ++i) {
          dtm_out[((size_t)a * T + t) * D + pos + i] = m1[((size_t)a * T + t) * nd + i];
          dt_ig_out[((size_t)a * T + t) * D + pos + i] = m2[((size_t)a * T + t) * nd + i];
        }
    pos += nd;
  }
  if (rc == 0) cat_start[K] = pos;
  __builtin_free(gi); __builtin_free(sub); __builtin_free(da); __builtin_free(ga); __builtin_free(cr); __builtin_free(m1);
  return rc;
}

/* ---- COCO accumulate for one category ---------------------------------------------------------------------------------------
 * pycocotools COCOeval.accumulate (reference cocoapi/PythonAPI/pycocotools/cocoeval.py:315-420) for the records of ONE category:
 * per maxDets the first max_det detections of every image, all of them ranked by a stable descending sort of the score (:366-372),
 * per area range and IoU threshold the running true / false positive counts, recall = tp / npig, precision = tp / (fp + tp + eps),
 * the precision envelope from the right, and the precision at the first position whose recall reaches each recall threshold
 * (np.searchsorted(rc, recThrs, side="left"); 0 beyond the last) (:381-412).  Inputs are the category's records laid end to end:
 * scores [n], lens [nrec] (detections per image, summing to n), dtm / dt_ig [A][T][n] bytes, n_gt [A] (counted ground truth).
 * Outputs, written only where n_gt[a] > 0 (the caller pre-fills -1): precision_out [A][M][T][R], recall_out [A][M][T]. */
static void merge_sort_desc(int *idx, int *tmp, int n, const double *score) {
  for (int w = 1; w < n; w *= 2) {
    for (int lo = 0; lo < n; lo += 2 * w) {
      int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n, i = lo, j = mid, k = lo;
      while (i < mid && j < hi) tmp[k++] = (score[idx[j]] > score[idx[i]]) ? idx[j++] : idx[i++];     /* ties: the left (earlier) one first */
      while (i < mid) tmp[k++] = idx[i++];
      while (j < hi) tmp[k++] = idx[j++];
    }
    for (int i = 0; i < n; ++i) idx[i] = tmp[i];
  }
}

int dib_coco_accumulate_cat(const double *scores, const int *lens, int nrec, const unsigned char *dtm, const unsigned char *dt_ig,
                            const int *n_gt, int A, int T, const int *max_dets, int M, const double *rec_thrs, int R,
                            double *precision_out, double *recall_out) {
  if (nrec < 0 || A <= 0 || T <= 0 || M <= 0 || R <= 0 || !n_gt || !max_dets || !rec_thrs || !precision_out || !recall_out) return -1;
  long long n = 0;
  for (int i = 0; i < nrec; ++i) { if (lens[i] < 0) return -1; n += lens[i]; }
  if (n > 0 && (!scores || !dtm || !dt_ig)) return -1;
  const size_t cap = (size_t)(n > 0 ? n : 1);
  int *idx = (int *)__builtin_malloc(cap * sizeof(int)), *tmp = (int *)__builtin_malloc(cap * sizeof(int));
  double *rc = (double *)__builtin_malloc(cap * sizeof(double)), *pr = (double *)__builtin_malloc(cap * sizeof(double));
  if (!idx || !tmp || !rc || !pr) { __builtin_free(idx); __builtin_free(tmp); __builtin_free(rc); __builtin_free(pr); return -3; }
  const double eps = 2.220446049250313e-16;                       /* np.spacing(1) */
  for (int m = 0; m < M; ++m) {
    int nd = 0;
    long long off = 0;
    for (int i = 0; i < nrec; ++i) {
      const int take = lens[i] < max_dets[m] ? lens[i] : max_dets[m];
      for (int j = 0; j < take; ++j) idx[nd++] = (int)(off + j);
      off += lens[i];
    }
    merge_sort_desc(idx, tmp, nd, scores);
    for (int a = 0; a < A; ++a) {
      if (n_gt[a] == 0) continue;
      const double npig = (double)n_gt[a];
      for (int t = 0; t < T; ++t) {
        const unsigned char *mt = dtm + ((size_t)a * T + t) * (size_t)n, *ig = dt_ig + ((size_t)a * T + t) * (size_t)n;
        long long tp = 0, fp = 0;
        for (int i = 0; i < nd; ++i) {
          const int d = idx[i];
          if (!ig[d]) { if (mt[d]) ++tp; else ++fp; }
          rc[i] = (double)tp / npig;
          pr[i] = (double)tp / (((double)fp + (double)tp) + eps);
        }
        recall_out[((size_t)a * M + m) * T + t] = nd ? rc[nd - 1] : 0.0;
        for (int i = nd - 1; i > 0; --i)
          if (pr[i] > pr[i - 1]) pr[i - 1] = pr[i];
        double *q = precision_out + (((size_t)a * M + m) * T + t) * (size_t)R;
        int pos = 0;
        for (int r = 0; r < R; ++r) {                               /* rec_thrs ascend, rc is non-decreasing: one forward walk */
          while (pos < nd && rc[pos] < rec_thrs[r]) ++pos;
          q[r] = pos < nd ? pr[pos] : 0.0;
        }
      }
    }
  }
  __builtin_free(idx); __builtin_free(tmp); __builtin_free(rc); __builtin_free(pr);
  return 0;
}

/* ---- segmentation masks of ConvertCocoPolysToMask (reference coco_utils.py:34-49: frPyObjects + decode + any) -----------------
 * A polygon is rasterised the way pycocotools does it (cocoapi/common/maskApi.c:162-218, rleFrPoly), restated:
 *   1. vertices to a grid five times finer than the pixels, (int)(5 v + 0.5) per coordinate (C truncation, as there);
 *   2. every edge walked along its longer axis, one grid point per step, the other coordinate (int)(start + slope * t + 0.5);
 *      an edge that runs backwards along its longer axis is walked from its far end so that the points come out in order;
 *   3. wherever the walk's x changes, the crossing of a pixel-column boundary is kept if it falls on a pixel centre's column
 *      (x5 -> (x5 + 0.5) / 5 - 0.5 integral and inside [0, w - 1]), with the row clamped to [0, h] and rounded up;
 *   4. the crossings, as column-major positions x * h + y, sorted: consecutive differences are the run lengths of a column-major
 *      mask that starts with zeros (the image's end h * w closes the last run); zero-length runs merge their neighbours.
 * The runs are decoded straight into the row-major byte mask [h][w], OR-ing (the reference takes `any` over an object's
 * polygons).  Returns 0, -1 on bad arguments, -3 out of memory. */
static int dib_cmp_u32(const void *a, const void *b) {
  const unsigned x = *(const unsigned *)a, y = *(const unsigned *)b;
  return x > y ? 1 : (x < y ? -1 : 0);
}

/* runs of a column-major mask (first run zeros) OR-ed into the row-major mask */
static void dib_runs_or(const unsigned *runs, long n_runs, long h, long w, unsigned char *mask) {
  const unsigned long long total = (unsigned long long)h * (unsigned long long)w;
  unsigned long long at = 0;
  int v = 0;
  for (long r = 0; r < n_runs && at < total; ++r, v = !v) {
    unsigned long long end = at + runs[r];
    if (end > total) end = total;
    if (v) {          /* column by column: one division per column segment, not per pixel */
      unsigned long long p = at;
      while (p < end) {
        const unsigned long long x = p / (unsigned long long)h, y0 = p % (unsigned long long)h;
        unsigned long long n = (unsigned long long)h - y0;
        if (n > end - p) n = end - p;
        unsigned char *col = mask + y0 * (unsigned long long)w + x;
        for (unsigned long long k = 0; k < n; ++k) col[k * (unsigned long long)w] = 1;
        p += n;
      }
    }
    at = end;
  }
}

int dib_mask_or_runs(const unsigned *runs, long n_runs, long h, long w, unsigned char *mask) {
  if (!runs || n_runs < 0 || h <= 0 || w <= 0 || !mask) return -1;
  dib_runs_or(runs, n_runs, h, w, mask);
  return 0;
}

int dib_mask_or_polygon(const double *xy, long k, long h, long w, unsigned char *mask) {
  if (!xy || k <= 0 || h <= 0 || w <= 0 || !mask || (unsigned long long)h * (unsigned long long)w >= 0xffffffffull) return -1;
  const double scale = 5;
  int *vx = (int *)__builtin_malloc(sizeof(int) * (size_t)(k + 1)), *vy = (int *)__builtin_malloc(sizeof(int) * (size_t)(k + 1));
  if (!vx || !vy) { __builtin_free(vx); __builtin_free(vy); return -3; }
  for (long j = 0; j < k; ++j) { vx[j] = (int)(scale * xy[2 * j] + .5); vy[j] = (int)(scale * xy[2 * j + 1] + .5); }
  vx[k] = vx[0]; vy[k] = vy[0];
  size_t n_pts = 0;
  for (long j = 0; j < k; ++j) {
    const int ax = __builtin_abs(vx[j] - vx[j + 1]), ay = __builtin_abs(vy[j] - vy[j + 1]);
    n_pts += (size_t)(ax > ay ? ax : ay) + 1;
  }
  int *px = (int *)__builtin_malloc(sizeof(int) * (n_pts + 1)), *py = (int *)__builtin_malloc(sizeof(int) * (n_pts + 1));
  unsigned *pos = (unsigned *)__builtin_malloc(sizeof(unsigned) * (n_pts + 2));
  unsigned *runs = (unsigned *)__builtin_malloc(sizeof(unsigned) * (n_pts + 2));
  if (!px || !py || !pos || !runs) { __builtin_free(vx); __builtin_free(vy); __builtin_free(px); __builtin_free(py); __builtin_free(pos); __builtin_free(runs); return -3; }
  size_t m = 0;
  for (long j = 0; j < k; ++j) {          /* step 2: the dense boundary */
    int x0 = vx[j], x1 = vx[j + 1], y0 = vy[j], y1 = vy[j + 1];
    const int dx = __builtin_abs(x1 - x0), dy = __builtin_abs(y0 - y1);
    const int along_x = dx >= dy;
    const int backwards = (along_x && x0 > x1) || (!along_x && y0 > y1);
    if (backwards) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
    const double slope = along_x ? (double)(y1 - y0) / dx : (double)(x1 - x0) / dy;      /* dx == dy == 0: NaN, used with t = 0 only (as there) */
    const int len = along_x ? dx : dy;
    for (int d = 0; d <= len; ++d) {
      const int t = backwards ? len - d : d;
      if (along_x) { px[m] = t + x0; py[m] = (int)(y0 + slope * t + .5); }
      else { py[m] = t + y0; px[m] = (int)(x0 + slope * t + .5); }
      ++m;
    }
  }
  size_t n_cross = 0;
  for (size_t j = 1; j < m; ++j) {        /* step 3: crossings of pixel-column boundaries */
    if (px[j] == px[j - 1]) continue;
    double xd = (double)(px[j] < px[j - 1] ? px[j] : px[j] - 1);
    xd = (xd + .5) / scale - .5;
    if (__builtin_floor(xd) != xd || xd < 0 || xd > (double)(w - 1)) continue;
    double yd = (double)(py[j] < py[j - 1] ? py[j] : py[j - 1]);
    yd = (yd + .5) / scale - .5;
    if (yd < 0) yd = 0; else if (yd > (double)h) yd = (double)h;
    yd = __builtin_ceil(yd);
    pos[n_cross++] = (unsigned)((int)xd * (int)h + (int)yd);
  }
  pos[n_cross++] = (unsigned)((unsigned long long)h * (unsigned long long)w);
  qsort(pos, n_cross, sizeof(unsigned), dib_cmp_u32);
  unsigned prev = 0;
  for (size_t j = 0; j < n_cross; ++j) { const unsigned t = pos[j]; pos[j] -= prev; prev = t; }     /* step 4: differences */
  size_t n_runs = 0, j = 0;
  runs[n_runs++] = pos[j++];
  while (j < n_cross) {
    if (pos[j] > 0) runs[n_runs++] = pos[j++];
    else { ++j; if (j < n_cross) runs[n_runs - 1] += pos[j++]; }          /* an empty run joins the runs on either side of it */
  }
  dib_runs_or(runs, (long)n_runs, h, w, mask);
  __builtin_free(vx); __builtin_free(vy); __builtin_free(px); __builtin_free(py); __builtin_free(pos); __builtin_free(runs);
  return 0;
}
