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

#ifdef __cplusplus
}
#endif
#endif
