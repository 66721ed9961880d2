// Tap compaction for gfx950: one 1024-thread workgroup per K x K PSF, single pass over the data.
//
// Replaces, for a whole batch and without any host synchronisation, the reference's
//   psf_GPU = psf_GPU / psf_GPU.sum();  non_zero_points = psf_GPU.nonzero()
// (models/blur_functions.py:98,63 and again utils.py:372-374) plus the min/max of the tap
// coordinates that expand_targets needs (utils.py:376-380), and cuts the tap list into the
// bounded segments the tiled blur stages in LDS.
//
// fp16 semantics: the sum is formed EXACTLY (every finite fp16 is a multiple of 2^-24, so the
// total is an int64 in those units) and rounded once to fp16 (round-to-nearest-even); the
// division is an IEEE fp32 divide rounded to fp16, which equals a correctly rounded fp16 divide
// (24 >= 2*11+2).  Order of the output taps = row-major, the order of torch.nonzero: every thread
// owns K*K/1024 CONSECUTIVE elements, so thread order is element order.
#include "dib_compact_dev.h"
#include <hip/hip_ext.h>
#include <vector>

namespace dib {

constexpr int CT = 1024;          // threads per PSF
constexpr int STAGE_TAPS = 4096;  // (row<<8|col) of the first taps are staged in LDS for the segmenter

template <typename T, int K>
__global__ __launch_bounds__(CT) void psf_compact_kernel(PsfPtrs ptrs, int normalize, int *__restrict__ tables) {
  using E = Elem<T>;
  constexpr int N = K * K, EPT = N / CT, LK = (K == 128) ? 7 : 8, NWAVE = CT / 64;
  __shared__ typename E::Acc s_part[NWAVE];
  __shared__ int s_wtot[NWAVE];
  __shared__ int s_ext[4];  // rmin rmax cmin cmax
  __shared__ unsigned short s_rc[STAGE_TAPS];
  __shared__ unsigned s_wb[STAGE_TAPS];      // weight bits (raw before the division, final after)
  __shared__ int s_flag;
  __shared__ unsigned s_vg[32];              // scratch of the vertical-run grouping (COMPACT_VRUNS)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T *p = reinterpret_cast<const T *>(ptrs.p[blockIdx.x]) + (size_t)tid * EPT;
  int *tab = tables + (size_t)blockIdx.x * table_words(K);


  // ---- one vectorised read of this thread's EPT consecutive elements ---------------------------
  T v[EPT];
  {
    constexpr int VEC = 16 / sizeof(T);  // elements per 16-byte load
    const uint4 *p4 = reinterpret_cast<const uint4 *>(p);
#pragma unroll
    for (int i = 0; i < EPT / VEC; ++i) {
      uint4 q = p4[i];
      const T *e = reinterpret_cast<const T *>(&q);
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[i * VEC + k] = e[k];
    }
  }
  if (tid < 4) s_ext[tid] = (tid & 1) ? -1 : K;

  // ================================ fast path ==================================================
  // A PSF is a thin curve (15..272 non-zeros of 16,384): compact the RAW non-zeros first, then sum,
  // divide and test only those -- one element per thread instead of EPT slots per thread.  It gives
  // up (and the general path below redoes everything) when the non-zeros do not fit the LDS stage,
  // when the sum is 0 / NaN (then every element becomes a NaN tap) or when a weight underflows to
  // zero in the division (that tap has to vanish, which shifts all later positions).
  const int e0f = tid * EPT;
  int ntaps = 0;
  bool fast_done = false;
  T total = T(1.0f);
  {
    unsigned long long rawmask = 0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) rawmask |= (unsigned long long)E::nonzero(v[i]) << i;
    const int rcnt = __popcll(rawmask);
    const bool wave_busy = __builtin_amdgcn_ballot_w64(rcnt != 0) != 0;
    const int rincl = wave_busy ? wave_scan_incl_i32(rcnt, lane) : 0;
    if (lane == 63) s_wtot[wave] = rincl;
    if (tid == 0) s_flag = 0;
    __syncthreads();                                                     // A
    const int wt = lane < NWAVE ? s_wtot[lane] : 0;
    const int wincl = wave_scan_incl_i32(wt, lane);
    const int n_raw = __builtin_amdgcn_readlane(wincl, NWAVE - 1);
    const int rbase = __builtin_amdgcn_readlane(wincl - wt, wave);
    const int rpos = rbase + rincl - rcnt;                               // this thread's first raw tap
    bool ok = n_raw <= STAGE_TAPS;
    if (ok && wave_busy) {
      int q = rpos;
#pragma unroll
      for (int i = 0; i < EPT; ++i)
        if (rawmask & (1ull << i)) {
          s_rc[q] = (unsigned short)((((e0f + i) >> LK) << 8) | ((e0f + i) & (K - 1)));
          s_wb[q] = E::bits(v[i]);
          ++q;
        }
    }
    __syncthreads();                                                     // B: raw taps staged in row-major order
    if (ok && (normalize & 1)) {
      typename E::Acc acc = 0;
      for (int j = tid; j < n_raw; j += CT) acc += E::lift(E::from_bits(s_wb[j]));
      const bool wbusy = __builtin_amdgcn_ballot_w64(tid < n_raw) != 0;
      if (wbusy) acc = wave_sum(acc);
      if (lane == 0) s_part[wave] = acc;
      __syncthreads();                                                   // C
      typename E::Acc part = lane < NWAVE ? s_part[lane] : typename E::Acc(0);
      total = E::finish(wave_total(part));
      ok = (float)total == (float)total && E::nonzero(total);
    }
    if (ok) {
      uint2 *taps_f = reinterpret_cast<uint2 *>(tab + table_taps_off(K));
      for (int j = tid; j < n_raw; j += CT) {
        T w = E::from_bits(s_wb[j]);
        if (normalize & 1) w = E::div(w, total);
        if (!E::nonzero(w)) s_flag = 1;                                  // underflow: give up
        taps_f[j] = make_uint2((unsigned)s_rc[j], E::bits(w));
        s_wb[j] = E::bits(w);
      }
    }
    __syncthreads();                                                     // D
    ok = ok && s_flag == 0;
    if (ok) {
      constexpr int TPRF = K / EPT;  // threads per PSF row
      if ((tid % TPRF) == 0) tab[table_rowptr_off() + tid / TPRF] = rpos;
      if (wave == 0) {
        // extents: rows from the first / last tap (row-major order), columns by a reduction
        int cmn = K, cmx = -1;
        for (int j = lane; j < n_raw; j += 64) { const int c = s_rc[j] & 255; cmn = min(cmn, c); cmx = max(cmx, c); }
        cmn = wave_min_i32(cmn); cmx = wave_max_i32(cmx);
        if (lane == 0) {
          tab[table_rowptr_off() + K] = n_raw;
          tab[HDR_NTAPS] = n_raw;
          tab[HDR_RMIN] = n_raw ? (s_rc[0] >> 8) : K; tab[HDR_RMAX] = n_raw ? (s_rc[n_raw - 1] >> 8) : -1;
          tab[HDR_CMIN] = cmn; tab[HDR_CMAX] = cmx;
          tab[HDR_K] = K | ((normalize & COMPACT_LARGE_WINDOW) ? 1 << 16 : 0); tab[HDR_SUM] = (int)E::bits(total);
        }
      }
      ntaps = n_raw;
      fast_done = true;
    }
  }
  uint2 *taps = reinterpret_cast<uint2 *>(tab + table_taps_off(K));
  if (!fast_done) {
  // ================================ general path ===============================================
  // A PSF is a thin curve: most of the 16 waves hold nothing but zeros.  One ballot per wave settles that
  // (`busy`); a busy wave still skips every element slot that is zero in all its lanes.  Lifting, dividing
  // and testing all 16,384 elements one by one used to be most of this kernel's time.
  bool any = false;
#pragma unroll
  for (int i = 0; i < EPT; ++i) any = any || E::nonzero(v[i]);
  const bool busy = __builtin_amdgcn_ballot_w64(any) != 0;   // wave-uniform

  // ---- sum ------------------------------------------------------------------------------------
  total = T(1.0f);
  if (normalize & 1) {
    typename E::Acc acc = 0;
    if (busy) {
#pragma unroll
      for (int i = 0; i < EPT; ++i)
        if (__builtin_amdgcn_ballot_w64(E::nonzero(v[i])) != 0) acc += E::lift(v[i]);
      acc = wave_sum(acc);
    }
    if (lane == 0) s_part[wave] = acc;
    __syncthreads();
    // every wave adds the 16 partials itself (lane k takes partial k): no second barrier, no serial loop
    typename E::Acc part = lane < NWAVE ? s_part[lane] : typename E::Acc(0);
    total = E::finish(wave_total(part));
  } else {
    __syncthreads();   // s_ext is initialised before any wave's atomics below
  }

  // ---- weights, non-zero mask, extents ----------------------------------------------------------
  unsigned long long mask = 0;
  int rmin = K, rmax = -1, cmin = K, cmax = -1;
  const int e0 = tid * EPT;
  // 0 / total is 0 unless total is 0 or NaN (then the reference's psf / psf.sum() is NaN everywhere and
  // every element becomes a tap): only in that case are all-zero slots divided too
  const bool sane = (float)total == (float)total && E::nonzero(total);
  if (busy || !sane) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      if (sane && __builtin_amdgcn_ballot_w64(E::nonzero(v[i])) == 0) continue;
      if (normalize & 1) v[i] = E::div(v[i], total);
      if (E::nonzero(v[i])) {
        mask |= 1ull << i;
        const int r = (e0 + i) >> LK, c = (e0 + i) & (K - 1);
        rmin = min(rmin, r); rmax = max(rmax, r); cmin = min(cmin, c); cmax = max(cmax, c);
      }
    }
  }
  const int cnt = __popcll(mask);
  int incl = cnt;
  if (busy || !sane) {
    rmin = wave_min_i32(rmin); rmax = wave_max_i32(rmax);
    cmin = wave_min_i32(cmin); cmax = wave_max_i32(cmax);
    if (lane == 0) {
      atomicMin(&s_ext[0], rmin); atomicMax(&s_ext[1], rmax);
      atomicMin(&s_ext[2], cmin); atomicMax(&s_ext[3], cmax);
    }
    // ---- inclusive scan of the per-thread counts inside the wave ----------------------------------
    incl = wave_scan_incl_i32(cnt, lane);
  }
  if (lane == 63) s_wtot[wave] = incl;
  __syncthreads();
  // wave totals -> bases: lane k takes wave k's total, a 16-lane scan gives every wave its base
  int wt = lane < NWAVE ? s_wtot[lane] : 0;
  const int wincl = wave_scan_incl_i32(wt, lane);
  ntaps = __builtin_amdgcn_readlane(wincl, NWAVE - 1);
  const int wbase = __builtin_amdgcn_readlane(wincl - wt, wave);
  int pos = wbase + incl - cnt;

  // ---- header, CSR row pointers, taps -----------------------------------------------------------------
  constexpr int TPR = K / EPT;  // threads per PSF row
  if ((tid % TPR) == 0) tab[table_rowptr_off() + tid / TPR] = pos;
  if (tid == 0) {
    tab[table_rowptr_off() + K] = ntaps;
    tab[HDR_NTAPS] = ntaps;
    tab[HDR_RMIN] = s_ext[0]; tab[HDR_RMAX] = s_ext[1]; tab[HDR_CMIN] = s_ext[2]; tab[HDR_CMAX] = s_ext[3];
    tab[HDR_K] = K | ((normalize & COMPACT_LARGE_WINDOW) ? 1 << 16 : 0); tab[HDR_SUM] = (int)E::bits(total);
  }
  if (__builtin_amdgcn_ballot_w64(mask != 0) != 0)
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    if (mask & (1ull << i)) {
      const unsigned rc = (unsigned)(((e0 + i) >> LK) << 8 | ((e0 + i) & (K - 1)));
      taps[pos] = make_uint2(rc, E::bits(v[i]));
      if (pos < STAGE_TAPS) { s_rc[pos] = (unsigned short)rc; s_wb[pos] = E::bits(v[i]); }
      ++pos;
    }
  }
  __threadfence_block();
  __syncthreads();
  }  // general path
  if (wave != 0) return;
  if (normalize & COMPACT_NO_SEGMENTS) return;  // diagnostics: skip the segmentation

  // ---- segmentation (wave 0): dib_compact_dev.h ----
  segment_taps<false, STAGE_TAPS, false, true>(tab, K, normalize, ntaps, (const lds_u16 *)s_rc, (const lds_u32 *)s_wb, lane, nullptr, (lds_u32 *)s_vg,
                                               sizeof(T) == 2);
}

}  // namespace dib

extern "C" size_t dib_tap_table_bytes(int K) {
  if (K != 128 && K != 256) return 0;
  return (size_t)dib::table_words(K) * sizeof(int);
}

extern "C" size_t dib_tap_tables_bytes(int K, int B) {
  if ((K != 128 && K != 256) || B < 0) return 0;
  return (size_t)dib::table_words(K) * B * sizeof(int);
}

// any_order: the launches carry hipExtAnyOrderLaunch -- no barrier in front of them, so they may start while the kernel queued before
// them on the same stream is still running (dib_blur_step with DIB_STEP_PSFS_COMPLETE: the previous step's blur).  The next ordinary
// launch on the stream still waits for them.
int dib::compact_launch(const void *const *ptrs, int dtype, int B, int K, int normalize, int *tables, hipStream_t s, bool any_order) {
  const size_t stride = (size_t)dib::table_words(K);
  const int lflags = any_order ? hipExtAnyOrderLaunch : 0;
  for (int b0 = 0; b0 < B; b0 += dib::MAX_BATCH) {
    dib::PsfPtrs pp;
    const int n = B - b0 < dib::MAX_BATCH ? B - b0 : dib::MAX_BATCH;
    for (int i = 0; i < n; ++i) {
      if (!ptrs[b0 + i] || ((uintptr_t)ptrs[b0 + i] & 15) != 0) { dib::set_error("dib_psf_compact: PSF %d is null or not 16-byte aligned", b0 + i); return DIB_EINVAL; }
      pp.p[i] = ptrs[b0 + i];
    }
    int *t = tables + (size_t)b0 * stride;
    // `normalize` doubles as the flag word: bit 3 (DIB_COMPACT_LARGE_WINDOW) selects the large-window segmentation, anything
    // else that is non-zero means "divide by the sum first"
    const int flags = ((normalize & ~(DIB_COMPACT_LARGE_WINDOW | DIB_COMPACT_VRUNS)) ? COMPACT_NORMALIZE : 0) | ((normalize & DIB_COMPACT_LARGE_WINDOW) ? COMPACT_LARGE_WINDOW : 0) |
                      ((normalize & DIB_COMPACT_VRUNS) ? COMPACT_VRUNS : 0);
    if (dtype == DIB_F16 && K == 128) hipExtLaunchKernelGGL((dib::psf_compact_kernel<__half, 128>), dim3(n), dim3(dib::CT), 0, s, nullptr, nullptr, lflags, pp, flags, t);
    else if (dtype == DIB_F16) hipExtLaunchKernelGGL((dib::psf_compact_kernel<__half, 256>), dim3(n), dim3(dib::CT), 0, s, nullptr, nullptr, lflags, pp, flags, t);
    else if (K == 128) hipExtLaunchKernelGGL((dib::psf_compact_kernel<float, 128>), dim3(n), dim3(dib::CT), 0, s, nullptr, nullptr, lflags, pp, flags, t);
    else hipExtLaunchKernelGGL((dib::psf_compact_kernel<float, 256>), dim3(n), dim3(dib::CT), 0, s, nullptr, nullptr, lflags, pp, flags, t);
  }
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
static int launch_compact(const void *const *ptrs, int dtype, int B, int K, int normalize, int *tables, hipStream_t s) {
  return dib::compact_launch(ptrs, dtype, B, K, normalize, tables, s, false);
}

static int check_compact_args(const void *p, void *tables_dev, int dtype, int B, int K) {
  if (!p || !tables_dev || B < 0) { dib::set_error("dib_psf_compact: null pointer or negative batch"); return DIB_EINVAL; }
  if (K != 128 && K != 256) { dib::set_error("dib_psf_compact: K must be 128 or 256, got %d", K); return DIB_EINVAL; }
  if (dtype != DIB_F16 && dtype != DIB_F32) { dib::set_error("dib_psf_compact: unknown dtype %d", dtype); return DIB_EINVAL; }
  return DIB_OK;
}

extern "C" int dib_psf_compact(const void *psf_dev, int dtype, int B, int K, int normalize, void *tables_dev, void *stream) {
  int rc = check_compact_args(psf_dev, tables_dev, dtype, B, K);
  if (rc != DIB_OK || B == 0) return rc;
  const size_t bytes = (size_t)K * K * (dtype == DIB_F16 ? 2 : 4);
  std::vector<const void *> ptrs((size_t)B);
  for (int i = 0; i < B; ++i) ptrs[i] = (const char *)psf_dev + (size_t)i * bytes;
  return launch_compact(ptrs.data(), dtype, B, K, normalize, (int *)tables_dev, (hipStream_t)stream);
}

extern "C" int dib_psf_compact_list(const void *const *psf_ptrs, int dtype, int B, int K, int normalize, void *tables_dev,
                                    void *stream) {
  int rc = check_compact_args(psf_ptrs, tables_dev, dtype, B, K);
  if (rc != DIB_OK || B == 0) return rc;
  return launch_compact(psf_ptrs, dtype, B, K, normalize, (int *)tables_dev, (hipStream_t)stream);
}

// Test hook (not part of the drop-in boundary): the 256-thread compaction of the blur step's single launch
// (dib_compact_dev.h: compact_psf_f16_wg256, write-through stores) on its own, into caller tables, so that the parity tests
// can compare its tables with the stand-alone kernel's word by word.  fp16 PSFs, K = 128, B <= 32.
namespace dib {
__global__ __launch_bounds__(256, 8) void psf_compact_wg256_kernel(PsfPtrs ptrs, int flags, int *__restrict__ tables) {
  extern __shared__ unsigned pool[];
  compact_psf_f16_wg256<true>(ptrs.p[blockIdx.x], flags, tables + (size_t)blockIdx.x * table_words(128), (lds_u32 *)pool, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6));
}
}  // namespace dib
extern "C" int dib_debug_compact_wg256(const void *const *psf_ptrs, int B, int normalize, void *tables_dev, void *stream) {
  if (!psf_ptrs || !tables_dev || B < 1 || B > dib::MAX_BATCH) { dib::set_error("dib_debug_compact_wg256: bad arguments"); return DIB_EINVAL; }
  dib::PsfPtrs pp;
  for (int i = 0; i < B; ++i) pp.p[i] = psf_ptrs[i];
  const int flags = ((normalize & ~DIB_COMPACT_LARGE_WINDOW) ? dib::COMPACT_NORMALIZE : 0) | ((normalize & DIB_COMPACT_LARGE_WINDOW) ? dib::COMPACT_LARGE_WINDOW : 0);
  hipLaunchKernelGGL(dib::psf_compact_wg256_kernel, dim3(B), dim3(256), 19712, (hipStream_t)stream, pp, flags, (int *)tables_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
