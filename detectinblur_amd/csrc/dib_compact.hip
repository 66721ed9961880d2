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
#include "dib_common.h"
#include <hip/hip_fp16.h>
#include <vector>

namespace dib {

__device__ inline long long half_bits_to_fixed(unsigned short h) {
  // value * 2^24 as an exact integer (inf/nan map to a huge sentinel that poisons the sum)
  int sign = h >> 15, e = (h >> 10) & 31, m = h & 1023;
  long long v;
  if (e == 0) v = m;                       // subnormal: m * 2^-24
  else if (e == 31) v = (1ll << 60);       // inf / nan: out of contract
  else v = (long long)(m | 1024) << (e - 1);
  return sign ? -v : v;
}

__device__ inline unsigned short fixed_to_half_bits(long long total) {
  // exact integer in units of 2^-24 -> fp16, round to nearest even
  unsigned short sign = total < 0 ? 0x8000 : 0;
  unsigned long long m = total < 0 ? (unsigned long long)(-total) : (unsigned long long)total;
  if (m == 0) return sign;
  int nbits = 64 - __clzll((long long)m);
  if (nbits <= 10) return sign | (unsigned short)m;            // subnormal, exact
  int shift = nbits - 11;                                       // keep 11 significant bits
  unsigned long long q = m >> shift, rem = m & ((1ull << shift) - 1);
  if (shift > 0) {
    unsigned long long half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) q++;
  }
  if (q == 2048) { q = 1024; shift++; }
  int e = shift + 1;                                            // biased exponent (value = q * 2^(shift-24))
  if (e >= 31) return sign | 0x7c00;                            // overflow -> inf
  return sign | (unsigned short)((e << 10) | (q & 1023));
}

template <typename T> struct Elem;
template <> struct Elem<__half> {
  using Acc = long long;
  static __device__ Acc lift(__half v) { return half_bits_to_fixed(__half_as_ushort(v)); }
  static __device__ __half finish(Acc a) { return __ushort_as_half(fixed_to_half_bits(a)); }
  static __device__ __half div(__half a, __half b) { return __float2half_rn(__half2float(a) / __half2float(b)); }
  static __device__ bool nonzero(__half v) { return (__half_as_ushort(v) & 0x7fff) != 0; }
  static __device__ unsigned bits(__half v) { return __half_as_ushort(v); }
};
template <> struct Elem<float> {
  // fp32 PSFs (manual_blur with fp32 operands): torch's fp32 sum order is implementation
  // defined; this path accumulates in fp64 in a fixed order and rounds once.
  using Acc = double;
  static __device__ Acc lift(float v) { return (double)v; }
  static __device__ float finish(Acc a) { return (float)a; }
  static __device__ float div(float a, float b) { return a / b; }
  static __device__ bool nonzero(float v) { return v != 0.0f; }
  static __device__ unsigned bits(float v) { return __float_as_uint(v); }
};

template <typename A> __device__ inline A wave_sum(A v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

constexpr int CT = 1024;          // threads per PSF
constexpr int STAGE_TAPS = 4096;  // (row<<8|col) of the first taps are staged in LDS for the segmenter

// PSF pointers travel by value in the kernel-argument buffer: a batch whose PSFs live in separate
// tensors (the reference's `psfs_GPU` list, engine.py:84) needs no torch.stack copy.
struct PsfPtrs { const void *p[MAX_BATCH]; };

template <typename T, int K>
__global__ __launch_bounds__(CT) void psf_compact_kernel(PsfPtrs ptrs, int normalize, int *__restrict__ tables) {
  using E = Elem<T>;
  constexpr int N = K * K, EPT = N / CT, LK = (K == 128) ? 7 : 8, NWAVE = CT / 64;
  __shared__ typename E::Acc s_part[NWAVE];
  __shared__ int s_wtot[NWAVE];
  __shared__ int s_ext[4];  // rmin rmax cmin cmax
  __shared__ T s_sum;
  __shared__ unsigned short s_rc[STAGE_TAPS];
  __shared__ unsigned short s_w16[STAGE_TAPS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T *p = reinterpret_cast<const T *>(ptrs.p[blockIdx.x]) + (size_t)tid * EPT;
  int *tab = tables + (size_t)blockIdx.x * table_words(K);

  // scheduler trailer behind the last table: the blur's tile-queue tickets start from zero
  if (blockIdx.x == 0 && tid < SCHED_WORDS && (normalize & 8)) tables[(size_t)gridDim.x * table_words(K) + tid] = 0;

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

  // ---- sum ------------------------------------------------------------------------------------
  T total = T(1.0f);
  if (normalize & 1) {
    // A PSF is a thin curve: of the 64 x EPT elements a wave holds, a few dozen are non-zero.  Zeros
    // add nothing, so element slots that are zero in every lane are skipped with one ballot
    // (lifting and dividing all 16,384 elements used to be a third of this kernel's time).
    typename E::Acc acc = 0;
#pragma unroll
    for (int i = 0; i < EPT; ++i)
      if (__builtin_amdgcn_ballot_w64(E::nonzero(v[i])) != 0) acc += E::lift(v[i]);
    acc = wave_sum(acc);
    if (lane == 0) s_part[wave] = acc;
    __syncthreads();
    if (tid == 0) {
      typename E::Acc a = 0;
      for (int k = 0; k < NWAVE; ++k) a += s_part[k];
      s_sum = E::finish(a);
    }
    __syncthreads();
    total = s_sum;
  } else {
    __syncthreads();
  }

  // ---- weights, non-zero mask, extents ----------------------------------------------------------
  unsigned long long mask = 0;
  int rmin = K, rmax = -1, cmin = K, cmax = -1;
  const int e0 = tid * EPT;
  // 0 / total is 0 unless total is 0 or NaN (then the reference's psf / psf.sum() is NaN everywhere and
  // every element becomes a tap): only in that case are all-zero slots divided too
  const bool sane = (float)total == (float)total && E::nonzero(total);
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
  const int cnt = __popcll(mask);
  for (int off = 32; off > 0; off >>= 1) {
    rmin = min(rmin, __shfl_down(rmin, off, 64)); rmax = max(rmax, __shfl_down(rmax, off, 64));
    cmin = min(cmin, __shfl_down(cmin, off, 64)); cmax = max(cmax, __shfl_down(cmax, off, 64));
  }
  if (lane == 0) {
    atomicMin(&s_ext[0], rmin); atomicMax(&s_ext[1], rmax);
    atomicMin(&s_ext[2], cmin); atomicMax(&s_ext[3], cmax);
  }

  // ---- exclusive scan of the per-thread counts ----------------------------------------------------
  int incl = cnt;
  for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
  if (lane == 63) s_wtot[wave] = incl;
  __syncthreads();
  int wbase = 0, ntaps = 0;
  for (int k = 0; k < NWAVE; ++k) { if (k < wave) wbase += s_wtot[k]; ntaps += s_wtot[k]; }
  int pos = wbase + incl - cnt;

  // ---- header, CSR row pointers, taps -----------------------------------------------------------------
  constexpr int TPR = K / EPT;  // threads per PSF row
  if ((tid % TPR) == 0) tab[table_rowptr_off() + tid / TPR] = pos;
  if (tid == 0) {
    tab[table_rowptr_off() + K] = ntaps;
    tab[HDR_NTAPS] = ntaps;
    tab[HDR_RMIN] = s_ext[0]; tab[HDR_RMAX] = s_ext[1]; tab[HDR_CMIN] = s_ext[2]; tab[HDR_CMAX] = s_ext[3];
    tab[HDR_K] = K; tab[HDR_SUM] = (int)E::bits(total);
  }
  uint2 *taps = reinterpret_cast<uint2 *>(tab + table_taps_off(K));
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    if (mask & (1ull << i)) {
      const unsigned rc = (unsigned)(((e0 + i) >> LK) << 8 | ((e0 + i) & (K - 1)));
      taps[pos] = make_uint2(rc, E::bits(v[i]));
      if (pos < STAGE_TAPS) { s_rc[pos] = (unsigned short)rc; s_w16[pos] = (unsigned short)E::bits(v[i]); }
      ++pos;
    }
  }
  __threadfence_block();
  __syncthreads();
  if (wave != 0) return;
  if (normalize & 4) return;  // diagnostics: skip the segmentation

  // ---- segmentation (wave 0): greedy runs with row span <= SEG_ROWS and column span <= SEG_COLS --------
  uint4 *segs = reinterpret_cast<uint4 *>(tab + table_segs_off(K));
  unsigned *ltaps = reinterpret_cast<unsigned *>(tab + table_ltaps_off(K));
  // per-tap LDS offsets of a closed segment [s0, s1) with last row rl and last column cmx
  auto emit_ltaps = [&](int s0, int s1, int rl, int cmx) {
    for (int j = s0 + lane; j < s1; j += 64) {
      unsigned rcj, wj;
      if (j < STAGE_TAPS) { rcj = s_rc[j]; wj = s_w16[j]; }
      else { const uint2 tp = taps[j]; rcj = tp.x & 0xffffu; wj = tp.y & 0xffffu; }
      const int rj = rcj >> 8, cj = rcj & 255;
      ltaps[j] = (unsigned)(((rl - rj) * WIN_PITCH + (cmx - cj)) * 8) | (wj << 16);
    }
  };
  int nseg = 0, seg_start = 0, seg_r0 = 0, seg_rlast = 0, car_cmin = 1 << 20, car_cmax = -1;
  for (int base = 0; base < ntaps; base += 64) {
    const int i = base + lane;
    const bool valid = i < ntaps;
    unsigned rc = 0;
    if (valid) rc = (i < STAGE_TAPS) ? s_rc[i] : (taps[i].x & 0xffffu);
    const int r = rc >> 8, c = rc & 255;
    if (base == 0) seg_r0 = __shfl(r, 0, 64);
    int lo = 0;
    while (true) {
      // inclusive prefix min / max of c over lanes [lo, lane], joined with the open segment's carry
      int pm = (valid && lane >= lo) ? c : (1 << 20), px = (valid && lane >= lo) ? c : -1;
      for (int off = 1; off < 64; off <<= 1) {
        int a = __shfl_up(pm, off, 64), b = __shfl_up(px, off, 64);
        if (lane >= off) { pm = min(pm, a); px = max(px, b); }
      }
      pm = min(pm, car_cmin); px = max(px, car_cmax);
      const bool bad = valid && lane >= lo && ((r - seg_r0 > SEG_ROWS) || (px - pm > SEG_COLS));
      const unsigned long long fail = __ballot(bad);
      const unsigned long long vmask = __ballot(valid);
      const int last_valid = 63 - __clzll((long long)vmask);  // vmask != 0 inside the loop
      if (fail == 0) {
        car_cmin = __shfl(pm, last_valid, 64); car_cmax = __shfl(px, last_valid, 64);
        seg_rlast = __shfl(r, last_valid, 64);
        break;
      }
      const int f = __ffsll((long long)fail) - 1;  // tap base+f opens a new segment
      int cmn = car_cmin, cmx = car_cmax, rl = seg_rlast;
      if (f > lo) { cmn = __shfl(pm, f - 1, 64); cmx = __shfl(px, f - 1, 64); rl = __shfl(r, f - 1, 64); }
      if (lane == 0) segs[nseg] = make_uint4(seg_start, base + f, (seg_r0 << 8) | rl, (cmn << 8) | cmx);
      emit_ltaps(seg_start, base + f, rl, cmx);
      ++nseg;
      seg_start = base + f;
      seg_r0 = __shfl(r, f, 64);
      seg_rlast = seg_r0;
      car_cmin = 1 << 20; car_cmax = -1;
      lo = f;
    }
  }
  if (ntaps > 0) {
    if (lane == 0) segs[nseg] = make_uint4(seg_start, ntaps, (seg_r0 << 8) | seg_rlast, (car_cmin << 8) | car_cmax);
    emit_ltaps(seg_start, ntaps, seg_rlast, car_cmax);
    ++nseg;
  }
  if (lane < 8) ltaps[ntaps + lane] = 0;  // the blur's scalar prefetch runs up to two taps past the end
  if (lane == 0) tab[HDR_NSEGS] = nseg;
}

}  // namespace dib

extern "C" size_t dib_tap_table_bytes(int K) {
  if (K != 128 && K != 256) return 0;
  return (size_t)dib::table_words(K) * sizeof(int);
}

extern "C" size_t dib_tap_tables_bytes(int K, int B) {
  if ((K != 128 && K != 256) || B < 0) return 0;
  return ((size_t)dib::table_words(K) * B + dib::SCHED_WORDS) * sizeof(int);
}

static int launch_compact(const void *const *ptrs, int dtype, int B, int K, int normalize, int *tables, hipStream_t s) {
  const size_t stride = (size_t)dib::table_words(K);
  for (int b0 = 0; b0 < B; b0 += dib::MAX_BATCH) {
    dib::PsfPtrs pp;
    const int n = B - b0 < dib::MAX_BATCH ? B - b0 : dib::MAX_BATCH;
    for (int i = 0; i < n; ++i) {
      if (!ptrs[b0 + i] || ((uintptr_t)ptrs[b0 + i] & 15) != 0) { dib::set_error("dib_psf_compact: PSF %d is null or not 16-byte aligned", b0 + i); return DIB_EINVAL; }
      pp.p[i] = ptrs[b0 + i];
    }
    int *t = tables + (size_t)b0 * stride;
    // bit 3: this launch owns the scheduler trailer behind the LAST table (only the final chunk does)
    const int flags = (normalize ? 1 : 0) | ((b0 + n == B) ? 8 : 0);
    if (dtype == DIB_F16 && K == 128) hipLaunchKernelGGL((dib::psf_compact_kernel<__half, 128>), dim3(n), dim3(dib::CT), 0, s, pp, flags, t);
    else if (dtype == DIB_F16) hipLaunchKernelGGL((dib::psf_compact_kernel<__half, 256>), dim3(n), dim3(dib::CT), 0, s, pp, flags, t);
    else if (K == 128) hipLaunchKernelGGL((dib::psf_compact_kernel<float, 128>), dim3(n), dim3(dib::CT), 0, s, pp, flags, t);
    else hipLaunchKernelGGL((dib::psf_compact_kernel<float, 256>), dim3(n), dim3(dib::CT), 0, s, pp, flags, t);
  }
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
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
