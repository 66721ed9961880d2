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
#include <hip/hip_ext.h>
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
  static __device__ __half from_bits(unsigned b) { return __ushort_as_half((unsigned short)b); }
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
  static __device__ float from_bits(unsigned b) { return __uint_as_float(b); }
};

template <typename A> __device__ inline A wave_sum(A v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---- wave64 reductions / scan on the DPP path -------------------------------------------------------
// __shfl_* compile to ds_bpermute_b32: ~100 cycles of latency each, and a 6-step reduction is a chain
// of them (x2 for 64-bit values).  Row-level DPP moves run at VALU rate; the four row results are
// collected with v_readlane.  dpp0: lanes without a source read 0; dppk: they keep `keep`.
template <int CTRL> __device__ inline int dpp0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ inline int dppk(int v, int keep) { return __builtin_amdgcn_update_dpp(keep, v, CTRL, 0xf, 0xf, false); }
constexpr int QUAD_SWAP1 = 0xB1, QUAD_SWAP2 = 0x4E, ROW_SHR1 = 0x111, ROW_SHR2 = 0x112, ROW_SHR4 = 0x114, ROW_SHR8 = 0x118;

__device__ inline int wave_sum_i32(int x) {   // wave-uniform result
  x += dpp0<QUAD_SWAP1>(x); x += dpp0<QUAD_SWAP2>(x);   // every lane: its quad's sum
  x += dpp0<ROW_SHR4>(x); x += dpp0<ROW_SHR8>(x);       // lanes 12..15 of a row: the row's sum
  return __builtin_amdgcn_readlane(x, 15) + __builtin_amdgcn_readlane(x, 31) + __builtin_amdgcn_readlane(x, 47) +
         __builtin_amdgcn_readlane(x, 63);
}
__device__ inline int wave_min_i32(int x) {
  x = min(x, dppk<QUAD_SWAP1>(x, x)); x = min(x, dppk<QUAD_SWAP2>(x, x));
  x = min(x, dppk<ROW_SHR4>(x, x)); x = min(x, dppk<ROW_SHR8>(x, x));
  return min(min(__builtin_amdgcn_readlane(x, 15), __builtin_amdgcn_readlane(x, 31)),
             min(__builtin_amdgcn_readlane(x, 47), __builtin_amdgcn_readlane(x, 63)));
}
__device__ inline int wave_max_i32(int x) {
  x = max(x, dppk<QUAD_SWAP1>(x, x)); x = max(x, dppk<QUAD_SWAP2>(x, x));
  x = max(x, dppk<ROW_SHR4>(x, x)); x = max(x, dppk<ROW_SHR8>(x, x));
  return max(max(__builtin_amdgcn_readlane(x, 15), __builtin_amdgcn_readlane(x, 31)),
             max(__builtin_amdgcn_readlane(x, 47), __builtin_amdgcn_readlane(x, 63)));
}
__device__ inline int wave_scan_incl_i32(int x, int lane) {   // inclusive prefix sum over the 64 lanes
  x += dpp0<ROW_SHR1>(x); x += dpp0<ROW_SHR2>(x); x += dpp0<ROW_SHR4>(x); x += dpp0<ROW_SHR8>(x);
  const int r0 = __builtin_amdgcn_readlane(x, 15), r1 = __builtin_amdgcn_readlane(x, 31), r2 = __builtin_amdgcn_readlane(x, 47);
  const int row = lane >> 4;
  return x + (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
}
__device__ inline int wave_scan_min_i32(int x, int lane) {   // inclusive prefix minimum (identity: INT_MAX)
  constexpr int ID = 0x7fffffff;
  x = min(x, dppk<ROW_SHR1>(x, ID)); x = min(x, dppk<ROW_SHR2>(x, ID)); x = min(x, dppk<ROW_SHR4>(x, ID)); x = min(x, dppk<ROW_SHR8>(x, ID));
  const int r0 = __builtin_amdgcn_readlane(x, 15), r1 = __builtin_amdgcn_readlane(x, 31), r2 = __builtin_amdgcn_readlane(x, 47);
  const int row = lane >> 4;
  return min(x, min(row > 0 ? r0 : ID, min(row > 1 ? r1 : ID, row > 2 ? r2 : ID)));
}
__device__ inline int wave_scan_max_i32(int x, int lane) {   // inclusive prefix maximum (identity: INT_MIN)
  constexpr int ID = (int)0x80000000;
  x = max(x, dppk<ROW_SHR1>(x, ID)); x = max(x, dppk<ROW_SHR2>(x, ID)); x = max(x, dppk<ROW_SHR4>(x, ID)); x = max(x, dppk<ROW_SHR8>(x, ID));
  const int r0 = __builtin_amdgcn_readlane(x, 15), r1 = __builtin_amdgcn_readlane(x, 31), r2 = __builtin_amdgcn_readlane(x, 47);
  const int row = lane >> 4;
  return max(x, max(row > 0 ? r0 : ID, max(row > 1 ? r1 : ID, row > 2 ? r2 : ID)));
}
// exact 64-bit sum as three limbs (21 + 21 + 22 bits): every limb's 64-lane sum fits 32 bits, and the
// recombination is arithmetic mod 2^64, so negative (two's complement) inputs come out right
__device__ inline long long wave_sum(long long v) {
  const unsigned long long u = (unsigned long long)v;
  const unsigned long long s0 = (unsigned)wave_sum_i32((int)(u & 0x1fffffu));
  const unsigned long long s1 = (unsigned)wave_sum_i32((int)((u >> 21) & 0x1fffffu));
  const unsigned long long s2 = (unsigned)wave_sum_i32((int)(u >> 42));
  return (long long)(s0 + (s1 << 21) + (s2 << 42));
}

// sum over the wave, same value in every lane
__device__ inline long long wave_total(long long v) { return wave_sum(v); }                 // limb path: already uniform
__device__ inline double wave_total(double v) { return __shfl(wave_sum(v), 0, 64); }         // shuffle path: lane 0 holds it

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
  __shared__ unsigned short s_rc[STAGE_TAPS];
  __shared__ unsigned s_wb[STAGE_TAPS];      // weight bits (raw before the division, final after)
  __shared__ int s_flag;

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

  // ---- segmentation (wave 0): greedy runs with row span <= SEG_ROWS and column span <= SEG_COLS --------
  uint4 *segs = reinterpret_cast<uint4 *>(tab + table_segs_off(K));
  unsigned *ltaps = reinterpret_cast<unsigned *>(tab + table_ltaps_off(K));
  unsigned *ltaps_q = reinterpret_cast<unsigned *>(tab + table_ltaps_q_off(K));
  // segment limits and the quad window's row pitch of the geometry this table is compacted for (dib_common.h)
  const bool large = (normalize & COMPACT_LARGE_WINDOW) != 0;
  const int seg_rows = large ? SEG_ROWS_L : SEG_ROWS, seg_cols = large ? SEG_COLS_L : SEG_COLS, qpitch = large ? QUAD_PITCH_L : QUAD_PITCH;
  // per-tap LDS offsets of a closed segment [s0, s1) with last row rl and last column cmx
  auto emit_ltaps = [&](int s0, int s1, int rl, int cmx) {
    for (int j = s0 + lane; j < s1; j += 64) {
      unsigned rcj, wj;
      if (j < STAGE_TAPS) { rcj = s_rc[j]; wj = s_wb[j] & 0xffffu; }
      else { const uint2 tp = taps[j]; rcj = tp.x & 0xffffu; wj = tp.y & 0xffffu; }
      const int rj = rcj >> 8, cj = rcj & 255;
      ltaps[j] = (unsigned)(((rl - rj) * WIN_PITCH + (cmx - cj)) * 8) | (wj << 16);
      ltaps_q[j] = (unsigned)(((rl - rj) * qpitch + (cmx - cj)) * 8) | (wj << 16);
    }
  };
  int nseg = 0, seg_start = 0, seg_r0 = 0, seg_rlast = 0, car_cmin = 1 << 20, car_cmax = -1;
  for (int base = 0; base < ntaps; base += 64) {
    const int i = base + lane;
    const bool valid = i < ntaps;
    unsigned rc = 0;
    if (valid) rc = (i < STAGE_TAPS) ? s_rc[i] : (taps[i].x & 0xffffu);
    const int r = rc >> 8, c = rc & 255;
    if (base == 0) seg_r0 = __builtin_amdgcn_readlane(r, 0);
    int lo = 0;
    while (true) {
      // inclusive prefix min / max of c over lanes [lo, lane], joined with the open segment's carry
      int pm = wave_scan_min_i32((valid && lane >= lo) ? c : (1 << 20), lane);
      int px = wave_scan_max_i32((valid && lane >= lo) ? c : -1, lane);
      pm = min(pm, car_cmin); px = max(px, car_cmax);
      const bool bad = valid && lane >= lo && ((r - seg_r0 > seg_rows) || (px - pm > seg_cols));
      const unsigned long long fail = __ballot(bad);
      const unsigned long long vmask = __ballot(valid);
      const int last_valid = 63 - __clzll((long long)vmask);  // vmask != 0 inside the loop
      if (fail == 0) {
        car_cmin = __builtin_amdgcn_readlane(pm, last_valid); car_cmax = __builtin_amdgcn_readlane(px, last_valid);
        seg_rlast = __builtin_amdgcn_readlane(r, last_valid);
        break;
      }
      const int f = __ffsll((long long)fail) - 1;  // tap base+f opens a new segment
      int cmn = car_cmin, cmx = car_cmax, rl = seg_rlast;
      if (f > lo) { cmn = __builtin_amdgcn_readlane(pm, f - 1); cmx = __builtin_amdgcn_readlane(px, f - 1); rl = __builtin_amdgcn_readlane(r, f - 1); }
      if (lane == 0) segs[nseg] = make_uint4(seg_start, base + f, (seg_r0 << 8) | rl, (cmn << 8) | cmx);
      emit_ltaps(seg_start, base + f, rl, cmx);
      ++nseg;
      seg_start = base + f;
      seg_r0 = __builtin_amdgcn_readlane(r, f);
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
  if (lane < 8) ltaps[ntaps + lane] = ltaps_q[ntaps + lane] = 0;  // the blur's scalar prefetch runs up to two taps past the end
  if (lane == 0) tab[HDR_NSEGS] = nseg;
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
    const int flags = ((normalize & ~DIB_COMPACT_LARGE_WINDOW) ? COMPACT_NORMALIZE : 0) | ((normalize & DIB_COMPACT_LARGE_WINDOW) ? COMPACT_LARGE_WINDOW : 0);
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
