// Tap compaction for gfx950: one 256-thread workgroup per K x K PSF.
//
// Replaces, for a whole batch and without any host synchronisation, the reference's
//   psf_GPU = psf_GPU / psf_GPU.sum();  non_zero_points = psf_GPU.nonzero()
// (models/blur_functions.py:98,63 and again utils.py:372-374) plus the min/max of the tap
// coordinates that expand_targets needs (utils.py:376-380).
//
// fp16 semantics: the sum is formed EXACTLY (every finite fp16 is a multiple of 2^-24, so the
// total is an int64 in those units) and rounded once to fp16 (round-to-nearest-even); the
// division is an IEEE fp32 divide rounded to fp16, which equals a correctly rounded fp16 divide
// (24 >= 2*11+2).  Order of the output taps = row-major, the order of torch.nonzero.
#include "dib_common.h"
#include <hip/hip_fp16.h>

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

template <typename T>
__global__ __launch_bounds__(256) void psf_compact_kernel(const T *__restrict__ psf, int K, int normalize,
                                                          int *__restrict__ tables) {
  using E = Elem<T>;
  __shared__ typename E::Acc s_part[4];
  __shared__ int s_cnt[1024 + 1];  // per 64-element chunk (K*K/64 <= 1024)
  __shared__ int s_ext[4];         // rmin rmax cmin cmax
  __shared__ T s_sum;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = K * K, nchunks = n / 64, lk = (K == 128) ? 7 : 8;
  const T *p = psf + (size_t)blockIdx.x * n;
  int *tab = tables + (size_t)blockIdx.x * table_words(K);

  // ---- pass 1: sum -------------------------------------------------------------------
  T total = T(1.0f);
  if (normalize) {
    typename E::Acc acc = 0;
    for (int i = tid; i < n; i += 256) acc += E::lift(p[i]);
    acc = wave_sum(acc);
    if (lane == 0) s_part[wave] = acc;
    __syncthreads();
    if (tid == 0) s_sum = E::finish(((s_part[0] + s_part[1]) + s_part[2]) + s_part[3]);
    __syncthreads();
    total = s_sum;
  }
  if (tid < 4) s_ext[tid] = (tid & 1) ? -1 : K;
  __syncthreads();

  // ---- pass 2: per-chunk non-zero counts + extents -------------------------------------
  int rmin = K, rmax = -1, cmin = K, cmax = -1;
  for (int ch = wave; ch < nchunks; ch += 4) {
    int i = ch * 64 + lane;
    T w = normalize ? E::div(p[i], total) : p[i];
    bool nz = E::nonzero(w);
    unsigned long long m = __ballot(nz);
    if (lane == 0) s_cnt[ch] = __popcll(m);
    if (nz) {
      int r = i >> lk, c = i & (K - 1);
      rmin = min(rmin, r); rmax = max(rmax, r); cmin = min(cmin, c); cmax = max(cmax, c);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    rmin = min(rmin, __shfl_down(rmin, off, 64)); rmax = max(rmax, __shfl_down(rmax, off, 64));
    cmin = min(cmin, __shfl_down(cmin, off, 64)); cmax = max(cmax, __shfl_down(cmax, off, 64));
  }
  if (lane == 0) {
    atomicMin(&s_ext[0], rmin); atomicMax(&s_ext[1], rmax);
    atomicMin(&s_ext[2], cmin); atomicMax(&s_ext[3], cmax);
  }
  __syncthreads();

  // ---- exclusive scan of the chunk counts (<= 1024 entries, 4 per thread) ---------------
  {
    int base = tid * 4, v[4], s = 0;
    for (int k = 0; k < 4; ++k) { v[k] = (base + k < nchunks) ? s_cnt[base + k] : 0; s += v[k]; }
    // inclusive scan of s across 256 threads: wave scan + cross-wave fix-up
    int incl = s;
    for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    __shared__ int s_wtot[4];
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int k = 0; k < wave; ++k) wbase += s_wtot[k];
    int excl = wbase + incl - s;
    __syncthreads();
    for (int k = 0; k < 4; ++k) { if (base + k < nchunks) s_cnt[base + k] = excl; excl += v[k]; }
    if (tid == 255) s_cnt[nchunks] = excl;   // total (thread 255 owns the last chunks)
    __syncthreads();
  }
  const int ntaps = s_cnt[nchunks];

  // ---- header + CSR row pointers ---------------------------------------------------------
  if (tid == 0) {
    tab[HDR_NTAPS] = ntaps;
    tab[HDR_RMIN] = s_ext[0]; tab[HDR_RMAX] = s_ext[1]; tab[HDR_CMIN] = s_ext[2]; tab[HDR_CMAX] = s_ext[3];
    tab[HDR_K] = K; tab[HDR_SUM] = (int)E::bits(total); tab[HDR_FLAGS] = 0;
  }
  const int cpr = K / 64;  // chunks per PSF row
  for (int r = tid; r <= K; r += 256) tab[table_rowptr_off() + r] = s_cnt[min(r * cpr, nchunks)];

  // ---- pass 3: ordered write ----------------------------------------------------------------
  uint2 *taps = reinterpret_cast<uint2 *>(tab + table_taps_off(K));
  for (int ch = wave; ch < nchunks; ch += 4) {
    int i = ch * 64 + lane;
    T w = normalize ? E::div(p[i], total) : p[i];
    bool nz = E::nonzero(w);
    unsigned long long m = __ballot(nz);
    if (nz) {
      int pos = s_cnt[ch] + __popcll(m & ((1ull << lane) - 1));
      int r = i >> lk, c = i & (K - 1);
      taps[pos] = make_uint2((unsigned)(r << 8 | c), E::bits(w));
    }
  }
}

}  // namespace dib

extern "C" size_t dib_tap_table_bytes(int K) {
  if (K != 128 && K != 256) return 0;
  return (size_t)dib::table_words(K) * sizeof(int);
}

extern "C" int dib_psf_compact(const void *psf_dev, int dtype, int B, int K, int normalize, void *tables_dev,
                               void *stream) {
  if (!psf_dev || !tables_dev || B < 0) { dib::set_error("dib_psf_compact: null pointer or negative batch"); return DIB_EINVAL; }
  if (K != 128 && K != 256) { dib::set_error("dib_psf_compact: K must be 128 or 256, got %d", K); return DIB_EINVAL; }
  if (B == 0) return DIB_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DIB_F16)
    hipLaunchKernelGGL(dib::psf_compact_kernel<__half>, dim3(B), dim3(256), 0, s, (const __half *)psf_dev, K, normalize, (int *)tables_dev);
  else if (dtype == DIB_F32)
    hipLaunchKernelGGL(dib::psf_compact_kernel<float>, dim3(B), dim3(256), 0, s, (const float *)psf_dev, K, normalize, (int *)tables_dev);
  else { dib::set_error("dib_psf_compact: unknown dtype %d", dtype); return DIB_EINVAL; }
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
