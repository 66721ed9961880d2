// Sparse PSF (x) image correlation for gfx950 (CDNA4) -- the `--gpu_blur` hot loop.
//
// Reference: models/blur_functions.py:11-69 (`manual_blur`): pad, then for EVERY non-zero tap a
// full-image torch.roll + mul + add_ (about 10 launches, 2 host syncs and 73 MB of HBM traffic per
// tap).  Here: one launch per batch, every source pixel fetched from HBM once (plus halo re-reads
// served by L2), the padded image never materialised (index math), taps broadcast from SGPRs.
//
// Arithmetic contract (DIB_ACC_BITEXACT): per output element, taps in row-major order,
//     acc = rnd16(acc + rnd16(P * w))            (v_pk_mul_f16 + v_pk_add_f16, never an FMA)
// which is bit-identical to the reference's Half tensors.
//
// Tiled fp16 kernel -- memory layout in LDS ("split-column" layout):
//   A workgroup (4 waves) owns a 256-wide output tile of ONE channel.  Lane l of every wave owns
//   the four columns x0 + l + 64k (k = 0..3), packed as two fp16x2 registers per row.  LDS row q
//   holds the source window row as 8-byte words:  word j = { P[j], P[j+64], P[j+128], P[j+192] },
//   j in [0, 64+ex), where P is the (virtually padded) source row starting at column
//   x0 + pb - cmax and ex = cmax - cmin is the PSF's column extent.  A tap (r, c) is then ONE
//   aligned, bank-conflict-free ds_read_b64 at word  lane + (cmax - c)  for ANY column shift --
//   odd shifts included, which a plain row-major fp16 layout cannot do with aligned packed reads.
//   Row shifts are LDS row offsets; the R rows a lane owns use compile-time immediate offsets, so
//   a tap costs one v_add (address) + R x { ds_read_b64, 2 v_pk_mul_f16, 2 v_pk_add_f16 }.
//   PSF rows are processed in bands of at most (LDS rows - tile rows) so any row extent fits.
//
// Roofline: HBM-bound by design (12.8 MB algorithmic bytes per 3x800x1333 image); per tap-pixel
// the kernel spends 1/4 LDS read + 1 packed VALU op, which balances HBM time at ~35-50 taps.
#include "dib_common.h"
#include <hip/hip_fp16.h>

namespace dib {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int TILE_W = 256;
constexpr int LDS_BYTES = 96 * 96 * 8;  // 73,728 B: two workgroups per CU

// Class A: column extent <= 32 (low exposure, the training default): 64-row tiles.
// Class B: column extent <= 128 (everything a 128-wide PSF can hold): 32-row tiles.
template <int PQ_, int R_, int LROWS_> struct TileCfg {
  static constexpr int PQ = PQ_;        // LDS row pitch in 8-byte words (>= 64 + ex)
  static constexpr int R = R_;          // rows per lane
  static constexpr int TH = 4 * R_;     // tile rows (4 waves)
  static constexpr int LROWS = LROWS_;  // LDS rows; band extent = LROWS - TH
  static constexpr int U = (PQ_ + 63) / 64;  // 8-byte words a lane fills per LDS row
  static_assert(PQ_ * LROWS_ * 8 <= LDS_BYTES, "LDS budget");
};
using CfgA = TileCfg<96, 16, 96>;
using CfgB = TileCfg<192, 8, 48>;

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
// 64-bit asm operands must be scalar integers: hipcc (ROCm 7.2) aliases both lanes of a
// 2 x 32-bit vector operand of an inline-asm "=v" output to the same register.
typedef unsigned long long u2;

// ---- LDS reads of the tap loop, hand-issued -------------------------------------------------
// hipcc merges neighbouring 8-byte LDS reads into ds_read2_b64 (half the LDS rate of ds_read_b64
// on gfx950, MI355X_MICROARCH.md LDS table) and waits right behind each one.  The reads are
// therefore issued as plain ds_read_b64 with immediate row offsets from inline asm, one tap
// ahead of the arithmetic, and waited for with an explicit lgkmcnt(0) that carries the
// destination registers as in/out operands so no consumer can be scheduled above it.
template <int OFF> __device__ __forceinline__ void lds_rd64(u2 &dst, unsigned addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int PITCH_BYTES, int R, int I = 0> __device__ __forceinline__ void lds_rd_rows(u2 (&buf)[R], unsigned addr) {
  if constexpr (I < R) {
    lds_rd64<I * PITCH_BYTES>(buf[I], addr);
    lds_rd_rows<PITCH_BYTES, R, I + 1>(buf, addr);
  }
}
template <int R> __device__ __forceinline__ void lds_wait(u2 (&b)[R]) {
  static_assert(R == 8 || R == 16, "R");
  if constexpr (R == 16) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    asm volatile("" : "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15]));
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
  }
}

template <typename Cfg, bool ZERO>
__device__ __forceinline__ void blur_tile_f16(const ImageDesc &d, const int *__restrict__ tab, int K, int ch,
                                              int tx, int ty, uint2 *lds) {
#pragma clang fp contract(off)
  constexpr int PQ = Cfg::PQ, R = Cfg::R, TH = Cfg::TH, BAND = Cfg::LROWS - Cfg::TH, U = Cfg::U;
  constexpr int G = 4;  // LDS rows a wave fills per batch of loads (8*G loads in flight per lane)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = d.H, W = d.W;
  const int pb = K / 2 - 1, pa = K / 2;
  const int mode = ZERO ? PAD_ZERO : (K > 129 ? PAD_REPLICATE : PAD_REFLECT);
  const int rmin = tab[HDR_RMIN], rmax = tab[HDR_RMAX], cmin = tab[HDR_CMIN], cmax = tab[HDR_CMAX];
  const int *rowptr = tab + table_rowptr_off();
  const uint2 *taps = reinterpret_cast<const uint2 *>(tab + table_taps_off(K));
  const int x0 = tx * TILE_W, y0 = ty * TH;
  const char *src = reinterpret_cast<const char *>(d.in) + (size_t)ch * H * W * 2;
  __half *dst = reinterpret_cast<__half *>(d.out) + (size_t)ch * H * W;

  const int pqused = 64 + (cmax - cmin);  // words per LDS row actually used
  // Byte offsets (within a source row) of the U words this lane fills per LDS row: once per tile.
  unsigned coff[U][4];
  unsigned cmask[U][2];  // zero-padding masks per packed pair (ZERO only)
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bool z;
      coff[u][k] = 2u * (unsigned)map_coord(x0 + pb - cmax + lane + 64 * u + 64 * k, W, pa, pb, mode, z);
      unsigned m = z ? 0u : 0xffffu;
      if (k & 1) cmask[u][k >> 1] |= m << 16; else cmask[u][k >> 1] = m;
    }
  }

  h2 acc[R][2];
#pragma unroll
  for (int i = 0; i < R; ++i) { acc[i][0] = h2{0, 0}; acc[i][1] = h2{0, 0}; }

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
  const unsigned lane_addr = lds0 + (unsigned)((wave * R) * PQ + lane) * 8u;

  for (int rb = rmin; rb <= rmax;) {
    const int re = min(rmax, rb + BAND);  // band of PSF rows [rb, re]
    const int t0 = rowptr[rb], t1 = rowptr[re + 1];
    if (t0 == t1) { rb = re + 1; continue; }  // (uniform) empty band
    const int nrows = TH + (re - rb);
    const int s_top = y0 + pb - re;  // virtual source row held in LDS row 0

    // ---- fill: each wave takes G consecutive LDS rows per pass; row maps are scalar ------------
    for (int qb = wave * G; qb < nrows; qb += 4 * G) {
      us2 v[G][U][2];
      bool zrow[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        int sr = map_coord(s_top + min(qb + g, nrows - 1), H, pa, pb, mode, zrow[g]);
        const char *row = src + (size_t)sr * W * 2;
        // loads are unconditional (offsets are clamped in-bounds); only the LDS write is predicated,
        // so no wait lands inside a divergent block
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            v[g][u][k >> 1][k & 1] = *reinterpret_cast<const unsigned short *>(row + coff[u][k]);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (qb + g < nrows) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (lane + 64 * u < pqused) {
              unsigned lo = __builtin_bit_cast(unsigned, v[g][u][0]), hi = __builtin_bit_cast(unsigned, v[g][u][1]);
              if (ZERO) {
                lo &= cmask[u][0]; hi &= cmask[u][1];
                if (zrow[g]) { lo = 0; hi = 0; }
              }
              lds[(qb + g) * PQ + lane + 64 * u] = make_uint2(lo, hi);
            }
          }
        }
      }
    }
    __syncthreads();

    // ---- accumulate: taps t0..t1 in row-major order, scalar-broadcast, LDS reads one tap ahead ----
    {
      u2 bx[R], by[R];
      auto issue = [&](u2(&buf)[R], uint2 tp) {
        const int r = tp.x >> 8, c = tp.x & 255;
        lds_rd_rows<PQ * 8, R>(buf, lane_addr + (unsigned)(((re - r) * PQ + (cmax - c)) * 8));
      };
      auto madd = [&](u2(&buf)[R], uint2 tp) {
        const unsigned wb = tp.y & 0xffffu;
        const h2 w = __builtin_bit_cast(h2, wb | (wb << 16));
        // all products first, then all sums: independent packed ops back to back (no RAW stalls)
        h2 p0[R], p1[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
          p0[i] = __builtin_bit_cast(h2, (unsigned)buf[i]) * w;
          p1[i] = __builtin_bit_cast(h2, (unsigned)(buf[i] >> 32)) * w;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
          acc[i][0] = acc[i][0] + p0[i];
          acc[i][1] = acc[i][1] + p1[i];
        }
      };
      int t = t0;
      uint2 ta = taps[t], tb = ta;
      issue(bx, ta);
      while (true) {
        const bool has_b = t + 1 < t1;
        if (has_b) tb = taps[t + 1];
        lds_wait<R>(bx);
        if (has_b) issue(by, tb);
        madd(bx, ta);
        if (!has_b) break;
        const bool has_a = t + 2 < t1;
        if (has_a) ta = taps[t + 2];
        lds_wait<R>(by);
        if (has_a) issue(bx, ta);
        madd(by, tb);
        if (!has_a) break;
        t += 2;
      }
    }
    __syncthreads();
    rb = re + 1;
  }

  // ---- store: lane owns columns x0 + lane + 64k ------------------------------------------------
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int y = y0 + wave * R + i;
    if (y < H) {
      unsigned short *orow = reinterpret_cast<unsigned short *>(dst) + (size_t)y * W + x0 + lane;
      const int xr = W - x0 - lane;  // columns remaining
      // halves are extracted with integer ops: hipcc (ROCm 7.2) stored the LOW half twice when the
      // high element of the fp16x2 accumulator was taken with a vector subscript
      const unsigned a = __builtin_bit_cast(unsigned, acc[i][0]), b = __builtin_bit_cast(unsigned, acc[i][1]);
      if (xr > 0) orow[0] = (unsigned short)(a & 0xffffu);
      if (xr > 64) orow[64] = (unsigned short)(a >> 16);
      if (xr > 128) orow[128] = (unsigned short)(b & 0xffffu);
      if (xr > 192) orow[192] = (unsigned short)(b >> 16);
    }
  }
}

// Grid: one workgroup per (image, channel, 256x32 tile slot).  Images whose PSF falls in the
// 64-row class use every other slot; the others exit at once.  Images whose column extent exceeds
// 128 (only possible with a 256-wide PSF) are left to blur_generic_kernel.
__global__ __launch_bounds__(256, 2) void blur_tiled_f16_kernel(BlurBatch batch, const int *__restrict__ tables,
                                                                int K) {
  extern __shared__ uint2 lds[];
  int b = blockIdx.x, i = 0;
  while (i + 1 < batch.n && b >= batch.img[i + 1].tile_begin) ++i;
  const ImageDesc &d = batch.img[i];
  int local = b - d.tile_begin;
  const int per_ch = d.tiles_x * d.tiles_y32;
  const int ch = local / per_ch;
  local -= ch * per_ch;
  const int ty = local / d.tiles_x, tx = local - ty * d.tiles_x;
  const int *tab = tables + (size_t)d.table * table_words(K);
  const int ex = tab[HDR_CMAX] - tab[HDR_CMIN];
  // ntaps == 0 gives ex < 0: class A with an empty tap loop writes the zeros the reference returns
  const bool zero = pad_mode_for(K, d.H, d.W) == PAD_ZERO;
  if (ex <= CfgA::PQ - 64) {
    if (ty * CfgA::TH >= d.H) return;  // unused slot of the 64-row class
    if (zero) blur_tile_f16<CfgA, true>(d, tab, K, ch, tx, ty, lds);
    else blur_tile_f16<CfgA, false>(d, tab, K, ch, tx, ty, lds);
  } else if (ex <= CfgB::PQ - 64) {
    if (zero) blur_tile_f16<CfgB, true>(d, tab, K, ch, tx, ty, lds);
    else blur_tile_f16<CfgB, false>(d, tab, K, ch, tx, ty, lds);
  }
}

// ---------------------------------------------------------------------------------------------
// Generic kernel: any K, any extent, fp16 or fp32, straight from global memory.  One thread per
// output element.  Used for fp32 images, for 256-wide PSFs whose column extent exceeds 128, and
// as an independent second implementation in the parity tests.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Arith;
template <> struct Arith<__half> {
  using V = _Float16;
  static __device__ V weight(unsigned bits) { return __builtin_bit_cast(_Float16, (unsigned short)(bits & 0xffff)); }
};
template <> struct Arith<float> {
  using V = float;
  static __device__ V weight(unsigned bits) { return __uint_as_float(bits); }
};

template <typename T, bool ONLY_WIDE>
__global__ __launch_bounds__(256) void blur_generic_kernel(BlurBatch batch, const int *__restrict__ tables, int K) {
#pragma clang fp contract(off)
  using V = typename Arith<T>::V;
  int b = blockIdx.x, i = 0;
  while (i + 1 < batch.n && b >= batch.img[i + 1].tile_begin) ++i;
  const ImageDesc &d = batch.img[i];
  const int *tab = tables + (size_t)d.table * table_words(K);
  if (ONLY_WIDE && tab[HDR_CMAX] - tab[HDR_CMIN] <= CfgB::PQ - 64) return;
  const int H = d.H, W = d.W, pb = K / 2 - 1, pa = K / 2, mode = pad_mode_for(K, H, W);
  const long long n = (long long)d.C * H * W;
  const long long e = (long long)(b - d.tile_begin) * 256 + threadIdx.x;
  if (e >= n) return;
  const int x = (int)(e % W), y = (int)((e / W) % H), ch = (int)(e / ((long long)W * H));
  const V *src = reinterpret_cast<const V *>(d.in) + (size_t)ch * H * W;
  const int ntaps = tab[HDR_NTAPS];
  const uint2 *taps = reinterpret_cast<const uint2 *>(tab + table_taps_off(K));
  V acc = 0;
  for (int t = 0; t < ntaps; ++t) {
    const uint2 tap = taps[t];
    const int r = tap.x >> 8, c = tap.x & 255;
    bool zr, zc;
    const int sy = map_coord(y + pb - r, H, pa, pb, mode, zr);
    const int sx = map_coord(x + pb - c, W, pa, pb, mode, zc);
    V p = (zr || zc) ? V(0) : src[(size_t)sy * W + sx];
    V prod = p * Arith<T>::weight(tap.y);
    acc = acc + prod;
  }
  reinterpret_cast<V *>(d.out)[e] = acc;
}

}  // namespace dib

using namespace dib;

extern "C" int dib_sparse_blur(const void *const *in_dev, void *const *out_dev, const int *C, const int *H,
                               const int *W, const int *table_index, int B, int dtype, const void *tables_dev,
                               int K, int acc_mode, void *stream) {
  if (B < 0 || (B > 0 && (!in_dev || !out_dev || !C || !H || !W || !table_index || !tables_dev))) {
    set_error("dib_sparse_blur: null pointer or negative batch");
    return DIB_EINVAL;
  }
  if (K != 128 && K != 256) { set_error("dib_sparse_blur: K must be 128 or 256, got %d", K); return DIB_EINVAL; }
  if (dtype != DIB_F16 && dtype != DIB_F32) { set_error("dib_sparse_blur: unknown dtype %d", dtype); return DIB_EINVAL; }
  if (acc_mode != DIB_ACC_BITEXACT) { set_error("dib_sparse_blur: accumulation mode %d not available", acc_mode); return DIB_EINVAL; }
  for (int i = 0; i < B; ++i) {
    if (table_index[i] < 0) continue;
    if (!in_dev[i] || !out_dev[i] || C[i] <= 0 || H[i] <= 0 || W[i] <= 0) {
      set_error("dib_sparse_blur: image %d has a null pointer or empty shape", i);
      return DIB_EINVAL;
    }
    if (in_dev[i] == out_dev[i]) { set_error("dib_sparse_blur: image %d: out aliases in", i); return DIB_EINVAL; }
    // F.pad(mode='reflect') raises unless pad < dim (blur_functions.py:59 with pads 63/64)
    if (K == 128 && !(H[i] < 64 || W[i] < 64) && (H[i] == 64 || W[i] == 64)) {
      set_error("Padding size should be less than the corresponding input dimension (image %d is %dx%d)", i, H[i], W[i]);
      return DIB_ESHAPE;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    DIB_HIP_CHECK(hipFuncSetAttribute((const void *)blur_tiled_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set = true;
  }
  int i = 0;
  while (i < B) {
    BlurBatch tiled, generic;
    tiled.n = generic.n = 0;
    int tiles = 0, gblocks = 0;
    for (; i < B && tiled.n < MAX_BATCH; ++i) {
      if (table_index[i] < 0) continue;
      ImageDesc d;
      d.in = in_dev[i]; d.out = out_dev[i]; d.C = C[i]; d.H = H[i]; d.W = W[i]; d.table = table_index[i];
      d.tiles_x = (W[i] + TILE_W - 1) / TILE_W;
      d.tiles_y32 = (H[i] + 31) / 32;
      d.tile_begin = tiles;
      tiles += d.C * d.tiles_x * d.tiles_y32;
      tiled.img[tiled.n++] = d;
      long long n = (long long)C[i] * H[i] * W[i];
      d.tile_begin = gblocks;
      gblocks += (int)((n + 255) / 256);
      generic.img[generic.n++] = d;
    }
    if (tiled.n == 0) break;
    tiled.total_tiles = tiles;
    generic.total_tiles = gblocks;
    if (dtype == DIB_F16) {
      hipLaunchKernelGGL(blur_tiled_f16_kernel, dim3(tiles), dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K);
      if (K == 256)  // column extents beyond 128 cannot be tiled; those images take the generic path
        hipLaunchKernelGGL((blur_generic_kernel<__half, true>), dim3(gblocks), dim3(256), 0, s, generic, (const int *)tables_dev, K);
    } else {
      hipLaunchKernelGGL((blur_generic_kernel<float, false>), dim3(gblocks), dim3(256), 0, s, generic, (const int *)tables_dev, K);
    }
    DIB_HIP_CHECK(hipGetLastError());
  }
  return DIB_OK;
}

// Test hook (not part of the drop-in boundary): runs the generic kernel on fp16 images so the
// parity tests can compare two independent device implementations.
extern "C" int dib_sparse_blur_generic(const void *in_dev, void *out_dev, int C, int H, int W, int dtype,
                                       const void *table_dev, int K, void *stream) {
  if (!in_dev || !out_dev || !table_dev) { set_error("dib_sparse_blur_generic: null pointer"); return DIB_EINVAL; }
  BlurBatch g;
  g.n = 1;
  ImageDesc d;
  d.in = in_dev; d.out = out_dev; d.C = C; d.H = H; d.W = W; d.table = 0; d.tile_begin = 0; d.tiles_x = d.tiles_y32 = 0;
  g.img[0] = d;
  int blocks = (int)(((long long)C * H * W + 255) / 256);
  g.total_tiles = blocks;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DIB_F16)
    hipLaunchKernelGGL((blur_generic_kernel<__half, false>), dim3(blocks), dim3(256), 0, s, g, (const int *)table_dev, K);
  else
    hipLaunchKernelGGL((blur_generic_kernel<float, false>), dim3(blocks), dim3(256), 0, s, g, (const int *)table_dev, K);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
