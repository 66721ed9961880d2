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
//   A workgroup (4 waves) owns a 256 x 32 output tile of ONE channel.  Lane l of every wave owns
//   the four columns x0 + l + 64k (k = 0..3), packed as two fp16x2 registers per row.  LDS row q
//   holds the source window row as 8-byte words:  word j = { P[j], P[j+64], P[j+128], P[j+192] },
//   j in [0, 64+ex), where P is the (virtually padded) source row starting at column
//   x0 + pb - cmax and ex = cmax - cmin is the column extent of the tap SEGMENT being processed.
//   A tap (r, c) is then ONE aligned, bank-conflict-free ds_read_b64 at word  lane + (cmax - c)
//   for ANY column shift -- odd shifts included, which a plain row-major fp16 layout cannot do
//   with aligned packed reads.  Row shifts are LDS row offsets; the R rows a lane owns use
//   compile-time immediate offsets, so a tap costs one v_add (address) +
//   R x { ds_read_b64, 2 v_pk_mul_f16, 2 v_pk_add_f16 }.
//   The tap list arrives cut into segments (dib_compact.hip) whose bounding box is at most
//   17 PSF rows x 33 PSF columns, so ONE small LDS window (48 rows x 96 words = 36 KB, four
//   workgroups per CU) serves any PSF; a wide or tall PSF simply takes several fill+accumulate
//   rounds, in tap order, with the accumulators staying in registers.
//
// Roofline: HBM-bound by design (12.8 MB algorithmic bytes per 3x800x1333 image); per tap-pixel
// the kernel spends 1/4 LDS read + 1 packed VALU op, which balances HBM time at ~35-50 taps.
#include "dib_common.h"
#include <hip/hip_fp16.h>

namespace dib {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int TILE_W = 256;
constexpr int PQ = WIN_PITCH;        // LDS row pitch in 8-byte words (96)
constexpr int R = 8;                // rows per lane
constexpr int NW = 4;               // waves per workgroup
constexpr int TH = NW * R;          // tile rows (32)
constexpr int LROWS = TH + SEG_ROWS;  // LDS rows per window (48)
constexpr int WIN_WORDS = LROWS * PQ;
constexpr int LDS_BYTES = 2 * WIN_WORDS * 8;  // two windows (double buffer): 73,728 B, 2 workgroups per CU
constexpr int G = (LROWS + NW - 1) / NW;  // LDS rows a wave fills: all of them in ONE batch of loads (12)
constexpr int WG_PER_CU = 2;

// 64-bit asm operands must be scalar integers: hipcc (ROCm 7.2) aliases both lanes of a
// 2 x 32-bit vector operand of an inline-asm "=v" output to the same register.
typedef unsigned long long u2;

// ---- the tap loop's memory side, hand-issued ---------------------------------------------------
// Left to hipcc, the tap loop serialises: it merges the 8-byte LDS reads into ds_read2_b64 (half
// the LDS rate of ds_read_b64 on gfx950), sinks the scalar tap load down to its first use and waits
// right behind every access.  One inline-asm block per tap therefore does the whole memory side:
//     s_waitcnt lgkmcnt(0)        data of tap t (issued one block ago) and ltap t+1 have arrived
//     A = B ; B = C               rotate the scalar tap words inside the asm, after the wait
//     8 x ds_read_b64             data of tap t+1, immediate row offsets, into the OTHER buffer
//     s_load_dword C              ltap t+2
// and the 32 packed multiply/adds of tap t follow in C++ while those accesses are in flight.  The
// buffer of tap t and the tap words are in/out operands of the block, so nothing that consumes them
// can be scheduled above the wait.  lgkmcnt is shared by LDS and SMEM, vmcnt is never touched: it
// belongs to the window loads of the NEXT item, in flight across the whole loop.
static_assert(WIN_PITCH * 8 == 768, "the asm below hard-codes the LDS row pitch");

// Operands: %0-%7 destination buffer, %8 A, %9 B, %10 C, %11 byte offset of the next ltap to load,
// %12 scalar temp, %13 LDS address, %14 ltaps base, %15 this lane's base address in the window.
#define DIB_TAP_OPERANDS(buf)                                                                              \
  "=&v"(buf[0]), "=&v"(buf[1]), "=&v"(buf[2]), "=&v"(buf[3]), "=&v"(buf[4]), "=&v"(buf[5]), "=&v"(buf[6]),  \
      "=&v"(buf[7]), "+s"(A), "+s"(B), "+s"(C), "+s"(toff), "=&s"(stmp), "=&v"(vaddr)
#define DIB_RD8                                                                                            \
  "ds_read_b64 %0, %13 offset:0\n\tds_read_b64 %1, %13 offset:768\n\tds_read_b64 %2, %13 offset:1536\n\t"   \
  "ds_read_b64 %3, %13 offset:2304\n\tds_read_b64 %4, %13 offset:3072\n\tds_read_b64 %5, %13 offset:3840\n\t" \
  "ds_read_b64 %6, %13 offset:4608\n\tds_read_b64 %7, %13 offset:5376\n\t"

// first block of a segment: fetch ltaps t0 and t0+1, issue the reads of tap t0, start loading t0+2
__device__ __forceinline__ void tap_first(u2 (&buf)[R], unsigned &A, unsigned &B, unsigned &C, unsigned &toff,
                                          unsigned long long ltaps, unsigned lane_addr) {
  unsigned stmp, vaddr;
  asm volatile(
      "s_load_dword %9, %14, %11\n\t"
      "s_add_u32 %11, %11, 4\n\t"
      "s_load_dword %10, %14, %11\n\t"
      "s_add_u32 %11, %11, 4\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_and_b32 %12, %9, 0xffff\n\t"
      "v_add_u32 %13, %12, %15\n\t" DIB_RD8
      : DIB_TAP_OPERANDS(buf)
      : "s"(ltaps), "v"(lane_addr)
      : "scc");
}
// steady state: data of the previous block and ltap C have arrived; rotate A<-B<-C; issue the
// reads of the new B into `buf`; start loading the following ltap into C
__device__ __forceinline__ void tap_next(u2 (&buf)[R], unsigned &A, unsigned &B, unsigned &C, unsigned &toff,
                                         unsigned long long ltaps, unsigned lane_addr) {
  unsigned stmp, vaddr;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_mov_b32 %8, %9\n\t"
      "s_mov_b32 %9, %10\n\t"
      "s_and_b32 %12, %9, 0xffff\n\t"
      "v_add_u32 %13, %12, %15\n\t" DIB_RD8
      "s_load_dword %10, %14, %11\n\t"
      "s_add_u32 %11, %11, 4"
      : DIB_TAP_OPERANDS(buf)
      : "s"(ltaps), "v"(lane_addr)
      : "scc");
  // inline-asm results count as divergent in hipcc's uniformity analysis; carried around the loop
  // they would land in VGPRs and could not feed the next block's "s" operands
  A = __builtin_amdgcn_readfirstlane(A); B = __builtin_amdgcn_readfirstlane(B);
  C = __builtin_amdgcn_readfirstlane(C); toff = __builtin_amdgcn_readfirstlane(toff);
}
// last tap of a segment: nothing left to issue
__device__ __forceinline__ void tap_last(unsigned &A, unsigned &B) {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, %1" : "+s"(A), "+s"(B));
}
// Ordering of the arithmetic: every v_pk_mul_f16 of tap t takes its weight from A, and A is an
// output of the block that holds the wait for tap t's data -- so no multiply can be scheduled
// above that wait, and the data buffers need no (copy-inducing) pass-through operands.

// Diagnostic stamps (nullptr in every product launch): shader-clock readings of lane 0 of wave 0.
__device__ __forceinline__ void stamp(unsigned long long *dbg, int slot) {
  if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 8 + slot] = __builtin_readcyclecounter();
}

// One work item = one (tile, tap segment): a window of source rows to stage and a run of taps.
// Kept to 8 packed words: three items are live at once (being accumulated, being staged, being
// decoded) and they must all stay in SGPRs.
struct Item {
  int H, W;
  int tab_off;   // word offset of the image's tap table inside `tables`
  int where;     // img | ch << 8 | mode << 16 | first << 24 | last << 25 | valid << 26
  int xy;        // x0 | y0 << 16
  int seg;       // rl | cmax << 8 | nrows << 16 | pqused << 24
  int t0, t1;

  __device__ int img() const { return where & 255; }
  __device__ int ch() const { return (where >> 8) & 255; }
  __device__ int mode() const { return (where >> 16) & 255; }
  __device__ bool first() const { return (where >> 24) & 1; }
  __device__ bool last() const { return (where >> 25) & 1; }
  __device__ bool valid() const { return (where >> 26) & 1; }
  __device__ int x0() const { return xy & 0xffff; }
  __device__ int y0() const { return (unsigned)xy >> 16; }
  __device__ int rl() const { return seg & 255; }
  __device__ int cmax() const { return (seg >> 8) & 255; }
  __device__ int nrows() const { return (seg >> 16) & 255; }
  __device__ int pqused() const { return (unsigned)seg >> 24; }
};

// n / d for 0 <= n < 2^24 with rcp = 1.0f / d: one multiply and a one-step correction instead of
// the ~40-instruction integer-division expansion
__device__ __forceinline__ int fast_div(int n, int d, float rcp) {
  int q = (int)((float)n * rcp);
  const int r = n - q * d;
  if (r >= d) ++q;
  else if (r < 0) --q;
  return q;
}

// Walks the (tile, segment) items of this workgroup: tiles wg, wg + nwg, ... of the flattened
// [image][channel][ty][tx] list (a stride of nwg mixes all images into every workgroup, which
// balances PSFs of different tap counts), segments in order inside each tile.  Tiles only move
// forward, so the image lookup is a cursor that advances, not a search.
struct ItemWalker {
  const BlurBatch &batch;
  const int *tables;
  int K, tile, stride, seg, nsegs;
  int img, img_begin, img_end;  // image cursor: tiles [img_begin, img_end) belong to batch.img[img]
  int tiles_x, per_ch;
  float rcp_tx, rcp_pc;
  const uint4 *segs;
  Item cur;

  __device__ ItemWalker(const BlurBatch &b, const int *t, int K_, int first_tile, int stride_)
      : batch(b), tables(t), K(K_), tile(first_tile), stride(stride_), seg(0), nsegs(0), img(-1), img_begin(0), img_end(0),
        tiles_x(1), per_ch(1), rcp_tx(1.f), rcp_pc(1.f), segs(nullptr) {
    cur.where = 0;
    open_tile();
  }
  __device__ void open_tile() {
    if (tile >= batch.total_tiles) { cur.where = 0; return; }
    while (tile >= img_end) {  // enter the next image (uniform; at most batch.n times per workgroup)
      ++img;
      const ImageDesc &d = batch.img[img];
      img_begin = d.tile_begin;
      tiles_x = d.tiles_x;
      per_ch = d.tiles_x * d.tiles_y;
      img_end = img_begin + d.C * per_ch;
      rcp_tx = 1.0f / (float)tiles_x;
      rcp_pc = 1.0f / (float)per_ch;
      const int *tab = tables + (size_t)d.table * table_words(K);
      nsegs = tab[HDR_NSEGS];
      segs = reinterpret_cast<const uint4 *>(tab + table_segs_off(K));
      cur.H = d.H; cur.W = d.W;
      cur.tab_off = d.table * table_words(K);
    }
    int local = tile - img_begin;
    const int ch = fast_div(local, per_ch, rcp_pc);
    local -= ch * per_ch;
    const int ty = fast_div(local, tiles_x, rcp_tx), tx = local - ty * tiles_x;
    cur.where = img | (ch << 8) | (pad_mode_for(K, cur.H, cur.W) << 16) | (1 << 26);
    cur.xy = (tx * TILE_W) | ((ty * TH) << 16);
    seg = 0;
    load_seg();
  }
  __device__ void load_seg() {
    cur.where &= ~(3 << 24);
    if (seg == 0) cur.where |= 1 << 24;
    if (nsegs == 0) {  // no taps at all: the tile is written as zeros
      cur.t0 = cur.t1 = 0; cur.seg = 0; cur.where |= 1 << 25;
      return;
    }
    const uint4 sg = segs[seg];
    cur.t0 = sg.x; cur.t1 = sg.y;
    const int rf = sg.z >> 8, rl = sg.z & 255, cmin = sg.w >> 8, cmax = sg.w & 255;
    cur.seg = rl | (cmax << 8) | ((TH + (rl - rf)) << 16) | ((64 + (cmax - cmin)) << 24);
    if (seg + 1 >= nsegs) cur.where |= 1 << 25;
  }
  __device__ void advance() {
    if (!cur.valid()) return;
    if (seg + 1 < nsegs) { ++seg; load_seg(); }
    else { tile += stride; open_tile(); }
  }
};

// The walker's values are wave-uniform by construction (they derive from blockIdx and kernel
// arguments), but after the struct has travelled through the software-pipelined loop hipcc no
// longer proves it and falls back to per-lane loads for the tap list.  v_readfirstlane pins every
// field to an SGPR again: taps come in through scalar loads and never touch vmcnt, which must
// stay reserved for the window loads that are in flight across the arithmetic.  Items carry
// integer handles only; pointers are re-derived from the kernel arguments so that they keep
// their global address space (a pointer rebuilt from integers degrades to flat loads).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ Item uniform_item(const Item &a) {
  Item b;
  b.H = uni(a.H); b.W = uni(a.W); b.tab_off = uni(a.tab_off); b.where = uni(a.where);
  b.xy = uni(a.xy); b.seg = uni(a.seg); b.t0 = uni(a.t0); b.t1 = uni(a.t1);
  return b;
}

// Raw source values of one window as they come back from memory: per LDS row the five fp16
// values P[lane + 64k], k = 0..4 (word j and word j+64 of the split layout share three of them).
struct FillRegs {
  unsigned short v[G][5];
  unsigned zmask;  // bit k: column k is zero padding; bit 8+g: row g is zero padding (PAD_ZERO only)
};

// Wave-uniform buffer descriptor of one channel plane (base, byte size): loads and stores then take
// a 32-bit per-lane byte offset (voffset) plus a scalar row offset (soffset) -- no 64-bit address
// arithmetic on the vector ALU, which the arithmetic of the co-resident wave keeps busy.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void *img_base, int ch, int H, int W) {
  const unsigned long long a = (unsigned long long)img_base + (unsigned long long)ch * H * W * 2ull;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, H * W * 2, 0x00020000);
}

__device__ __forceinline__ void issue_fill(const BlurBatch &batch, int K, const Item &it, FillRegs &f, int wave, int lane) {
  const int pb = K / 2 - 1, pa = K / 2, mode = it.mode();
  const __amdgpu_buffer_rsrc_t rsrc = plane_rsrc(batch.img[it.img()].in, it.ch(), it.H, it.W);
  unsigned coff[5];
  f.zmask = 0;
  const int c_first = it.x0() + pb - it.cmax();  // virtual column of P[0]
  if (c_first >= 0 && c_first + 63 + 256 <= it.W - 1) {
    // interior tile: no reflection, no clamping
#pragma unroll
    for (int k = 0; k < 5; ++k) coff[k] = 2u * (unsigned)(c_first + lane + 64 * k);
  } else {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      bool z;
      coff[k] = 2u * (unsigned)map_coord(c_first + lane + 64 * k, it.W, pa, pb, mode, z);
      if (z) f.zmask |= 1u << k;
    }
  }
  const int qb = wave * G, r_first = it.y0() + pb - it.rl() + qb;  // virtual row of this wave's first LDS row
  const int w2 = it.W * 2;
  if (r_first >= 0 && r_first + G - 1 <= it.H - 1) {
    // every row of the batch lies inside the image (rows past the window's end are loaded but never
    // written to LDS); unconditional loads, all G*5 in flight before any is used
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int soff = (r_first + g) * w2;
#pragma unroll
      for (int k = 0; k < 5; ++k) f.v[g][k] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, coff[k], soff, 0);
    }
  } else {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bool zr;
      const int sr = map_coord(r_first - qb + min(qb + g, max(it.nrows() - 1, 0)), it.H, pa, pb, mode, zr);
      if (zr) f.zmask |= 1u << (8 + g);
      const int soff = __builtin_amdgcn_readfirstlane(sr * w2);
#pragma unroll
      for (int k = 0; k < 5; ++k) f.v[g][k] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, coff[k], soff, 0);
    }
  }
}

template <bool ZERO>
__device__ __forceinline__ void commit_fill(const Item &it, const FillRegs &f, uint2 *win, int wave, int lane) {
  const int qb = wave * G, nrows = it.nrows();
  const bool second = lane + 64 < it.pqused();
  uint2 *wp = win + qb * PQ + lane;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (qb + g < nrows) {  // wave-uniform
      unsigned c[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        c[k] = f.v[g][k];
        if (ZERO && (((f.zmask >> k) & 1u) || ((f.zmask >> (8 + g)) & 1u))) c[k] = 0;
      }
      wp[g * PQ] = make_uint2(c[0] | (c[1] << 16), c[2] | (c[3] << 16));
      if (second) wp[g * PQ + 64] = make_uint2(c[1] | (c[2] << 16), c[3] | (c[4] << 16));
    }
  }
}

// taps t0..t1 of the item in row-major order; memory side in the asm blocks above
__device__ __forceinline__ void accumulate(const int *__restrict__ tables, int K, const Item &it, h2 (&acc)[R][2], unsigned lane_addr) {
#pragma clang fp contract(off)
  const int n = __builtin_amdgcn_readfirstlane(it.t1 - it.t0);
  if (n <= 0) return;
  // asm "s" operands must be provably uniform: rebuild the (integer) address from v_readfirstlane halves
  const unsigned long long la = (unsigned long long)(tables + it.tab_off + table_ltaps_off(K));
  const unsigned long long ltaps = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(la >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((unsigned)la);
  u2 bx[R], by[R];
  unsigned A = 0, B = 0, C = 0, toff = (unsigned)__builtin_amdgcn_readfirstlane(it.t0 * 4);
  auto madd = [&](u2(&buf)[R], unsigned tapw) {
    const unsigned wb = tapw >> 16;
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
  tap_first(bx, A, B, C, toff, ltaps, lane_addr);
  int i = 0;
  while (true) {
    if (i == n - 1) { tap_last(A, B); madd(bx, A); break; }
    tap_next(by, A, B, C, toff, ltaps, lane_addr);
    madd(bx, A);
    ++i;
    if (i == n - 1) { tap_last(A, B); madd(by, A); break; }
    tap_next(bx, A, B, C, toff, ltaps, lane_addr);
    madd(by, A);
    ++i;
  }
}

__device__ __forceinline__ void store_tile(const BlurBatch &batch, const Item &it, const h2 (&acc)[R][2], int wave, int lane) {
  const __amdgpu_buffer_rsrc_t rsrc = plane_rsrc(batch.img[it.img()].out, it.ch(), it.H, it.W);
  const int xr = it.W - it.x0() - lane;  // columns remaining for this lane
  const unsigned voff = 2u * (unsigned)(it.x0() + lane);
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int y = it.y0() + wave * R + i;
    if (y < it.H) {  // wave-uniform
      const int soff = y * it.W * 2;
      // halves are extracted with integer ops: hipcc (ROCm 7.2) stored the LOW half twice when the
      // high element of the fp16x2 accumulator was taken with a vector subscript
      const unsigned a = __builtin_bit_cast(unsigned, acc[i][0]), b = __builtin_bit_cast(unsigned, acc[i][1]);
      if (xr > 0) __builtin_amdgcn_raw_buffer_store_b16((short)(a & 0xffffu), rsrc, voff, soff, 0);
      if (xr > 64) __builtin_amdgcn_raw_buffer_store_b16((short)(a >> 16), rsrc, voff + 128u, soff, 0);
      if (xr > 128) __builtin_amdgcn_raw_buffer_store_b16((short)(b & 0xffffu), rsrc, voff + 256u, soff, 0);
      if (xr > 192) __builtin_amdgcn_raw_buffer_store_b16((short)(b >> 16), rsrc, voff + 384u, soff, 0);
    }
  }
}

// Persistent kernel: WG_PER_CU workgroups per CU, each walking its own list of (tile, segment)
// items.  Software pipeline, one barrier per item: while item n is accumulated out of LDS window
// n%2, the global loads of item n+1 are in flight in registers; they are packed into window
// (n+1)%2 once the arithmetic of item n has been issued.  A window is only rewritten two items
// after it was last read, and every wave passes the barrier in between.
__global__ __launch_bounds__(256, WG_PER_CU) void blur_tiled_f16_kernel(BlurBatch batch, const int *__restrict__ tables,
                                                                        int K, unsigned long long *dbg) {
  extern __shared__ uint2 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  ItemWalker walk(batch, tables, K, blockIdx.x, gridDim.x);
  if (!walk.cur.valid()) return;

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
  const unsigned lane_off = (unsigned)((wave * R) * PQ + lane) * 8u;
  h2 acc[R][2];
  FillRegs regs;

  stamp(dbg, 0);
  Item it = uniform_item(walk.cur);
  issue_fill(batch, K, it, regs, wave, lane);
  walk.advance();
  Item nxt = uniform_item(walk.cur);
  if (it.mode() == PAD_ZERO) commit_fill<true>(it, regs, lds, wave, lane);
  else commit_fill<false>(it, regs, lds, wave, lane);
  __syncthreads();
  stamp(dbg, 1);
  int par = 0, nitems = 0;
  unsigned long long c_walk = 0, c_issue = 0, c_acc = 0, c_store = 0, c_commit = 0, c_bar = 0, tq = 0;
#define DIB_TICK(var) do { if (dbg) { unsigned long long now_ = __builtin_readcyclecounter(); var += now_ - tq; tq = now_; } } while (0)
  if (dbg) tq = __builtin_readcyclecounter();
  while (true) {
    if (nxt.valid()) issue_fill(batch, K, nxt, regs, wave, lane);   // loads fly during the arithmetic below
    DIB_TICK(c_issue);
    // decode the item after next now: its scalar loads complete under the arithmetic as well
    walk.advance();
    const Item nxt2 = uniform_item(walk.cur);
    DIB_TICK(c_walk);
    if (it.first()) {
#pragma unroll
      for (int i = 0; i < R; ++i) { acc[i][0] = h2{0, 0}; acc[i][1] = h2{0, 0}; }
    }
    accumulate(tables, K, it, acc, lds0 + (unsigned)par * (WIN_WORDS * 8) + lane_off);
    DIB_TICK(c_acc);
    if (it.last()) store_tile(batch, it, acc, wave, lane);
    DIB_TICK(c_store);
    ++nitems;
    if (!nxt.valid()) break;
    par ^= 1;
    if (nxt.mode() == PAD_ZERO) commit_fill<true>(nxt, regs, lds + par * WIN_WORDS, wave, lane);
    else commit_fill<false>(nxt, regs, lds + par * WIN_WORDS, wave, lane);
    DIB_TICK(c_commit);
    __syncthreads();
    DIB_TICK(c_bar);
    it = nxt;
    nxt = nxt2;
  }
  if (dbg && threadIdx.x == 0) {
    dbg[(size_t)blockIdx.x * 8 + 4] = (c_issue << 32) | (c_acc & 0xffffffffull);
    dbg[(size_t)blockIdx.x * 8 + 5] = (c_store << 32) | (c_commit & 0xffffffffull);
    dbg[(size_t)blockIdx.x * 8 + 6] = (c_walk << 32) | (c_bar & 0xffffffffull);
  }
  stamp(dbg, 2);
  if (dbg && threadIdx.x == 0) {
    dbg[(size_t)blockIdx.x * 8 + 3] = (unsigned long long)nitems;
    dbg[(size_t)blockIdx.x * 8 + 7] = wall_clock64();
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

// Diagnostics only: when set, the tiled kernel records per-workgroup phase stamps (8 x u64 each).
static unsigned long long *g_stamp_buffer = nullptr;
extern "C" void dib_debug_set_stamp_buffer(void *dev_ptr) { g_stamp_buffer = (unsigned long long *)dev_ptr; }

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
  static int g_num_cus = 256;
  if (!attr_set) {
    DIB_HIP_CHECK(hipFuncSetAttribute((const void *)blur_tiled_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    int dev = 0, cus = 0;
    DIB_HIP_CHECK(hipGetDevice(&dev));
    DIB_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus > 0) g_num_cus = cus;
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
      d.tiles_y = (H[i] + TH - 1) / TH;
      d.tile_begin = tiles;
      tiles += d.C * d.tiles_x * d.tiles_y;
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
      int nwg = g_num_cus * WG_PER_CU;
      if (nwg > tiles) nwg = tiles;
      hipLaunchKernelGGL(blur_tiled_f16_kernel, dim3(nwg), dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K,
                         g_stamp_buffer);
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
  d.in = in_dev; d.out = out_dev; d.C = C; d.H = H; d.W = W; d.table = 0; d.tile_begin = 0; d.tiles_x = d.tiles_y = 0;
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
