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
//   A workgroup owns a 256 x 32 output tile of ONE channel.  Lane l of every wave owns the four
//   columns x0 + l + 64k (k = 0..3), packed as two fp16x2 registers per row.  LDS row q holds the
//   source window row as 8-byte words:  word j = { P[j], P[j+64], P[j+128], P[j+192] },
//   j in [0, 64+ex), where P is the (virtually padded) source row starting at column
//   x0 + pb - cmax and ex = cmax - cmin is the column extent of the tap SEGMENT being processed.
//   A tap (r, c) is then ONE aligned, bank-conflict-free ds_read_b64 at word  lane + (cmax - c)
//   for ANY column shift -- odd shifts included.  (Measured on MI355X: a ds_read_b64 that is not
//   8-byte aligned runs 22x slower, so a plain row-major fp16 window is not an option.)
//   Row shifts are LDS row offsets; the R rows a lane owns use compile-time immediate offsets.
//   The tap list arrives cut into segments (dib_compact.hip) whose bounding box is at most
//   17 PSF rows x 33 PSF columns, so ONE small LDS window (48 rows x 96 words = 36 KB, four
//   workgroups per CU) serves any PSF; a wide or tall PSF simply takes several fill+accumulate
//   rounds, in tap order, with the accumulators staying in registers.
//
// What bounds it (measured, scratch/ubench): v_pk_mul_f16 / v_pk_add_f16 issue at HALF rate on
// gfx950 (1.9 ns per wave-instruction per SIMD at full occupancy, 3.2 ns with one wave), so the
// bit-exact contract costs 2 x 1.9 ns per 128 pixel-taps: ~32 us for the BASELINE batch, above its
// 16 us HBM time.  Occupancy is the lever: four workgroups per CU, waves that stall on their
// window loads or on a barrier leave the VALU to the others.
#include "dib_common.h"
#include <hip/hip_fp16.h>

namespace dib {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int TILE_W = 256;
constexpr int PQ = WIN_PITCH;         // LDS row pitch in 8-byte words (96)
constexpr int TH = 32;                // tile rows
constexpr int LROWS = TH + SEG_ROWS;  // LDS rows per window (48)
constexpr int LDS_BYTES = LROWS * PQ * 8;  // 36,864 B: four workgroups per CU
static_assert(WIN_PITCH * 8 == 768, "the asm below hard-codes the LDS row pitch");

// 64-bit asm operands must be scalar integers: hipcc (ROCm 7.2) aliases both lanes of a
// 2 x 32-bit vector operand of an inline-asm "=v" output to the same register.
typedef unsigned long long u2;

// ---- the tap loop, hand-written ------------------------------------------------------------------
// Left to hipcc the loop serialises (it merges the 8-byte LDS reads into half-rate ds_read2_b64,
// sinks the scalar tap load to its first use and waits right behind every access), and stitching
// it from several asm statements costs ~20 scalar instructions per tap in glue (asm results count
// as divergent, so loop-carried scalars bounce through VGPRs).  The whole loop of one tap segment
// is therefore ONE asm statement with fixed buffer registers:
//     v[64:79]  buffer A: the 8 rows x 8 bytes of the current tap, multiplied in place
//     v[80:95]  buffer B: same for the following tap (the two alternate)
//     v96       LDS address
//   per tap:  s_waitcnt lgkmcnt(0)          data of this tap + ltap word of the next one arrived
//             8 x ds_read_b64 -> other buf  next tap's data (address = its ltap offset + lane base)
//             s_load_dword                  ltap word of the tap after next
//             16 x v_pk_mul_f16, 16 x v_pk_add_f16 on this tap's data while all of that is in flight
// 7 scalar + 1 vector instruction of overhead per 32 packed operations.  Scalar instructions matter
// here: with one or two waves per SIMD in their tap loops (the usual case, the others are filling or
// storing) a wave's own SALU work is on its critical path, it is not hidden behind another wave's
// VALU.  Hence: the weight is read straight from the high half of the ltap SGPR (op_sel, no
// shift + pack), the LDS address is one v_mad_u32_u16 (no scalar mask), the loop ends on the borrow
// of the counter's decrement, and the last tap does not branch around its (harmless, zero-padded)
// look-ahead -- 13 -> 7 scalar instructions per tap, -2.5 % kernel time, bit-identical results.
// (Also built and measured: a x3-unrolled loop with THREE data buffers -- LDS look-ahead of two taps,
// `s_waitcnt lgkmcnt(8)` -- that takes its ltap words from a VGPR with v_readlane, 4 scalar instructions
// per tap and no scalar-memory loads at all.  Bit-identical, but 3 % slower: LDS latency is already
// covered by one tap of look-ahead, and the readlane -> address -> ds_read chain opens every tap.)
// ltap word = byte offset of the tap's source word in the window (low 16) | fp16 weight (high 16).
// Operands: %0-%15 accumulators, %16 byte offset of the next ltap, %17 taps left, %18 A (tap being
// multiplied), %19 B (tap being fetched), %20 C (tap in flight), %21 scalar temp, %22 ltaps, %23 lane base.
#define DIB_MUL(b, i) "v_pk_mul_f16 v" #b ", %18, v" #b " op_sel:[1,0] op_sel_hi:[1,1]\n\t"
#define DIB_ADD(b, i) "v_pk_add_f16 %" #i ", %" #i ", v" #b "\n\t"
#define DIB_MADD_A                                                                                          \
  DIB_MUL(64, 0) DIB_MUL(65, 1) DIB_MUL(66, 2) DIB_MUL(67, 3) DIB_MUL(68, 4) DIB_MUL(69, 5) DIB_MUL(70, 6) DIB_MUL(71, 7) \
  DIB_MUL(72, 8) DIB_MUL(73, 9) DIB_MUL(74, 10) DIB_MUL(75, 11) DIB_MUL(76, 12) DIB_MUL(77, 13) DIB_MUL(78, 14) DIB_MUL(79, 15) \
  DIB_ADD(64, 0) DIB_ADD(65, 1) DIB_ADD(66, 2) DIB_ADD(67, 3) DIB_ADD(68, 4) DIB_ADD(69, 5) DIB_ADD(70, 6) DIB_ADD(71, 7) \
  DIB_ADD(72, 8) DIB_ADD(73, 9) DIB_ADD(74, 10) DIB_ADD(75, 11) DIB_ADD(76, 12) DIB_ADD(77, 13) DIB_ADD(78, 14) DIB_ADD(79, 15)
#define DIB_MADD_B                                                                                          \
  DIB_MUL(80, 0) DIB_MUL(81, 1) DIB_MUL(82, 2) DIB_MUL(83, 3) DIB_MUL(84, 4) DIB_MUL(85, 5) DIB_MUL(86, 6) DIB_MUL(87, 7) \
  DIB_MUL(88, 8) DIB_MUL(89, 9) DIB_MUL(90, 10) DIB_MUL(91, 11) DIB_MUL(92, 12) DIB_MUL(93, 13) DIB_MUL(94, 14) DIB_MUL(95, 15) \
  DIB_ADD(80, 0) DIB_ADD(81, 1) DIB_ADD(82, 2) DIB_ADD(83, 3) DIB_ADD(84, 4) DIB_ADD(85, 5) DIB_ADD(86, 6) DIB_ADD(87, 7) \
  DIB_ADD(88, 8) DIB_ADD(89, 9) DIB_ADD(90, 10) DIB_ADD(91, 11) DIB_ADD(92, 12) DIB_ADD(93, 13) DIB_ADD(94, 14) DIB_ADD(95, 15)
// DIB_ACC_FMA16: the same loop with ONE packed fused multiply-add per register (one rounding per tap
// instead of two): half the tap arithmetic, not the reference's arithmetic.
#define DIB_FMA(b, i) "v_pk_fma_f16 %" #i ", %18, v" #b ", %" #i " op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
#define DIB_FMADD_A                                                                                         \
  DIB_FMA(64, 0) DIB_FMA(65, 1) DIB_FMA(66, 2) DIB_FMA(67, 3) DIB_FMA(68, 4) DIB_FMA(69, 5) DIB_FMA(70, 6) DIB_FMA(71, 7) \
  DIB_FMA(72, 8) DIB_FMA(73, 9) DIB_FMA(74, 10) DIB_FMA(75, 11) DIB_FMA(76, 12) DIB_FMA(77, 13) DIB_FMA(78, 14) DIB_FMA(79, 15)
#define DIB_FMADD_B                                                                                         \
  DIB_FMA(80, 0) DIB_FMA(81, 1) DIB_FMA(82, 2) DIB_FMA(83, 3) DIB_FMA(84, 4) DIB_FMA(85, 5) DIB_FMA(86, 6) DIB_FMA(87, 7) \
  DIB_FMA(88, 8) DIB_FMA(89, 9) DIB_FMA(90, 10) DIB_FMA(91, 11) DIB_FMA(92, 12) DIB_FMA(93, 13) DIB_FMA(94, 14) DIB_FMA(95, 15)
#define DIB_READ8(base)                                                                                      \
  "v_mad_u32_u16 v96, %19, 1, %23\n\t"                                                                       \
  "ds_read_b64 v[" #base ":" #base "+1], v96\n\tds_read_b64 v[" #base "+2:" #base "+3], v96 offset:768\n\t"   \
  "ds_read_b64 v[" #base "+4:" #base "+5], v96 offset:1536\n\tds_read_b64 v[" #base "+6:" #base "+7], v96 offset:2304\n\t" \
  "ds_read_b64 v[" #base "+8:" #base "+9], v96 offset:3072\n\tds_read_b64 v[" #base "+10:" #base "+11], v96 offset:3840\n\t" \
  "ds_read_b64 v[" #base "+12:" #base "+13], v96 offset:4608\n\tds_read_b64 v[" #base "+14:" #base "+15], v96 offset:5376\n\t"
#define DIB_NEXTTAP "s_load_dword %20, %22, %16\n\ts_add_u32 %16, %16, 4\n\t"

// acc[i][0] / acc[i][1] (i = 0..7): the packed fp16 accumulators of this lane's 8 rows x 4 columns
template <bool FUSED>
__device__ __forceinline__ void tap_loop_r8(h2 (&acc)[8][2], unsigned long long ltaps, int t0, int n, unsigned lane_addr) {
  // cnt = taps left minus one: the borrow of its decrement ends the loop (n >= 1 in every segment)
  unsigned toff = (unsigned)__builtin_amdgcn_readfirstlane(t0 * 4), cnt = (unsigned)__builtin_amdgcn_readfirstlane(n - 1);
  unsigned sA, sB, sC, st;
  unsigned a[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[2 * i] = __builtin_bit_cast(unsigned, acc[i][0]); a[2 * i + 1] = __builtin_bit_cast(unsigned, acc[i][1]); }
#define DIB_R8_ASM(ARITH_A, ARITH_B) \
  asm volatile( \
      /* vmcnt(0): window loads whose values were never used (rows past the window's end) may still be */ \
      /* in flight, and hipcc is free to have put their destinations into the registers clobbered */ \
      /* here -- it waits before ITS OWN next write to such a register, but not before this asm's. */ \
      /* prologue: ltap[t0] -> B, ltap[t0+1] -> C, data of tap t0 -> buffer A */ \
      "s_load_dword %19, %22, %16\n\ts_add_u32 %16, %16, 4\n\t" DIB_NEXTTAP \
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" DIB_READ8(64) \
      "Ldib_loop%=:\n\t" \
      /* ---- tap in buffer A ---- */ \
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %18, %19\n\ts_mov_b32 %19, %20\n\t" \
      DIB_READ8(80) DIB_NEXTTAP \
      ARITH_A \
      "s_sub_u32 %17, %17, 1\n\ts_cbranch_scc1 Ldib_done%=\n\t" \
      /* ---- tap in buffer B ---- */ \
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %18, %19\n\ts_mov_b32 %19, %20\n\t" \
      DIB_READ8(64) DIB_NEXTTAP \
      ARITH_B \
      "s_sub_u32 %17, %17, 1\n\ts_cbranch_scc0 Ldib_loop%=\n\t" \
      "Ldib_done%=:\n\t" \
      "s_waitcnt lgkmcnt(0)" \
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), \
        "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+s"(toff), "+s"(cnt), "=&s"(sA), "=&s"(sB), \
        "=&s"(sC), "=&s"(st) \
      : "s"(ltaps), "v"(lane_addr) \
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", \
        "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "scc", \
        "memory")
  if constexpr (FUSED) { DIB_R8_ASM(DIB_FMADD_A, DIB_FMADD_B); } else { DIB_R8_ASM(DIB_MADD_A, DIB_MADD_B); }
#undef DIB_R8_ASM
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i][0] = __builtin_bit_cast(h2, a[2 * i]); acc[i][1] = __builtin_bit_cast(h2, a[2 * i + 1]); }
}
#undef DIB_MUL
#undef DIB_ADD
#undef DIB_MADD_A
#undef DIB_MADD_B
#undef DIB_FMA
#undef DIB_FMADD_A
#undef DIB_FMADD_B
#undef DIB_READ8
#undef DIB_NEXTTAP

// ---- the same loop for 8 waves x 4 rows per lane (buffers v[40:47] / v[48:55], address v56) ----------
#define DIB_MADD(b0, b1, b2, b3, b4, b5, b6, b7)                                                        \
  "s_lshr_b32 %13, %10, 16\n\ts_pack_ll_b32_b16 %13, %13, %13\n\t"                                        \
  "v_pk_mul_f16 " b0 ", %13, " b0 "\n\tv_pk_mul_f16 " b1 ", %13, " b1 "\n\tv_pk_mul_f16 " b2 ", %13, " b2 "\n\t" \
  "v_pk_mul_f16 " b3 ", %13, " b3 "\n\tv_pk_mul_f16 " b4 ", %13, " b4 "\n\tv_pk_mul_f16 " b5 ", %13, " b5 "\n\t" \
  "v_pk_mul_f16 " b6 ", %13, " b6 "\n\tv_pk_mul_f16 " b7 ", %13, " b7 "\n\t"                              \
  "v_pk_add_f16 %0, %0, " b0 "\n\tv_pk_add_f16 %1, %1, " b1 "\n\tv_pk_add_f16 %2, %2, " b2 "\n\t"          \
  "v_pk_add_f16 %3, %3, " b3 "\n\tv_pk_add_f16 %4, %4, " b4 "\n\tv_pk_add_f16 %5, %5, " b5 "\n\t"          \
  "v_pk_add_f16 %6, %6, " b6 "\n\tv_pk_add_f16 %7, %7, " b7 "\n\t"
#define DIB_READ4(lo0, lo1, lo2, lo3)                                                                    \
  "s_and_b32 %13, %11, 0xffff\n\tv_add_u32 v56, %13, %15\n\t"                                             \
  "ds_read_b64 " lo0 ", v56\n\tds_read_b64 " lo1 ", v56 offset:768\n\t"                                   \
  "ds_read_b64 " lo2 ", v56 offset:1536\n\tds_read_b64 " lo3 ", v56 offset:2304\n\t"
#define DIB_NEXTTAP "s_load_dword %12, %14, %8\n\ts_add_u32 %8, %8, 4\n\t"

// acc[i][0] / acc[i][1] (i = 0..3): the packed fp16 accumulators of this lane's 4 rows x 4 columns
__device__ __forceinline__ void tap_loop_r4(h2 (&acc)[4][2], unsigned long long ltaps, int t0, int n, unsigned lane_addr) {
  unsigned toff = (unsigned)__builtin_amdgcn_readfirstlane(t0 * 4), cnt = (unsigned)__builtin_amdgcn_readfirstlane(n);
  unsigned sA, sB, sC, st;
  unsigned a0 = __builtin_bit_cast(unsigned, acc[0][0]), a1 = __builtin_bit_cast(unsigned, acc[0][1]);
  unsigned a2 = __builtin_bit_cast(unsigned, acc[1][0]), a3 = __builtin_bit_cast(unsigned, acc[1][1]);
  unsigned a4 = __builtin_bit_cast(unsigned, acc[2][0]), a5 = __builtin_bit_cast(unsigned, acc[2][1]);
  unsigned a6 = __builtin_bit_cast(unsigned, acc[3][0]), a7 = __builtin_bit_cast(unsigned, acc[3][1]);
  asm volatile(
      // vmcnt(0): window loads whose values were never used (rows past the window's end) may still be
      // in flight, and hipcc is free to have put their destinations into the registers clobbered
      // here -- it waits before ITS OWN next write to such a register, but not before this asm's.
      // prologue: ltap[t0] -> B, ltap[t0+1] -> C, data of tap t0 -> buffer A
      "s_load_dword %11, %14, %8\n\ts_add_u32 %8, %8, 4\n\t" DIB_NEXTTAP
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" DIB_READ4("v[40:41]", "v[42:43]", "v[44:45]", "v[46:47]")
      "Ldib_loop%=:\n\t"
      // ---- tap in buffer A ----
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %10, %11\n\ts_mov_b32 %11, %12\n\t"
      "s_cmp_eq_u32 %9, 1\n\ts_cbranch_scc1 Ldib_lastA%=\n\t"
      DIB_READ4("v[48:49]", "v[50:51]", "v[52:53]", "v[54:55]") DIB_NEXTTAP
      "Ldib_lastA%=:\n\t"
      DIB_MADD("v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47")
      "s_sub_u32 %9, %9, 1\n\ts_cmp_eq_u32 %9, 0\n\ts_cbranch_scc1 Ldib_done%=\n\t"
      // ---- tap in buffer B ----
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %10, %11\n\ts_mov_b32 %11, %12\n\t"
      "s_cmp_eq_u32 %9, 1\n\ts_cbranch_scc1 Ldib_lastB%=\n\t"
      DIB_READ4("v[40:41]", "v[42:43]", "v[44:45]", "v[46:47]") DIB_NEXTTAP
      "Ldib_lastB%=:\n\t"
      DIB_MADD("v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55")
      "s_sub_u32 %9, %9, 1\n\ts_cmp_eq_u32 %9, 0\n\ts_cbranch_scc0 Ldib_loop%=\n\t"
      "Ldib_done%=:\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(toff), "+s"(cnt),
        "=&s"(sA), "=&s"(sB), "=&s"(sC), "=&s"(st)
      : "s"(ltaps), "v"(lane_addr)
      : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
        "v56", "scc", "memory");
  acc[0][0] = __builtin_bit_cast(h2, a0); acc[0][1] = __builtin_bit_cast(h2, a1);
  acc[1][0] = __builtin_bit_cast(h2, a2); acc[1][1] = __builtin_bit_cast(h2, a3);
  acc[2][0] = __builtin_bit_cast(h2, a4); acc[2][1] = __builtin_bit_cast(h2, a5);
  acc[3][0] = __builtin_bit_cast(h2, a6); acc[3][1] = __builtin_bit_cast(h2, a7);
}
#undef DIB_MADD
#undef DIB_READ4
#undef DIB_NEXTTAP

// DIB_ACC_FP32: acc32 = acc32 + float(P) * float(w), taps in the same order, ONE rounding to fp16 at the
// store.  The product of two fp16 values is exact in fp32 (11 + 11 <= 24 significand bits), so fused
// and unfused forms agree and the result is reproducible bit for bit on any IEEE fp32 machine (the
// oracle restates it with numpy float32).  Plain C++: this mode trades the hand-scheduled loop for
// accuracy (error vs exact arithmetic ~2^-12 relative instead of ~ntaps * 2^-12).
template <int R>
__device__ __forceinline__ void tap_loop_fp32(float (&acc)[R][4], const unsigned *__restrict__ ltaps, int t0, int n,
                                              unsigned lane_addr) {
#pragma clang fp contract(off)
#pragma unroll 2
  for (int t = t0; t < t0 + n; ++t) {
    const unsigned lt = ltaps[t];
    const float w = (float)__builtin_bit_cast(_Float16, (unsigned short)(lt >> 16));
    const unsigned a = lane_addr + (lt & 0xffffu);
#pragma unroll
    for (int i = 0; i < R; ++i) {
      typedef unsigned uvec2 __attribute__((ext_vector_type(2)));
      const uvec2 q = *(const __attribute__((address_space(3))) uvec2 *)(size_t)(a + (unsigned)(i * PQ * 8));
      const float p0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(q.x & 0xffffu));
      const float p1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(q.x >> 16));
      const float p2 = (float)__builtin_bit_cast(_Float16, (unsigned short)(q.y & 0xffffu));
      const float p3 = (float)__builtin_bit_cast(_Float16, (unsigned short)(q.y >> 16));
      // explicit fma: the product is exact in fp32, so fused == unfused bit for bit, and hipcc can fold
      // the fp16 -> fp32 conversions of both factors into v_fma_mix_f32
      acc[i][0] = __builtin_fmaf(p0, w, acc[i][0]); acc[i][1] = __builtin_fmaf(p1, w, acc[i][1]);
      acc[i][2] = __builtin_fmaf(p2, w, acc[i][2]); acc[i][3] = __builtin_fmaf(p3, w, acc[i][3]);
    }
  }
}

// Diagnostic stamps (nullptr in every product launch): shader-clock readings of lane 0 of wave 0.
__device__ __forceinline__ void stamp(unsigned long long *dbg, int slot) {
  if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 8 + slot] = __builtin_readcyclecounter();
}

// Wave-uniform buffer descriptor of one channel plane (base, byte size): loads and stores then take
// a 32-bit per-lane byte offset (voffset) plus a scalar row offset (soffset) -- no 64-bit address
// arithmetic on the vector ALU, which the arithmetic of the co-resident waves keeps busy.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void *img_base, int ch, int H, int W) {
  const unsigned long long a = (unsigned long long)img_base + (unsigned long long)ch * H * W * 2ull;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, H * W * 2, 0x00020000);
}

// One workgroup = one (image, channel, 256 x 32 tile).  NW waves, R = 32 / NW rows per lane.
template <int NW, bool ZERO, int ACC>
__device__ __forceinline__ void blur_tile_f16(const ImageDesc &d, const int *__restrict__ tab, int K, int ch, int tx,
                                              int ty, uint2 *lds, unsigned long long *dbg) {
#pragma clang fp contract(off)
  constexpr int R = TH / NW;              // rows per lane
  constexpr int G = (LROWS + NW - 1) / NW;  // LDS rows a wave fills: all of them in ONE batch of loads
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = d.H, W = d.W;
  const int pb = K / 2 - 1, pa = K / 2;
  const int mode = ZERO ? PAD_ZERO : (K > 129 ? PAD_REPLICATE : PAD_REFLECT);
  const int nsegs = tab[HDR_NSEGS];
  const uint4 *segs = reinterpret_cast<const uint4 *>(tab + table_segs_off(K));
  const unsigned long long la = (unsigned long long)(tab + table_ltaps_off(K));
  const unsigned long long ltaps = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(la >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((unsigned)la);
  const int x0 = tx * TILE_W, y0 = ty * TH;
  const __amdgpu_buffer_rsrc_t in_rsrc = plane_rsrc(d.in, ch, H, W);
  const int w2 = W * 2;

  h2 acc[R][2];
  float acc32[R][4];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    acc[i][0] = h2{0, 0}; acc[i][1] = h2{0, 0};
    acc32[i][0] = acc32[i][1] = acc32[i][2] = acc32[i][3] = 0.f;
  }

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
  const unsigned lane_addr = lds0 + (unsigned)((wave * R) * PQ + lane) * 8u;
  const int qb = wave * G;
  uint2 *wp = lds + qb * PQ + lane;

  stamp(dbg, 0);
  if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 8 + 4] = wall_clock64();
  for (int sg = 0; sg < nsegs; ++sg) {
    const uint4 seg = segs[sg];
    const int t0 = seg.x, n = (int)(seg.y - seg.x);
    const int rf = seg.z >> 8, rl = seg.z & 255, cmin = seg.w >> 8, cmax = seg.w & 255;
    const int nrows = TH + (rl - rf);
    const bool second = lane + 64 < 64 + (cmax - cmin);

    // ---- fill: per LDS row the five values P[lane + 64k] (word j and word j+64 share three) ------
    {
      unsigned coff[5];
      unsigned zmask = 0;
      const int c_first = x0 + pb - cmax;  // virtual column of P[0]
      if (!ZERO && c_first >= 0 && c_first + 63 + 256 <= W - 1) {  // interior: no reflection, no clamping
#pragma unroll
        for (int k = 0; k < 5; ++k) coff[k] = 2u * (unsigned)(c_first + lane + 64 * k);
      } else {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          bool z;
          coff[k] = 2u * (unsigned)map_coord(c_first + lane + 64 * k, W, pa, pb, mode, z);
          if (z) zmask |= 1u << k;
        }
      }
      unsigned short v[G][5];
      const int r_first = y0 + pb - rl + qb;  // virtual row of this wave's first LDS row
      if (!ZERO && r_first >= 0 && r_first + G - 1 <= H - 1) {
        // all rows inside the image (rows past the window's end are loaded but never written to LDS)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int soff = (r_first + g) * w2;
#pragma unroll
          for (int k = 0; k < 5; ++k) v[g][k] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(in_rsrc, coff[k], soff, 0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          bool zr;
          const int sr = map_coord(r_first - qb + min(qb + g, nrows - 1), H, pa, pb, mode, zr);
          if (zr) zmask |= 1u << (8 + g);
          const int soff = __builtin_amdgcn_readfirstlane(sr * w2);
#pragma unroll
          for (int k = 0; k < 5; ++k) v[g][k] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(in_rsrc, coff[k], soff, 0);
        }
      }
      if (sg > 0) __syncthreads();  // every wave is done reading the previous segment's window
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (qb + g < nrows) {  // wave-uniform
          unsigned c[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            c[k] = v[g][k];
            if (ZERO && (((zmask >> k) & 1u) || ((zmask >> (8 + g)) & 1u))) c[k] = 0;
          }
          wp[g * PQ] = make_uint2(c[0] | (c[1] << 16), c[2] | (c[3] << 16));
          if (second) wp[g * PQ + 64] = make_uint2(c[1] | (c[2] << 16), c[3] | (c[4] << 16));
        }
      }
    }
    __syncthreads();
    if (sg == 0) stamp(dbg, 1);

    // ---- accumulate: taps of the segment in row-major order (hand-written loop above) -----------------
    // the table pointer itself (kernel-argument derived, provably uniform): hipcc then fetches the ltap words
    // with scalar loads; a pointer rebuilt from an integer would go through per-lane flat loads
    if constexpr (ACC == DIB_ACC_FP32) tap_loop_fp32<R>(acc32, reinterpret_cast<const unsigned *>(tab + table_ltaps_off(K)), t0, n, lane_addr);
    else if constexpr (R == 8) tap_loop_r8<ACC == DIB_ACC_FMA16>(acc, ltaps, t0, n, lane_addr);
    else tap_loop_r4(acc, ltaps, t0, n, lane_addr);
    if (sg == 0) stamp(dbg, 2);
  }

  // ---- store: lane owns columns x0 + lane + 64k ------------------------------------------------
  {
    const __amdgpu_buffer_rsrc_t out_rsrc = plane_rsrc(d.out, ch, H, W);
    const int xr = W - x0 - lane;  // columns remaining for this lane
    const unsigned voff = 2u * (unsigned)(x0 + lane);
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int y = y0 + wave * R + i;
      if (y < H) {  // wave-uniform
        const int soff = y * w2;
        // halves are extracted with integer ops: hipcc (ROCm 7.2) stored the LOW half twice when the
        // high element of the fp16x2 accumulator was taken with a vector subscript
        unsigned a = __builtin_bit_cast(unsigned, acc[i][0]), b = __builtin_bit_cast(unsigned, acc[i][1]);
        if constexpr (ACC == DIB_ACC_FP32) {
          a = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)acc32[i][0]) |
              ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)acc32[i][1]) << 16);
          b = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)acc32[i][2]) |
              ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)acc32[i][3]) << 16);
        }
        if (xr > 0) __builtin_amdgcn_raw_buffer_store_b16((short)(a & 0xffffu), out_rsrc, voff, soff, 0);
        if (xr > 64) __builtin_amdgcn_raw_buffer_store_b16((short)(a >> 16), out_rsrc, voff + 128u, soff, 0);
        if (xr > 128) __builtin_amdgcn_raw_buffer_store_b16((short)(b & 0xffffu), out_rsrc, voff + 256u, soff, 0);
        if (xr > 192) __builtin_amdgcn_raw_buffer_store_b16((short)(b >> 16), out_rsrc, voff + 384u, soff, 0);
      }
    }
  }
  stamp(dbg, 3);
  if (dbg && threadIdx.x == 0) {
    dbg[(size_t)blockIdx.x * 8 + 5] = wall_clock64();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dbg[(size_t)blockIdx.x * 8 + 6] = ((unsigned long long)xcc << 32) | hwid;
  }
}

// Grid: one 256-thread workgroup per (image, channel, 256 x 32 tile); four workgroups fit a CU (36 KB
// of LDS each) and the hardware dispatcher refills a CU as soon as a workgroup retires.  The tile a
// workgroup takes is NOT blockIdx in the flattened [image][channel][ty][tx] order: see "Tile order"
// at the top of the kernel (per-XCD bands of every image).
// (Measured alternatives, scratch/: a persistent kernel pulling tiles from an XCD-sharded atomic
// queue was 30-60 % slower -- the returning atomics and the per-tile descriptor fetches sit on the
// critical path of every tile, and static striding loses ~20 us to tap-count imbalance; 8-wave
// workgroups double the wave launches, whose rate bounds this kernel at ~32 us for 28,800 waves.
// An L2 look-ahead -- every workgroup touching, one dword per 128-byte line, the window of the tile
// 8..256 positions further down its XCD's list, with the tap loop no longer draining vmcnt -- made
// the kernel 7 % SLOWER at every distance: the fill is not waiting on HBM latency.)
template <int NW, int TPW, int ACC = DIB_ACC_BITEXACT>
__global__ __launch_bounds__(64 * NW, NW) void blur_tiled_f16_kernel(BlurBatch batch, const int *__restrict__ tables, int K,
                                                                     unsigned long long *dbg) {
  extern __shared__ uint2 lds[];
  // Tile order.  Workgroups with equal blockIdx % 8 share an XCD (and its 4 MB L2); hardware hands
  // consecutive ids to different XCDs, so with a flat tile order no two neighbouring tiles ever
  // share an L2 and every halo row comes from HBM again.  Instead each XCD group x walks, image after
  // image, the x-th eighth of that image's tiles: neighbouring tiles run on one XCD close in time
  // (halo rows hit in L2), and every XCD still gets an equal share of every image, so PSFs of
  // different tap counts do not unbalance the XCDs.  (A performance choice only: any tile order
  // gives the same output.)
  int i = -1, local = 0;
  if (batch.xcd_bands) {
    const int x = blockIdx.x & 7, l = blockIdx.x >> 3;
    int cum = 0;
    for (int k = 0; k < batch.n; ++k) {
      const int T = batch.tile_begin[k + 1] - batch.tile_begin[k];
      const int lo = (x * T) >> 3, hi = ((x + 1) * T) >> 3;
      if (i < 0 && l < cum + (hi - lo)) { i = k; local = lo + (l - cum); }
      cum += hi - lo;
    }
    if (i < 0) return;  // the grid is 8 x the longest per-XCD list
  } else {
    const int tile = blockIdx.x;
    i = 0;
#pragma unroll
    for (int k = 1; k < MAX_BATCH; ++k)
      if (k < batch.n && tile >= batch.tile_begin[k]) i = k;
    local = tile - batch.tile_begin[i];
  }
  const ImageDesc &d = batch.img[i];
  const int per_ch = d.tiles_x * d.tiles_y;   // tiles_y counts STACKS of TPW vertically adjacent tiles
  const int ch = local / per_ch;
  local -= ch * per_ch;
  const int sy = local / d.tiles_x, tx = local - sy * d.tiles_x;
  const int *tab = tables + (size_t)d.table * table_words(K);
  const bool zero = pad_mode_for(K, d.H, d.W) == PAD_ZERO;
  for (int rep = 0; rep < TPW; ++rep) {
    const int ty = sy * TPW + rep;
    if (ty * TH >= d.H) break;
    if (rep > 0) __syncthreads();  // the previous tile's window reads are over
    if (zero) blur_tile_f16<NW, true, ACC>(d, tab, K, ch, tx, ty, lds, dbg);
    else blur_tile_f16<NW, false, ACC>(d, tab, K, ch, tx, ty, lds, dbg);
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

template <typename T, int ACC>
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
  float acc32 = 0.f;  // DIB_ACC_FP32 (fp16 images): exact products, fp32 running sum, one final rounding
  for (int t = 0; t < ntaps; ++t) {
    const uint2 tap = taps[t];
    const int r = tap.x >> 8, c = tap.x & 255;
    bool zr, zc;
    const int sy = map_coord(y + pb - r, H, pa, pb, mode, zr);
    const int sx = map_coord(x + pb - c, W, pa, pb, mode, zc);
    V p = (zr || zc) ? V(0) : src[(size_t)sy * W + sx];
    if constexpr (ACC == DIB_ACC_FP32) {
      acc32 = acc32 + (float)p * (float)Arith<T>::weight(tap.y);
    } else if constexpr (ACC == DIB_ACC_FMA16) {
      acc = __builtin_fmaf16(p, Arith<T>::weight(tap.y), acc);   // native fp16 fma: one rounding
    } else {
      V prod = p * Arith<T>::weight(tap.y);
      acc = acc + prod;
    }
  }
  reinterpret_cast<V *>(d.out)[e] = ACC == DIB_ACC_FP32 ? (V)acc32 : acc;
}

}  // namespace dib

using namespace dib;

// Diagnostics only: when set, the tiled kernel records per-workgroup phase stamps (8 x u64 each).
static unsigned long long *g_stamp_buffer = nullptr;
extern "C" void dib_debug_set_stamp_buffer(void *dev_ptr) { g_stamp_buffer = (unsigned long long *)dev_ptr; }
// Tuning knobs (all variants are bit-identical): waves per workgroup (4 or 8), tiles per workgroup (1 or 2)
static int g_nw = 4, g_tpw = 1, g_xcd_bands = 1;
extern "C" void dib_debug_set_tile_order(int xcd_bands) { g_xcd_bands = xcd_bands ? 1 : 0; }
extern "C" void dib_debug_set_variant(int nw, int tpw) { g_nw = (nw == 4) ? 4 : 8; g_tpw = (tpw == 2) ? 2 : 1; }


extern "C" int dib_sparse_blur(const void *const *in_dev, void *const *out_dev, const int *C, const int *H,
                               const int *W, const int *table_index, int B, int dtype, void *tables_dev,
                               int num_tables, int K, int acc_mode, void *stream) {
  if (B < 0 || (B > 0 && (!in_dev || !out_dev || !C || !H || !W || !table_index || !tables_dev))) {
    set_error("dib_sparse_blur: null pointer or negative batch");
    return DIB_EINVAL;
  }
  if (K != 128 && K != 256) { set_error("dib_sparse_blur: K must be 128 or 256, got %d", K); return DIB_EINVAL; }
  if (dtype != DIB_F16 && dtype != DIB_F32) { set_error("dib_sparse_blur: unknown dtype %d", dtype); return DIB_EINVAL; }
  if (acc_mode != DIB_ACC_BITEXACT && acc_mode != DIB_ACC_FP32 && acc_mode != DIB_ACC_FMA16) { set_error("dib_sparse_blur: unknown accumulation mode %d", acc_mode); return DIB_EINVAL; }
  if (acc_mode != DIB_ACC_BITEXACT && dtype != DIB_F16) { set_error("dib_sparse_blur: DIB_ACC_FP32 / DIB_ACC_FMA16 apply to fp16 images only (fp32 images already accumulate in fp32)"); return DIB_EINVAL; }
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
  if (num_tables <= 0) { set_error("dib_sparse_blur: num_tables must be positive"); return DIB_EINVAL; }
  for (int i = 0; i < B; ++i)
    if (table_index[i] >= num_tables) { set_error("dib_sparse_blur: table_index[%d] = %d out of range", i, table_index[i]); return DIB_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    DIB_HIP_CHECK(hipFuncSetAttribute((const void *)blur_tiled_f16_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DIB_HIP_CHECK(hipFuncSetAttribute((const void *)blur_tiled_f16_kernel<4, 1, DIB_ACC_FP32>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DIB_HIP_CHECK(hipFuncSetAttribute((const void *)blur_tiled_f16_kernel<4, 1, DIB_ACC_FMA16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DIB_HIP_CHECK(hipFuncSetAttribute((const void *)blur_tiled_f16_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DIB_HIP_CHECK(hipFuncSetAttribute((const void *)blur_tiled_f16_kernel<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DIB_HIP_CHECK(hipFuncSetAttribute((const void *)blur_tiled_f16_kernel<8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set = true;
  }
  const int tpw = (acc_mode != DIB_ACC_BITEXACT) ? 1 : g_tpw;  // the other modes exist as <4, 1> only
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
      d.tiles_y = (H[i] + TH * tpw - 1) / (TH * tpw);
      d.tile_begin = tiles;
      tiled.tile_begin[tiled.n] = tiles;
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
    tiled.xcd_bands = g_xcd_bands;
    generic.xcd_bands = 0;
    int grid = tiles;
    if (g_xcd_bands) {   // 8 x the longest per-XCD tile list (lists differ by at most one tile per image)
      int longest = 0;
      for (int x = 0; x < 8; ++x) {
        int len = 0;
        for (int k = 0; k < tiled.n; ++k) {
          const int T = (k + 1 < tiled.n ? tiled.tile_begin[k + 1] : tiles) - tiled.tile_begin[k];
          len += (((x + 1) * T) >> 3) - ((x * T) >> 3);
        }
        longest = len > longest ? len : longest;
      }
      grid = 8 * longest;
    }
    if (dtype == DIB_F16) {
      for (int k = tiled.n; k <= MAX_BATCH; ++k) tiled.tile_begin[k] = tiles;
      if (acc_mode == DIB_ACC_FP32) hipLaunchKernelGGL((blur_tiled_f16_kernel<4, 1, DIB_ACC_FP32>), dim3(grid), dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
      else if (acc_mode == DIB_ACC_FMA16) hipLaunchKernelGGL((blur_tiled_f16_kernel<4, 1, DIB_ACC_FMA16>), dim3(grid), dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
      else if (g_nw == 4 && g_tpw == 1) hipLaunchKernelGGL((blur_tiled_f16_kernel<4, 1>), dim3(grid), dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
      else if (g_nw == 4) hipLaunchKernelGGL((blur_tiled_f16_kernel<4, 2>), dim3(grid), dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
      else if (g_tpw == 1) hipLaunchKernelGGL((blur_tiled_f16_kernel<8, 1>), dim3(grid), dim3(512), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
      else hipLaunchKernelGGL((blur_tiled_f16_kernel<8, 2>), dim3(grid), dim3(512), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
    } else {
      hipLaunchKernelGGL((blur_generic_kernel<float, DIB_ACC_BITEXACT>), dim3(gblocks), dim3(256), 0, s, generic, (const int *)tables_dev, K);
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
  g.xcd_bands = 0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DIB_F16)
    hipLaunchKernelGGL((blur_generic_kernel<__half, DIB_ACC_BITEXACT>), dim3(blocks), dim3(256), 0, s, g, (const int *)table_dev, K);
  else if (dtype == DIB_F32)
    hipLaunchKernelGGL((blur_generic_kernel<float, DIB_ACC_BITEXACT>), dim3(blocks), dim3(256), 0, s, g, (const int *)table_dev, K);
  else if (dtype == 2)  // fp16 image, DIB_ACC_FP32 arithmetic
    hipLaunchKernelGGL((blur_generic_kernel<__half, DIB_ACC_FP32>), dim3(blocks), dim3(256), 0, s, g, (const int *)table_dev, K);
  else  // dtype 3: fp16 image, DIB_ACC_FMA16 arithmetic
    hipLaunchKernelGGL((blur_generic_kernel<__half, DIB_ACC_FMA16>), dim3(blocks), dim3(256), 0, s, g, (const int *)table_dev, K);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
