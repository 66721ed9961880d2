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
// What bounds it (measured, scratch/ubench): a packed fp16 instruction occupies a SIMD for 4 cycles
// per wave (1.9 ns at the clock the chip holds; 3.2 ns with one wave alone), the same per-element rate as
// fp32 (MI355X_MICROARCH.md: SIMD-32, v_fma_f32 2 cycles per wave).  The bit-exact contract needs a
// multiply AND an add per 2 pixel-taps: 15.1 M wave-instructions for the BASELINE batch = ~26 us of
// pure VALU time on 1024 SIMDs, above its 16 us HBM time.  Everything else -- window fills, stores,
// launch ramp -- has to hide behind that arithmetic: see "Persistent streaming kernel" below.
#include "dib_common.h"
#include <hip/hip_fp16.h>
#include <mutex>

namespace dib {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int TILE_W = 256;
constexpr int PQ = WIN_PITCH;         // LDS row pitch in 8-byte words (96)
constexpr int TH = 32;                // tile rows
constexpr int LROWS = TH + SEG_ROWS;  // LDS rows per window (48)
constexpr int LDS_BYTES = LROWS * PQ * 8;  // 36,864 B: four workgroups per CU
constexpr int STREAM_LDS_BYTES = LDS_BYTES + 16;  // + the ticket mailbox of the persistent kernel
constexpr int WG_PER_CU = 3;   // persistent kernel: 3 x 8 waves per CU = 6 per SIMD at <= 80 registers per lane
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
// here: with one or two waves per SIMD in their tap loops a wave's own SALU work is on its critical
// path, it is not hidden behind another wave's VALU.  Hence: the weight is read straight from the high
// half of the ltap SGPR (op_sel, no shift + pack), the LDS address is one v_mad_u32_u16 (no scalar mask),
// the loop ends on the borrow of the counter's decrement, and the last tap does not branch around its
// (harmless, zero-padded) look-ahead.
// (Also built and measured: a x3-unrolled loop with THREE data buffers -- LDS look-ahead of two taps,
// `s_waitcnt lgkmcnt(8)` -- that takes its ltap words from a VGPR with v_readlane.  Bit-identical, 3 % slower:
// LDS latency is already covered by one tap of look-ahead.)
// ltap word = byte offset of the tap's source word in the window (low 16) | fp16 weight (high 16).
// Operands: %0-%15 accumulators, %16 byte offset of the next ltap, %17 taps left, %18 A (tap being
// multiplied), %19 B (tap being fetched), %20 C (tap in flight), %21 scalar temp, %22 ltaps, %23 lane base.
//
// HALF variant: a tile whose valid columns all lie in its first 128 (the right-hand edge column of an
// image whose width is not a multiple of 256: 1333 = 5 x 256 + 53) only has the packed registers
// {P[j], P[j+64]}: 4-byte LDS reads, 8 multiplies + 8 adds per tap instead of 16 + 16.
#define DIB_MUL(b) "v_pk_mul_f16 v" #b ", %18, v" #b " op_sel:[1,0] op_sel_hi:[1,1]\n\t"
#define DIB_ADD(b, i) "v_pk_add_f16 %" #i ", %" #i ", v" #b "\n\t"
#define DIB_FMA(b, i) "v_pk_fma_f16 %" #i ", %18, v" #b ", %" #i " op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
#define DIB_MADD_A                                                                                          \
  DIB_MUL(64) DIB_MUL(65) DIB_MUL(66) DIB_MUL(67) DIB_MUL(68) DIB_MUL(69) DIB_MUL(70) DIB_MUL(71) \
  DIB_MUL(72) DIB_MUL(73) DIB_MUL(74) DIB_MUL(75) DIB_MUL(76) DIB_MUL(77) DIB_MUL(78) DIB_MUL(79) \
  DIB_ADD(64, 0) DIB_ADD(65, 1) DIB_ADD(66, 2) DIB_ADD(67, 3) DIB_ADD(68, 4) DIB_ADD(69, 5) DIB_ADD(70, 6) DIB_ADD(71, 7) \
  DIB_ADD(72, 8) DIB_ADD(73, 9) DIB_ADD(74, 10) DIB_ADD(75, 11) DIB_ADD(76, 12) DIB_ADD(77, 13) DIB_ADD(78, 14) DIB_ADD(79, 15)
#define DIB_MADD_B                                                                                          \
  DIB_MUL(80) DIB_MUL(81) DIB_MUL(82) DIB_MUL(83) DIB_MUL(84) DIB_MUL(85) DIB_MUL(86) DIB_MUL(87) \
  DIB_MUL(88) DIB_MUL(89) DIB_MUL(90) DIB_MUL(91) DIB_MUL(92) DIB_MUL(93) DIB_MUL(94) DIB_MUL(95) \
  DIB_ADD(80, 0) DIB_ADD(81, 1) DIB_ADD(82, 2) DIB_ADD(83, 3) DIB_ADD(84, 4) DIB_ADD(85, 5) DIB_ADD(86, 6) DIB_ADD(87, 7) \
  DIB_ADD(88, 8) DIB_ADD(89, 9) DIB_ADD(90, 10) DIB_ADD(91, 11) DIB_ADD(92, 12) DIB_ADD(93, 13) DIB_ADD(94, 14) DIB_ADD(95, 15)
// DIB_ACC_FMA16: the same loop with ONE packed fused multiply-add per register (one rounding per tap
// instead of two): half the tap arithmetic, not the reference's arithmetic.
#define DIB_FMADD_A                                                                                         \
  DIB_FMA(64, 0) DIB_FMA(65, 1) DIB_FMA(66, 2) DIB_FMA(67, 3) DIB_FMA(68, 4) DIB_FMA(69, 5) DIB_FMA(70, 6) DIB_FMA(71, 7) \
  DIB_FMA(72, 8) DIB_FMA(73, 9) DIB_FMA(74, 10) DIB_FMA(75, 11) DIB_FMA(76, 12) DIB_FMA(77, 13) DIB_FMA(78, 14) DIB_FMA(79, 15)
#define DIB_FMADD_B                                                                                         \
  DIB_FMA(80, 0) DIB_FMA(81, 1) DIB_FMA(82, 2) DIB_FMA(83, 3) DIB_FMA(84, 4) DIB_FMA(85, 5) DIB_FMA(86, 6) DIB_FMA(87, 7) \
  DIB_FMA(88, 8) DIB_FMA(89, 9) DIB_FMA(90, 10) DIB_FMA(91, 11) DIB_FMA(92, 12) DIB_FMA(93, 13) DIB_FMA(94, 14) DIB_FMA(95, 15)
// HALF: row i of the tap sits in v[base + i]; it feeds the even accumulators (acc[i][0] = operand 2i)
#define DIB_MADDH_A                                                                                         \
  DIB_MUL(64) DIB_MUL(65) DIB_MUL(66) DIB_MUL(67) DIB_MUL(68) DIB_MUL(69) DIB_MUL(70) DIB_MUL(71) \
  DIB_ADD(64, 0) DIB_ADD(65, 2) DIB_ADD(66, 4) DIB_ADD(67, 6) DIB_ADD(68, 8) DIB_ADD(69, 10) DIB_ADD(70, 12) DIB_ADD(71, 14)
#define DIB_MADDH_B                                                                                         \
  DIB_MUL(80) DIB_MUL(81) DIB_MUL(82) DIB_MUL(83) DIB_MUL(84) DIB_MUL(85) DIB_MUL(86) DIB_MUL(87) \
  DIB_ADD(80, 0) DIB_ADD(81, 2) DIB_ADD(82, 4) DIB_ADD(83, 6) DIB_ADD(84, 8) DIB_ADD(85, 10) DIB_ADD(86, 12) DIB_ADD(87, 14)
#define DIB_FMADDH_A                                                                                        \
  DIB_FMA(64, 0) DIB_FMA(65, 2) DIB_FMA(66, 4) DIB_FMA(67, 6) DIB_FMA(68, 8) DIB_FMA(69, 10) DIB_FMA(70, 12) DIB_FMA(71, 14)
#define DIB_FMADDH_B                                                                                        \
  DIB_FMA(80, 0) DIB_FMA(81, 2) DIB_FMA(82, 4) DIB_FMA(83, 6) DIB_FMA(84, 8) DIB_FMA(85, 10) DIB_FMA(86, 12) DIB_FMA(87, 14)
#define DIB_READ8(base)                                                                                      \
  "v_mad_u32_u16 v96, %19, 1, %23\n\t"                                                                       \
  "ds_read_b64 v[" #base ":" #base "+1], v96\n\tds_read_b64 v[" #base "+2:" #base "+3], v96 offset:768\n\t"   \
  "ds_read_b64 v[" #base "+4:" #base "+5], v96 offset:1536\n\tds_read_b64 v[" #base "+6:" #base "+7], v96 offset:2304\n\t" \
  "ds_read_b64 v[" #base "+8:" #base "+9], v96 offset:3072\n\tds_read_b64 v[" #base "+10:" #base "+11], v96 offset:3840\n\t" \
  "ds_read_b64 v[" #base "+12:" #base "+13], v96 offset:4608\n\tds_read_b64 v[" #base "+14:" #base "+15], v96 offset:5376\n\t"
#define DIB_READ8H(base)                                                                                     \
  "v_mad_u32_u16 v96, %19, 1, %23\n\t"                                                                       \
  "ds_read_b32 v[" #base "], v96\n\tds_read_b32 v[" #base "+1], v96 offset:768\n\t"                           \
  "ds_read_b32 v[" #base "+2], v96 offset:1536\n\tds_read_b32 v[" #base "+3], v96 offset:2304\n\t"            \
  "ds_read_b32 v[" #base "+4], v96 offset:3072\n\tds_read_b32 v[" #base "+5], v96 offset:3840\n\t"            \
  "ds_read_b32 v[" #base "+6], v96 offset:4608\n\tds_read_b32 v[" #base "+7], v96 offset:5376\n\t"
#define DIB_NEXTTAP "s_load_dword %20, %22, %16\n\ts_add_u32 %16, %16, 4\n\t"

// acc[i][0] / acc[i][1] (i = 0..7): the packed fp16 accumulators of this lane's 8 rows x 4 columns.
// DRAIN: wait for every outstanding vector-memory operation before the loop (the one-tile-per-workgroup
// kernel: window loads whose values were never used -- rows past the window's end -- may still be in
// flight, and hipcc is free to have put their destinations into the registers clobbered here; it waits
// before ITS OWN next write to such a register, but not before this asm's).  The streaming kernel keeps
// the NEXT window's loads in flight across the loop; every one of them is consumed afterwards, so none
// can sit in a clobbered register.
template <bool FUSED, bool HALF, bool DRAIN>
__device__ __forceinline__ void tap_loop_r8(h2 (&acc)[8][2], unsigned long long ltaps, int t0, int n, unsigned lane_addr) {
  // cnt = taps left minus one: the borrow of its decrement ends the loop (n >= 1 in every segment)
  unsigned toff = (unsigned)__builtin_amdgcn_readfirstlane(t0 * 4), cnt = (unsigned)__builtin_amdgcn_readfirstlane(n - 1);
  unsigned sA, sB, sC, st;
  unsigned a[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[2 * i] = __builtin_bit_cast(unsigned, acc[i][0]); a[2 * i + 1] = __builtin_bit_cast(unsigned, acc[i][1]); }
#define DIB_R8_ASM(WAIT0, RD_A, RD_B, ARITH_A, ARITH_B) \
  asm volatile( \
      /* prologue: ltap[t0] -> B, ltap[t0+1] -> C, data of tap t0 -> buffer A */ \
      "s_load_dword %19, %22, %16\n\ts_add_u32 %16, %16, 4\n\t" DIB_NEXTTAP \
      WAIT0 RD_A \
      "Ldib_loop%=:\n\t" \
      /* ---- tap in buffer A ---- */ \
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %18, %19\n\ts_mov_b32 %19, %20\n\t" \
      RD_B DIB_NEXTTAP \
      ARITH_A \
      "s_sub_u32 %17, %17, 1\n\ts_cbranch_scc1 Ldib_done%=\n\t" \
      /* ---- tap in buffer B ---- */ \
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %18, %19\n\ts_mov_b32 %19, %20\n\t" \
      RD_A DIB_NEXTTAP \
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
#define DIB_W_DRAIN "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
#define DIB_W_KEEP "s_waitcnt lgkmcnt(0)\n\t"
  if constexpr (HALF) {
    if constexpr (FUSED) {
      if constexpr (DRAIN) { DIB_R8_ASM(DIB_W_DRAIN, DIB_READ8H(64), DIB_READ8H(80), DIB_FMADDH_A, DIB_FMADDH_B); }
      else { DIB_R8_ASM(DIB_W_KEEP, DIB_READ8H(64), DIB_READ8H(80), DIB_FMADDH_A, DIB_FMADDH_B); }
    } else {
      if constexpr (DRAIN) { DIB_R8_ASM(DIB_W_DRAIN, DIB_READ8H(64), DIB_READ8H(80), DIB_MADDH_A, DIB_MADDH_B); }
      else { DIB_R8_ASM(DIB_W_KEEP, DIB_READ8H(64), DIB_READ8H(80), DIB_MADDH_A, DIB_MADDH_B); }
    }
  } else {
    if constexpr (FUSED) {
      if constexpr (DRAIN) { DIB_R8_ASM(DIB_W_DRAIN, DIB_READ8(64), DIB_READ8(80), DIB_FMADD_A, DIB_FMADD_B); }
      else { DIB_R8_ASM(DIB_W_KEEP, DIB_READ8(64), DIB_READ8(80), DIB_FMADD_A, DIB_FMADD_B); }
    } else {
      if constexpr (DRAIN) { DIB_R8_ASM(DIB_W_DRAIN, DIB_READ8(64), DIB_READ8(80), DIB_MADD_A, DIB_MADD_B); }
      else { DIB_R8_ASM(DIB_W_KEEP, DIB_READ8(64), DIB_READ8(80), DIB_MADD_A, DIB_MADD_B); }
    }
  }
#undef DIB_W_DRAIN
#undef DIB_W_KEEP
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
#undef DIB_MADDH_A
#undef DIB_MADDH_B
#undef DIB_FMADDH_A
#undef DIB_FMADDH_B
#undef DIB_READ8
#undef DIB_READ8H
#undef DIB_NEXTTAP

// ---- the same loop for 4 rows per lane (8-wave workgroups of the persistent kernel) -----------------------------------
// buffers v[64:71] / v[72:79], LDS address v48.  Operands: %0-%7 accumulators, %8 byte offset of the next ltap,
// %9 taps left, %10 A, %11 B, %12 C, %13 scalar temp, %14 ltaps, %15 lane base.
#define DIB4_MUL(b) "v_pk_mul_f16 v" #b ", %10, v" #b " op_sel:[1,0] op_sel_hi:[1,1]\n\t"
#define DIB4_ADD(b, i) "v_pk_add_f16 %" #i ", %" #i ", v" #b "\n\t"
#define DIB4_FMA(b, i) "v_pk_fma_f16 %" #i ", %10, v" #b ", %" #i " op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
#define DIB4_MADD_A DIB4_MUL(32) DIB4_MUL(33) DIB4_MUL(34) DIB4_MUL(35) DIB4_MUL(36) DIB4_MUL(37) DIB4_MUL(38) DIB4_MUL(39) \
  DIB4_ADD(32, 0) DIB4_ADD(33, 1) DIB4_ADD(34, 2) DIB4_ADD(35, 3) DIB4_ADD(36, 4) DIB4_ADD(37, 5) DIB4_ADD(38, 6) DIB4_ADD(39, 7)
#define DIB4_MADD_B DIB4_MUL(40) DIB4_MUL(41) DIB4_MUL(42) DIB4_MUL(43) DIB4_MUL(44) DIB4_MUL(45) DIB4_MUL(46) DIB4_MUL(47) \
  DIB4_ADD(40, 0) DIB4_ADD(41, 1) DIB4_ADD(42, 2) DIB4_ADD(43, 3) DIB4_ADD(44, 4) DIB4_ADD(45, 5) DIB4_ADD(46, 6) DIB4_ADD(47, 7)
#define DIB4_FMADD_A DIB4_FMA(32, 0) DIB4_FMA(33, 1) DIB4_FMA(34, 2) DIB4_FMA(35, 3) DIB4_FMA(36, 4) DIB4_FMA(37, 5) DIB4_FMA(38, 6) DIB4_FMA(39, 7)
#define DIB4_FMADD_B DIB4_FMA(40, 0) DIB4_FMA(41, 1) DIB4_FMA(42, 2) DIB4_FMA(43, 3) DIB4_FMA(44, 4) DIB4_FMA(45, 5) DIB4_FMA(46, 6) DIB4_FMA(47, 7)
#define DIB4_MADDH_A DIB4_MUL(32) DIB4_MUL(33) DIB4_MUL(34) DIB4_MUL(35) DIB4_ADD(32, 0) DIB4_ADD(33, 2) DIB4_ADD(34, 4) DIB4_ADD(35, 6)
#define DIB4_MADDH_B DIB4_MUL(40) DIB4_MUL(41) DIB4_MUL(42) DIB4_MUL(43) DIB4_ADD(40, 0) DIB4_ADD(41, 2) DIB4_ADD(42, 4) DIB4_ADD(43, 6)
#define DIB4_FMADDH_A DIB4_FMA(32, 0) DIB4_FMA(33, 2) DIB4_FMA(34, 4) DIB4_FMA(35, 6)
#define DIB4_FMADDH_B DIB4_FMA(40, 0) DIB4_FMA(41, 2) DIB4_FMA(42, 4) DIB4_FMA(43, 6)
#define DIB4_READ(base)                                                                                      \
  "v_mad_u32_u16 v48, %11, 1, %15\n\t"                                                                       \
  "ds_read_b64 v[" #base ":" #base "+1], v48\n\tds_read_b64 v[" #base "+2:" #base "+3], v48 offset:768\n\t"   \
  "ds_read_b64 v[" #base "+4:" #base "+5], v48 offset:1536\n\tds_read_b64 v[" #base "+6:" #base "+7], v48 offset:2304\n\t"
#define DIB4_READH(base)                                                                                     \
  "v_mad_u32_u16 v48, %11, 1, %15\n\t"                                                                       \
  "ds_read_b32 v[" #base "], v48\n\tds_read_b32 v[" #base "+1], v48 offset:768\n\t"                           \
  "ds_read_b32 v[" #base "+2], v48 offset:1536\n\tds_read_b32 v[" #base "+3], v48 offset:2304\n\t"
#define DIB4_NEXTTAP "s_load_dword %12, %14, %8\n\ts_add_u32 %8, %8, 4\n\t"
template <bool FUSED, bool HALF>
__device__ __forceinline__ void tap_loop_r4(h2 (&acc)[4][2], unsigned long long ltaps, int t0, int n, unsigned lane_addr) {
  unsigned toff = (unsigned)__builtin_amdgcn_readfirstlane(t0 * 4), cnt = (unsigned)__builtin_amdgcn_readfirstlane(n - 1);
  unsigned sA, sB, sC, st;
  unsigned a[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[2 * i] = __builtin_bit_cast(unsigned, acc[i][0]); a[2 * i + 1] = __builtin_bit_cast(unsigned, acc[i][1]); }
#define DIB_R4_ASM(RD_A, RD_B, ARITH_A, ARITH_B) \
  asm volatile( \
      "s_load_dword %11, %14, %8\n\ts_add_u32 %8, %8, 4\n\t" DIB4_NEXTTAP \
      "s_waitcnt lgkmcnt(0)\n\t" RD_A \
      "Ldib4_loop%=:\n\t" \
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %10, %11\n\ts_mov_b32 %11, %12\n\t" \
      RD_B DIB4_NEXTTAP \
      ARITH_A \
      "s_sub_u32 %9, %9, 1\n\ts_cbranch_scc1 Ldib4_done%=\n\t" \
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %10, %11\n\ts_mov_b32 %11, %12\n\t" \
      RD_A DIB4_NEXTTAP \
      ARITH_B \
      "s_sub_u32 %9, %9, 1\n\ts_cbranch_scc0 Ldib4_loop%=\n\t" \
      "Ldib4_done%=:\n\t" \
      "s_waitcnt lgkmcnt(0)" \
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+s"(toff), "+s"(cnt), \
        "=&s"(sA), "=&s"(sB), "=&s"(sC), "=&s"(st) \
      : "s"(ltaps), "v"(lane_addr) \
      : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "scc", \
        "memory")
  if constexpr (HALF) {
    if constexpr (FUSED) { DIB_R4_ASM(DIB4_READH(32), DIB4_READH(40), DIB4_FMADDH_A, DIB4_FMADDH_B); }
    else { DIB_R4_ASM(DIB4_READH(32), DIB4_READH(40), DIB4_MADDH_A, DIB4_MADDH_B); }
  } else {
    if constexpr (FUSED) { DIB_R4_ASM(DIB4_READ(32), DIB4_READ(40), DIB4_FMADD_A, DIB4_FMADD_B); }
    else { DIB_R4_ASM(DIB4_READ(32), DIB4_READ(40), DIB4_MADD_A, DIB4_MADD_B); }
  }
#undef DIB_R4_ASM
#pragma unroll
  for (int i = 0; i < 4; ++i) { acc[i][0] = __builtin_bit_cast(h2, a[2 * i]); acc[i][1] = __builtin_bit_cast(h2, a[2 * i + 1]); }
}

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

// Scalar loads of table words, written out: inside the persistent kernel's loop hipcc will not use the scalar
// cache for them (stores, atomics and asm statements in the loop count as possible clobbers of *tables), and its
// vector-load fallback waits with vmcnt(0) -- on the ticket in flight and on every look-ahead load.
__device__ __forceinline__ unsigned sload_u32(const void *p) {
  unsigned r;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"((unsigned long long)p));
  return r;
}
__device__ __forceinline__ uint4 sload_u128(const void *p) {
  unsigned __int128 r;
  asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"((unsigned long long)p));
  return make_uint4((unsigned)r, (unsigned)(r >> 32), (unsigned)(r >> 64), (unsigned)(r >> 96));
}

constexpr int NW = 4;                       // one-tile-per-workgroup kernel: waves per workgroup
constexpr int R = TH / NW;                  //   rows per lane (8)
constexpr int G = (LROWS + NW - 1) / NW;    //   LDS rows a wave fills (12): all of them in ONE batch of loads
constexpr int SNW = 8;                      // persistent kernel: waves per workgroup
constexpr int SR = TH / SNW;                //   rows per lane (4)
constexpr int SG = LROWS / SNW;             //   LDS rows a wave fills (6)

// The window of one (tile, tap segment) as a wave sees it: what to load, and later what to write to LDS.
struct Window {
  int t0, n;          // taps of the segment
  int rf, rl, cmin, cmax;
};
__device__ __forceinline__ Window window_of(const uint4 seg) {
  Window w;
  w.t0 = seg.x; w.n = (int)(seg.y - seg.x);
  w.rf = seg.z >> 8; w.rl = seg.z & 255; w.cmin = seg.w >> 8; w.cmax = seg.w & 255;
  return w;
}

// Branch-free form of map_coord (dib_common.h) for the fill below: same mapping, `zero` = the read is a zero fill.
__device__ __forceinline__ int map_coord_sel(int s, int n, int pa, int pb, int mode, bool &zero) {
  s = (s == -pa) ? n + pb : s;                                    // torch.roll's wrap row / column
  const int a = s < 0 ? -s : s;
  const int refl = a > n - 1 ? 2 * (n - 1) - a : a;
  const int m = mode == PAD_REFLECT ? refl : s;
  zero = mode == PAD_ZERO && (s < 0 || s > n - 1);
  return min(max(m, 0), n - 1);
}

// ---- fill, part 1: issue this wave's G x 5 window loads ---------------------------------------------------------
// Per LDS row the five values P[lane + 64k] (word j and word j+64 share three).  The values stay in flight in v[][];
// zmask collects the zero-fill flags (PAD_ZERO images only): bit k = column k of this lane, bit 8+g = row g of this
// wave.  Address generation is the expensive part of a fill (a CU has ONE scalar unit for its twelve waves), so
// windows that lie inside the image -- all but the border tiles -- take the short forms: one multiply and eleven
// adds for the rows, one vector offset plus compile-time immediates for the columns.
template <int G>
__device__ __forceinline__ void issue_window_loads(unsigned (&v)[G][5], unsigned &zmask, const __amdgpu_buffer_rsrc_t in_rsrc,
                                                   const Window &w, int x0, int y0, int H, int W, int K, int mode, int lane, int qb) {
  const int pb = K / 2 - 1, pa = K / 2;
  const int w2 = W * 2;
  unsigned coff[5];
  int soff[G];
  zmask = 0;
  const int c_first = x0 + pb - w.cmax;  // virtual column of P[0]
  const int r_first = y0 + pb - w.rl;    // virtual row of the window's first LDS row
  const bool zero_mode = mode == PAD_ZERO;
  if (!zero_mode && c_first >= 0 && c_first + 63 + 256 <= W - 1) {
    const unsigned c0 = 2u * (unsigned)(c_first + lane);
#pragma unroll
    for (int k = 0; k < 5; ++k) coff[k] = c0 + 128u * k;
  } else {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      bool z;
      coff[k] = 2u * (unsigned)map_coord_sel(c_first + lane + 64 * k, W, pa, pb, mode, z);
      zmask |= z ? 1u << k : 0u;
    }
  }
  if (!zero_mode && r_first >= 0 && r_first + LROWS - 1 <= H - 1) {
    const int s0 = (r_first + qb) * w2;
#pragma unroll
    for (int g = 0; g < G; ++g) soff[g] = s0 + g * w2;
  } else {
    const int nrows = TH + (w.rl - w.rf);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bool zr;
      // rows past the window's end repeat its last row: loaded and written, read by no tap
      const int sr = map_coord_sel(r_first + min(qb + g, nrows - 1), H, pa, pb, mode, zr);
      zmask |= zr ? 1u << (8 + g) : 0u;
      soff[g] = sr * w2;
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int so = __builtin_amdgcn_readfirstlane(soff[g]);
#pragma unroll
    for (int k = 0; k < 5; ++k) v[g][k] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(in_rsrc, coff[k], so, 0);
  }
}

// ---- fill, part 2: the loaded values -> LDS words {P[j], P[j+64], P[j+128], P[j+192]} ---------------------
// word j from c0..c3, word j+64 from c1..c4 (only lanes below the segment's column extent have one).
// wp: LDS byte address of this lane's word in this wave's first row.
typedef unsigned uvec2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uvec2_t lds_u2;
template <bool MASKED, int G>
__device__ __forceinline__ void write_window_rows(unsigned wp, unsigned (&v)[G][5], unsigned zmask, bool second) {
#pragma unroll
  for (int g = 0; g < G; ++g) {
    // The values were loaded an iteration ago and must stay 60 separate registers in flight until here: without
    // this fence hipcc hoists the packing below across the loop's back edge to right behind the loads (halving its
    // live registers) and waits for every load there -- before the tap loop the loads were meant to hide behind.
    asm volatile("" : "+v"(v[g][0]), "+v"(v[g][1]), "+v"(v[g][2]), "+v"(v[g][3]), "+v"(v[g][4]));
    if (MASKED) {
#pragma unroll
      for (int k = 0; k < 5; ++k)
        if (((zmask >> k) & 1u) || ((zmask >> (8 + g)) & 1u)) v[g][k] = 0;
    }
    const unsigned w0 = v[g][0] | (v[g][1] << 16), w1 = v[g][2] | (v[g][3] << 16);
    *(lds_u2 *)(size_t)(wp + (unsigned)(g * PQ * 8)) = uvec2_t{w0, w1};
    // {c1, c2} and {c3, c4}: funnel shifts of the words above (one VALU instruction each)
    if (second)
      *(lds_u2 *)(size_t)(wp + (unsigned)((g * PQ + 64) * 8)) = uvec2_t{__builtin_amdgcn_alignbit(w1, w0, 16), __builtin_amdgcn_alignbit(v[g][4], w1, 16)};
  }
}
template <int G>
__device__ __forceinline__ void write_window(unsigned wp, unsigned (&v)[G][5], unsigned zmask, bool second) {
  if (__builtin_amdgcn_ballot_w64(zmask != 0) != 0) write_window_rows<true, G>(wp, v, zmask, second);   // PAD_ZERO images only
  else write_window_rows<false, G>(wp, v, zmask, second);
}

// ---- store: lane owns columns x0 + lane + 64k ---------------------------------------------------------
// No execution masks: the plane's buffer descriptor covers exactly H * W * 2 bytes and the hardware drops a buffer
// store whose per-lane offset lies past that range, so lanes whose column is outside the image get an out-of-range
// offset once per tile.  (The scalar row offset takes no part in the range check: rows below the image are skipped.)
template <int ACC, int R>
__device__ __forceinline__ void store_tile(const h2 (&acc)[R][2], const float (&acc32)[R][4], void *out_base, int ch, int H, int W, int x0,
                                           int y0, int lane, int wave) {
  const __amdgpu_buffer_rsrc_t out_rsrc = plane_rsrc(out_base, ch, H, W);
  const int w2 = W * 2;
  const int xr = W - x0 - lane;  // columns remaining for this lane
  const unsigned voff = 2u * (unsigned)(x0 + lane), oob = 0x7ffffff0u;
  const unsigned vo0 = xr > 0 ? voff : oob, vo1 = xr > 64 ? voff + 128u : oob, vo2 = xr > 128 ? voff + 256u : oob,
                 vo3 = xr > 192 ? voff + 384u : oob;
  const int s0 = (y0 + wave * R) * w2;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    if (y0 + wave * R + i >= H) break;   // wave-uniform
    const int soff = s0 + i * w2;
    // halves are extracted with integer ops: hipcc (ROCm 7.2) stored the LOW half twice when the
    // high element of the fp16x2 accumulator was taken with a vector subscript
    unsigned a = __builtin_bit_cast(unsigned, acc[i][0]), b = __builtin_bit_cast(unsigned, acc[i][1]);
    if constexpr (ACC == DIB_ACC_FP32) {
      a = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)acc32[i][0]) |
          ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)acc32[i][1]) << 16);
      b = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)acc32[i][2]) |
          ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)acc32[i][3]) << 16);
    }
    __builtin_amdgcn_raw_buffer_store_b16((short)(a & 0xffffu), out_rsrc, vo0, soff, 0);
    __builtin_amdgcn_raw_buffer_store_b16((short)(a >> 16), out_rsrc, vo1, soff, 0);
    __builtin_amdgcn_raw_buffer_store_b16((short)(b & 0xffffu), out_rsrc, vo2, soff, 0);
    __builtin_amdgcn_raw_buffer_store_b16((short)(b >> 16), out_rsrc, vo3, soff, 0);
  }
}

// =============================================================================================================
// One-tile-per-workgroup kernel: one workgroup = one (image, channel, 256 x 32 tile); fill -> taps -> store, the
// latencies of one workgroup's phases hidden only by the other three workgroups of its CU.  Kept as the second
// tiled implementation (DIB_ACC_FP32 runs on it; the parity tests compare it with the streaming kernel).
// =============================================================================================================
template <int ACC>
__device__ __forceinline__ void blur_tile_f16(const ImageDesc &d, const int *__restrict__ tab, int K, int ch, int tx,
                                              int ty, uint2 *lds, unsigned long long *dbg) {
#pragma clang fp contract(off)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = d.H, W = d.W;
  const int mode = pad_mode_for(K, H, W);
  const int nsegs = tab[HDR_NSEGS];
  const uint4 *segs = reinterpret_cast<const uint4 *>(tab + table_segs_off(K));
  const unsigned long long la = (unsigned long long)(tab + table_ltaps_off(K));
  const unsigned long long ltaps = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(la >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((unsigned)la);
  const int x0 = tx * TILE_W, y0 = ty * TH;
  const __amdgpu_buffer_rsrc_t in_rsrc = plane_rsrc(d.in, ch, H, W);

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
  const unsigned wp = lds0 + (unsigned)(qb * PQ + lane) * 8u;

  stamp(dbg, 0);
  for (int sg = 0; sg < nsegs; ++sg) {
    const Window w = window_of(segs[sg]);
    unsigned v[G][5];
    unsigned zmask;
    issue_window_loads(v, zmask, in_rsrc, w, x0, y0, H, W, K, mode, lane, qb);
    if (sg > 0) __syncthreads();  // every wave is done reading the previous segment's window
    write_window(wp, v, zmask, lane < w.cmax - w.cmin);
    __syncthreads();
    if (sg == 0) stamp(dbg, 1);
    // the table pointer itself (kernel-argument derived, provably uniform): hipcc then fetches the ltap words
    // with scalar loads; a pointer rebuilt from an integer would go through per-lane flat loads
    if constexpr (ACC == DIB_ACC_FP32) tap_loop_fp32<R>(acc32, reinterpret_cast<const unsigned *>(tab + table_ltaps_off(K)), w.t0, w.n, lane_addr);
    else tap_loop_r8<ACC == DIB_ACC_FMA16, false, true>(acc, ltaps, w.t0, w.n, lane_addr);
    if (sg == 0) stamp(dbg, 2);
  }
  store_tile<ACC, R>(acc, acc32, d.out, ch, H, W, x0, y0, lane, wave);
  stamp(dbg, 3);
}

// Tile order.  Workgroups with equal blockIdx % 8 share an XCD (and its 4 MB L2); hardware hands consecutive ids
// to different XCDs, so with a flat tile order no two neighbouring tiles ever share an L2 and every halo row comes
// from HBM again (measured 1.85 x the algorithmic traffic).  Instead the tiles of every image are cut into eight
// contiguous bands, band x of every image forming XCD x's LIST: neighbouring tiles run on one XCD close in time
// (halo rows hit in L2), and every XCD still gets an equal share of every image, so PSFs of different tap counts do
// not unbalance the XCDs.  (A performance choice only: any tile order gives the same output.)
// list_lookup: entry t of list x -> (image, tile index inside the image); false past the end of the list.
__device__ __forceinline__ bool list_lookup(const BlurBatch &batch, int x, int t, int &img, int &local) {
  for (int k = 0; k < batch.n; ++k) {
    const int T = batch.tile_begin[k + 1] - batch.tile_begin[k];
    const int lo = (x * T) >> 3, hi = ((x + 1) * T) >> 3;
    if (t < hi - lo) { img = k; local = lo + t; return true; }
    t -= hi - lo;
  }
  return false;
}

template <int ACC = DIB_ACC_BITEXACT>
__global__ __launch_bounds__(64 * NW, NW) void blur_tiled_f16_kernel(BlurBatch batch, const int *__restrict__ tables, int K,
                                                                     unsigned long long *dbg) {
  extern __shared__ uint2 lds[];
  int i, local;
  if (batch.xcd_bands) {
    if (!list_lookup(batch, blockIdx.x & 7, blockIdx.x >> 3, i, local)) return;  // the grid is 8 x the longest list
  } else {
    const int tile = blockIdx.x;
    i = 0;
#pragma unroll
    for (int k = 1; k < MAX_BATCH; ++k)
      if (k < batch.n && tile >= batch.tile_begin[k]) i = k;
    local = tile - batch.tile_begin[i];
  }
  const ImageDesc &d = batch.img[i];
  const int per_ch = d.tiles_x * d.tiles_y;
  const int ch = local / per_ch;
  local -= ch * per_ch;
  const int ty = local / d.tiles_x, tx = local - ty * d.tiles_x;
  blur_tile_f16<ACC>(d, tables + (size_t)d.table * table_words(K), K, ch, tx, ty, lds, dbg);
}

// =============================================================================================================
// Persistent streaming kernel (the default for fp16 images): 3 workgroups per CU live for the whole launch and
// walk their share of the eight per-XCD tile lists (see "Tile order").  What it buys over one tile per workgroup:
//   * the window loads of tile n+1 are ISSUED before the tap loop of tile n and land while it runs (60 values in
//     flight per lane, registers only: LDS keeps one window).  A wave then spends its life inside tap loops; fills
//     cost their issue slots, not their latency -- from HBM as from L2;
//   * 768 workgroup launches instead of 3600 (the dispatcher starts ~1 wave/ns: 14,400 waves were 16 us of
//     dispatch) and no per-tile relaunch gap;
//   * stores are deferred by one tile and go out in FRONT of the look-ahead loads (see below);
//   * tiles whose valid columns fit in 128 (the right-hand edge of 1333 = 5 x 256 + 53) run the HALF tap loop.
// Schedule: STATIC.  Workgroup b serves list b % 8 (workgroups with equal b % 8 share an XCD and its L2 under the
// round-robin placement the hardware uses; a speed matter only -- every entry of every list is served exactly once
// whatever the placement) and takes entries b / 8, b / 8 + S, b / 8 + 2 S, ... with S = gridDim / 8.  Lists are in
// descriptor order, heaviest image first (blur_functions hands them over that way), so every workgroup gets a
// heavy-to-light spread and the odd extra tile is a light one.  (Built and measured first: dynamic tickets from
// per-XCD atomic heads, drawn two tiles ahead.  A returning device-scope atomic on this path takes ~10 us under the
// kernel's own load -- twice a tap loop -- so every tile waited for its ticket: 139 us per launch against 96 us for
// the same code with static tickets.)
// Per-tile bookkeeping is kept off the scalar unit (ONE per CU, shared by its twelve waves and by the 7 scalar
// instructions of every tap): the image a tile belongs to changes ~8 times in a workgroup's life, so its descriptor,
// table header and first segment sit in SGPRs (ImageCtx) and are re-read only then; tile -> (channel, row, column)
// uses host-computed reciprocals instead of integer division.
// =============================================================================================================
// entry of an image's tile index -> channel and origin (reciprocals from the host: no integer division)
__device__ __forceinline__ void tile_geom(const ImageDesc &d, int local, int &ch, int &x0, int &y0) {
  const int per_ch = d.tiles_x * d.tiles_y;
  // inv == 0 stands for a divisor of 1 (its reciprocal 2^32 does not fit)
  ch = d.inv_per_ch ? (int)__umulhi((unsigned)local, d.inv_per_ch) : local;
  local -= ch * per_ch;
  const int ty = d.inv_tiles_x ? (int)__umulhi((unsigned)local, d.inv_tiles_x) : local, tx = local - ty * d.tiles_x;
  x0 = tx * TILE_W; y0 = ty * TH;
}

template <int ACC>
__global__ __launch_bounds__(64 * SNW, WG_PER_CU * SNW / 4) void blur_stream_f16_kernel(BlurBatch batch, QueueInfo qi, const int *__restrict__ tables, int K,
                                                                             unsigned long long *dbg, int prio) {
#pragma clang fp contract(off)
  extern __shared__ uint2 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
  const unsigned lane_addr = lds0 + (unsigned)((wave * SR) * PQ + lane) * 8u;
  const int qb = wave * SG;
  const unsigned wp = lds0 + (unsigned)(qb * PQ + lane) * 8u;

  const int list = (int)(blockIdx.x & 7u);
  const int stride = (int)(gridDim.x >> 3);          // the host launches a multiple of 8 workgroups
  const int len = qi.len[list];
  const int j = (int)(blockIdx.x >> 3);             // this workgroup's column in the schedule
  // Round r of the schedule hands entries [r * stride, (r + 1) * stride) to the workgroups of the list, in
  // alternating direction ("snake"): entries are sorted heavy to light, so a workgroup that got the heaviest entry of
  // one round gets the lightest of the next, and the sums even out.
  int round = 0;
  auto entry_of = [&](int r) { return r * stride + ((r & 1) ? stride - 1 - j : j); };
  int entry = entry_of(0);                          // list entry of the tile whose window is (about to be) in flight
  if (entry >= len) return;

  // Only INDICES travel from tile to tile (image, tile inside the image): whatever a phase needs of an image it
  // re-reads from the kernel arguments through the scalar cache.  A context kept in registers instead (pointers,
  // shapes, table header, first segment: ~50 SGPRs for the three tiles in flight) spilled into VGPR lanes and paid a
  // v_readlane in front of most instructions of the fill, store and issue phases.
  int lk = 0, l_lo = 0, l_hi = 0, l_first = 0;      // look-ahead cursor: image lk holds list entries [l_lo, l_hi)
  auto advance_to = [&](int e) {                    // entries only grow, so images only advance
    for (;;) {
      const int T = batch.tile_begin[lk + 1] - batch.tile_begin[lk];
      l_first = (list * T) >> 3;
      l_hi = l_lo + ((((list + 1) * T) >> 3) - l_first);
      if (e < l_hi) break;
      l_lo = l_hi; ++lk;
    }
  };

  unsigned v[SG][5];
  unsigned zmask = 0;
  h2 acc[SR][2], out[SR][2];
#pragma unroll
  for (int i = 0; i < SR; ++i) { acc[i][0] = h2{0, 0}; acc[i][1] = h2{0, 0}; out[i][0] = h2{0, 0}; out[i][1] = h2{0, 0}; }
  const float none32[SR][4] = {};
  Window win;                                       // the window whose loads are in flight in v[][]

  // loads of (image k, tile local, segment sg) -> v[][]; leaves the segment's description in `win`
  auto issue = [&](int k, int local, int sg) {
    const ImageDesc &d = batch.img[k];
    int ch, x0, y0;
    tile_geom(d, local, ch, x0, y0);
    const int *tab = tables + (size_t)d.table * table_words(K);
    win = window_of(sload_u128(reinterpret_cast<const uint4 *>(tab + table_segs_off(K)) + sg));
    issue_window_loads(v, zmask, plane_rsrc(d.in, ch, d.H, d.W), win, x0, y0, d.H, d.W, K, pad_mode_for(K, d.H, d.W), lane, qb);
  };

  // ---- prologue: first tile, its first window in flight ------------------------------------------------------------
  advance_to(entry);
  int ck = lk, clocal = l_first + (entry - l_lo);   // tile being computed
  int pk = 0, plocal = 0;                           // tile waiting to be stored
  bool have_prev = false;
  issue(ck, clocal, 0);

  // diagnostics (dbg == nullptr in every product launch): shader cycles this workgroup's wave 0 spent per phase
  unsigned long long ph_fill = 0, ph_store = 0, ph_issue = 0, ph_taps = 0, t_mark = 0, ntiles = 0;
  const unsigned long long t_begin = dbg ? __builtin_readcyclecounter() : 0;
  auto lap = [&](unsigned long long &bucket) {
    if (dbg) { const unsigned long long now = __builtin_readcyclecounter(); bucket += now - t_mark; t_mark = now; }
  };
  t_mark = t_begin;

  // Stores are DEFERRED by one tile: the results of tile n-1 wait in out[][] and are stored behind barrier B of tile n,
  // in FRONT of the look-ahead loads.  Vector-memory operations retire in issue order (one vmcnt for loads and stores):
  // stores issued behind the look-ahead loads would have to complete (write acknowledgement) before the next window
  // can be written; issued in front of them they have a whole tap loop to drain.
  while (true) {
    const ImageDesc &cd = batch.img[ck];
    const int *c_tab = tables + (size_t)cd.table * table_words(K);
    const int c_nsegs = (int)sload_u32(c_tab + HDR_NSEGS);
    const unsigned long long la = (unsigned long long)(c_tab + table_ltaps_off(K));
    const unsigned long long ltaps = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(la >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((unsigned)la);
    int c_ch, c_x0, c_y0;
    tile_geom(cd, clocal, c_ch, c_x0, c_y0);
    const bool half = cd.W - c_x0 <= 128;           // every valid column of the tile lies in its first 128
    const int next_entry = entry_of(round + 1);
    const bool more = next_entry < len;
    int nk = ck, nlocal = clocal;
    for (int sg = 0; sg < c_nsegs; ++sg) {
      const Window w = win;                       // in flight in v[][]
      __syncthreads();                            // A: every wave is done reading the previous window
      write_window(wp, v, zmask, lane < w.cmax - w.cmin);
      __syncthreads();                            // B: window complete
      lap(ph_fill);
      if (sg == 0 && have_prev) {
        const ImageDesc &pd = batch.img[pk];
        int p_ch, p_x0, p_y0;
        tile_geom(pd, plocal, p_ch, p_x0, p_y0);
        store_tile<DIB_ACC_BITEXACT, SR>(out, none32, pd.out, p_ch, pd.H, pd.W, p_x0, p_y0, lane, wave);
      }
      lap(ph_store);
      // ---- look-ahead: the next window's loads go out before this one's tap loop ----------------------------------
      if (sg + 1 < c_nsegs) {                     // next segment of this tile (PSFs wider / taller than one window)
        issue(ck, clocal, sg + 1);
      } else if (more) {
        advance_to(next_entry);
        nk = lk; nlocal = l_first + (next_entry - l_lo);
        issue(nk, nlocal, 0);
      }
      lap(ph_issue);
      if (prio == 1) __builtin_amdgcn_s_setprio(0); else if (prio == 2) __builtin_amdgcn_s_setprio(3);
      // ---- taps of this segment -------------------------------------------------------------------------------------
      if (half) tap_loop_r4<ACC == DIB_ACC_FMA16, true>(acc, ltaps, w.t0, w.n, lane_addr);
      else tap_loop_r4<ACC == DIB_ACC_FMA16, false>(acc, ltaps, w.t0, w.n, lane_addr);
      if (prio == 1) __builtin_amdgcn_s_setprio(3); else if (prio == 2) __builtin_amdgcn_s_setprio(0);
      lap(ph_taps);
    }
#pragma unroll
    for (int i = 0; i < SR; ++i) {
      out[i][0] = acc[i][0]; out[i][1] = acc[i][1];
      acc[i][0] = h2{0, 0}; acc[i][1] = h2{0, 0};
    }
    pk = ck; plocal = clocal; have_prev = true;
    ++ntiles;
    if (!more) break;
    entry = next_entry; ++round; ck = nk; clocal = nlocal;
  }
  {
    const ImageDesc &pd = batch.img[pk];
    int p_ch, p_x0, p_y0;
    tile_geom(pd, plocal, p_ch, p_x0, p_y0);
    store_tile<DIB_ACC_BITEXACT, SR>(out, none32, pd.out, p_ch, pd.H, pd.W, p_x0, p_y0, lane, wave);
  }
  if (dbg && tid == 0) {
    unsigned long long *o = dbg + (size_t)blockIdx.x * 8;
    o[0] = ph_fill; o[1] = ph_store; o[2] = ph_issue; o[3] = ph_taps; o[4] = 0;
    o[5] = __builtin_readcyclecounter() - t_begin; o[6] = ntiles; o[7] = (unsigned long long)list;
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

// Diagnostics only: when set, the one-tile-per-workgroup kernel records per-workgroup phase stamps (8 x u64 each).
static unsigned long long *g_stamp_buffer = nullptr;
extern "C" void dib_debug_set_stamp_buffer(void *dev_ptr) { g_stamp_buffer = (unsigned long long *)dev_ptr; }
// Which tiled implementation serves fp16 images (both bit-identical; tests/test_blur_gpu.py compares them):
// 0 = persistent streaming kernel (default), 1 = one tile per workgroup.  `xcd_bands` = 0 gives the latter a flat
// tile order (the traffic experiment of DESIGN.md section 4).
static int g_variant = 0, g_xcd_bands = 1, g_prio = 0;
extern "C" void dib_debug_set_tile_order(int xcd_bands) { g_xcd_bands = xcd_bands ? 1 : 0; }
extern "C" void dib_debug_set_variant(int variant, int prio) { g_variant = variant == 1 ? 1 : 0; g_prio = prio; }

namespace {
// Per-device launch state: the dynamic-LDS opt-in is a per-device function attribute, and the persistent grid is
// sized from the device's CU count.  Guarded by a mutex: entry points may be called from several host threads.
struct DeviceState { bool ready = false; int cus = 0; };
std::mutex g_dev_mutex;
DeviceState g_dev[64];

template <typename Kern> hipError_t opt_in(Kern k, int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int prepare_device(int *cus) {
  int dev = 0;
  DIB_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { set_error("dib_sparse_blur: device index %d out of range", dev); return DIB_EINVAL; }
  std::lock_guard<std::mutex> lock(g_dev_mutex);
  DeviceState &st = g_dev[dev];
  if (!st.ready) {
    DIB_HIP_CHECK(opt_in(blur_tiled_f16_kernel<DIB_ACC_BITEXACT>, LDS_BYTES));
    DIB_HIP_CHECK(opt_in(blur_tiled_f16_kernel<DIB_ACC_FP32>, LDS_BYTES));
    DIB_HIP_CHECK(opt_in(blur_tiled_f16_kernel<DIB_ACC_FMA16>, LDS_BYTES));
    DIB_HIP_CHECK(opt_in(blur_stream_f16_kernel<DIB_ACC_BITEXACT>, STREAM_LDS_BYTES));
    DIB_HIP_CHECK(opt_in(blur_stream_f16_kernel<DIB_ACC_FMA16>, STREAM_LDS_BYTES));
    DIB_HIP_CHECK(hipDeviceGetAttribute(&st.cus, hipDeviceAttributeMultiprocessorCount, dev));
    st.ready = true;
  }
  *cus = st.cus;
  return DIB_OK;
}
}  // namespace

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
  int cus = 0;
  if (int rc = prepare_device(&cus)) return rc;
  int i = 0;
  while (i < B) {
    BlurBatch tiled, generic;
    tiled.n = generic.n = 0;
    int tiles = 0, gblocks = 0;
    bool small_images = true;   // the persistent kernel's reciprocal tile arithmetic holds
    for (; i < B && tiled.n < MAX_BATCH; ++i) {
      if (table_index[i] < 0) continue;
      ImageDesc d;
      d.in = in_dev[i]; d.out = out_dev[i]; d.C = C[i]; d.H = H[i]; d.W = W[i]; d.table = table_index[i];
      d.tiles_x = (W[i] + TILE_W - 1) / TILE_W;
      d.tiles_y = (H[i] + TH - 1) / TH;
      // ceil(2^32 / d): umulhi(n, inv) == n / d for n * d < 2^32, i.e. for every tile index of an image with < 65,536 tiles
      d.inv_tiles_x = (unsigned)((0x100000000ull + (unsigned)d.tiles_x - 1) / (unsigned)d.tiles_x);
      d.inv_per_ch = (unsigned)((0x100000000ull + (unsigned)(d.tiles_x * d.tiles_y) - 1) / (unsigned)(d.tiles_x * d.tiles_y));
      small_images = small_images && (long long)d.C * d.tiles_x * d.tiles_y < 65536;
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
    if (dtype == DIB_F16) {
      for (int k = tiled.n; k <= MAX_BATCH; ++k) tiled.tile_begin[k] = tiles;
      // the eight per-XCD lists (lists differ by at most one tile per image)
      QueueInfo qi;
      int longest = 0;
      for (int x = 0; x < 8; ++x) {
        int len = 0;
        for (int k = 0; k < tiled.n; ++k) {
          const int T = tiled.tile_begin[k + 1] - tiled.tile_begin[k];
          len += (((x + 1) * T) >> 3) - ((x * T) >> 3);
        }
        qi.len[x] = len;
        longest = len > longest ? len : longest;
      }
      const bool streaming = g_variant == 0 && acc_mode != DIB_ACC_FP32 && small_images;
      if (streaming) {
        // persistent: every workgroup resident at once, one share of workgroups per list.  The share is the smallest
        // one that needs no more rounds than the largest possible share would: every workgroup then serves the same
        // number of tiles (+- 1) instead of a few of them serving one more.
        const int share_max = WG_PER_CU * cus / 8 > 0 ? WG_PER_CU * cus / 8 : 1;
        const int rounds = (longest + share_max - 1) / share_max;
        const int grid = 8 * ((longest + rounds - 1) / rounds);
        if (acc_mode == DIB_ACC_FMA16) hipLaunchKernelGGL((blur_stream_f16_kernel<DIB_ACC_FMA16>), dim3(grid), dim3(64 * SNW), STREAM_LDS_BYTES, s, tiled, qi, (const int *)tables_dev, K, g_stamp_buffer, g_prio);
        else hipLaunchKernelGGL((blur_stream_f16_kernel<DIB_ACC_BITEXACT>), dim3(grid), dim3(64 * SNW), STREAM_LDS_BYTES, s, tiled, qi, (const int *)tables_dev, K, g_stamp_buffer, g_prio);
      } else {
        const int grid = g_xcd_bands ? 8 * longest : tiles;
        if (acc_mode == DIB_ACC_FP32) hipLaunchKernelGGL((blur_tiled_f16_kernel<DIB_ACC_FP32>), dim3(grid), dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
        else if (acc_mode == DIB_ACC_FMA16) hipLaunchKernelGGL((blur_tiled_f16_kernel<DIB_ACC_FMA16>), dim3(grid), dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
        else hipLaunchKernelGGL((blur_tiled_f16_kernel<DIB_ACC_BITEXACT>), dim3(grid), dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
      }
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
