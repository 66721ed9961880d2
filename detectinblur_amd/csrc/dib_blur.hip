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
// Tiled fp16 kernels -- the "split-column" idea: a lane owns columns a fixed distance apart and the LDS window stores, per
// position j, the pixels of exactly those columns next to each other.  A tap (r, c) is then one ALIGNED LDS read at
// position  lane + (cmax - c)  for ANY column shift, odd ones included (measured on MI355X: a ds_read_b64 that is not
// 8-byte aligned runs 22x slower, so a plain row-major fp16 window is not an option).  Row shifts are LDS row offsets;
// the rows a lane owns use compile-time immediate offsets.  The tap list arrives cut into segments (dib_compact.hip)
// whose bounding box is at most 13 PSF rows x 25 PSF columns, so ONE small LDS window serves any PSF: a wide or tall PSF
// takes several fill + accumulate rounds, in tap order, with the accumulators staying in registers.
//   default ("quad") shape  128 x 32 tile, lane = 32 h + j owns columns j + {0, 32, 64, 96} of 4 rows, 8-byte elements
//                           {P[k], P[k+32] | P[k+64], P[k+96]}, 44 rows x 56 elements = 19.7 KB: 8 workgroups per CU
//   256-wide shape          256 x 32 tile, lane owns columns l + 64k (k = 0..3) of 8 rows, 8-byte words, 44 x 96 words =
//                           33.8 KB: 4 workgroups per CU; what DIB_ACC_FP32 runs on, and the second implementation the
//                           tests compare with the default one bit for bit
//
// What bounds it (measured: profiles/r2_blur_tap_slope.txt): a packed fp16 instruction occupies a SIMD for 4 cycles per
// wave, and the bit-exact contract needs a multiply AND an add per 2 pixel-taps: 16.9 M wave-instructions for the BASELINE
// batch = 30.8 us of pure vector-ALU time on 1024 SIMDs, above its 13 us of HBM time.  Everything else -- window fills,
// stores, launch ramp, drain -- has to hide behind that arithmetic, i.e. behind the other workgroups of the CU: hence
// eight of them per CU, and hence the instruction diet of everything outside the tap loop.
#include "dib_compact_dev.h"
#include <mutex>
#include <vector>
#include <stdlib.h>
#include <string.h>

// ---- the device's status word: what a kernel of this file writes instead of trapping ---------------------------------
// Two things can go wrong inside a launch that the host cannot see when it enqueues it: an in-launch hand-off that does not
// arrive within its poll budget (blur_step_f16_kernel) and a tap table compacted for the other LDS window geometry.  A trap
// would take the whole GPU context -- a training job's -- with it.  Instead the workgroup stores a code into a block of PINNED
// HOST memory the library owns (one per device, set up by prepare_device; the pointer sits in this code-object global, so no
// kernel carries it in a register) and ends; its tile stays unwritten.  The next dib_blur_step / dib_sparse_blur call on the
// device finds the code without any synchronisation, reports it (DIB_ETIMEOUT / DIB_EINVAL) and, for a hand-off, takes the
// single launch out of service for that device: include/dib.h, "Device status".
extern "C" { __device__ __attribute__((used)) unsigned *dib_status_word = nullptr; }
#define DIB_STATUS_HANDOFF 1u     /* word 0: code, word 1: detail (the hand-off's tag / the table's geometry word) */
#define DIB_STATUS_GEOMETRY 2u
#define DIB_STATUS_WORDS 16
#ifdef DIB_STEP_POLLSTATS
// Diagnostic build only (scratch/t_step_contention.py): the most polls any workgroup of any launch needed, [0] for its PSF's first
// segment, [1] for the counter in front of its first tap loop; read and cleared through dib_debug_poll_stats.
extern "C" { __device__ __attribute__((used)) unsigned dib_poll_stats[4] = {0, 0, 0, 0}; }
#endif

namespace dib {

__device__ __forceinline__ void report_and_exit(unsigned code, unsigned detail) {
  typedef __attribute__((address_space(1))) unsigned gu32;
  gu32 *st = (gu32 *)(size_t)*(unsigned *volatile *)&dib_status_word;
  __hip_atomic_store(st + 1, detail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(st, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_endpgm();
}

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int TILE_W = 256;
constexpr int PQ = WIN_PITCH;         // LDS row pitch in 8-byte words (96)
constexpr int TH = 32;                // tile rows
constexpr int LROWS = TH + SEG_ROWS;  // LDS rows per window (44)
constexpr int LDS_BYTES = LROWS * PQ * 8;  // 33,792 B: four workgroups per CU
static_assert(WIN_PITCH * 8 == 768, "the asm below hard-codes the LDS row pitch");

// 64-bit asm operands must be scalar integers: hipcc (ROCm 7.2) aliases both lanes of a
// 2 x 32-bit vector operand of an inline-asm "=v" output to the same register.
typedef unsigned long long u2;

// ---- -DDIB_PORTABLE_TAPS: the tap loops and the step's wait as plain C++ ---------------------------------------------------------
// The hand-written loops below name vector and scalar registers and lean on what hipcc 7.2 does around them; a compiler that
// allocates differently turns tests/test_kernel_resources.py red.  `make portable` (csrc/Makefile) builds libdib_hip_portable.so
// with every asm loop replaced by the C++ restatement next to it: the SAME operations on the same operands in the same order
// (bit-identical results -- the GPU suite runs green on it: DIB_HIP_LIB=.../libdib_hip_portable.so python -m pytest tests -m gpu),
// scheduled by the compiler, slower and outside the register budget.  A way to keep working, not a product path.
#ifdef DIB_PORTABLE_TAPS
typedef unsigned uvec2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uvec2 lds_b64(unsigned addr) { return *(const __attribute__((address_space(3))) uvec2 *)(size_t)addr; }
__device__ __forceinline__ unsigned lds_b32(unsigned addr) { return *(const __attribute__((address_space(3))) unsigned *)(size_t)addr; }
__device__ __forceinline__ _Float16 half_of(unsigned bits) { return __builtin_bit_cast(_Float16, (unsigned short)(bits & 0xffffu)); }
// one tap on one packed register: the reference's multiply, rounded, then its add, rounded -- or (FUSED) one fused multiply-add
template <bool FUSED>
__device__ __forceinline__ h2 tap_pk(h2 acc, unsigned p_bits, _Float16 w) {
#pragma clang fp contract(off)
  const h2 p = __builtin_bit_cast(h2, p_bits);
  if constexpr (FUSED) return h2{(_Float16)__builtin_fmaf16(w, p.x, acc.x), (_Float16)__builtin_fmaf16(w, p.y, acc.y)};
  else { const h2 t = h2{(_Float16)(w * p.x), (_Float16)(w * p.y)}; return h2{(_Float16)(acc.x + t.x), (_Float16)(acc.y + t.y)}; }
}
#endif

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
// The loop opens with vmcnt(0): a window load whose value hipcc found no use for may still be in flight, and
// hipcc is free to have put its destination into a register clobbered here; it waits before ITS OWN next write
// to such a register, but not before this asm's.
template <bool FUSED, bool HALF>
__device__ __forceinline__ void tap_loop_r8(h2 (&acc)[8][2], unsigned long long ltaps, int t0, int n, unsigned lane_addr) {
#ifdef DIB_PORTABLE_TAPS
  {
    const unsigned *lt = reinterpret_cast<const unsigned *>(ltaps);
    for (int t = t0; t < t0 + n; ++t) {
      const unsigned word = lt[t];
      const _Float16 w = half_of(word >> 16);
      const unsigned at = lane_addr + (word & 0xffffu);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (HALF) acc[i][0] = tap_pk<FUSED>(acc[i][0], lds_b32(at + 768u * i), w);
        else { const uvec2 e = lds_b64(at + 768u * i); acc[i][0] = tap_pk<FUSED>(acc[i][0], e.x, w); acc[i][1] = tap_pk<FUSED>(acc[i][1], e.y, w); }
      }
    }
    return;
  }
#endif
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
  if constexpr (HALF) {
    if constexpr (FUSED) { DIB_R8_ASM(DIB_W_DRAIN, DIB_READ8H(64), DIB_READ8H(80), DIB_FMADDH_A, DIB_FMADDH_B); }
    else { DIB_R8_ASM(DIB_W_DRAIN, DIB_READ8H(64), DIB_READ8H(80), DIB_MADDH_A, DIB_MADDH_B); }
  } else {
    if constexpr (FUSED) { DIB_R8_ASM(DIB_W_DRAIN, DIB_READ8(64), DIB_READ8(80), DIB_FMADD_A, DIB_FMADD_B); }
    else { DIB_R8_ASM(DIB_W_DRAIN, DIB_READ8(64), DIB_READ8(80), DIB_MADD_A, DIB_MADD_B); }
  }
#undef DIB_W_DRAIN
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
  if (dbg && threadIdx.x == 0) dbg[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + slot] = __builtin_readcyclecounter();
}

// Wave-uniform buffer descriptor of one channel plane (base, byte size): loads and stores then take
// a 32-bit per-lane byte offset (voffset) plus a scalar row offset (soffset) -- no 64-bit address
// arithmetic on the vector ALU, which the arithmetic of the co-resident waves keeps busy.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void *img_base, int ch, int H, int W) {
  const unsigned long long a = (unsigned long long)img_base + (unsigned long long)ch * H * W * 2ull;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, H * W * 2, 0x00020000);
}

constexpr int NW = 4;                       // waves per workgroup
constexpr int R = TH / NW;                  // rows per lane (8)
constexpr int G = (LROWS + NW - 1) / NW;    // LDS rows a wave fills (11): all of them in ONE batch of loads

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
// No execution masks and no branches: the hardware drops a buffer store whose per-lane offset lies past the
// descriptor's range.  Lanes whose column is outside the image get an out-of-range offset once per tile; a row
// below the image (bottom tiles of an image whose height is not a multiple of 32) is stored through a descriptor of
// range zero.  Always exactly 4 x R store instructions: the look-ahead of the strip loop relies on that count
// (vector-memory operations retire in issue order; hipcc can only wait for "all but the last 32" if it is sure of the 32).
template <int ACC, int R>
__device__ __forceinline__ void store_tile(const h2 (&acc)[R][2], const float (&acc32)[R][4], void *out_base, int ch, int H, int W, int x0,
                                           int y0, int lane, int wave) {
  const unsigned long long pa = (unsigned long long)out_base + (unsigned long long)ch * H * W * 2ull;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pa), hi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
  void *plane = (void *)(((unsigned long long)hi << 32) | lo);
  const int w2 = W * 2;
  const int xr = W - x0 - lane;  // columns remaining for this lane
  const unsigned voff = 2u * (unsigned)(x0 + lane), oob = 0x7ffffff0u;
  const unsigned vo0 = xr > 0 ? voff : oob, vo1 = xr > 64 ? voff + 128u : oob, vo2 = xr > 128 ? voff + 256u : oob,
                 vo3 = xr > 192 ? voff + 384u : oob;
  const int yb = y0 + wave * R;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(plane, 0, yb + i < H ? H * w2 : 0, 0x00020000);
    const int soff = (yb + i) * w2;
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
// The tiled kernel: one workgroup = one (image, channel, 256 x 32 tile); fill -> taps -> store, the latencies of one
// workgroup's phases hidden by the other three workgroups of its CU.
// Round 2 built the alternative in full -- a PERSISTENT kernel (3 workgroups per CU for the whole launch) that issues
// the window loads of tile n+1 before the tap loop of tile n (60 values in flight per lane), defers the stores of
// tile n-1 behind the next barrier so they do not sit in front of those loads in the in-order vmcnt queue, runs a
// static snake schedule over the per-XCD lists, and keeps only tile indices in SGPRs -- in a 4-wave x 8-row and an
// 8-wave x 4-row form, both bit-identical to this kernel (scratch/blur_stream_experiment.hip, scratch/stamps_stream.py).
// Measured on the BASELINE batch: 62.6 us (4 x 8, tap loops at raised priority) and 70.8 us (8 x 4) against 52 us
// here.  Why: (1) the look-ahead costs 60 + 16 registers, i.e. one wave per SIMD (168 registers, 3 waves), and the
// packed-fp16 issue rate of a SIMD grows with the number of waves that are inside tap loops at the same time
// (scratch/ubench/ub_clk.hip: 5.1 / 4.4 / 3.0 / 2.6 cycles per instruction with 1 / 2 / 3 / 8 waves); (2) a dynamic
// tile queue is not affordable: a returning device-scope atomic took ~10 us under the kernel's own load, twice a tap
// loop (139 us per launch with tickets drawn one tile ahead); (3) with static shares the slowest workgroup ran 25 %
// longer than the mean.  The hardware dispatcher of THIS kernel is the better tile queue.
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
  const int x0 = tx * TILE_W;
  const bool half = W - x0 <= 128;   // every valid column lies in the tile's first 128: the HALF tap loop
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
  unsigned v[G][5];
  unsigned zmask;

  stamp(dbg, 0);
  const int y0 = ty * TH;
  for (int sg = 0; sg < nsegs; ++sg) {
    const Window w = window_of(segs[sg]);
    issue_window_loads(v, zmask, in_rsrc, w, x0, y0, H, W, K, mode, lane, qb);
    if (sg > 0) __syncthreads();  // every wave is done reading the previous segment's window
    write_window(wp, v, zmask, lane < w.cmax - w.cmin);
    __syncthreads();
    if (sg == 0) stamp(dbg, 1);
    // the table pointer itself (kernel-argument derived, provably uniform): hipcc then fetches the ltap words
    // with scalar loads; a pointer rebuilt from an integer would go through per-lane flat loads
    if constexpr (ACC == DIB_ACC_FP32) tap_loop_fp32<R>(acc32, reinterpret_cast<const unsigned *>(tab + table_ltaps_off(K)), w.t0, w.n, lane_addr);
    else if (half) tap_loop_r8<ACC == DIB_ACC_FMA16, true>(acc, ltaps, w.t0, w.n, lane_addr);
    else tap_loop_r8<ACC == DIB_ACC_FMA16, false>(acc, ltaps, w.t0, w.n, lane_addr);
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
// Grid: y = image (descriptor order: heaviest first), x = 8 t + list.  The image index comes with the workgroup, so the
// prologue's chain of dependent scalar loads is descriptor -> table header / first segment (two round trips; a lookup
// through a flattened tile index cost a third).  band_entry: entry t of list x of an image with T tiles -> tile index
// inside the image; false past the end of the band (the grid's x extent is 8 x the longest band of the launch).
__device__ __forceinline__ bool band_entry(int T, int x, int t, int &local) {
  const int lo = (x * T) >> 3, hi = ((x + 1) * T) >> 3;
  local = lo + t;
  return t < hi - lo;
}

// =============================================================================================
// Default ("quad") shape: 128 x 32 tiles, 4 waves, eight workgroups per CU (the kernel is a closed system: a CU's slots
// each run dispatch -> prologue -> fill -> taps -> store in sequence, so slots are what buys throughput: 3 / 4 / 8 slots
// gave 61 / 51 / 46 us on the BASELINE batch).  A lane owns FOUR columns of FOUR rows -- lane = 32 h + j computes
// columns {j, j+32, j+64, j+96} of rows 4h .. 4h+3 of its wave's 8 tile rows (a wave FILLS 11 of the window's 44 rows) -- and the window is stored as 8-byte elements
// e[k] = {P[k], P[k+32] | P[k+64], P[k+96]}  (k = 0 .. 31 + SEG_COLS), so that one tap is 4 x ds_read_b64 per lane: the
// LDS serves 256 B per clock for 8-byte reads and 128 B for 4-byte ones (MI355X_MICROARCH.md, LDS table), and the first
// 128-wide shape of round 2 (4-byte words {P[j], P[j+64]}, 8 x ds_read_b32 per tap; scratch/blur_narrow_shape.hip) ran
// its tap phase at the LDS's rate, not the vector ALU's.
// Element k + dcol is 8-byte aligned for every tap column, and the 32 lanes of an LDS lane group read 32
// consecutive elements = all 64 banks once.  44 rows x 56 elements x 8 B = 19,712 B: eight workgroups per CU.
// =============================================================================================
constexpr int QTILE_W = 128;
constexpr int QPITCH = QUAD_PITCH * 8;        // bytes per LDS row (448)
constexpr int QLDS_BYTES = LROWS * QPITCH;    // 19,712 B
static_assert(QPITCH == 448, "the asm below hard-codes the LDS row pitch");
static_assert(LROWS % NW == 0, "every wave fills the same number of window rows");
// The two window geometries of the quad shape (dib_common.h): L = false the standard one above, L = true the LARGE one
// (21 x 64 segments, 52 rows x 96 elements = 39,936 B, 4 workgroups per CU) for launches that leave the chip's slots empty.
template <bool L> struct QGeom {
  static constexpr int PITCH_EL = L ? QUAD_PITCH_L : QUAD_PITCH;   // elements per LDS row
  static constexpr int PITCH = PITCH_EL * 8;                       // bytes per LDS row
  static constexpr int ROWS = TH + (L ? SEG_ROWS_L : SEG_ROWS);    // LDS rows
  static constexpr int GQ = ROWS / NW;                             // rows a wave fills
  static constexpr int BYTES = ROWS * PITCH;
};
static_assert(QGeom<false>::PITCH == QPITCH && QGeom<false>::ROWS == LROWS && QGeom<false>::BYTES == QLDS_BYTES, "standard geometry");
static_assert(QGeom<true>::PITCH == 768 && QGeom<true>::ROWS % NW == 0 && QGeom<true>::BYTES == 39936, "the asm below hard-codes the large window's row pitch");

// Tap loop of the quad shape: 4 x 8-byte reads per tap, rows i = 0..3 land in v[base+2i : base+2i+1] = the operands of
// accumulators 2i (columns j, j+32) and 2i+1 (columns j+64, j+96); the LDS address is one v_mad_u32_u16 (low 16 bits of
// the ltap word + lane base).  Same arithmetic and the same one-tap LDS look-ahead as the 256-wide shape's loop; the scalar side
// is leaner because every instruction a wave issues costs launch time here (measured by padding the loop: +0.24 us per
// million scalar, +0.69 per million vector, +1.25 per million LDS instructions): the ltap words
// arrive two at a time (s_load_dwordx2) into three fixed register pairs P = s[36:37], Q = s[38:39], R = s[40:41] that the
// 6-way unrolled body addresses by name, so nothing is moved between "current / next / in flight" registers: per tap
// 1 wait + 1/2 load + 1/2 add + 1 subtract + 1 branch instead of 7 scalar instructions.  A pair is reloaded in the tap
// after its last use and first used three taps later (with two pairs and one tap of cover the waits ran into the
// scalar loads: no faster than the old loop).
//   tap 6k  : load R <- w[6k+4..5] | read Y at offset(P.hi) | multiply-add X by weight(P.lo)
//   tap 6k+1:                        read X at offset(Q.lo) | multiply-add Y by weight(P.hi)
//   tap 6k+2: load P <- w[6k+6..7] | read Y at offset(Q.hi) | multiply-add X by weight(Q.lo)
//   tap 6k+3:                        read X at offset(R.lo) | multiply-add Y by weight(Q.hi)
//   tap 6k+4: load Q <- w[6k+8..9] | read Y at offset(R.hi) | multiply-add X by weight(R.lo)
//   tap 6k+5:                        read X at offset(P.lo) | multiply-add Y by weight(R.hi)
// Operands: %0-%7 accumulators, %8 byte offset of the next ltap pair, %9 taps left, %10 ltaps, %11 lane base.
#define DIBQ_ADD(b, i) "v_pk_add_f16 %" #i ", %" #i ", v" #b "\n\t"
#define DIBQ_MUL(b, W) "v_pk_mul_f16 v" #b ", " W ", v" #b " op_sel:[1,0] op_sel_hi:[1,1]\n\t"
#define DIBQ_FMA(b, i, W) "v_pk_fma_f16 %" #i ", " W ", v" #b ", %" #i " op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
#define DIBQ_MADD_X(W) DIBQ_MUL(32, W) DIBQ_MUL(33, W) DIBQ_MUL(34, W) DIBQ_MUL(35, W) DIBQ_MUL(36, W) DIBQ_MUL(37, W) DIBQ_MUL(38, W) DIBQ_MUL(39, W) \
  DIBQ_ADD(32, 0) DIBQ_ADD(33, 1) DIBQ_ADD(34, 2) DIBQ_ADD(35, 3) DIBQ_ADD(36, 4) DIBQ_ADD(37, 5) DIBQ_ADD(38, 6) DIBQ_ADD(39, 7)
#define DIBQ_MADD_Y(W) DIBQ_MUL(40, W) DIBQ_MUL(41, W) DIBQ_MUL(42, W) DIBQ_MUL(43, W) DIBQ_MUL(44, W) DIBQ_MUL(45, W) DIBQ_MUL(46, W) DIBQ_MUL(47, W) \
  DIBQ_ADD(40, 0) DIBQ_ADD(41, 1) DIBQ_ADD(42, 2) DIBQ_ADD(43, 3) DIBQ_ADD(44, 4) DIBQ_ADD(45, 5) DIBQ_ADD(46, 6) DIBQ_ADD(47, 7)
#define DIBQ_FMADD_X(W) DIBQ_FMA(32, 0, W) DIBQ_FMA(33, 1, W) DIBQ_FMA(34, 2, W) DIBQ_FMA(35, 3, W) DIBQ_FMA(36, 4, W) DIBQ_FMA(37, 5, W) DIBQ_FMA(38, 6, W) DIBQ_FMA(39, 7, W)
#define DIBQ_FMADD_Y(W) DIBQ_FMA(40, 0, W) DIBQ_FMA(41, 1, W) DIBQ_FMA(42, 2, W) DIBQ_FMA(43, 3, W) DIBQ_FMA(44, 4, W) DIBQ_FMA(45, 5, W) DIBQ_FMA(46, 6, W) DIBQ_FMA(47, 7, W)
#define DIBQ_READ(base, OFF)                                                                                 \
  "v_mad_u32_u16 v48, " OFF ", 1, %11\n\t"                                                                    \
  "ds_read_b64 v[" #base ":" #base "+1], v48\n\tds_read_b64 v[" #base "+2:" #base "+3], v48 offset:448\n\t"    \
  "ds_read_b64 v[" #base "+4:" #base "+5], v48 offset:896\n\tds_read_b64 v[" #base "+6:" #base "+7], v48 offset:1344\n\t"
// HALF variant: a tile with at most 64 valid columns (the right-hand edge of an image whose width is not a multiple of
// 128: 1333 = 10 x 128 + 53) only needs the first word {P[k], P[k+32]} of every element: 4-byte reads, 4 + 4 instead of
// 8 + 8 arithmetic instructions per tap, on the even registers / accumulators.
#define DIBQ_READH(base, OFF)                                                                                \
  "v_mad_u32_u16 v48, " OFF ", 1, %11\n\t"                                                                    \
  "ds_read_b32 v[" #base "], v48\n\tds_read_b32 v[" #base "+2], v48 offset:448\n\t"                            \
  "ds_read_b32 v[" #base "+4], v48 offset:896\n\tds_read_b32 v[" #base "+6], v48 offset:1344\n\t"
#define DIBQ_READ_L(base, OFF)                                                                               \
  "v_mad_u32_u16 v48, " OFF ", 1, %11\n\t"                                                                    \
  "ds_read_b64 v[" #base ":" #base "+1], v48\n\tds_read_b64 v[" #base "+2:" #base "+3], v48 offset:768\n\t"    \
  "ds_read_b64 v[" #base "+4:" #base "+5], v48 offset:1536\n\tds_read_b64 v[" #base "+6:" #base "+7], v48 offset:2304\n\t"
#define DIBQ_READH_L(base, OFF)                                                                              \
  "v_mad_u32_u16 v48, " OFF ", 1, %11\n\t"                                                                    \
  "ds_read_b32 v[" #base "], v48\n\tds_read_b32 v[" #base "+2], v48 offset:768\n\t"                            \
  "ds_read_b32 v[" #base "+4], v48 offset:1536\n\tds_read_b32 v[" #base "+6], v48 offset:2304\n\t"
#define DIBQ_MADDH_X(W) DIBQ_MUL(32, W) DIBQ_MUL(34, W) DIBQ_MUL(36, W) DIBQ_MUL(38, W) DIBQ_ADD(32, 0) DIBQ_ADD(34, 2) DIBQ_ADD(36, 4) DIBQ_ADD(38, 6)
#define DIBQ_MADDH_Y(W) DIBQ_MUL(40, W) DIBQ_MUL(42, W) DIBQ_MUL(44, W) DIBQ_MUL(46, W) DIBQ_ADD(40, 0) DIBQ_ADD(42, 2) DIBQ_ADD(44, 4) DIBQ_ADD(46, 6)
#define DIBQ_FMADDH_X(W) DIBQ_FMA(32, 0, W) DIBQ_FMA(34, 2, W) DIBQ_FMA(36, 4, W) DIBQ_FMA(38, 6, W)
#define DIBQ_FMADDH_Y(W) DIBQ_FMA(40, 0, W) DIBQ_FMA(42, 2, W) DIBQ_FMA(44, 4, W) DIBQ_FMA(46, 6, W)
#define DIBQ_LOAD(PAIR) "s_load_dwordx2 " PAIR ", %10, %8\n\ts_add_u32 %8, %8, 8\n\t"
#define DIBQ_NEXT(LABEL) "s_sub_u32 %9, %9, 1\n\ts_cbranch_scc1 " LABEL "\n\t"
template <bool FUSED, bool HALF, bool L = false>
__device__ __forceinline__ void tap_loop_quad(h2 (&acc)[8], unsigned long long ltaps, int t0, int n, unsigned lane_addr) {
#ifdef DIB_PORTABLE_TAPS
  {
    constexpr unsigned PITCH_B = L ? 768u : 448u;
    const unsigned *lt = reinterpret_cast<const unsigned *>(ltaps);
    for (int t = t0; t < t0 + n; ++t) {
      const unsigned word = lt[t];
      const _Float16 w = half_of(word >> 16);
      const unsigned at = lane_addr + (word & 0xffffu);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (HALF) acc[2 * i] = tap_pk<FUSED>(acc[2 * i], lds_b32(at + PITCH_B * i), w);
        else { const uvec2 e = lds_b64(at + PITCH_B * i); acc[2 * i] = tap_pk<FUSED>(acc[2 * i], e.x, w); acc[2 * i + 1] = tap_pk<FUSED>(acc[2 * i + 1], e.y, w); }
      }
    }
    return;
  }
#endif
  unsigned toff = (unsigned)__builtin_amdgcn_readfirstlane(t0 * 4), cnt = (unsigned)__builtin_amdgcn_readfirstlane(n - 1);
  unsigned a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(unsigned, acc[i]);
#define DIB_RQ_ASM(RD, ARITH_X, ARITH_Y) \
  asm volatile( \
      DIBQ_LOAD("s[36:37]") DIBQ_LOAD("s[38:39]") "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" RD(32, "s36") \
      "Ldibq_loop%=:\n\t" \
      "s_waitcnt lgkmcnt(0)\n\t" DIBQ_LOAD("s[40:41]") RD(40, "s37") ARITH_X("s36") DIBQ_NEXT("Ldibq_done%=") \
      "s_waitcnt lgkmcnt(0)\n\t" RD(32, "s38") ARITH_Y("s37") DIBQ_NEXT("Ldibq_done%=") \
      "s_waitcnt lgkmcnt(0)\n\t" DIBQ_LOAD("s[36:37]") RD(40, "s39") ARITH_X("s38") DIBQ_NEXT("Ldibq_done%=") \
      "s_waitcnt lgkmcnt(0)\n\t" RD(32, "s40") ARITH_Y("s39") DIBQ_NEXT("Ldibq_done%=") \
      "s_waitcnt lgkmcnt(0)\n\t" DIBQ_LOAD("s[38:39]") RD(40, "s41") ARITH_X("s40") DIBQ_NEXT("Ldibq_done%=") \
      "s_waitcnt lgkmcnt(0)\n\t" RD(32, "s36") ARITH_Y("s41") \
      "s_sub_u32 %9, %9, 1\n\ts_cbranch_scc0 Ldibq_loop%=\n\t" \
      "Ldibq_done%=:\n\t" \
      "s_waitcnt lgkmcnt(0)" \
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+s"(toff), "+s"(cnt) \
      : "s"(ltaps), "v"(lane_addr) \
      : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", \
        "s36", "s37", "s38", "s39", "s40", "s41", "scc", "memory")
  if constexpr (L && FUSED && HALF) { DIB_RQ_ASM(DIBQ_READH_L, DIBQ_FMADDH_X, DIBQ_FMADDH_Y); }
  else if constexpr (L && FUSED) { DIB_RQ_ASM(DIBQ_READ_L, DIBQ_FMADD_X, DIBQ_FMADD_Y); }
  else if constexpr (L && HALF) { DIB_RQ_ASM(DIBQ_READH_L, DIBQ_MADDH_X, DIBQ_MADDH_Y); }
  else if constexpr (L) { DIB_RQ_ASM(DIBQ_READ_L, DIBQ_MADD_X, DIBQ_MADD_Y); }
  else if constexpr (FUSED && HALF) { DIB_RQ_ASM(DIBQ_READH, DIBQ_FMADDH_X, DIBQ_FMADDH_Y); }
  else if constexpr (FUSED) { DIB_RQ_ASM(DIBQ_READ, DIBQ_FMADD_X, DIBQ_FMADD_Y); }
  else if constexpr (HALF) { DIB_RQ_ASM(DIBQ_READH, DIBQ_MADDH_X, DIBQ_MADDH_Y); }
  else { DIB_RQ_ASM(DIBQ_READ, DIBQ_MADD_X, DIBQ_MADD_Y); }
#undef DIB_RQ_ASM
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = __builtin_bit_cast(h2, a[i]);
}

// DIB_ACC_FP32 on the quad shape: acc32 = fma(float(P), float(w), acc32) per pixel and tap, taps in the same order, ONE rounding
// to fp16 at the store -- the product of two fp16 values is exact in fp32, so this equals the unfused form bit for bit (the
// oracle restates it with numpy float32).  One v_fma_mix_f32 per pixel-tap (both fp16 factors converted inside the
// instruction, the weight straight from the high half of the ltap's scalar register, the pixel from either half of the 8-byte
// element: op_sel), i.e. HALF the vector-ALU time of the bit-exact mode's multiply + add; same window, same reads, same
// one-tap LDS look-ahead and 6-way unrolled scalar side as tap_loop_quad.  Sixteen fp32 accumulators per lane: acc[4 i + m] =
// row i, column j + 32 m.  Operands: %0-%15 accumulators, %16 byte offset of the next ltap pair, %17 taps left, %18 ltaps,
// %19 lane base.
#define DIBF_MIX(b, i, hs, W) "v_fma_mix_f32 %" #i ", " W ", v" #b ", %" #i " op_sel:[1," #hs ",0] op_sel_hi:[1,1,0]\n\t"
#define DIBF_ROWS_X(W) DIBF_MIX(32, 0, 0, W) DIBF_MIX(32, 1, 1, W) DIBF_MIX(33, 2, 0, W) DIBF_MIX(33, 3, 1, W) DIBF_MIX(34, 4, 0, W) DIBF_MIX(34, 5, 1, W) DIBF_MIX(35, 6, 0, W) DIBF_MIX(35, 7, 1, W) \
  DIBF_MIX(36, 8, 0, W) DIBF_MIX(36, 9, 1, W) DIBF_MIX(37, 10, 0, W) DIBF_MIX(37, 11, 1, W) DIBF_MIX(38, 12, 0, W) DIBF_MIX(38, 13, 1, W) DIBF_MIX(39, 14, 0, W) DIBF_MIX(39, 15, 1, W)
#define DIBF_ROWS_Y(W) DIBF_MIX(40, 0, 0, W) DIBF_MIX(40, 1, 1, W) DIBF_MIX(41, 2, 0, W) DIBF_MIX(41, 3, 1, W) DIBF_MIX(42, 4, 0, W) DIBF_MIX(42, 5, 1, W) DIBF_MIX(43, 6, 0, W) DIBF_MIX(43, 7, 1, W) \
  DIBF_MIX(44, 8, 0, W) DIBF_MIX(44, 9, 1, W) DIBF_MIX(45, 10, 0, W) DIBF_MIX(45, 11, 1, W) DIBF_MIX(46, 12, 0, W) DIBF_MIX(46, 13, 1, W) DIBF_MIX(47, 14, 0, W) DIBF_MIX(47, 15, 1, W)
// HALF (at most 64 valid columns): the first word {P[k], P[k+32]} of every element only
#define DIBF_ROWSH_X(W) DIBF_MIX(32, 0, 0, W) DIBF_MIX(32, 1, 1, W) DIBF_MIX(34, 4, 0, W) DIBF_MIX(34, 5, 1, W) DIBF_MIX(36, 8, 0, W) DIBF_MIX(36, 9, 1, W) DIBF_MIX(38, 12, 0, W) DIBF_MIX(38, 13, 1, W)
#define DIBF_ROWSH_Y(W) DIBF_MIX(40, 0, 0, W) DIBF_MIX(40, 1, 1, W) DIBF_MIX(42, 4, 0, W) DIBF_MIX(42, 5, 1, W) DIBF_MIX(44, 8, 0, W) DIBF_MIX(44, 9, 1, W) DIBF_MIX(46, 12, 0, W) DIBF_MIX(46, 13, 1, W)
#define DIBF_READ(base, OFF, P1, P2, P3)                                                                      \
  "v_mad_u32_u16 v48, " OFF ", 1, %19\n\t"                                                                    \
  "ds_read_b64 v[" #base ":" #base "+1], v48\n\tds_read_b64 v[" #base "+2:" #base "+3], v48 offset:" P1 "\n\t"  \
  "ds_read_b64 v[" #base "+4:" #base "+5], v48 offset:" P2 "\n\tds_read_b64 v[" #base "+6:" #base "+7], v48 offset:" P3 "\n\t"
#define DIBF_READH(base, OFF, P1, P2, P3)                                                                     \
  "v_mad_u32_u16 v48, " OFF ", 1, %19\n\t"                                                                    \
  "ds_read_b32 v[" #base "], v48\n\tds_read_b32 v[" #base "+2], v48 offset:" P1 "\n\t"                         \
  "ds_read_b32 v[" #base "+4], v48 offset:" P2 "\n\tds_read_b32 v[" #base "+6], v48 offset:" P3 "\n\t"
#define DIBF_RD_S(base, OFF) DIBF_READ(base, OFF, "448", "896", "1344")
#define DIBF_RD_L(base, OFF) DIBF_READ(base, OFF, "768", "1536", "2304")
#define DIBF_RDH_S(base, OFF) DIBF_READH(base, OFF, "448", "896", "1344")
#define DIBF_RDH_L(base, OFF) DIBF_READH(base, OFF, "768", "1536", "2304")
#define DIBF_LOAD(PAIR) "s_load_dwordx2 " PAIR ", %18, %16\n\ts_add_u32 %16, %16, 8\n\t"
#define DIBF_NEXT(LABEL) "s_sub_u32 %17, %17, 1\n\ts_cbranch_scc1 " LABEL "\n\t"
template <bool HALF, bool L>
__device__ __forceinline__ void tap_loop_quad_fp32(float (&acc)[16], unsigned long long ltaps, int t0, int n, unsigned lane_addr) {
#ifdef DIB_PORTABLE_TAPS
  {
    constexpr unsigned PITCH_B = L ? 768u : 448u;
    const unsigned *lt = reinterpret_cast<const unsigned *>(ltaps);
    for (int t = t0; t < t0 + n; ++t) {
      const unsigned word = lt[t];
      const float w = (float)half_of(word >> 16);
      const unsigned at = lane_addr + (word & 0xffffu);
#pragma unroll
      for (int i = 0; i < 4; ++i) {       // the product of two fp16 values is exact in fp32: fused == unfused
        uvec2 e;
        if constexpr (HALF) { e.x = lds_b32(at + PITCH_B * i); e.y = 0; } else e = lds_b64(at + PITCH_B * i);
        acc[4 * i] = __builtin_fmaf(w, (float)half_of(e.x), acc[4 * i]);
        acc[4 * i + 1] = __builtin_fmaf(w, (float)half_of(e.x >> 16), acc[4 * i + 1]);
        if constexpr (!HALF) {
          acc[4 * i + 2] = __builtin_fmaf(w, (float)half_of(e.y), acc[4 * i + 2]);
          acc[4 * i + 3] = __builtin_fmaf(w, (float)half_of(e.y >> 16), acc[4 * i + 3]);
        }
      }
    }
    return;
  }
#endif
  unsigned toff = (unsigned)__builtin_amdgcn_readfirstlane(t0 * 4), cnt = (unsigned)__builtin_amdgcn_readfirstlane(n - 1);
#define DIB_RF_ASM(RD, ARITH_X, ARITH_Y) \
  asm volatile( \
      DIBF_LOAD("s[36:37]") DIBF_LOAD("s[38:39]") "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" RD(32, "s36") \
      "Ldibf_loop%=:\n\t" \
      "s_waitcnt lgkmcnt(0)\n\t" DIBF_LOAD("s[40:41]") RD(40, "s37") ARITH_X("s36") DIBF_NEXT("Ldibf_done%=") \
      "s_waitcnt lgkmcnt(0)\n\t" RD(32, "s38") ARITH_Y("s37") DIBF_NEXT("Ldibf_done%=") \
      "s_waitcnt lgkmcnt(0)\n\t" DIBF_LOAD("s[36:37]") RD(40, "s39") ARITH_X("s38") DIBF_NEXT("Ldibf_done%=") \
      "s_waitcnt lgkmcnt(0)\n\t" RD(32, "s40") ARITH_Y("s39") DIBF_NEXT("Ldibf_done%=") \
      "s_waitcnt lgkmcnt(0)\n\t" DIBF_LOAD("s[38:39]") RD(40, "s41") ARITH_X("s40") DIBF_NEXT("Ldibf_done%=") \
      "s_waitcnt lgkmcnt(0)\n\t" RD(32, "s36") ARITH_Y("s41") \
      "s_sub_u32 %17, %17, 1\n\ts_cbranch_scc0 Ldibf_loop%=\n\t" \
      "Ldibf_done%=:\n\t" \
      "s_waitcnt lgkmcnt(0)" \
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), \
        "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]), "+s"(toff), "+s"(cnt) \
      : "s"(ltaps), "v"(lane_addr) \
      : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", \
        "s36", "s37", "s38", "s39", "s40", "s41", "scc", "memory")
  if constexpr (L && HALF) { DIB_RF_ASM(DIBF_RDH_L, DIBF_ROWSH_X, DIBF_ROWSH_Y); }
  else if constexpr (L) { DIB_RF_ASM(DIBF_RD_L, DIBF_ROWS_X, DIBF_ROWS_Y); }
  else if constexpr (HALF) { DIB_RF_ASM(DIBF_RDH_S, DIBF_ROWSH_X, DIBF_ROWSH_Y); }
  else { DIB_RF_ASM(DIBF_RD_S, DIBF_ROWS_X, DIBF_ROWS_Y); }
#undef DIB_RF_ASM
}

// ---- DIB_ACC_FAST16: the tap loop over VERTICAL-RUN GROUPS (dib_common.h: vgroups) -----------------------------------------------
// The tolerance mode may accumulate a window's taps in any order, so they arrive regrouped: n <= 4 taps of one PSF column in
// consecutive PSF rows.  A lane's four output rows need window rows j .. j + 3 for tap j of the group, i.e. n + 3 rows for the
// whole group instead of 4 n: 5 / 6 / 7 ds_read_b64 for 2 / 3 / 4 taps.  That is what this mode was short of: with one v_pk_fma_f16
// per register and tap the vector ALU (8 x 4 cycles per wave-tap) and the LDS (4 x 512 B per wave-tap at 256 B per clock and CU, four
// SIMDs reading) need the SAME 32 cycles per tap, and DIB_ACC_FMA16's loop ran at 56 (profiles/r2_blur_tap_slope.txt).
// A segment's groups come sorted by size, fours first, so the loop is four straight-line bodies -- no size tests inside a body --
// and changes body at most three times per segment: per group one wait, one scalar load, one LDS address, n + 3 reads, 8 n
// multiply-adds and five scalar instructions, where the plain loop spends five scalar instructions per TAP (this mode is as
// sensitive to scalar instructions as to vector ones: 0.6 us of launch time per million of either).
//   buffers  X = v[32:45], Y = v[46:59]: rows 0 .. 6 of a group (row k in v[base + 2 k : base + 2 k + 1]), v60 the LDS address
//   records  S0 = s[36:39] with X, S1 = s[40:43] with Y: the record of the group being multiplied also names the NEXT group's
//            offset and size (x), so its rows can be requested before its own record has arrived
//   a step   wait (this group's rows, this group's record) | request the next group's record into the other set | request the
//            next group's rows (n + 3 of them: the next group is no larger) | multiply-add this group | same size next? go on
// Taps of a group run from its lowest PSF row to its highest (j = n - 1 .. 0): the order the oracle restates (tests/test_fast16_gpu.py).
// Operands: %0-%7 accumulators, %8 byte offset of the next record to request, %9 the section's base, %10 lane base.
#define VR_LO "op_sel:[0,0,0] op_sel_hi:[0,1,1]"
#define VR_HI "op_sel:[1,0,0] op_sel_hi:[1,1,1]"
#define VR_FMA(i, W, SEL, B, k) "v_pk_fma_f16 %" #i ", " W ", v[" #B "+" #k "], %" #i " " SEL "\n\t"
// tap j of the group in buffer B: window rows j .. j + 3 -> accumulators (2 i, 2 i + 1) of output row i
#define VR_TAP(B, j, W, SEL) VR_FMA(0, W, SEL, B, 2*j) VR_FMA(1, W, SEL, B, 2*j+1) VR_FMA(2, W, SEL, B, 2*j+2) VR_FMA(3, W, SEL, B, 2*j+3) \
  VR_FMA(4, W, SEL, B, 2*j+4) VR_FMA(5, W, SEL, B, 2*j+5) VR_FMA(6, W, SEL, B, 2*j+6) VR_FMA(7, W, SEL, B, 2*j+7)
#define VR_TAPH(B, j, W, SEL) VR_FMA(0, W, SEL, B, 2*j) VR_FMA(2, W, SEL, B, 2*j+2) VR_FMA(4, W, SEL, B, 2*j+4) VR_FMA(6, W, SEL, B, 2*j+6)
#define VR_RD(B, k, off) "ds_read_b64 v[" #B "+2*" #k ":" #B "+2*" #k "+1], v60 offset:" #off "\n\t"
#define VR_RDH(B, k, off) "ds_read_b32 v[" #B "+2*" #k "], v60 offset:" #off "\n\t"
// the rows of a group of N taps at LDS offset OFF (the low 16 bits of a scalar) into buffer B
#define VR_READS4(RD, B, OFF) "v_mad_u32_u16 v60, " OFF ", 1, %10\n\t" RD(B, 0, 0) RD(B, 1, 448) RD(B, 2, 896) RD(B, 3, 1344) RD(B, 4, 1792) RD(B, 5, 2240) RD(B, 6, 2688)
#define VR_READS3(RD, B, OFF) "v_mad_u32_u16 v60, " OFF ", 1, %10\n\t" RD(B, 0, 0) RD(B, 1, 448) RD(B, 2, 896) RD(B, 3, 1344) RD(B, 4, 1792) RD(B, 5, 2240)
#define VR_READS2(RD, B, OFF) "v_mad_u32_u16 v60, " OFF ", 1, %10\n\t" RD(B, 0, 0) RD(B, 1, 448) RD(B, 2, 896) RD(B, 3, 1344) RD(B, 4, 1792)
#define VR_READS1(RD, B, OFF) "v_mad_u32_u16 v60, " OFF ", 1, %10\n\t" RD(B, 0, 0) RD(B, 1, 448) RD(B, 2, 896) RD(B, 3, 1344)
// the 8 N multiply-adds of the group in buffer B, weights w0 | w1 << 16 in W01, w2 | w3 << 16 in W23
#define VR_ARITH4(TAP, B, W01, W23) TAP(B, 3, W23, VR_HI) TAP(B, 2, W23, VR_LO) TAP(B, 1, W01, VR_HI) TAP(B, 0, W01, VR_LO)
#define VR_ARITH3(TAP, B, W01, W23) TAP(B, 2, W23, VR_LO) TAP(B, 1, W01, VR_HI) TAP(B, 0, W01, VR_LO)
#define VR_ARITH2(TAP, B, W01, W23) TAP(B, 1, W01, VR_HI) TAP(B, 0, W01, VR_LO)
#define VR_ARITH1(TAP, B, W01, W23) TAP(B, 0, W01, VR_LO)
// one step of the body for groups of N taps (NM1 = N - 1 as a string): the group in buffer BC (record C0 .. C2) is multiplied while
// the next group's rows go to BN and its record to the OTHER register set; P / Q: this step's and the other step's parity letter
#define VR_STEP(N, NM1, RD, TAP, BC, BN, C0, C1, C2, OTHER, P, Q)                                            \
  "Lvr_c" #N P "%=:\n\t"                                                                                    \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
  "s_load_dwordx4 " OTHER ", %9, %8\n\ts_add_u32 %8, %8, 16\n\t"                                            \
  VR_READS##N(RD, BN, C0)                                                                                   \
  "s_bfe_u32 s46, " C0 ", 0x30010\n\t"                                                                      \
  VR_ARITH##N(TAP, BC, C1, C2)                                                                              \
  "s_cmp_eq_u32 s46, " NM1 "\n\ts_cbranch_scc1 Lvr_c" #N Q "%=\n\t"                                         \
  "s_branch Lvr_d" Q "%=\n\t"
#define VR_BODY(N, NM1, RD, TAP)                                                                             \
  VR_STEP(N, NM1, RD, TAP, 32, 46, "s36", "s37", "s38", "s[40:43]", "x", "y")                               \
  VR_STEP(N, NM1, RD, TAP, 46, 32, "s40", "s41", "s42", "s[36:39]", "y", "x")
// which body runs the group whose size code is in s46 (4: the segment is done), by the buffer its rows are in
#define VR_DISPATCH(P)                                                                                       \
  "Lvr_d" P "%=:\n\t"                                                                                       \
  "s_cmp_eq_u32 s46, 4\n\ts_cbranch_scc1 Lvr_done%=\n\t"                                                    \
  "s_cmp_eq_u32 s46, 3\n\ts_cbranch_scc1 Lvr_c4" P "%=\n\t"                                                 \
  "s_cmp_eq_u32 s46, 2\n\ts_cbranch_scc1 Lvr_c3" P "%=\n\t"                                                 \
  "s_cmp_eq_u32 s46, 1\n\ts_cbranch_scc1 Lvr_c2" P "%=\n\t"                                                 \
  "s_branch Lvr_c1" P "%=\n\t"
template <bool HALF>
__device__ __forceinline__ void tap_loop_quad_vrun(h2 (&acc)[8], unsigned long long vgroups, int g0, unsigned lane_addr) {
#ifdef DIB_PORTABLE_TAPS
  {
    const uint4 *rec = reinterpret_cast<const uint4 *>(vgroups) + g0;
    uint4 r = rec[0];
    unsigned off = r.w & 0xffffu, code = (r.w >> 16) & 7u;      // the first record names its own group
    for (int g = 0;; ++g) {
      const int n = (int)code + 1;
      uvec2 row[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        row[k] = uvec2{0u, 0u};
        if (k < n + 3) { if constexpr (HALF) row[k].x = lds_b32(lane_addr + off + 448u * k); else row[k] = lds_b64(lane_addr + off + 448u * k); }
      }
#pragma unroll
      for (int j = 3; j >= 0; --j) {         // taps from the group's highest index down (the asm's order, the oracle's tap_order_vruns)
        if (j >= n) continue;
        const _Float16 w = half_of((j < 2 ? r.y : r.z) >> (16 * (j & 1)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] = tap_pk<true>(acc[2 * i], row[j + i].x, w);
          if constexpr (!HALF) acc[2 * i + 1] = tap_pk<true>(acc[2 * i + 1], row[j + i].y, w);
        }
      }
      off = r.x & 0xffffu; code = (r.x >> 16) & 7u;
      if (code == 4u) break;
      r = rec[g + 1];
    }
    return;
  }
#endif
  unsigned toff = (unsigned)__builtin_amdgcn_readfirstlane(g0 * 16);
  unsigned a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(unsigned, acc[i]);
#define DIB_VR_ASM(RD, TAP) \
  asm volatile( \
      /* the segment's first record: its own offset and size sit in w; its rows are requested for the largest size (rows past a \
         smaller group's are read and not used: the window's own rows, or zeros past the workgroup's LDS) */ \
      "s_load_dwordx4 s[36:39], %9, %8\n\ts_add_u32 %8, %8, 16\n\t" \
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
      "s_bfe_u32 s46, s39, 0x30010\n\t" \
      VR_READS4(RD, 32, "s39") \
      "s_branch Lvr_dx%=\n\t" \
      VR_BODY(4, "3", RD, TAP) VR_BODY(3, "2", RD, TAP) VR_BODY(2, "1", RD, TAP) VR_BODY(1, "0", RD, TAP) \
      VR_DISPATCH("x") VR_DISPATCH("y") \
      "Lvr_done%=:\n\t" \
      "s_waitcnt lgkmcnt(0)" \
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+s"(toff) \
      : "s"(vgroups), "v"(lane_addr) \
      : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", \
        "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s46", \
        "scc", "memory")
  if constexpr (HALF) { DIB_VR_ASM(VR_RDH, VR_TAPH); }
  else { DIB_VR_ASM(VR_RD, VR_TAP); }
#undef DIB_VR_ASM
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = __builtin_bit_cast(h2, a[i]);
}

// STEP (the blur step's single launch, blur_step_f16_kernel): with `early` set the segment count and the first segment come
// from the caller (the compaction's early record: the table itself must not be touched yet) and `wait_tables()` is called
// once, between the first window's barrier and the first tap loop -- the first reader of the table's offsets.
struct NoWait { __device__ void operator()() const {} };
// NORM (dib_sparse_blur_normalized): the store phase writes float((blurred - mean) / std) into the detector's zero-padded fp32
// batch (planar, or channels-last) instead of the fp16 image: the blur and dib_normalize_pad in one launch, for batches whose
// images need no resize (reference engine.py:107-110 + models/net_transforms.py:112-121, :238-247).  Same operations in the same
// order as the two launches (fp16 -> fp32, subtract, IEEE divide): bit-identical.
struct NormArgs {
  float mean[MAX_BATCH][4], std[MAX_BATCH][4];   // per image (descriptor order) and channel
  int Hp, Wp;                                     // the batch's padded height and width: pixels outside the image become 0
  int nhwc;                                       // 1: channels-last batch [B][Hp][Wp][3]; the launch then orders tiles channel-fastest
};
// LDS byte address of a __shared__ array (the tile function takes the integer: with a generic pointer to LDS handed through the
// call hipcc 7.2 emitted an illegal null test in some instantiations: "V_CMP_NE_U32_e32 0, $src_shared_base")
__device__ __forceinline__ unsigned lds_addr(unsigned *shared) { return (unsigned)(size_t)(__attribute__((address_space(3))) char *)shared; }
template <int ACC, bool L = false, bool STEP = false, typename Wait = NoWait, bool NORM = false>
__device__ __forceinline__ void blur_quad_tile_f16(const ImageDesc &d, const int *__restrict__ tab, int K, int ch, int tx, int ty,
                                                   const unsigned lds0, const int wave, const int early = 0, const int nsegs0 = 0,
                                                   const uint4 seg0 = uint4{0, 0, 0, 0}, const Wait wait_tables = Wait(),
                                                   const NormArgs *na = nullptr, const int img = 0) {
#pragma clang fp contract(off)
  constexpr int GQ = QGeom<L>::GQ;            // LDS rows a wave fills (11; large window: 15)
  constexpr int QPITCH = QGeom<L>::PITCH, QUAD_PITCH = QGeom<L>::PITCH_EL, LROWS = QGeom<L>::ROWS;   // shadow the standard geometry's constants
  const int H = d.H, W = d.W, w2 = W * 2;
  const int mode = pad_mode_for(K, H, W);
  const int pb = K / 2 - 1, pa = K / 2;
  // second (and last) scalar round trip of the prologue: segment count and first segment, requested together.  Written
  // out because hipcc turns these into vector loads + v_readfirstlane once an asm statement precedes them.
  const uint4 *segs = reinterpret_cast<const uint4 *>(tab + table_segs_off(K));
  int nsegs;
  int use_vruns = 0;
  uint4 seg;
  if (STEP && early) {
    nsegs = nsegs0 | 1 << 30;       // bit 30: "wait for the tables in front of the first tap loop" (no register of its own)
    seg = seg0;
  } else {
    // header words 4..7 in one request (cmax, K | geometry << 16, sum, segment count) next to the first segment
    unsigned __int128 r, hd;
    asm volatile("s_load_dwordx4 %0, %2, 0x10\n\ts_load_dwordx4 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(hd), "=&s"(r) : "s"((unsigned long long)tab), "s"((unsigned long long)segs));
    seg = make_uint4((unsigned)r, (unsigned)(r >> 32), (unsigned)(r >> 64), (unsigned)(r >> 96));
    nsegs = (int)(unsigned)(hd >> 96);
    // a table compacted for the other window geometry (include/dib.h: DIB_COMPACT_LARGE_WINDOW / DIB_WINDOW_LARGE must agree)
    // holds offsets for another LDS pitch: garbage pixels, silently.  The workgroup leaves its tile unwritten instead and says so in
    // the device's status word (the next call on the device returns DIB_EINVAL; blur_ops.sparse_blur checks on the host first).
    // (the step's single launch compacts its own tables, for the standard window: nothing to check there)
    if constexpr (!STEP) {
      const unsigned kw = (unsigned)(hd >> 32) >> 16;      // bit 0: large window, bit 1: the vertical-run groups are there
      if ((kw & 1u) != (L ? 1u : 0u)) report_and_exit(DIB_STATUS_GEOMETRY, (unsigned)(hd >> 32));
      // DIB_ACC_FAST16 on a table without groups (compacted without DIB_COMPACT_VRUNS, or a PSF beyond the compaction's LDS
      // stage: > 4,096 taps): the plain fused loop, taps in row-major order -- the same arithmetic, DIB_ACC_FMA16's result
      if constexpr (ACC == DIB_ACC_FAST16) use_vruns = (int)(kw >> 1) & 1;
    }
  }
  static_assert(HDR_NSEGS == 7 && HDR_K == 5 && HDR_WORDS == 8, "header words 4..7 in the asm above");
  const unsigned long long la = (unsigned long long)(tab + table_ltaps_q_off(K));
  const unsigned long long ltaps = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(la >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((unsigned)la);
  // STEP: the tile's origin lives in ONE scalar register (ty << 16 | tx; tiles per channel < 2^16, dib_sparse_blur checks) and is
  // unpacked where it is needed -- not kept as x0 and y0 across the window fill, where the step kernel has no register left.
  const int x0c = tx * QTILE_W, y0c = ty * TH, pxy = ty << 16 | tx;
  auto x0f = [&]() -> int { if constexpr (STEP) { int p = pxy; asm volatile("" : "+s"(p)); return (p & 0xffff) * QTILE_W; } else return x0c; };
  auto y0f = [&]() -> int { if constexpr (STEP) { int p = pxy; asm volatile("" : "+s"(p)); return (int)((unsigned)p >> 16) * TH; } else return y0c; };
  const __amdgpu_buffer_rsrc_t in_rsrc = plane_rsrc(d.in, ch, H, W);
  h2 acc[8];
  float acc32[16];       // DIB_ACC_FP32: row i, column j + 32 m at [4 i + m]
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = h2{0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) acc32[i] = 0.f;
  const int qb = wave * GQ;
  // Lane-derived values of the later phases are recomputed there from an opaque copy of the lane index (2-3 instructions):
  // hoisted to here they would stay live across the 44 outstanding window loads and spill (64 registers = 8 waves / SIMD).
  // (the lane index itself comes from v_mbcnt: no input register to keep either)
  auto fresh_lane = [&]() { int l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return l; };
  typedef unsigned lds_u2v __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) lds_u2v lds_u2;

#ifdef DIB_TIMELINE
  if (!L && !STEP && wave == 0 && fresh_lane() == 0)
    *(__attribute__((address_space(3))) unsigned long long *)(size_t)(lds0 + QGeom<false>::BYTES + 32) = __builtin_amdgcn_s_memrealtime();
#endif
  for (int sg = 0; sg < (STEP ? (nsegs & 0xffffff) : nsegs); ++sg) {
    if (sg > 0) seg = segs[sg];
    const Window w = window_of(seg);
    const int lane = fresh_lane();
    const int x0 = x0f(), y0 = y0f();
    // ---- fill: per LDS row the four values P[lane + 32 m] of element `lane` (large window: six values -- element 64 + lane,
    // owned by lanes 0-31, is {P[lane + 64], P[lane + 96], P[lane + 128], P[lane + 160]}, the first two shared with element lane)
    constexpr int NK = L ? 6 : 4;
    // DIB_ACC_FP32 (sixteen fp32 accumulators instead of eight packed ones) fills in TWO parts of six and five rows: with all
    // 44 values of a fill in flight next to the accumulators the kernel needs 73 registers, i.e. six waves per SIMD instead of
    // eight (42.5 us on the BASELINE batch against 33 us for the same arithmetic volume in FMA16 mode: slots are what hide a
    // workgroup's phases behind the others').
    constexpr int PARTS = (ACC == DIB_ACC_FP32 && !L) ? 2 : 1, GP = (GQ + PARTS - 1) / PARTS;
    short v[GP][NK];
    unsigned coff[NK];
    int soff[GQ];
    unsigned zmask = 0;
    const int c_first = x0 + pb - w.cmax, r_first = y0 + pb - w.rl;
    const bool zero_mode = mode == PAD_ZERO;
    if (!zero_mode && c_first >= 0 && c_first + 63 + 32 * (NK - 1) <= W - 1) {
      const unsigned c0 = 2u * (unsigned)(c_first + lane);
#pragma unroll
      for (int k = 0; k < NK; ++k) coff[k] = c0 + 64u * k;
    } else {
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        bool z;
        coff[k] = 2u * (unsigned)map_coord_sel(c_first + lane + 32 * k, W, pa, pb, mode, z);
        zmask |= z ? 1u << k : 0u;
      }
    }
    if (!zero_mode && r_first >= 0 && r_first + LROWS - 1 <= H - 1) {
      const int s0 = (r_first + qb) * w2;
#pragma unroll
      for (int g = 0; g < GQ; ++g) soff[g] = s0 + g * w2;
    } else {
      const int nrows = TH + (w.rl - w.rf);
#pragma unroll
      for (int g = 0; g < GQ; ++g) {
        bool zr;
        const int sr = map_coord_sel(r_first + min(qb + g, nrows - 1), H, pa, pb, mode, zr);
        zmask |= zr ? 1u << (8 + g) : 0u;
        soff[g] = sr * w2;
      }
    }
    const bool masked = __builtin_amdgcn_ballot_w64(zmask != 0) != 0;   // PAD_ZERO images only
#pragma unroll
    for (int part = 0; part < PARTS; ++part) {
#pragma unroll
      for (int g = 0; g < GP; ++g) {
        if (part * GP + g >= GQ) continue;
        const int so = __builtin_amdgcn_readfirstlane(soff[part * GP + g]);
#pragma unroll
        for (int k = 0; k < NK; ++k) v[g][k] = __builtin_amdgcn_raw_buffer_load_b16(in_rsrc, coff[k], so, 0);
      }
#ifdef DIB_TIMELINE
      if (!L && !STEP && sg == 0 && part == 0 && wave == 0 && fresh_lane() == 0)
        *(__attribute__((address_space(3))) unsigned long long *)(size_t)(lds0 + QGeom<false>::BYTES + 40) = __builtin_amdgcn_s_memrealtime();
#endif
      if (part == 0 && sg > 0) __syncthreads();  // every wave is done reading the previous window
      // Elements 0 .. 31 + column extent are read by the taps; writing all 56 of a row costs the same (an LDS store is
      // priced per instruction), lanes 56-63 own no element.  The two 16-bit values of a word are merged by v_perm_b32
      // straight from the load registers (the packing does not care what their high halves hold).
      const int wl = fresh_lane();
      const unsigned wp = lds0 + (unsigned)((qb + part * GP) * QPITCH + wl * 8);
      if constexpr (L) {     // elements lane (all lanes) and 64 + lane (lanes 0-31); zero-fill flags applied per value
#pragma unroll
        for (int g = 0; g < GP; ++g) {
          if (part * GP + g >= GQ) continue;
          unsigned u[NK];
#pragma unroll
          for (int k = 0; k < NK; ++k)
            u[k] = (masked && (((zmask >> k) & 1u) || ((zmask >> (8 + part * GP + g)) & 1u))) ? 0u : (unsigned)(unsigned short)v[g][k];
          lds_u2v e;
          e.x = u[0] | (u[1] << 16);
          e.y = u[2] | (u[3] << 16);
          *(lds_u2 *)(size_t)(wp + (unsigned)(g * QPITCH)) = e;
          if (wl < 32) {
            lds_u2v f;
            f.x = u[2] | (u[3] << 16);
            f.y = u[4] | (u[5] << 16);
            *(lds_u2 *)(size_t)(wp + (unsigned)(g * QPITCH + 512)) = f;
          }
        }
      } else if (wl < QUAD_PITCH) {
        if (!masked) {
#pragma unroll
          for (int g = 0; g < GP; ++g) {
            if (part * GP + g >= GQ) continue;
            lds_u2v e;
            typedef short s2v __attribute__((ext_vector_type(2)));
            e.x = __builtin_bit_cast(unsigned, s2v{v[g][0], v[g][1]});
            e.y = __builtin_bit_cast(unsigned, s2v{v[g][2], v[g][3]});
            *(lds_u2 *)(size_t)(wp + (unsigned)(g * QPITCH)) = e;
          }
        } else {
#pragma unroll
          for (int g = 0; g < GP; ++g) {
            if (part * GP + g >= GQ) continue;
            unsigned u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              u[k] = (((zmask >> k) & 1u) || ((zmask >> (8 + part * GP + g)) & 1u)) ? 0u : (unsigned)(unsigned short)v[g][k];
            lds_u2v e;
            e.x = u[0] | (u[1] << 16);
            e.y = u[2] | (u[3] << 16);
            *(lds_u2 *)(size_t)(wp + (unsigned)(g * QPITCH)) = e;
          }
        }
      }
    }
    __syncthreads();
#ifdef DIB_TIMELINE
    if (!L && !STEP && sg == 0 && wave == 0 && fresh_lane() == 0)
      *(__attribute__((address_space(3))) unsigned long long *)(size_t)(lds0 + QGeom<false>::BYTES + 16) = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (STEP) { if (sg == 0 && (nsegs >> 30)) wait_tables(); }
    const int tl = fresh_lane();
    const unsigned lane_addr = lds0 + (unsigned)((wave * 8 + (tl >> 5) * 4) * QPITCH + (tl & 31) * 8);
    if constexpr (ACC == DIB_ACC_FP32) {
      if (W - x0f() <= 64) tap_loop_quad_fp32<true, L>(acc32, ltaps, w.t0, w.n, lane_addr);
      else tap_loop_quad_fp32<false, L>(acc32, ltaps, w.t0, w.n, lane_addr);
    } else if constexpr (ACC == DIB_ACC_FAST16) {
      // the segment's vertical-run groups: records from index t0 on (dib_common.h: vgroups)
      const unsigned long long va = (unsigned long long)(tab + table_vgroups_off(K));
      const unsigned long long vg = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(va >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)va);
      if (!use_vruns) {
        if (W - x0f() <= 64) tap_loop_quad<true, true, L>(acc, ltaps, w.t0, w.n, lane_addr);
        else tap_loop_quad<true, false, L>(acc, ltaps, w.t0, w.n, lane_addr);
      } else if (W - x0f() <= 64) tap_loop_quad_vrun<true>(acc, vg, w.t0, lane_addr);
      else tap_loop_quad_vrun<false>(acc, vg, w.t0, lane_addr);
    } else {
      if (W - x0f() <= 64) tap_loop_quad<ACC == DIB_ACC_FMA16, true, L>(acc, ltaps, w.t0, w.n, lane_addr);
      else tap_loop_quad<ACC == DIB_ACC_FMA16, false, L>(acc, ltaps, w.t0, w.n, lane_addr);
    }
  }
  if constexpr (ACC == DIB_ACC_FP32) {      // the one rounding of this mode
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = h2{(_Float16)acc32[2 * i], (_Float16)acc32[2 * i + 1]};
  }
#ifdef DIB_TIMELINE
  if (!L && !STEP && wave == 0 && fresh_lane() == 0)
    *(__attribute__((address_space(3))) unsigned long long *)(size_t)(lds0 + QGeom<false>::BYTES + 24) = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- store: a store instruction writes 32 columns of row y (lanes 0-31) and of row y + 4 (lanes 32-63); lanes
  // outside the image get an out-of-range buffer offset and are dropped by the range check ----------------------------
  if constexpr (NORM) {
    const int Hp = na->Hp, Wp = na->Wp, nhwc = na->nhwc;
    const float m = na->mean[img][ch], sd = na->std[img][ch];
    const unsigned long long a = (unsigned long long)d.out + (nhwc ? (unsigned long long)ch * 4ull : (unsigned long long)ch * Hp * Wp * 4ull);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, Hp * Wp * 4 * (nhwc ? 3 : 1) - (nhwc ? ch * 4 : 0), 0x00020000);
    const int x0 = x0f(), y0 = y0f();
    const int sl = fresh_lane();
    const int yl = y0 + wave * 8 + (sl >> 5) * 4, xl = x0 + (sl & 31);
    const unsigned px = nhwc ? 12u : 4u;      // bytes between horizontally adjacent pixels of one channel
    const unsigned base = (unsigned)(yl * Wp + xl) * px, oob = 0x7ffffff0u;
    const bool inside = x0 + QTILE_W <= W && y0 + TH <= H;     // every pixel of the tile lies inside the image
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned ro = base + (unsigned)(i * Wp) * px;
      const float v[4] = {(float)acc[2 * i].x, (float)acc[2 * i].y, (float)acc[2 * i + 1].x, (float)acc[2 * i + 1].y};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = xl + 32 * q, row = yl + i;
        // inside the image: the normalised pixel; between the image and the batch's padded extent: the padding's zero; beyond: no store
        const bool in_img = inside || (row < H && col < W);
        const float f = in_img ? (v[q] - m) / sd : 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, f), out_rsrc, (in_img || (row < Hp && col < Wp)) ? ro + 32u * q * px : oob, 0, 0);
      }
    }
  } else {
    // STEP: the plane offset ch * H * W * 2 is formed again here (from an opaque copy of ch) instead of staying alive from the
    // input descriptor's across the segment loop: two scalar registers the step kernel does not have (it spilled them).
    int chs = ch;
    if constexpr (STEP) asm volatile("" : "+s"(chs));
    const __amdgpu_buffer_rsrc_t out_rsrc = plane_rsrc(d.out, chs, H, W);
    const int x0 = x0f(), y0 = y0f();
    const int sl = fresh_lane();
    const int yl = y0 + wave * 8 + (sl >> 5) * 4, xl = x0 + (sl & 31);
    const unsigned base = (unsigned)(yl * w2 + xl * 2), oob = 0x7ffffff0u;
    if (x0 + QTILE_W <= W && y0 + TH <= H) {
      // tile inside the image (5 of 6 at 800 x 1333): one address register, the row in the scalar offset, the column
      // block in the immediate offset, the high halves stored straight from the registers -- no vector instruction at all
      // (the general form below costs ~45 per wave).  Written out: there is no builtin for the d16_hi stores.
      typedef int i4v __attribute__((ext_vector_type(4)));
      const unsigned long long pa2 = (unsigned long long)d.out + (unsigned long long)chs * H * W * 2ull;   // as plane_rsrc
      const i4v rs = {__builtin_amdgcn_readfirstlane((int)(unsigned)pa2), __builtin_amdgcn_readfirstlane((int)(unsigned)(pa2 >> 32)) & 0xffff,
                      H * W * 2, 0x00020000};
      int so = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("buffer_store_short %0, %2, %3, %4 offen\n\t"
                     "buffer_store_short_d16_hi %0, %2, %3, %4 offen offset:64\n\t"
                     "buffer_store_short %1, %2, %3, %4 offen offset:128\n\t"
                     "buffer_store_short_d16_hi %1, %2, %3, %4 offen offset:192"
                     :: "v"(acc[2 * i]), "v"(acc[2 * i + 1]), "v"(base), "s"(rs), "s"(so) : "memory");
        so += w2;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool row_ok = yl + i < H;
      const unsigned a0 = __builtin_bit_cast(unsigned, acc[2 * i]), a1 = __builtin_bit_cast(unsigned, acc[2 * i + 1]);
      const unsigned ro = base + (unsigned)(i * w2);
      __builtin_amdgcn_raw_buffer_store_b16((short)(a0 & 0xffffu), out_rsrc, row_ok && xl < W ? ro : oob, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b16((short)(a0 >> 16), out_rsrc, row_ok && xl + 32 < W ? ro + 64u : oob, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b16((short)(a1 & 0xffffu), out_rsrc, row_ok && xl + 64 < W ? ro + 128u : oob, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b16((short)(a1 >> 16), out_rsrc, row_ok && xl + 96 < W ? ro + 192u : oob, 0, 0);
    }
  }
}

#ifdef DIB_TIMELINE
// scratch/timeline.py, scratch/timeline_native.py: per-workgroup residency (100 MHz wall clock), where it ran and when wave 0 passed
// its phases (prologue done, first window's loads issued, first window ready, taps done); never defined in the product build.
// The begin stamp and the record index wait in 16 extra bytes of LDS and the buffer pointer is a device global read (volatile) at the end, so
// that no value stays live across the tile function: a few more live SGPRs push the allocation from 80 to 96 (+16 for the trap
// handler's) and cost a wave per SIMD, and the measurement would not be of the shipped kernel.
__device__ unsigned long long *g_timeline;
#define DIB_TL_SLOT (*(unsigned long long *volatile *)&g_timeline)
constexpr int TL_WORD = QLDS_BYTES / 4;
#endif

// KC: the PSF canvas (128 or 256) as a compile-time constant -- table offsets, pads and the padding mode fold, ~50 scalar
// instructions of every workgroup's prologue.  An instruction on a workgroup's serial path costs launch time out of
// proportion (8 waves per SIMD: ~10 cycles per instruction and wave, times 3.2 rounds of workgroups).
// FLAT (ragged batches, dib_common.h: FlatBands): a 1-D grid of exactly the working workgroups; the image comes from one more
// scalar load (the XCD list's 16 entry offsets) in front of the descriptor's.
template <int ACC, int KC, bool FLAT = false>
__global__ __launch_bounds__(256, 8) void blur_quad_f16_kernel(BlurBatch batch, FlatBands fb) {
  constexpr int K = KC;
  extern __shared__ unsigned nlds[];
  int img_i = blockIdx.y, entry = blockIdx.x >> 3;
  if constexpr (FLAT) {
    typedef int i16v __attribute__((ext_vector_type(16)));
    const i16v bb = *reinterpret_cast<const i16v *>(fb.begin[blockIdx.x & 7]);
    const int len = bb[15];
    if (entry >= len) return;
    // Against the dispatcher's round robin.  A ragged batch is ONE round of workgroups, all resident at once, and the hardware deals a
    // list's workgroups to its XCD's 32 CUs in turn: CU j runs workgroups j, j + 32, j + 64, ... of the list (measured: the per-CU
    // tap sums of profiles/r5_native_timeline.txt are exactly those of that model).  The list is sorted by weight (heaviest image
    // first), so dealt straight CU 0 gets the heaviest entry of EVERY stride of 32 -- and a workgroup of the left-over partial stride
    // on top: 277 taps against a mean of 241 on the native-size batch, and the CUs with a seventh workgroup are the ones that finish
    // last whatever their taps are.  Every FULL stride is therefore walked backwards: the CUs that get the partial stride's extra
    // workgroup (0, 1, ...) get the lightest entry of every stride before it (15.9 -> 15.1-15.3 us; walking only every other
    // stride backwards evens the tap sums out best, 259 at most, and gains 0.1-0.3 us less: profiles/r6_native_order.txt).  Which
    // TILE a workgroup computes is all that changes.
    {
      const int row = entry >> 5, full = len >> 5;
      if (row < full && row < 32 && ((fb.rev_mask >> row) & 1u)) entry ^= 31;
    }
    int start = bb[0];
    img_i = 0;
#pragma unroll
    for (int k = 1; k < FLAT_MAX; ++k) {
      const bool ge = entry >= bb[k];
      img_i += ge ? 1 : 0;
      start = ge ? bb[k] : start;
    }
    entry -= start;
  }
#ifdef DIB_TIMELINE
  if (threadIdx.x == 0) {   // 48 more bytes of LDS
    *(unsigned long long *)(nlds + TL_WORD) = __builtin_amdgcn_s_memrealtime();
    nlds[TL_WORD + 2] = FLAT ? blockIdx.x : blockIdx.y * 1024 + blockIdx.x;   // x extent of the BASELINE launch: 832
    nlds[TL_WORD + 3] = img_i;
  }
#endif
  // Prologue = two scalar round trips.  First: the whole descriptor, K and the table base, requested together (left to
  // hipcc the fields are fetched one use at a time, a wait in front of each: six dependent round trips per workgroup).
  const ImageDesc d = batch.img[img_i];
  asm volatile("" ::"s"(d.in), "s"(d.out), "s"(d.C), "s"(d.H), "s"(d.W), "s"(d.table), "s"(d.tiles_x), "s"(d.tiles_y),
               "s"(d.inv_per_ch), "s"(d.inv_tiles_x), "s"(d.tab));
  const int per_ch = d.tiles_x * d.tiles_y;
  int local;
  if (!band_entry(d.C * per_ch, blockIdx.x & 7, entry, local)) return;
  const int ch = magic_div(local, d.inv_per_ch);
  local -= ch * per_ch;
  const int ty = magic_div(local, d.inv_tiles_x), tx = local - ty * d.tiles_x;
  blur_quad_tile_f16<ACC>(d, d.tab, K, ch, tx, ty, lds_addr(nlds), __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6));
#ifdef DIB_TIMELINE
  if (threadIdx.x == 0) {
    unsigned long long *tl = DIB_TL_SLOT;
    if (tl) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned long long *o = tl + 8 * (size_t)*(volatile unsigned *)(nlds + TL_WORD + 2);
      o[0] = *(volatile unsigned long long *)(nlds + TL_WORD);
      o[1] = __builtin_amdgcn_s_memrealtime();
      o[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
      o[3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) | (unsigned long long)*(volatile unsigned *)(nlds + TL_WORD + 3) << 32;   // XCC id | image << 32
      o[4] = *(volatile unsigned long long *)(nlds + TL_WORD + 4); o[5] = *(volatile unsigned long long *)(nlds + TL_WORD + 6);   // first window ready, taps done
      o[6] = *(volatile unsigned long long *)(nlds + TL_WORD + 8); o[7] = *(volatile unsigned long long *)(nlds + TL_WORD + 10);  // prologue done, first window's loads issued
    }
  }
#endif
}

// =============================================================================================
// The blur STEP as ONE launch (dib_blur_step: tap compaction + blur; reference models/blur_functions.py:92-100).
// As two launches the step pays the compaction twice over: its own ~5.5 us (8 workgroups, a latency chain) and a dependent
// kernel boundary in front of the blur (BENCH_r04: 47.2 us per step around a 38.9 us blur).  Here the grid's first columns
// compact (one 256-thread workgroup per PSF, tables written through to memory with sc1 stores), every blur workgroup behind
// them waits until a counter says all tables of THIS launch are done, and the compaction runs under the launch's own ramp.
//   Grid: x = ncx + (the blur's x), y = image; of the ncx leading columns only row 0's first n_psf blocks work, the others
//   exit at once (ncx is a multiple of 8, so `x & 7` still names the XCD list of a blur block).
//   Hand-off (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"), in two stages.
//   (1) As soon as a PSF's segments are final its FIRST segment goes out as two 8-byte {data, tag} words (sc1 stores, 32 copies):
//   a blur workgroup of that PSF's image polls one copy with sc1 loads and fills its first window from it -- the launch's first
//   fill (every resident workgroup's window at once) then runs while the compaction still makes the per-tap offsets.
//   (2) Everything else: payload stored sc1 -> every storing wave's s_waitcnt vmcnt(0) -> workgroup barrier -> agent-scope add
//   to EVERY replica of the counter (one wave instruction, 32 lanes, 32 lines); the blur workgroup looks at the counter in
//   front of its first tap loop and touches the tables -- scalar loads through the scalar cache -- only afterwards.  No acquire is needed for those: the scalar cache and this XCD's L2 are
//   invalidated at the kernel's start and nobody reads a table line before the counter says so (ONE counter for all tables:
//   neighbouring tables share cache lines), so neither can hold a stale copy.  Workgroups dispatched after the first rounds
//   (linear index >= STEP_EARLY) first look at replica `x & 7` through the scalar cache: a stale value can only read low (the
//   counter is monotonic) and falls back to the sc1 poll; early ones must not do that (they would leave the stale line in the
//   scalar cache for the rest of the launch).
//   Forward progress: the compacting workgroups have the grid's lowest indices and are dispatched first (observed dispatch
//   order, relied on by every decoupled look-back scan; nothing the hardware promises).  A poll that does not succeed within its
//   budget (StepSync::poll_budget, ~1 s) neither hangs nor traps: the workgroup reports DIB_STATUS_HANDOFF through the device's
//   status word and ends, the host finds the word on its next call and takes the single launch out of service (dib_step.hip).
// =============================================================================================
constexpr unsigned STEP_EARLY = 4096;
constexpr int STEP_LDS_EXTRA = 16;      // {sync pointer, target, poll budget} for StepWait, behind the window (19,728 B: still eight workgroups per CU)
#ifdef DIB_STEP_STAMPS
// Diagnostic build only (scratch/t_step_stamps.py): 100 MHz wall-clock stamps of the compacting workgroups (8 words each) and of
// the blur workgroups of grid row 0 (4 words each, behind them).
__device__ unsigned long long *g_step_stamps;
#endif
// The counter once more, in front of the first tap loop of a blur workgroup that started on the early record (by then it has
// nearly always been reached, so a fresh value lands in the scalar cache; a stale one only costs the sc1 polls behind it).
// The counter's copy is chosen by the XCD the workgroup runs on (a hardware register).
struct StepWait {
  unsigned lds_words;     // LDS byte address of {sync lo, sync hi, target, poll budget}, written by the kernel in front of the tile function
  // ONE asm statement on the registers the tap loops clobber anyway (s[36:41], v[44:48]: free between two of them); the four
  // words it needs wait in 16 bytes of LDS behind the window, not in scalar registers: the kernel runs at 78 of the 80 that
  // eight waves per SIMD allow, and three more live across the window fill spilled.
  //   s[36:37] = sync + 128 * XCD, s38 = target, s39 = poll budget; ready when (int)(counter - target) >= 0: first through the
  //   scalar cache, then (rare) sc1 vector polls of the same copy.  A wave whose budget runs out (StepSync::poll_budget: ~1 s)
  //   does not trap: it writes DIB_STATUS_HANDOFF and the launch's tag into the device's status word (pinned host memory, found
  //   through the code object's `dib_status_word`) and ends; dib_blur_step reports it on its next call (include/dib.h).
  __device__ __forceinline__ void operator()() const {
    static_assert(STEP_REPLICA_WORDS * 4 == 128, "the shift below");
#ifdef DIB_PORTABLE_TAPS
    {
      const uvec2 sp = lds_b64(lds_words), tb = lds_b64(lds_words + 8);
      const unsigned xcc = __builtin_amdgcn_s_getreg((2 << 11) | 20);       // HW_REG_XCC_ID[2:0]
      const unsigned *ctr = reinterpret_cast<const unsigned *>((((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)sp.y) << 32) |
                                                                 (unsigned)__builtin_amdgcn_readfirstlane((int)sp.x)) + 128ull * xcc);
      const unsigned target = (unsigned)__builtin_amdgcn_readfirstlane((int)tb.x), budget = (unsigned)__builtin_amdgcn_readfirstlane((int)tb.y);
      for (unsigned spins = 0;; ++spins) {
        if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) return;
        if (spins >= budget) report_and_exit(DIB_STATUS_HANDOFF, target);
        __builtin_amdgcn_s_sleep(2);
      }
    }
#endif
    asm volatile(
        "v_mov_b32 v48, %0\n\t"
        "ds_read_b64 v[46:47], v48\n\t"
        "ds_read_b64 v[44:45], v48 offset:8\n\t"
        "s_getreg_b32 s39, hwreg(20, 0, 3)\n\t"              /* HW_REG_XCC_ID[2:0] */
        "s_lshl_b32 s39, s39, 7\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 s36, v46\n\t"
        "v_readfirstlane_b32 s37, v47\n\t"
        "v_readfirstlane_b32 s38, v44\n\t"
        "s_add_u32 s36, s36, s39\n\t"
        "s_addc_u32 s37, s37, 0\n\t"
        "v_readfirstlane_b32 s39, v45\n\t"
        "s_load_dword s40, s[36:37], 0x0\n\t"
        "s_mov_b32 s41, 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_sub_u32 s40, s40, s38\n\t"
        "s_cmp_lt_i32 s40, 0\n\t"
        "s_cbranch_scc0 Ldibw_done%=\n\t"
        "v_mov_b32 v48, 0\n\t"
        "Ldibw_poll%=:\n\t"
        "global_load_dword v48, v48, s[36:37] sc1\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_readfirstlane_b32 s40, v48\n\t"
        "v_mov_b32 v48, 0\n\t"
        "s_sub_u32 s40, s40, s38\n\t"
        "s_cmp_lt_i32 s40, 0\n\t"
        "s_cbranch_scc0 Ldibw_done%=\n\t"
        "s_sleep 2\n\t"
        "s_add_u32 s41, s41, 1\n\t"
        "s_cmp_lt_u32 s41, s39\n\t"
        "s_cbranch_scc1 Ldibw_poll%=\n\t"
        /* budget spent: {DIB_STATUS_HANDOFF, tag} -> the status word, and this wave ends (the s_add's literal sits 4 bytes, the
           s_addc's 12 bytes behind the address s_getpc returns: the offsets hipcc itself emits for this sequence) */
        "s_getpc_b64 s[36:37]\n\t"
        "s_add_u32 s36, s36, dib_status_word@rel32@lo+4\n\t"
        "s_addc_u32 s37, s37, dib_status_word@rel32@hi+12\n\t"
        "s_load_dwordx2 s[36:37], s[36:37], 0x0\n\t"
        "v_mov_b32 v47, s38\n\t"
        "v_mov_b32 v46, 1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "global_store_dword v48, v47, s[36:37] offset:4 sc0 sc1\n\t"
        "global_store_dword v48, v46, s[36:37] sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_endpgm\n\t"
        "Ldibw_done%=:\n\t"
#ifdef DIB_STEP_POLLSTATS
        "s_cmp_lt_u32 s41, 33\n\t"                /* only the long waits are recorded: thousands of same-address atomics per launch cost milliseconds */
        "s_cbranch_scc1 Ldibw_nostat%=\n\t"
        "s_getpc_b64 s[36:37]\n\t"
        "s_add_u32 s36, s36, dib_poll_stats@rel32@lo+4\n\t"
        "s_addc_u32 s37, s37, dib_poll_stats@rel32@hi+12\n\t"
        "v_mov_b32 v48, 0\n\t"
        "global_load_dword v47, v48, s[36:37] offset:4 sc1\n\t"      /* only a new maximum is recorded */
        "s_waitcnt vmcnt(0)\n\t"
        "v_readfirstlane_b32 s40, v47\n\t"
        "s_cmp_le_u32 s41, s40\n\t"
        "s_cbranch_scc1 Ldibw_nostat%=\n\t"
        "v_mov_b32 v47, s41\n\t"
        "global_atomic_umax v48, v47, s[36:37] offset:4\n\t"
        "Ldibw_nostat%=:\n\t"
#endif
        :: "s"(lds_words)
        : "s36", "s37", "s38", "s39", "s40", "s41", "v44", "v45", "v46", "v47", "v48", "scc", "memory");
  }
};
static_assert(DIB_STATUS_HANDOFF == 1u, "the v_mov_b32 v46, 1 above");

template <int ACC>
__global__ __launch_bounds__(256, 8) void blur_step_f16_kernel(BlurBatch batch, StepSync sy, PsfPtrs psfs) {
  constexpr int K = 128;
  extern __shared__ unsigned nlds[];
  // First scalar round trip: the launch's hand-off words and the image descriptor, requested together (a compacting
  // workgroup fetches a descriptor it does not need rather than every blur workgroup waiting twice).
  const ImageDesc d = batch.img[blockIdx.y];
  asm volatile("" ::"s"(d.in), "s"(d.out), "s"(d.C), "s"(d.H), "s"(d.W), "s"(d.table), "s"(d.tiles_x), "s"(d.tiles_y),
               "s"(d.inv_per_ch), "s"(d.inv_tiles_x), "s"(d.tab), "s"(sy.sync), "s"(sy.target), "s"(sy.ncx), "s"(sy.row), "s"(sy.rec));
  // the ONE consumer of the thread-index register in this kernel (dib_compact_dev.h: compact_psf_f16_wg256)
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (blockIdx.x < (unsigned)sy.ncx) {
    if (blockIdx.y == 0 && blockIdx.x < (unsigned)sy.n_psf && !(sy.flags & COMPACT_DEBUG_NOSIGNAL)) {
      unsigned *rec = sy.rec + (size_t)blockIdx.x * (STEP_REPLICAS * STEP_REC_WORDS);
#ifdef DIB_STEP_STAMPS
      unsigned long long *dbg = *(unsigned long long *volatile *)&g_step_stamps;
      if (dbg) dbg += 8 * blockIdx.x;
      compact_psf_f16_wg256<true, true>(psfs.p[blockIdx.x], sy.flags, sy.tables + (size_t)blockIdx.x * table_words(K), (lds_u32 *)nlds, wave, rec, sy.target, dbg);
#else
      compact_psf_f16_wg256<true, true>(psfs.p[blockIdx.x], sy.flags, sy.tables + (size_t)blockIdx.x * table_words(K), (lds_u32 *)nlds, wave, rec, sy.target);
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave: its sc1 stores have left
      __syncthreads();
      const int t = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#ifdef DIB_STEP_STAMPS
      if (dbg && t == 0) dbg[6] = __builtin_amdgcn_s_memrealtime();
#endif
      if (t < STEP_REPLICAS)
        __hip_atomic_fetch_add(sy.sync + t * STEP_REPLICA_WORDS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef DIB_STEP_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (dbg && t == 0) dbg[7] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    return;
  }
  const unsigned bx = blockIdx.x - (unsigned)sy.ncx;
  const int per_ch = d.tiles_x * d.tiles_y;
  int local;
  if (!band_entry(d.C * per_ch, bx & 7, bx >> 3, local)) __builtin_amdgcn_endpgm();
#ifdef DIB_STEP_STAMPS
  unsigned long long *bdbg = *(unsigned long long *volatile *)&g_step_stamps;
  if (bdbg && blockIdx.y == 0 && wave == 0) { bdbg += 8 * MAX_BATCH + 4 * bx; bdbg[0] = __builtin_amdgcn_s_memrealtime(); } else bdbg = nullptr;
#endif
  // ---- are the tables of this launch complete?  (the counter, through the scalar cache: late workgroups only) ----
  int early = 1;      // an integer in ONE scalar register (as a bool hipcc keeps it as a 64-bit lane mask and spills that)
  if (blockIdx.y * (unsigned)sy.row + blockIdx.x >= STEP_EARLY) {
    unsigned c;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c) : "s"(sy.sync + (blockIdx.x & 7) * STEP_REPLICA_WORDS) : "memory");
    early = (int)((c - sy.target) >> 31);
  }
  asm volatile("" : "+s"(early));
  // the counter once more, in front of the first tap loop of a workgroup that started early: StepWait below
  int nsegs0 = 0;
  uint4 seg0 = make_uint4(0, 0, 0, 0);
  if (early) {   // ---- the first segment of this image's PSF, from the compaction's early record (sc1 polls of one copy) ----
    const unsigned long long *pr = reinterpret_cast<const unsigned long long *>(sy.rec + ((size_t)d.table * STEP_REPLICAS + (blockIdx.x & (STEP_REPLICAS - 1))) * STEP_REC_WORDS);
#ifdef DIB_STEP_POLLSTATS
    unsigned pspins = 0;
#endif
    for (unsigned spins = 0;; ++spins) {
#ifdef DIB_STEP_POLLSTATS
      pspins = spins;
#endif
      const unsigned long long ra = __hip_atomic_load(pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long rb = __hip_atomic_load(pr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned ta = __builtin_amdgcn_readfirstlane((unsigned)(ra >> 32)), tb = __builtin_amdgcn_readfirstlane((unsigned)(rb >> 32));
      if (ta == sy.target && tb == sy.target) {
        const unsigned a = __builtin_amdgcn_readfirstlane((unsigned)ra), b2 = __builtin_amdgcn_readfirstlane((unsigned)rb);
        nsegs0 = (int)(a >> 16);
        seg0 = make_uint4(0u, a & 0xffffu, b2 >> 16, b2 & 0xffffu);
        break;
      }
      if (spins > sy.poll_budget) report_and_exit(DIB_STATUS_HANDOFF, sy.target);      // no trap: see dib_status_word
      __builtin_amdgcn_s_sleep(2);
    }
#ifdef DIB_STEP_POLLSTATS
    // (only a new maximum is recorded: thousands of same-address atomics per launch cost milliseconds)
    if (pspins > 32 && wave == 0 && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0 &&
        pspins > __hip_atomic_load(&dib_poll_stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(&dib_poll_stats[0], pspins);
#endif
    // what StepWait needs, into the 16 bytes of LDS behind the window (every wave's lane 0 writes the same three words)
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {
      nlds[QLDS_BYTES / 4] = (unsigned)(unsigned long long)sy.sync;
      nlds[QLDS_BYTES / 4 + 1] = (unsigned)((unsigned long long)sy.sync >> 32);
      nlds[QLDS_BYTES / 4 + 2] = sy.target;
      nlds[QLDS_BYTES / 4 + 3] = sy.poll_budget;
    }
    asm volatile("" ::: "memory");
  }
#ifdef DIB_STEP_STAMPS
  if (bdbg) bdbg[1] = __builtin_amdgcn_s_memrealtime();
#endif
  const int ch = magic_div(local, d.inv_per_ch);
  local -= ch * per_ch;
  const int ty = magic_div(local, d.inv_tiles_x), tx = local - ty * d.tiles_x;
  blur_quad_tile_f16<ACC, false, true>(d, d.tab, K, ch, tx, ty, lds_addr(nlds), wave, early, nsegs0, seg0, StepWait{(unsigned)(size_t)(__attribute__((address_space(3))) char *)nlds + QLDS_BYTES});
#ifdef DIB_STEP_STAMPS
  if (bdbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); bdbg[2] = __builtin_amdgcn_s_memrealtime(); }
#endif
  // A hard end: hipcc otherwise funnels every exit through one block placed behind the compaction code and keeps values of
  // this path alive for it across the tile function, whose window fill has no register to spare (a spill to scratch memory).
  __builtin_amdgcn_endpgm();
}

// The blur with the fused fp32 normalising store (dib_sparse_blur_normalized): blur_quad_f16_kernel's body, NORM store phase.
template <int ACC, int KC>
__global__ __launch_bounds__(256, 8) void blur_quad_f16_norm_kernel(BlurBatch batch, NormArgs na) {
  constexpr int K = KC;
  extern __shared__ unsigned nlds[];
  const ImageDesc d = batch.img[blockIdx.y];
  asm volatile("" ::"s"(d.in), "s"(d.out), "s"(d.C), "s"(d.H), "s"(d.W), "s"(d.table), "s"(d.tiles_x), "s"(d.tiles_y),
               "s"(d.inv_per_ch), "s"(d.inv_tiles_x), "s"(d.tab));
  const int per_ch = d.tiles_x * d.tiles_y;
  int local;
  if (!band_entry(d.C * per_ch, blockIdx.x & 7, blockIdx.x >> 3, local)) return;
  int ch;
  if (na.nhwc) {   // channel fastest: the three workgroups that complete a pixel's 12 bytes are dispatched back to back on one XCD
    const int t = magic_div(local, 0x55555556u);      // local / 3 (exact for local < 2^31)
    ch = local - 3 * t;
    local = t;
  } else {
    ch = magic_div(local, d.inv_per_ch);
    local -= ch * per_ch;
  }
  const int ty = magic_div(local, d.inv_tiles_x), tx = local - ty * d.tiles_x;
  blur_quad_tile_f16<ACC, false, false, NoWait, true>(d, d.tab, K, ch, tx, ty, lds_addr(nlds), __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), 0, 0,
                                                      uint4{0, 0, 0, 0}, NoWait(), &na, (int)blockIdx.y);
}

// DIB_ACC_FP32 on the default tiles: the same kernel with sixteen fp32 accumulators per lane instead of eight packed ones (the
// window fill goes in two parts to stay inside 64 registers: see the tile function).  Half the vector-ALU work of the bit-exact
// mode per pixel and tap.
template <int KC>
__global__ __launch_bounds__(256, 8) void blur_quad_f32acc_kernel(BlurBatch batch) {
  constexpr int K = KC;
  extern __shared__ unsigned nlds[];
  const ImageDesc d = batch.img[blockIdx.y];
  asm volatile("" ::"s"(d.in), "s"(d.out), "s"(d.C), "s"(d.H), "s"(d.W), "s"(d.table), "s"(d.tiles_x), "s"(d.tiles_y),
               "s"(d.inv_per_ch), "s"(d.inv_tiles_x), "s"(d.tab));
  const int per_ch = d.tiles_x * d.tiles_y;
  int local;
  if (!band_entry(d.C * per_ch, blockIdx.x & 7, blockIdx.x >> 3, local)) return;
  const int ch = magic_div(local, d.inv_per_ch);
  local -= ch * per_ch;
  const int ty = magic_div(local, d.inv_tiles_x), tx = local - ty * d.tiles_x;
  blur_quad_tile_f16<DIB_ACC_FP32>(d, d.tab, K, ch, tx, ty, lds_addr(nlds), __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6));
}

// The same tile function on the LARGE window (39.9 KB of LDS: 4 workgroups per CU, so up to 128 registers per lane cost
// nothing; the fill keeps 78 window values in flight instead of 44).
template <int ACC, int KC>
__global__ __launch_bounds__(256, 4) void blur_quad_large_f16_kernel(BlurBatch batch) {
  constexpr int K = KC;
  extern __shared__ unsigned nlds[];
  const ImageDesc d = batch.img[blockIdx.y];
  const int per_ch = d.tiles_x * d.tiles_y;
  int local;
  if (!band_entry(d.C * per_ch, blockIdx.x & 7, blockIdx.x >> 3, local)) return;
  const int ch = magic_div(local, d.inv_per_ch);
  local -= ch * per_ch;
  const int ty = magic_div(local, d.inv_tiles_x), tx = local - ty * d.tiles_x;
  blur_quad_tile_f16<ACC, true>(d, d.tab, K, ch, tx, ty, lds_addr(nlds), __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6));
}

template <int ACC>
__global__ __launch_bounds__(64 * NW, NW) void blur_tiled_f16_kernel(BlurBatch batch, const int *__restrict__ tables, int K,
                                                                     unsigned long long *dbg) {
  extern __shared__ uint2 lds[];
  const ImageDesc &d = batch.img[blockIdx.y];
  const int per_ch = d.tiles_x * d.tiles_y;
  int local;
  if (batch.xcd_bands) {
    if (!band_entry(d.C * per_ch, blockIdx.x & 7, blockIdx.x >> 3, local)) return;
  } else {
    local = blockIdx.x;              // flat order (the traffic experiment): tile index = block index
    if (local >= d.C * per_ch) return;
  }
  const int ch = magic_div(local, d.inv_per_ch);
  local -= ch * per_ch;
  const int ty = magic_div(local, d.inv_tiles_x), tx = local - ty * d.tiles_x;
  blur_tile_f16<ACC>(d, tables + (size_t)d.table * table_words(K), K, ch, tx, ty, lds, dbg);
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
extern "C" void dib_debug_set_stamp_buffer(void *dev_ptr) {
  g_stamp_buffer = (unsigned long long *)dev_ptr;
#ifdef DIB_STEP_STAMPS
  (void)hipMemcpyToSymbol(HIP_SYMBOL(dib::g_step_stamps), &g_stamp_buffer, sizeof(g_stamp_buffer));
#endif
#ifdef DIB_TIMELINE
  (void)hipMemcpyToSymbol(HIP_SYMBOL(dib::g_timeline), &g_stamp_buffer, sizeof(g_stamp_buffer));
#endif
}
#ifdef DIB_TIMELINE
#define TL_EXTRA 48
#else
#define TL_EXTRA 0
#endif
// Tile order of the tiled kernel: 1 = per-XCD bands (default), 0 = flat (the traffic experiment of DESIGN.md section 4).
static int g_xcd_bands = 1;
// Tile shape serving fp16 images in the bit-exact and FMA16 modes (all shapes bit-identical; tests/test_blur_gpu.py
// compares them): 0 = 128 x 32 "quad" tiles (8-byte LDS elements), 8 workgroups per CU (default), 1 = 256 x 32 tiles,
// 4 per CU (what DIB_ACC_FP32 always runs on: the second, independent tiled implementation).
static int shape_from_env() {   // DIB_BLUR_SHAPE=0|1 runs a whole test suite on one shape
  const char *e = getenv("DIB_BLUR_SHAPE");
  return e && e[0] == '1' && !e[1] ? 1 : 0;
}
static int g_shape = shape_from_env();
extern "C" void dib_debug_set_shape(int shape) { g_shape = shape == 1 ? 1 : 0; }
extern "C" void dib_debug_set_tile_order(int xcd_bands) { g_xcd_bands = xcd_bands ? 1 : 0; }
// Ragged batches on the default tiles: 1 = the 1-D grid of working workgroups (default), 0 = the 2-D grid (A/B runs, tests that
// compare the two).
static int g_flat_grid = !(getenv("DIB_FLAT_GRID") && getenv("DIB_FLAT_GRID")[0] == '0');
extern "C" void dib_debug_set_flat_grid(int on) { g_flat_grid = on ? 1 : 0; }
// ... and on it, every full stride of 32 workgroups of an XCD's list walked backwards (default; 0 in A/B runs)
static int g_flat_snake = !(getenv("DIB_FLAT_SNAKE") && getenv("DIB_FLAT_SNAKE")[0] == '0');
extern "C" void dib_debug_set_flat_snake(int on) { g_flat_snake = on ? 1 : 0; }
static int g_flat_mask = -1;      // experiments: an explicit stride mask
extern "C" void dib_debug_set_flat_mask(int mask) { g_flat_mask = mask; }

namespace {
// Per-device launch state: the dynamic-LDS opt-in is a per-device function attribute.  Guarded by a mutex: entry
// points may be called from several host threads.
struct DeviceState {
  bool ready = false;
  volatile unsigned *status = nullptr;   // DIB_STATUS_WORDS words of pinned host memory the device's kernels report into (dib_status_word)
  bool step_single_off = false;          // a hand-off timed out on this device: dib_blur_step stays on two launches
  unsigned handoff_timeouts = 0;
};
std::mutex g_dev_mutex;
DeviceState g_dev[64];

template <typename Kern> hipError_t opt_in(Kern k, int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int prepare_device() {
  int dev = 0;
  DIB_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { set_error("dib_sparse_blur: device index %d out of range", dev); return DIB_EINVAL; }
  std::lock_guard<std::mutex> lock(g_dev_mutex);
  DeviceState &st = g_dev[dev];
  if (!st.ready) {
    if (!st.status) {   // the status block: coherent pinned host memory, its device address into the code object's global
      void *host = nullptr, *devp = nullptr;
      DIB_HIP_CHECK(hipHostMalloc(&host, DIB_STATUS_WORDS * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
      memset(host, 0, DIB_STATUS_WORDS * sizeof(unsigned));
      DIB_HIP_CHECK(hipHostGetDevicePointer(&devp, host, 0));
      DIB_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(dib_status_word), &devp, sizeof(devp)));
      st.status = (volatile unsigned *)host;
    }
    DIB_HIP_CHECK(opt_in(blur_tiled_f16_kernel<DIB_ACC_BITEXACT>, LDS_BYTES));
    DIB_HIP_CHECK(opt_in(blur_tiled_f16_kernel<DIB_ACC_FP32>, LDS_BYTES));
    DIB_HIP_CHECK(opt_in(blur_tiled_f16_kernel<DIB_ACC_FMA16>, LDS_BYTES));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_kernel<DIB_ACC_BITEXACT, 128>), QLDS_BYTES + TL_EXTRA));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_kernel<DIB_ACC_FMA16, 128>), QLDS_BYTES + TL_EXTRA));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_kernel<DIB_ACC_BITEXACT, 128, true>), QLDS_BYTES + TL_EXTRA));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_kernel<DIB_ACC_FMA16, 128, true>), QLDS_BYTES + TL_EXTRA));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_kernel<DIB_ACC_FAST16, 128>), QLDS_BYTES + TL_EXTRA));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_kernel<DIB_ACC_FAST16, 128, true>), QLDS_BYTES + TL_EXTRA));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_kernel<DIB_ACC_BITEXACT, 256>), QLDS_BYTES + TL_EXTRA));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_kernel<DIB_ACC_FMA16, 256>), QLDS_BYTES + TL_EXTRA));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_norm_kernel<DIB_ACC_BITEXACT, 128>), QLDS_BYTES));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_norm_kernel<DIB_ACC_FMA16, 128>), QLDS_BYTES));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_norm_kernel<DIB_ACC_BITEXACT, 256>), QLDS_BYTES));
    DIB_HIP_CHECK(opt_in((blur_quad_f16_norm_kernel<DIB_ACC_FMA16, 256>), QLDS_BYTES));
    DIB_HIP_CHECK(opt_in((blur_quad_f32acc_kernel<128>), QLDS_BYTES));
    DIB_HIP_CHECK(opt_in((blur_quad_f32acc_kernel<256>), QLDS_BYTES));
    DIB_HIP_CHECK(opt_in((blur_step_f16_kernel<DIB_ACC_BITEXACT>), QLDS_BYTES + STEP_LDS_EXTRA));
    DIB_HIP_CHECK(opt_in((blur_step_f16_kernel<DIB_ACC_FMA16>), QLDS_BYTES + STEP_LDS_EXTRA));
    DIB_HIP_CHECK(opt_in((blur_quad_large_f16_kernel<DIB_ACC_BITEXACT, 128>), QGeom<true>::BYTES));
    DIB_HIP_CHECK(opt_in((blur_quad_large_f16_kernel<DIB_ACC_FMA16, 128>), QGeom<true>::BYTES));
    DIB_HIP_CHECK(opt_in((blur_quad_large_f16_kernel<DIB_ACC_BITEXACT, 256>), QGeom<true>::BYTES));
    DIB_HIP_CHECK(opt_in((blur_quad_large_f16_kernel<DIB_ACC_FMA16, 256>), QGeom<true>::BYTES));
    st.ready = true;
  }
  return DIB_OK;
}
}  // namespace

// ---- Device status (include/dib.h) ----------------------------------------------------------------------------------
// What the host does with a code a kernel left in the status word: report it once, clear it, and -- for a hand-off that timed
// out -- keep the device on compaction + blur as two launches from then on (whatever held the compacting workgroups back may
// do so again, and a step that leaves tiles unwritten is worse than one that is 2 us slower).
static int consume_status_locked(DeviceState &st, int dev, const char *who) {
  if (!st.status) return DIB_OK;
  const unsigned code = st.status[0];
  if (code == 0) return DIB_OK;
  const unsigned detail = st.status[1];
  st.status[1] = 0;
  st.status[0] = 0;
  if (code == DIB_STATUS_HANDOFF) {
    st.step_single_off = true;
    ++st.handoff_timeouts;
    set_error("%s: an earlier dib_blur_step on device %d ran out of its poll budget waiting for the in-launch tap compaction (tag %u): "
              "the images of that step are incomplete.  The single launch is now off for this device (compaction + blur as two "
              "launches from here on); re-issue the batch", who, dev, detail);
    return DIB_ETIMEOUT;
  }
  if (code == DIB_STATUS_GEOMETRY) {
    set_error("%s: an earlier launch on device %d was handed a tap table compacted for the %s LDS window (DIB_COMPACT_LARGE_WINDOW "
              "and DIB_WINDOW_LARGE / DIB_STEP_LARGE_WINDOW must agree): its images were left unwritten", who, dev,
              ((detail >> 16) & 1u) ? "large" : "standard");
    return DIB_EINVAL;
  }
  set_error("%s: unknown device status %u on device %d", who, code, dev);
  return DIB_EHIP;
}
int dib::consume_device_status(const char *who) {
  int dev = 0;
  DIB_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return DIB_OK;
  std::lock_guard<std::mutex> lock(g_dev_mutex);
  return consume_status_locked(g_dev[dev], dev, who);
}
unsigned dib::handoff_generation(int dev) {
  if (dev < 0 || dev >= 64) return 0;
  std::lock_guard<std::mutex> lock(g_dev_mutex);
  return g_dev[dev].handoff_timeouts;
}
extern "C" int dib_device_status(int clear) {
  int dev = 0;
  DIB_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return DIB_OK;
  std::lock_guard<std::mutex> lock(g_dev_mutex);
  if (clear) return consume_status_locked(g_dev[dev], dev, "dib_device_status");
  const unsigned code = g_dev[dev].status ? g_dev[dev].status[0] : 0u;
  return code == 0 ? DIB_OK : code == DIB_STATUS_HANDOFF ? DIB_ETIMEOUT : code == DIB_STATUS_GEOMETRY ? DIB_EINVAL : DIB_EHIP;
}
// Test hooks: the hand-off's poll budget (1 makes every first-round workgroup of a single launch give up: the compaction takes
// ~4 us), and whether the single launch is in service on the current device (-1 = ask only; returns the state before the call).
static unsigned g_poll_budget = 1u << 20;
static int g_step_nosignal = 0;
extern "C" void dib_debug_set_step_poll_budget(unsigned polls) { g_poll_budget = polls ? polls : 1u << 20; }
// the single launch's compacting workgroups do nothing: every blur workgroup of the launch runs out of its poll budget
extern "C" void dib_debug_set_step_nosignal(int on) { g_step_nosignal = on ? 1 : 0; }
extern "C" int dib_debug_step_single_launch(int on) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  std::lock_guard<std::mutex> lock(g_dev_mutex);
  const int was = g_dev[dev].step_single_off ? 0 : 1;
  if (on >= 0) g_dev[dev].step_single_off = on == 0;
  return was;
}
#ifdef DIB_STEP_POLLSTATS
extern "C" int dib_debug_poll_stats(unsigned *out4, int reset) {
  DIB_HIP_CHECK(hipMemcpyFromSymbol(out4, HIP_SYMBOL(dib_poll_stats), 4 * sizeof(unsigned)));
  if (reset) { const unsigned z[4] = {0, 0, 0, 0}; DIB_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(dib_poll_stats), z, sizeof(z))); }
  return DIB_OK;
}
#endif

namespace {
// Per-image argument checks shared by dib_sparse_blur and the blur step's single launch.
int check_images(const void *const *in_dev, void *const *out_dev, const int *C, const int *H, const int *W, const int *table_index, int B,
                 int K, int num_tables) {
  for (int i = 0; i < B; ++i) {
    if (table_index[i] < 0) continue;
    if (!in_dev[i] || !out_dev[i] || C[i] <= 0 || H[i] <= 0 || W[i] <= 0) {
      set_error("dib_sparse_blur: image %d has a null pointer or empty shape", i);
      return DIB_EINVAL;
    }
    if (in_dev[i] == out_dev[i]) { set_error("dib_sparse_blur: image %d: out aliases in", i); return DIB_EINVAL; }
    {   // the tile index split (magic_div) is exact while channels x tiles-per-channel^2 < 2^32: ~150 Mpixel per channel
      const unsigned long long per_ch = (unsigned long long)((W[i] + 127) / 128) * ((H[i] + 31) / 32);
      if ((unsigned long long)C[i] * per_ch * per_ch >= 0x100000000ull) {
        set_error("dib_sparse_blur: image %d (%d x %d x %d) is too large", i, C[i], H[i], W[i]);
        return DIB_ESHAPE;
      }
    }
    // F.pad(mode='reflect') raises unless pad < dim (blur_functions.py:59 with pads 63/64)
    if (K == 128 && !(H[i] < 64 || W[i] < 64) && (H[i] == 64 || W[i] == 64)) {
      set_error("Padding size should be less than the corresponding input dimension (image %d is %dx%d)", i, H[i], W[i]);
      return DIB_ESHAPE;
    }
  }
  if (num_tables <= 0) { set_error("dib_sparse_blur: num_tables must be positive"); return DIB_EINVAL; }
  for (int i = 0; i < B; ++i)
    if (table_index[i] >= num_tables) { set_error("dib_sparse_blur: table_index[%d] = %d out of range", i, table_index[i]); return DIB_EINVAL; }
  return DIB_OK;
}

// Descriptor of image i for the default ("quad") tiles.
ImageDesc quad_desc(const void *in, void *out, int C, int H, int W, int table, const int *tables, int K, int tile_begin) {
  ImageDesc d;
  d.in = in; d.out = out; d.C = C; d.H = H; d.W = W; d.table = table;
  d.tiles_x = (W + QTILE_W - 1) / QTILE_W;
  d.tiles_y = (H + TH - 1) / TH;
  d.inv_per_ch = magic_inverse((unsigned)(d.tiles_x * d.tiles_y));
  d.inv_tiles_x = magic_inverse((unsigned)d.tiles_x);
  d.tab = tables + (size_t)table * table_words(K);
  d.tile_begin = tile_begin;
  return d;
}

// x extent of the quad grid: 8 x the longest per-XCD band of the launch's images
int quad_grid_x(const BlurBatch &b) {
  int gx = 0;
  for (int k = 0; k < b.n; ++k) {
    const int T = b.img[k].C * b.img[k].tiles_x * b.img[k].tiles_y;
    int longest = 0;
    for (int x = 0; x < 8; ++x) {
      const int len = (((x + 1) * T) >> 3) - ((x * T) >> 3);
      longest = len > longest ? len : longest;
    }
    gx = 8 * longest > gx ? 8 * longest : gx;
  }
  return gx;
}
}  // namespace

// The step's single launch: off with DIB_STEP_FUSED=0 (A/B runs and the tests that compare the two paths).
static int step_fused_from_env() {
  const char *e = getenv("DIB_STEP_FUSED");
  return !(e && e[0] == '0' && !e[1]);
}
static int g_step_fused = step_fused_from_env();
extern "C" void dib_debug_set_step_fused(int on) { g_step_fused = on ? 1 : 0; }

int dib::blur_step_fused_launch(const void *const *psf_ptrs, int num_psfs, int normalize, const void *const *in_dev, void *const *out_dev,
                                const int *C, const int *H, const int *W, const int *table_index, int B, int acc_mode, int *tables,
                                unsigned *sync, unsigned *rec, unsigned target, hipStream_t s) {
  if (!g_step_fused || g_shape != 0 || num_psfs > MAX_BATCH || (acc_mode != DIB_ACC_BITEXACT && acc_mode != DIB_ACC_FMA16)) return 1;
  if (B < 0 || (B > 0 && (!in_dev || !out_dev || !C || !H || !W || !table_index))) {
    set_error("dib_blur_step: null pointer or negative batch");
    return DIB_EINVAL;
  }
  int active = 0;
  for (int i = 0; i < B; ++i) active += table_index[i] >= 0;
  if (active == 0 || active > MAX_BATCH) return 1;
  if (int rc = check_images(in_dev, out_dev, C, H, W, table_index, B, 128, num_psfs)) return rc;
  PsfPtrs pp;
  for (int i = 0; i < num_psfs; ++i) {
    if (!psf_ptrs[i] || ((uintptr_t)psf_ptrs[i] & 15) != 0) { set_error("dib_psf_compact: PSF %d is null or not 16-byte aligned", i); return DIB_EINVAL; }
    pp.p[i] = psf_ptrs[i];
  }
  if (int rc = prepare_device()) return rc;
  {   // a hand-off timed out on this device before: two launches
    int dev = 0;
    DIB_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    if (g_dev[dev].step_single_off) return 1;
  }
  BlurBatch tiled;
  tiled.n = 0;
  int tiles = 0;
  for (int i = 0; i < B; ++i) {
    if (table_index[i] < 0) continue;
    const ImageDesc d = quad_desc(in_dev[i], out_dev[i], C[i], H[i], W[i], table_index[i], tables, 128, tiles);
    tiled.tile_begin[tiled.n] = tiles;
    tiles += d.C * d.tiles_x * d.tiles_y;
    tiled.img[tiled.n++] = d;
  }
  for (int k = tiled.n; k <= MAX_BATCH; ++k) tiled.tile_begin[k] = tiles;
  tiled.total_tiles = tiles;
  tiled.xcd_bands = 1;
  StepSync sy;
  sy.sync = sync; sy.rec = rec; sy.target = target; sy.n_psf = num_psfs; sy.ncx = (num_psfs + 7) & ~7; sy.tables = tables;
  sy.row = sy.ncx + quad_grid_x(tiled);
  sy.poll_budget = g_poll_budget;
  sy.flags = (normalize & ~DIB_COMPACT_LARGE_WINDOW) ? COMPACT_NORMALIZE : 0;
  { static const int skip = getenv("DIB_STEP_DEBUG_SKIP") ? 1 : 0; if (skip) sy.flags |= COMPACT_DEBUG_SKIP; }   // diagnostics: the hand-off alone
  if (g_step_nosignal) sy.flags |= COMPACT_DEBUG_NOSIGNAL;     // tests: a launch whose tables never arrive
  const dim3 grid(sy.row, tiled.n);
  if (acc_mode == DIB_ACC_FMA16) hipLaunchKernelGGL((blur_step_f16_kernel<DIB_ACC_FMA16>), grid, dim3(256), QLDS_BYTES + STEP_LDS_EXTRA, s, tiled, sy, pp);
  else hipLaunchKernelGGL((blur_step_f16_kernel<DIB_ACC_BITEXACT>), grid, dim3(256), QLDS_BYTES + STEP_LDS_EXTRA, s, tiled, sy, pp);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" int dib_sparse_blur(const void *const *in_dev, void *const *out_dev, const int *C, const int *H,
                               const int *W, const int *table_index, int B, int dtype, void *tables_dev,
                               int num_tables, int K, int acc_mode, void *stream) {
  if (B < 0 || (B > 0 && (!in_dev || !out_dev || !C || !H || !W || !table_index || !tables_dev))) {
    set_error("dib_sparse_blur: null pointer or negative batch");
    return DIB_EINVAL;
  }
  if (K != 128 && K != 256) { set_error("dib_sparse_blur: K must be 128 or 256, got %d", K); return DIB_EINVAL; }
  if (dtype != DIB_F16 && dtype != DIB_F32) { set_error("dib_sparse_blur: unknown dtype %d", dtype); return DIB_EINVAL; }
  const bool large = (acc_mode & DIB_WINDOW_LARGE) != 0;      // the tables were compacted with DIB_COMPACT_LARGE_WINDOW
  acc_mode &= ~DIB_WINDOW_LARGE;
  if (large && (dtype != DIB_F16 || acc_mode == DIB_ACC_FP32 || g_shape != 0)) {
    set_error("dib_sparse_blur: DIB_WINDOW_LARGE serves fp16 images in DIB_ACC_BITEXACT / DIB_ACC_FMA16 on the default tile shape only");
    return DIB_EINVAL;
  }
  if (acc_mode != DIB_ACC_BITEXACT && acc_mode != DIB_ACC_FP32 && acc_mode != DIB_ACC_FMA16 && acc_mode != DIB_ACC_FAST16) { set_error("dib_sparse_blur: unknown accumulation mode %d", acc_mode); return DIB_EINVAL; }
  if (acc_mode != DIB_ACC_BITEXACT && dtype != DIB_F16) { set_error("dib_sparse_blur: DIB_ACC_FP32 / DIB_ACC_FMA16 / DIB_ACC_FAST16 apply to fp16 images only (fp32 images already accumulate in fp32)"); return DIB_EINVAL; }
  if (acc_mode == DIB_ACC_FAST16 && (K != 128 || large || g_shape != 0)) {
    set_error("dib_sparse_blur: DIB_ACC_FAST16 serves K = 128 on the default tiles and the standard window (tables compacted with DIB_COMPACT_VRUNS)");
    return DIB_EINVAL;
  }
  if (int rc = check_images(in_dev, out_dev, C, H, W, table_index, B, K, num_tables)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = prepare_device()) return rc;
  if (int rc = consume_device_status("dib_sparse_blur")) return rc;
  int i = 0;
  while (i < B) {
    BlurBatch tiled, generic;
    const bool quad = g_shape == 0 && dtype == DIB_F16;
    tiled.n = generic.n = 0;
    int tiles = 0, gblocks = 0;
    for (; i < B && tiled.n < MAX_BATCH; ++i) {
      if (table_index[i] < 0) continue;
      ImageDesc d;
      d.in = in_dev[i]; d.out = out_dev[i]; d.C = C[i]; d.H = H[i]; d.W = W[i]; d.table = table_index[i];
      d.tiles_x = quad ? (W[i] + QTILE_W - 1) / QTILE_W : (W[i] + TILE_W - 1) / TILE_W;
      d.tiles_y = (H[i] + TH - 1) / TH;
      d.inv_per_ch = magic_inverse((unsigned)(d.tiles_x * d.tiles_y));
      d.inv_tiles_x = magic_inverse((unsigned)d.tiles_x);
      d.tab = (const int *)tables_dev + (size_t)table_index[i] * table_words(K);
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
      // x extent: the longest image of the launch -- 8 x its longest band, or (flat order) its tile count
      int gx = 0;
      for (int k = 0; k < tiled.n; ++k) {
        const int T = tiled.tile_begin[k + 1] - tiled.tile_begin[k];
        int ext = T;
        if (g_xcd_bands || quad) {
          int longest = 0;
          for (int x = 0; x < 8; ++x) {
            const int len = (((x + 1) * T) >> 3) - ((x * T) >> 3);
            longest = len > longest ? len : longest;
          }
          ext = 8 * longest;
        }
        gx = ext > gx ? ext : gx;
      }
      dim3 grid(gx, tiled.n);
      // ragged batch on the default tiles: a 1-D grid of exactly the working workgroups (FlatBands, dib_common.h)
      FlatBands fb = {};
      bool flat = false;
      if (quad && !large && K == 128 && g_flat_grid && tiled.n > 1 && tiled.n <= FLAT_MAX && acc_mode != DIB_ACC_FP32) {    // (FAST16 included)
        for (int k = 1; k < tiled.n && !flat; ++k)
          flat = tiled.tile_begin[k + 1] - tiled.tile_begin[k] != tiled.tile_begin[1] - tiled.tile_begin[0];
        if (flat) {
          int longest = 0;
          for (int x = 0; x < 8; ++x) {
            int at = 0;
            for (int k = 0; k < 16; ++k) {
              fb.begin[x][k] = k < tiled.n ? at : 0x7fffffff;
              if (k < tiled.n) {
                const int T = tiled.tile_begin[k + 1] - tiled.tile_begin[k];
                at += (((x + 1) * T) >> 3) - ((x * T) >> 3);
              }
            }
            fb.begin[x][15] = at;
            longest = at > longest ? at : longest;
          }
          grid = dim3(8 * longest, 1);
          // strides walked backwards: all of them in the bit-exact mode (the kernel's comment; only FULL strides are looked at), none in
          // the tolerance modes, whose launch is not bound by a CU's vector-ALU work (profiles/r6_native_order.txt: 12.5 us straight,
          // 12.8 reversed); A/B runs set the mask
          fb.rev_mask = g_flat_mask >= 0 ? (unsigned)g_flat_mask : (g_flat_snake && acc_mode == DIB_ACC_BITEXACT ? 0xffffffffu : 0u);
        }
      }
#define DIB_LAUNCH_QUAD(ACCM)                                                                                             \
  do {                                                                                                                   \
    if (large && K == 128) hipLaunchKernelGGL((blur_quad_large_f16_kernel<ACCM, 128>), grid, dim3(256), QGeom<true>::BYTES, s, tiled); \
    else if (large) hipLaunchKernelGGL((blur_quad_large_f16_kernel<ACCM, 256>), grid, dim3(256), QGeom<true>::BYTES, s, tiled);       \
    else if (flat) hipLaunchKernelGGL((blur_quad_f16_kernel<ACCM, 128, true>), grid, dim3(256), QLDS_BYTES + TL_EXTRA, s, tiled, fb); \
    else if (K == 128) hipLaunchKernelGGL((blur_quad_f16_kernel<ACCM, 128>), grid, dim3(256), QLDS_BYTES + TL_EXTRA, s, tiled, fb); \
    else hipLaunchKernelGGL((blur_quad_f16_kernel<ACCM, 256>), grid, dim3(256), QLDS_BYTES + TL_EXTRA, s, tiled, fb);          \
  } while (0)
      if (quad && acc_mode == DIB_ACC_FAST16) {
        if (flat) hipLaunchKernelGGL((blur_quad_f16_kernel<DIB_ACC_FAST16, 128, true>), grid, dim3(256), QLDS_BYTES + TL_EXTRA, s, tiled, fb);
        else hipLaunchKernelGGL((blur_quad_f16_kernel<DIB_ACC_FAST16, 128>), grid, dim3(256), QLDS_BYTES + TL_EXTRA, s, tiled, fb);
      } else if (quad && acc_mode == DIB_ACC_FMA16) DIB_LAUNCH_QUAD(DIB_ACC_FMA16);
      else if (quad && acc_mode == DIB_ACC_FP32 && K == 128) hipLaunchKernelGGL((blur_quad_f32acc_kernel<128>), grid, dim3(256), QLDS_BYTES, s, tiled);
      else if (quad && acc_mode == DIB_ACC_FP32) hipLaunchKernelGGL((blur_quad_f32acc_kernel<256>), grid, dim3(256), QLDS_BYTES, s, tiled);
      else if (quad) DIB_LAUNCH_QUAD(DIB_ACC_BITEXACT);
#undef DIB_LAUNCH_QUAD
      else if (acc_mode == DIB_ACC_FP32) hipLaunchKernelGGL((blur_tiled_f16_kernel<DIB_ACC_FP32>), grid, dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
      else if (acc_mode == DIB_ACC_FMA16) hipLaunchKernelGGL((blur_tiled_f16_kernel<DIB_ACC_FMA16>), grid, dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
      else hipLaunchKernelGGL((blur_tiled_f16_kernel<DIB_ACC_BITEXACT>), grid, dim3(256), LDS_BYTES, s, tiled, (const int *)tables_dev, K, g_stamp_buffer);
    } else {
      hipLaunchKernelGGL((blur_generic_kernel<float, DIB_ACC_BITEXACT>), dim3(gblocks), dim3(256), 0, s, generic, (const int *)tables_dev, K);
    }
    DIB_HIP_CHECK(hipGetLastError());
  }
  return DIB_OK;
}

extern "C" int dib_sparse_blur_normalized(const void *const *in_dev, const int *H, const int *W, const int *table_index, const int *slot, int B,
                                          void *tables_dev, int num_tables, int K, int acc_mode, const float *mean, const float *std,
                                          float *out_dev, int Hp, int Wp, int channels_last, void *stream) {
  if (B <= 0 || !in_dev || !H || !W || !table_index || !tables_dev || !mean || !std || !out_dev) {
    set_error("dib_sparse_blur_normalized: null pointer or empty batch");
    return DIB_EINVAL;
  }
  if (K != 128 && K != 256) { set_error("dib_sparse_blur_normalized: K must be 128 or 256, got %d", K); return DIB_EINVAL; }
  if (acc_mode != DIB_ACC_BITEXACT && acc_mode != DIB_ACC_FMA16) { set_error("dib_sparse_blur_normalized: DIB_ACC_BITEXACT or DIB_ACC_FMA16"); return DIB_EINVAL; }
  if (Hp <= 0 || Wp <= 0 || (long long)Hp * Wp * 12 >= 0x7ffffff0ll) { set_error("dib_sparse_blur_normalized: bad padded size %d x %d", Hp, Wp); return DIB_EINVAL; }
  if (B > MAX_BATCH || g_shape != 0) return 1;
  std::vector<int> C((size_t)B, 3);
  std::vector<void *> outs((size_t)B);
  for (int i = 0; i < B; ++i) {
    const int sl = slot ? slot[i] : i;
    if (sl < 0 || sl >= B) { set_error("dib_sparse_blur_normalized: slot[%d] = %d out of range", i, sl); return DIB_EINVAL; }
    outs[i] = out_dev + (size_t)sl * 3 * Hp * Wp;
    if (table_index[i] < 0) return 1;                                         // an image that is not blurred: the unfused path
    if (H[i] > Hp || W[i] > Wp) { set_error("dib_sparse_blur_normalized: image %d (%d x %d) exceeds the batch (%d x %d)", i, H[i], W[i], Hp, Wp); return DIB_EINVAL; }
    // the padding is written by the tiles of the image: they have to cover the batch's extent
    if ((H[i] + TH - 1) / TH * TH < Hp || (W[i] + QTILE_W - 1) / QTILE_W * QTILE_W < Wp) return 1;
  }
  if (int rc = check_images(in_dev, outs.data(), C.data(), H, W, table_index, B, K, num_tables)) return rc;
  if (int rc = prepare_device()) return rc;
  if (int rc = consume_device_status("dib_sparse_blur_normalized")) return rc;
  BlurBatch tiled;
  NormArgs na;
  na.Hp = Hp; na.Wp = Wp; na.nhwc = channels_last ? 1 : 0;
  tiled.n = 0;
  int tiles = 0;
  for (int i = 0; i < B; ++i) {
    const ImageDesc d = quad_desc(in_dev[i], outs[i], 3, H[i], W[i], table_index[i], (const int *)tables_dev, K, tiles);
    tiled.tile_begin[tiled.n] = tiles;
    tiles += d.C * d.tiles_x * d.tiles_y;
    for (int c = 0; c < 3; ++c) { na.mean[tiled.n][c] = mean[i * 3 + c]; na.std[tiled.n][c] = std[i * 3 + c]; }
    na.mean[tiled.n][3] = 0.f; na.std[tiled.n][3] = 1.f;
    tiled.img[tiled.n++] = d;
  }
  for (int k = tiled.n; k <= MAX_BATCH; ++k) tiled.tile_begin[k] = tiles;
  tiled.total_tiles = tiles;
  tiled.xcd_bands = 1;
  const dim3 grid(quad_grid_x(tiled), tiled.n);
  hipStream_t s = (hipStream_t)stream;
  if (acc_mode == DIB_ACC_FMA16 && K == 128) hipLaunchKernelGGL((blur_quad_f16_norm_kernel<DIB_ACC_FMA16, 128>), grid, dim3(256), QLDS_BYTES, s, tiled, na);
  else if (acc_mode == DIB_ACC_FMA16) hipLaunchKernelGGL((blur_quad_f16_norm_kernel<DIB_ACC_FMA16, 256>), grid, dim3(256), QLDS_BYTES, s, tiled, na);
  else if (K == 128) hipLaunchKernelGGL((blur_quad_f16_norm_kernel<DIB_ACC_BITEXACT, 128>), grid, dim3(256), QLDS_BYTES, s, tiled, na);
  else hipLaunchKernelGGL((blur_quad_f16_norm_kernel<DIB_ACC_BITEXACT, 256>), grid, dim3(256), QLDS_BYTES, s, tiled, na);
  DIB_HIP_CHECK(hipGetLastError());
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
  d.in = in_dev; d.out = out_dev; d.C = C; d.H = H; d.W = W; d.table = 0; d.tile_begin = 0; d.tiles_x = d.tiles_y = 0; d.inv_per_ch = d.inv_tiles_x = 0; d.tab = (const int *)table_dev;
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
