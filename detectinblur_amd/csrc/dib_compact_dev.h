// Device code of the tap compaction (include/dib.h, `Tap tables`), shared by the stand-alone kernel (dib_compact.hip: one
// 1024-thread workgroup per PSF) and by the blur step's single launch (dib_blur.hip: blur_step_f16_kernel, whose first
// workgroups -- 256 threads each -- compact the batch's PSFs while the blur workgroups behind them wait for a counter).
#pragma once
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

// ---- stores of the table words ------------------------------------------------------------------------------------------
// WT = false: plain stores (the stand-alone kernel: the kernel boundary publishes them).  WT = true: agent-scope relaxed atomic
// stores = `global_store ... sc1`, written through the XCD's L2 to memory: what the in-launch hand-off of the blur step needs
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility": sc1 payload -> every storing wave's
// s_waitcnt vmcnt(0) -> workgroup barrier -> counter add), instead of a release fence that writes the whole L2 back (~1.7 us).
// The pointer is cast to the GLOBAL address space: on a flat pointer whose origin hipcc cannot see it expands the atomic into a
// run-time "is it LDS?" test, and hipcc 7.2 then dies on that test inside the step kernel ("Illegal instruction detected:
// V_CMP_NE_U32_e32 0, $src_shared_base").
template <bool WT> __device__ __forceinline__ void st32(void *p, unsigned v) {
  if constexpr (WT) __hip_atomic_store((__attribute__((address_space(1))) unsigned *)(size_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *(unsigned *)p = v;
}
template <bool WT> __device__ __forceinline__ void st64(void *p, unsigned lo, unsigned hi) {
  if constexpr (WT) __hip_atomic_store((__attribute__((address_space(1))) unsigned long long *)(size_t)p, ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *(uint2 *)p = make_uint2(lo, hi);
}
template <bool WT> __device__ __forceinline__ void st128(void *p, unsigned a, unsigned b, unsigned c, unsigned d) {
  if constexpr (WT) { st64<true>(p, a, b); st64<true>((char *)p + 8, c, d); }
  else *(uint4 *)p = make_uint4(a, b, c, d);
}

// LDS memory is named by pointers of the LDS address space throughout the compaction's device code.  (Generic pointers into the
// step kernel's dynamic LDS made hipcc 7.2 die now and then, depending on unrelated code: "Illegal instruction detected:
// V_CMP_NE_U32_e32 0, $src_shared_base" -- the null test of an address-space cast it failed to fold.)
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) unsigned short lds_u16;
typedef __attribute__((address_space(3))) long long lds_i64;
typedef unsigned lds_vec4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) lds_vec4 lds_u128;

// ---- segmentation (ONE wave): greedy runs with row span <= SEG_ROWS and column span <= SEG_COLS -----------------------------
// Cuts the row-major tap list into the segments the tiled blur stages in LDS and writes, per tap, its source offset inside
// that window for both window layouts (ltaps / ltaps_q), the 8 zero words behind them and HDR_NSEGS.  The first STAGE taps are
// read from LDS (s_rc = row << 8 | col, s_wb = weight bits), later ones from the table's own tap list.
// VG (the stand-alone kernel with COMPACT_VRUNS): every closed segment's vertical-run groups go to the table's `vgroups` section
// (dib_common.h) -- s_vg: 32 words of LDS scratch (per row of the segment: a column bit mask and the index of the row's first tap).
template <bool WT, int STAGE, bool FIRST = false, bool VG = false>   // FIRST: s_first (LDS, 5 words) receives the first entry and the count, as segment_positions leaves them
__device__ __forceinline__ void segment_taps(int *tab, int K, int flags, int ntaps, const lds_u16 *s_rc, const lds_u32 *s_wb, int lane,
                                              lds_int *s_first = nullptr, lds_u32 *s_vg = nullptr, bool half_weights = true) {
  const uint2 *taps = reinterpret_cast<const uint2 *>(tab + table_taps_off(K));
  char *segs = reinterpret_cast<char *>(tab + table_segs_off(K));
  unsigned *ltaps = reinterpret_cast<unsigned *>(tab + table_ltaps_off(K));
  unsigned *ltaps_q = reinterpret_cast<unsigned *>(tab + table_ltaps_q_off(K));
  // segment limits and the quad window's row pitch of the geometry this table is compacted for (dib_common.h)
  const bool large = (flags & COMPACT_LARGE_WINDOW) != 0;
  const int seg_rows = large ? SEG_ROWS_L : SEG_ROWS, seg_cols = large ? SEG_COLS_L : SEG_COLS, qpitch = large ? QUAD_PITCH_L : QUAD_PITCH;
  // per-tap LDS offsets of a closed segment [s0, s1) with last row rl and last column cmx
  auto emit_ltaps = [&](int s0, int s1, int rl, int cmx) {
    for (int j = s0 + lane; j < s1; j += 64) {
      unsigned rcj, wj;
      if (j < STAGE) { rcj = s_rc[j]; wj = s_wb[j] & 0xffffu; }
      else { const uint2 tp = taps[j]; rcj = tp.x & 0xffffu; wj = tp.y & 0xffffu; }
      const int rj = rcj >> 8, cj = rcj & 255;
      st32<WT>(ltaps + j, (unsigned)(((rl - rj) * WIN_PITCH + (cmx - cj)) * 8) | (wj << 16));
      st32<WT>(ltaps_q + j, (unsigned)(((rl - rj) * qpitch + (cmx - cj)) * 8) | (wj << 16));
    }
  };
  // ---- vertical-run groups of a closed segment [s0, s1): rows rf .. rl, columns cmn .. cmx (one wave, all 64 lanes here) ----
  bool vg_ok = VG && (flags & COMPACT_VRUNS) && !large && half_weights && K == 128;
  unsigned *vgroups = reinterpret_cast<unsigned *>(tab + table_vgroups_off(K));
  auto emit_vgroups = [&](int s0, int s1, int rf, int rl, int cmn, int cmx) {
    if constexpr (VG) {
      if (!vg_ok) return;
      if (s1 > STAGE) { vg_ok = false; return; }       // taps beyond the LDS stage: no groups for this table (the blur then refuses FAST16)
      lds_u32 *s_mask = s_vg;                           // [16] bit c - cmn of word r - rf: the segment has a tap at (r, c)
      lds_int *s_start = (lds_int *)(s_vg + 16);        // [16] index of the first tap of row r inside the staged list
      if (lane < 16) { s_mask[lane] = 0u; s_start[lane] = 0x7fffffff; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int b = s0; b < s1; b += 64) {
        const int j = b + lane;
        if (j < s1) {
          const unsigned rc = s_rc[j];
          __hip_atomic_fetch_or(&s_mask[(rc >> 8) - rf], 1u << ((rc & 255) - cmn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_min(&s_start[(rc >> 8) - rf], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      auto weight_at = [&](int rr, int cc) -> unsigned {       // fp16 weight bits of the segment's tap at (rf + rr, cmn + cc)
        return s_wb[s_start[rr] + __popc(s_mask[rr] & ((1u << cc) - 1u))] & 0xffffu;
      };
      const int nrows = rl - rf + 1;
      // A run of L taps is cut into L / 4 groups of four and one group of L % 4.  The segment's groups are stored SORTED BY SIZE,
      // fours first (the tap loop then runs one straight-line body per size: dib_blur.hip), inside a size in the row-major order of
      // the runs' first taps, a run's fours from its lowest rows up.  Two passes over the taps: count per size, then place.
      int cnt4 = 0, cnt3 = 0, cnt2 = 0, cnt1 = 0;
      auto run_of = [&](int j, int &rr, int &cc) -> int {     // length of the run that STARTS at staged tap j (0: j is not a run's first tap)
        const unsigned rc = s_rc[j];
        rr = (int)(rc >> 8) - rf; cc = (int)(rc & 255) - cmn;
        if (rr > 0 && ((s_mask[rr - 1] >> cc) & 1u)) return 0;
        int L = 1;
        while (rr + L < nrows && ((s_mask[rr + L] >> cc) & 1u)) ++L;
        return L;
      };
      for (int b = s0; b < s1; b += 64) {
        const int j = b + lane;
        int rr = 0, cc = 0;
        const int L = j < s1 ? run_of(j, rr, cc) : 0;
        cnt4 += wave_sum_i32(L >> 2);
        cnt3 += wave_sum_i32((L & 3) == 3); cnt2 += wave_sum_i32((L & 3) == 2); cnt1 += wave_sum_i32((L & 3) == 1);
      }
      const int total = cnt4 + cnt3 + cnt2 + cnt1;
      int base4 = s0, base3 = s0 + cnt4, base2 = base3 + cnt3, base1 = base2 + cnt2;     // next free slot per size
      // one group of n taps whose highest PSF row is segment row rb, column cc, into slot: its weights (tap j of a group reads window
      // rows j .. j + 3 from the group's offset: w[0] belongs to the group's HIGHEST PSF row), and its own offset | size into the
      // PREVIOUS slot's x (the tap loop fetches one group ahead) or, for the segment's first slot, into that slot's own w
      auto place = [&](int slot, int n, int rb, int cc) {
        unsigned w[VRUN_MAX] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < VRUN_MAX; ++t)
          if (t < n) w[t] = weight_at(rb - t, cc);
        const unsigned own = (unsigned)(((rl - rf - rb) * qpitch + (cmx - cmn - cc)) * 8) | (unsigned)(n - 1) << 16;
        if (slot > s0) st32<WT>(vgroups + 4 * (size_t)(slot - 1), own);
        else st32<WT>(vgroups + 4 * (size_t)slot + 3, own);
        st32<WT>(vgroups + 4 * (size_t)slot + 1, w[0] | w[1] << 16);
        st32<WT>(vgroups + 4 * (size_t)slot + 2, w[2] | w[3] << 16);
      };
      for (int b = s0; b < s1; b += 64) {
        const int j = b + lane;
        int rr = 0, cc = 0;
        const int L = j < s1 ? run_of(j, rr, cc) : 0;
        const int n4 = L >> 2, rem = L & 3;
        const int i4 = wave_scan_incl_i32(n4, lane), i3 = wave_scan_incl_i32(rem == 3, lane), i2 = wave_scan_incl_i32(rem == 2, lane),
                  i1 = wave_scan_incl_i32(rem == 1, lane);
        for (int g = 0; g < n4; ++g) place(base4 + i4 - n4 + g, 4, rr + 4 * g + 3, cc);
        if (rem == 3) place(base3 + i3 - 1, 3, rr + L - 1, cc);
        if (rem == 2) place(base2 + i2 - 1, 2, rr + L - 1, cc);
        if (rem == 1) place(base1 + i1 - 1, 1, rr + L - 1, cc);
        base4 += __builtin_amdgcn_readlane(i4, 63); base3 += __builtin_amdgcn_readlane(i3, 63);
        base2 += __builtin_amdgcn_readlane(i2, 63); base1 += __builtin_amdgcn_readlane(i1, 63);
      }
      // behind the last group: the end code (4) in place of a size
      if (lane == 0) st32<WT>(vgroups + 4 * (size_t)(s0 + total - 1), 4u << 16);
    }
  };
  int nseg = 0, seg_start = 0, seg_r0 = 0, seg_rlast = 0, car_cmin = 1 << 20, car_cmax = -1;
  for (int base = 0; base < ntaps; base += 64) {
    const int i = base + lane;
    const bool valid = i < ntaps;
    unsigned rc = 0;
    if (valid) rc = (i < STAGE) ? s_rc[i] : (taps[i].x & 0xffffu);
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
      if (lane == 0) {
        st128<WT>(segs + 16 * (size_t)nseg, seg_start, base + f, (seg_r0 << 8) | rl, (cmn << 8) | cmx);
        if (FIRST && nseg == 0) { s_first[0] = seg_start; s_first[1] = base + f; s_first[2] = (seg_r0 << 8) | rl; s_first[3] = (cmn << 8) | cmx; }
      }
      emit_ltaps(seg_start, base + f, rl, cmx);
      emit_vgroups(seg_start, base + f, seg_r0, rl, cmn, cmx);
      ++nseg;
      seg_start = base + f;
      seg_r0 = __builtin_amdgcn_readlane(r, f);
      seg_rlast = seg_r0;
      car_cmin = 1 << 20; car_cmax = -1;
      lo = f;
    }
  }
  if (ntaps > 0) {
    if (lane == 0) {
      st128<WT>(segs + 16 * (size_t)nseg, seg_start, ntaps, (seg_r0 << 8) | seg_rlast, (car_cmin << 8) | car_cmax);
      if (FIRST && nseg == 0) { s_first[0] = seg_start; s_first[1] = ntaps; s_first[2] = (seg_r0 << 8) | seg_rlast; s_first[3] = (car_cmin << 8) | car_cmax; }
    }
    emit_ltaps(seg_start, ntaps, seg_rlast, car_cmax);
    emit_vgroups(seg_start, ntaps, seg_r0, seg_rlast, car_cmin, car_cmax);
    ++nseg;
  } else if (FIRST && lane == 0) {
    s_first[0] = s_first[1] = s_first[2] = s_first[3] = 0;
  }
  if (FIRST && lane == 0) s_first[4] = nseg;
  if (lane < 8) { st32<WT>(ltaps + ntaps + lane, 0u); st32<WT>(ltaps_q + ntaps + lane, 0u); }  // the blur's scalar prefetch runs up to two taps past the end
  if (lane == 0) st32<WT>(tab + HDR_NSEGS, (unsigned)nseg);
  if constexpr (VG) {      // the header's K word again (same lane that wrote it before), now with the groups' validity
    if (lane == 0) st32<WT>(tab + HDR_K, (unsigned)K | (large ? 1u << 16 : 0u) | ((vg_ok && ntaps > 0) ? HDR_K_VRUNS : 0u));
  }
}

// The same cut on the tap POSITIONS alone (one wave; the taps' weights are still being summed and divided by other waves):
// writes the segment entries and HDR_NSEGS and leaves, in LDS, every tap's segment (s_seg) and every segment's
// r_last << 8 | cmax (s_sinfo), from which all threads then form the per-tap offsets at once.
template <bool WT>
__device__ __forceinline__ void segment_positions(int *tab, int K, int flags, int ntaps, const lds_u16 *s_rc, lds_u16 *s_seg,
                                                   lds_u32 *s_sinfo, lds_int *s_first, int lane) {
  char *segs = reinterpret_cast<char *>(tab + table_segs_off(K));
  const bool large = (flags & COMPACT_LARGE_WINDOW) != 0;
  const int seg_rows = large ? SEG_ROWS_L : SEG_ROWS, seg_cols = large ? SEG_COLS_L : SEG_COLS;
  int nseg = 0, seg_start = 0, seg_r0 = 0, seg_rlast = 0, car_cmin = 1 << 20, car_cmax = -1;
  for (int base = 0; base < ntaps; base += 64) {
    const int i = base + lane;
    const bool valid = i < ntaps;
    const unsigned rc = valid ? s_rc[i] : 0u;
    const int r = rc >> 8, c = rc & 255;
    if (base == 0) seg_r0 = __builtin_amdgcn_readlane(r, 0);
    int lo = 0;
    while (true) {
      int pm = wave_scan_min_i32((valid && lane >= lo) ? c : (1 << 20), lane);
      int px = wave_scan_max_i32((valid && lane >= lo) ? c : -1, lane);
      pm = min(pm, car_cmin); px = max(px, car_cmax);
      const bool bad = valid && lane >= lo && ((r - seg_r0 > seg_rows) || (px - pm > seg_cols));
      const unsigned long long fail = __ballot(bad);
      const unsigned long long vmask = __ballot(valid);
      const int last_valid = 63 - __clzll((long long)vmask);
      if (fail == 0) {
        if (valid && lane >= lo) s_seg[i] = (unsigned short)nseg;
        car_cmin = __builtin_amdgcn_readlane(pm, last_valid); car_cmax = __builtin_amdgcn_readlane(px, last_valid);
        seg_rlast = __builtin_amdgcn_readlane(r, last_valid);
        break;
      }
      const int f = __ffsll((long long)fail) - 1;  // tap base+f opens a new segment
      if (lane >= lo && lane < f) s_seg[i] = (unsigned short)nseg;
      int cmn = car_cmin, cmx = car_cmax, rl = seg_rlast;
      if (f > lo) { cmn = __builtin_amdgcn_readlane(pm, f - 1); cmx = __builtin_amdgcn_readlane(px, f - 1); rl = __builtin_amdgcn_readlane(r, f - 1); }
      if (lane == 0) {
        st128<WT>(segs + 16 * (size_t)nseg, seg_start, base + f, (seg_r0 << 8) | rl, (cmn << 8) | cmx);
        s_sinfo[nseg] = (unsigned)((rl << 8) | cmx);
        if (nseg == 0) { s_first[0] = seg_start; s_first[1] = base + f; s_first[2] = (seg_r0 << 8) | rl; s_first[3] = (cmn << 8) | cmx; }
      }
      ++nseg;
      seg_start = base + f;
      seg_r0 = __builtin_amdgcn_readlane(r, f);
      seg_rlast = seg_r0;
      car_cmin = 1 << 20; car_cmax = -1;
      lo = f;
    }
  }
  if (ntaps > 0) {
    if (lane == 0) {
      st128<WT>(segs + 16 * (size_t)nseg, seg_start, ntaps, (seg_r0 << 8) | seg_rlast, (car_cmin << 8) | car_cmax);
      s_sinfo[nseg] = (unsigned)((seg_rlast << 8) | car_cmax);
      if (nseg == 0) { s_first[0] = seg_start; s_first[1] = ntaps; s_first[2] = (seg_r0 << 8) | seg_rlast; s_first[3] = (car_cmin << 8) | car_cmax; }
    }
    ++nseg;
  } else if (lane == 0) {
    s_first[0] = s_first[1] = s_first[2] = s_first[3] = 0;
  }
  if (lane == 0) { st32<WT>(tab + HDR_NSEGS, (unsigned)nseg); s_first[4] = nseg; }
}

// ---- one fp16 128 x 128 PSF compacted by ONE 256-thread workgroup (the blur step's in-launch compaction) -------------------
// Same table as psf_compact_kernel<__half, 128> writes (tests/test_blur_step_gpu.py compares every word the blur and the box
// growth read).  Thread t reads the 16-byte pieces u = 256 i + t (i = 0..7) of the PSF: coalesced, all eight in flight at
// once; element order is (i, t, k).  A PSF is a thin curve, so nearly every (wave, i) pair holds nothing but zeros and skips
// its prefix scan.  Register diet: this code shares a kernel with the blur's tile function and must fit its 64 registers
// without scratch memory (a kernel with scratch pays for it in every workgroup's dispatch), so a piece that holds a non-zero
// goes to a small LDS list at once, with the position it will get inside its (piece, wave) pair, instead of staying in
// registers until the pairs' totals are known.
//   fast path (the only one real PSFs take): pieces with non-zeros listed in LDS -> raw non-zeros staged in row-major order ->
//     exact sum -> divide -> drop the taps whose weight underflowed to zero (rare: one more in-order pass of wave 0) -> tap
//     list, row pointers by binary search, header, segments;
//   general path (more than CSTAGE non-zeros or CHITS such pieces, or a sum that is 0 / NaN: then psf / psf.sum() makes EVERY
//     element a tap): two passes over the eight pieces per thread, re-read from memory, divide first, taps straight to the table.
// pool: the workgroup's dynamic LDS (16.7 KB used; the blur's window, 19.7 KB, is what the launch provides).
constexpr int CSTAGE = 1024, CHITS = 512;
// `wave`: the calling wave's number (0..3), from the caller -- in the blur step's kernel the thread-index register must have
// ONE consumer (a second one, however far from the blur's tile function, made hipcc keep that register alive across the tile
// function's window fill: one register too many there, a spill to scratch memory); the lane comes from v_mbcnt.
// EARLY / `early`: where to publish the first segment once it is final (see StepSync::rec); `tag`: this launch's stamp.
// (A template flag, not a null test of the pointer: hipcc 7.2 miscompiled that test inside the step kernel -- "Illegal instruction
// detected: V_CMP_NE_U32_e32 0, $src_shared_base".)
template <bool WT, bool EARLY = false>
__device__ __forceinline__ void compact_psf_f16_wg256(const void *psf, int flags, int *tab, lds_u32 *pool, const int wave, unsigned *early = nullptr,
                                                      unsigned tag = 0, unsigned long long *dbg = nullptr) {
#define DIB_CSTAMP(n) do { if (dbg && wave == 0 && lane == 0) dbg[n] = __builtin_amdgcn_s_memrealtime(); } while (0)
  using E = Elem<__half>;
  constexpr int K = 128, LK = 7, NCH = K * K / 8 / 256;   // 8 pieces of 16 bytes per thread
  lds_u16 *s_rc = (lds_u16 *)pool;                                             // [CSTAGE]   row << 8 | col
  lds_u32 *s_wb = pool + CSTAGE / 2;                                           // [CSTAGE]   weight bits
  lds_u128 *s_hit = (lds_u128 *)(pool + CSTAGE / 2 + CSTAGE);                  // [CHITS]    pieces that hold a non-zero
  lds_u32 *s_hmeta = pool + CSTAGE / 2 + CSTAGE + 4 * CHITS;                   // [CHITS]    mask | offset inside the pair << 8 | tid << 18 | i << 26
  lds_int *s_tot = (lds_int *)(s_hmeta + CHITS);                               // [NCH * 4]  non-zeros of (piece i, wave w), at i * 4 + w
  lds_int *s_base = s_tot + NCH * 4;                                             // [NCH * 4]  exclusive prefix of s_tot
  lds_i64 *s_part = (lds_i64 *)(s_base + NCH * 4);                         // [4]
  lds_int *s_misc = (lds_int *)(s_part + 4);                                   // [16] 0: a tap vanished, 1: taps kept, 2..5: extents, 6: pieces listed, 7: the sum's bits, 8..11: first segment, 12: segments
  static_assert(((CSTAGE / 2 + CSTAGE) * 4) % 16 == 0 && ((CSTAGE / 2 + CSTAGE + 5 * CHITS + 2 * NCH * 4) * 4) % 8 == 0, "alignment of s_hit / s_part");
  static_assert((CSTAGE / 2 + CSTAGE + 5 * CHITS + 2 * NCH * 4 + 8 + 16) * 4 <= 19712, "fits the blur's window");
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int tid = wave * 64 + lane;
  DIB_CSTAMP(0);
  if (flags & COMPACT_DEBUG_SKIP) {                   // diagnostics: the tables of an earlier launch stay as they are
    if (EARLY && wave == 0 && lane < STEP_REPLICAS) {  // (their first segment goes out with this launch's tag)
      const uint4 s0 = *reinterpret_cast<const uint4 *>(tab + table_segs_off(K));
      st64<true>(early + lane * STEP_REC_WORDS, (s0.y & 0xffffu) | (unsigned)tab[HDR_NSEGS] << 16, tag);
      st64<true>(early + lane * STEP_REC_WORDS + 2, s0.z << 16 | (s0.w & 0xffffu), tag);
    }
    return;
  }
  const uint4 *p4 = reinterpret_cast<const uint4 *>(psf);
  uint4 q[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) q[i] = p4[i * 256 + tid];
  if (tid < 16) s_misc[tid] = (tid < 2 || tid > 5) ? 0 : ((tid & 1) ? -1 : K);   // 2: rmin 3: rmax 4: cmin 5: cmax
  __syncthreads();
  auto nzmask = [](const uint4 &v) -> unsigned {
    unsigned b = 0;
    b |= (v.x & 0x7fffu) ? 1u : 0u; b |= (v.x & 0x7fff0000u) ? 2u : 0u;
    b |= (v.y & 0x7fffu) ? 4u : 0u; b |= (v.y & 0x7fff0000u) ? 8u : 0u;
    b |= (v.z & 0x7fffu) ? 16u : 0u; b |= (v.z & 0x7fff0000u) ? 32u : 0u;
    b |= (v.w & 0x7fffu) ? 64u : 0u; b |= (v.w & 0x7fff0000u) ? 128u : 0u;
    return b;
  };
  auto half_of = [](const uint4 &v, int k) -> unsigned {                       // bits of element k of a piece (no array: see above)
    const unsigned d = (k < 2) ? v.x : (k < 4) ? v.y : (k < 6) ? v.z : v.w;
    return (k & 1) ? d >> 16 : d & 0xffffu;
  };
  const int normalize = flags & COMPACT_NORMALIZE;
  // ---- raw non-zeros: per-piece masks, per-(piece, wave) inclusive scans, pieces with a non-zero -> LDS list -----------------
  unsigned long long bits = 0;              // 8 mask bits per piece (the general path's skip test)
  {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const unsigned b = ((q[i].x | q[i].y | q[i].z | q[i].w) & 0x7fff7fffu) ? nzmask(q[i]) : 0u;
      bits |= (unsigned long long)b << (8 * i);
      const int c = __popc(b);
      int incl = 0;
      if (__builtin_amdgcn_ballot_w64(b != 0) != 0) {
        incl = wave_scan_incl_i32(c, lane);
        if (b) {
          const int slot = __hip_atomic_fetch_add(&s_misc[6], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (slot < CHITS) { s_hit[slot] = lds_vec4{q[i].x, q[i].y, q[i].z, q[i].w}; s_hmeta[slot] = b | (unsigned)(incl - c) << 8 | (unsigned)tid << 18 | (unsigned)i << 26; }
        }
      }
      if (lane == 63) s_tot[i * 4 + wave] = incl;
    }
  }
  DIB_CSTAMP(1);
  __syncthreads();                                                                   // A
  int tv = lane < NCH * 4 ? s_tot[lane] : 0;
  int tincl = wave_scan_incl_i32(tv, lane);
  const int n_raw = __builtin_amdgcn_readlane(tincl, NCH * 4 - 1);
  const int nhits = s_misc[6];
  bool ok = n_raw <= CSTAGE && nhits <= CHITS;
  if (ok) {
    if (lane < NCH * 4) s_base[lane] = tincl - tv;     // every wave writes the same 32 values and reads back its own
    for (int h = tid; h < nhits; h += 256) {
      const lds_vec4 v = s_hit[h];
      const unsigned m = s_hmeta[h];
      const int i = m >> 26, t = (m >> 18) & 255;
      int pos = s_base[i * 4 + (t >> 6)] + (int)((m >> 8) & 1023);
      const int e0 = (i * 256 + t) * 8;
      const unsigned hw[8] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16, v.z & 0xffffu, v.z >> 16, v.w & 0xffffu, v.w >> 16};
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (m & (1u << k)) {
          s_rc[pos] = (unsigned short)((((e0 + k) >> LK) << 8) | ((e0 + k) & (K - 1)));
          s_wb[pos] = hw[k];
          ++pos;
        }
    }
  }
  __syncthreads();                                                                   // B: raw taps staged in row-major order
  DIB_CSTAMP(2);
  // ---- C: three things that do not need each other, on three waves: wave 0 cuts the (raw) tap POSITIONS into segments,
  // wave 1 forms the exact sum, waves 2 / 3 have nothing yet.  (One after the other on the whole workgroup these phases were
  // 2.4 us of the 6.5 us the blur's workgroups waited for this one: profiles/r5_step_timeline.txt.)
  lds_u16 *s_seg = (lds_u16 *)s_hit;                                           // [CSTAGE] segment of tap j   (the piece list is dead)
  lds_u32 *s_sinfo = (lds_u32 *)(s_seg + CSTAGE);                              // [CSTAGE] r_last << 8 | cmax of segment s
  static_assert(CSTAGE * 2 + CSTAGE * 4 <= CHITS * 16, "segment ids + infos fit the piece list's space");
  __half total = __float2half_rn(1.0f);
  // the first segment's record (wave 0, which wrote s_misc[8..12] itself: no barrier in between)
  auto publish_first = [&]() {
    if (EARLY && !(flags & COMPACT_NO_SEGMENTS) && lane < STEP_REPLICAS) {
      const unsigned a = (unsigned)s_misc[9] | (unsigned)s_misc[12] << 16;          // end of the first segment | segments
      const unsigned b = (unsigned)s_misc[10] << 16 | (unsigned)s_misc[11];         // r_first << 8 | r_last, cmin << 8 | cmax
      st64<true>(early + lane * STEP_REC_WORDS, a, tag);
      st64<true>(early + lane * STEP_REC_WORDS + 2, b, tag);
    }
  };
  // CSR row pointers of rows [r0, r1): first tap of the row, by binary search (the staged list is sorted by row << 8 | col)
  auto row_pointers = [&](int r, int n) {
    const unsigned key = (unsigned)r << 8;
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_rc[mid] < key) lo = mid + 1; else hi = mid; }
    st32<WT>(tab + table_rowptr_off() + r, (unsigned)lo);
  };
  // column extents of the first n staged taps -> s_misc[4], s_misc[5] (one wave)
  auto col_extents = [&](int n) {
    int cmn = K, cmx = -1;
    for (int j = lane; j < n; j += 64) { const int c = s_rc[j] & 255; cmn = min(cmn, c); cmx = max(cmx, c); }
    cmn = wave_min_i32(cmn); cmx = wave_max_i32(cmx);
    if (lane == 0) { s_misc[4] = cmn; s_misc[5] = cmx; }
  };
  if (ok) {
    if (wave == 0 && !(flags & COMPACT_NO_SEGMENTS)) segment_positions<WT>(tab, K, flags, n_raw, s_rc, s_seg, s_sinfo, s_misc + 8, lane);
    if (wave == 1) {   // the exact sum, then every weight divided by it (in place): a tap whose weight becomes zero vanishes
      bool lost = false, sane = true;
      if (normalize) {
        long long acc = 0;
        for (int j = lane; j < n_raw; j += 64) acc += E::lift(E::from_bits(s_wb[j]));
        const __half t = E::finish(wave_sum(acc));
        if (lane == 0) s_misc[7] = (int)E::bits(t);
        sane = (float)t == (float)t && E::nonzero(t);
        if (sane)
          for (int j = lane; j < n_raw; j += 64) {
            const __half w = E::div(E::from_bits(s_wb[j]), t);
            lost = lost || !E::nonzero(w);
            s_wb[j] = E::bits(w);
          }
      }
      if (__builtin_amdgcn_ballot_w64(lost) != 0 && lane == 0) s_misc[0] = 1;
      col_extents(n_raw);
    }
    if (wave >= 2) {   // rows 0..63 on wave 2, 64..128 on wave 3
      row_pointers((wave - 2) * 64 + lane, n_raw);
      if (wave == 3 && lane == 0) row_pointers(K, n_raw);
    }
  }
  DIB_CSTAMP(3);
  __syncthreads();                                                                   // C
  if (ok && normalize) {
    total = E::from_bits((unsigned)s_misc[7]);
    ok = (float)total == (float)total && E::nonzero(total);
  }
  int ntaps = 0;
  if (ok) {
    // ================================ fast path ==============================================================================
    // The first segment is final now unless a tap vanished: published at once (wave 0, which wrote the segments; 32 copies,
    // two 8-byte words {data, tag} each), the blur's workgroups fill their first window while the offsets are still being made.
    if (wave == 0 && s_misc[0] == 0) publish_first();
    ntaps = n_raw;
    unsigned *ltaps = reinterpret_cast<unsigned *>(tab + table_ltaps_off(K));
    unsigned *ltaps_q = reinterpret_cast<unsigned *>(tab + table_ltaps_q_off(K));
    if (s_misc[0]) {   // a weight underflowed to zero in the division: that tap vanishes, later ones move up (wave 0, in order)
      if (wave == 0) {
        int out = 0;
        for (int base = 0; base < n_raw; base += 64) {
          const int j = base + lane;
          const bool valid = j < n_raw;
          const unsigned w = valid ? s_wb[j] : 0u;
          const unsigned short rc = valid ? s_rc[j] : (unsigned short)0;
          const bool keep = valid && (w & 0x7fffu) != 0;
          const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
          const int pos = out + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0));
          if (keep) { s_rc[pos] = rc; s_wb[pos] = w; }     // pos <= j, and this round's reads are all done
          out += __popcll(km);
        }
        if (lane == 0) s_misc[1] = out;
      }
      __syncthreads();
      ntaps = s_misc[1];
      // segments, row pointers and extents were made with the vanished taps in them: once more
      if (wave == 0 && !(flags & COMPACT_NO_SEGMENTS)) { segment_taps<WT, CSTAGE, true>(tab, K, flags, ntaps, s_rc, s_wb, lane, s_misc + 8); publish_first(); }
      if (wave == 1) col_extents(ntaps);
      if (wave >= 2) {
        row_pointers((wave - 2) * 64 + lane, ntaps);
        if (wave == 3 && lane == 0) row_pointers(K, ntaps);
      }
      __syncthreads();
    } else if (!(flags & COMPACT_NO_SEGMENTS)) {
      // per-tap offsets of the tap's source word inside its segment's window, both layouts (see segment_taps)
      const int qpitch = (flags & COMPACT_LARGE_WINDOW) ? QUAD_PITCH_L : QUAD_PITCH;
      for (int j = tid; j < ntaps; j += 256) {
        const unsigned rc = s_rc[j], wj = s_wb[j] & 0xffffu, info = s_sinfo[s_seg[j]];
        const int dr = (int)(info >> 8) - (int)(rc >> 8), dc = (int)(info & 255) - (int)(rc & 255);
        st32<WT>(ltaps + j, (unsigned)((dr * WIN_PITCH + dc) * 8) | (wj << 16));
        st32<WT>(ltaps_q + j, (unsigned)((dr * qpitch + dc) * 8) | (wj << 16));
      }
      if (tid < 8) { st32<WT>(ltaps + ntaps + tid, 0u); st32<WT>(ltaps_q + ntaps + tid, 0u); }
    }
    char *taps = reinterpret_cast<char *>(tab + table_taps_off(K));
    for (int j = tid; j < ntaps; j += 256) st64<WT>(taps + 8 * (size_t)j, (unsigned)s_rc[j], s_wb[j]);
    if (tid == 64) {   // header (a lane of wave 1: wave 0 may still be busy in the rare path above)
      st32<WT>(tab + HDR_NTAPS, (unsigned)ntaps);
      st32<WT>(tab + HDR_RMIN, (unsigned)(ntaps ? (s_rc[0] >> 8) : K)); st32<WT>(tab + HDR_RMAX, (unsigned)(ntaps ? (s_rc[ntaps - 1] >> 8) : -1));
      st32<WT>(tab + HDR_CMIN, (unsigned)s_misc[4]); st32<WT>(tab + HDR_CMAX, (unsigned)s_misc[5]);
      st32<WT>(tab + HDR_K, (unsigned)(K | ((flags & COMPACT_LARGE_WINDOW) ? 1 << 16 : 0))); st32<WT>(tab + HDR_SUM, E::bits(total));
    }
    DIB_CSTAMP(4);
    return;
  } else {
    // ================================ general path ===========================================================================
    // Rolled loops that re-read the pieces from memory (L2 hits): nothing here may index the register arrays of the fast
    // path dynamically -- that would put them into scratch memory, for every workgroup of the launch.  The trip counts are
    // hidden from hipcc, which otherwise unrolls and vectorises the divisions below into ~100 live registers.
    int nch = NCH, nk = 8;
    asm volatile("" : "+s"(nch), "+s"(nk));
    // Phase C may already have stored row pointers and segments (a sum that turned out 0 / NaN): those stores have to be out
    // before other threads of the workgroup store to the same words below.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (normalize) {
      long long acc = 0;
  #pragma unroll 1
    for (int i = 0; i < nch; ++i) {
        if (((unsigned)(bits >> (8 * i)) & 0xffu) == 0) continue;
        const uint4 v = p4[i * 256 + tid];
#pragma unroll 1
        for (int k = 0; k < nk; ++k) acc += E::lift(E::from_bits(half_of(v, k)));
      }
      acc = wave_sum(acc);
      __syncthreads();               // s_part of the fast path's attempt has been read by every wave
      if (lane == 0) s_part[wave] = acc;
      __syncthreads();
      const long long part = lane < 4 ? s_part[lane] : 0ll;
      total = E::finish(wave_total(part));
    }
    // 0 / total is 0 unless total is 0 or NaN: then the reference's psf / psf.sum() is NaN (or inf) everywhere and every
    // element becomes a tap
    const bool sane = (float)total == (float)total && E::nonzero(total);
    auto weight = [&](unsigned raw) -> unsigned {          // bits of the tap's weight; 0 = no tap
      __half w = E::from_bits(raw);
      if (sane && !E::nonzero(w)) return 0u;
      if (normalize) w = E::div(w, total);
      return E::nonzero(w) ? E::bits(w) : 0u;
    };
    bits = 0;
    int rmin = K, rmax = -1, cmin = K, cmax = -1;
#pragma unroll 1
    for (int i = 0; i < nch; ++i) {
      const uint4 v = p4[i * 256 + tid];
      unsigned b = 0;
      const int e0 = (i * 256 + tid) * 8;
#pragma unroll 1
      for (int k = 0; k < nk; ++k)
        if (weight(half_of(v, k))) {
          b |= 1u << k;
          const int r = (e0 + k) >> LK, c = (e0 + k) & (K - 1);
          rmin = min(rmin, r); rmax = max(rmax, r); cmin = min(cmin, c); cmax = max(cmax, c);
        }
      bits |= (unsigned long long)b << (8 * i);
      const int sc = wave_scan_incl_i32(__popc(b), lane);
      if (lane == 63) s_tot[i * 4 + wave] = sc;
    }
    rmin = wave_min_i32(rmin); rmax = wave_max_i32(rmax); cmin = wave_min_i32(cmin); cmax = wave_max_i32(cmax);
    if (lane == 0) {
      __hip_atomic_fetch_min(&s_misc[2], rmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_max(&s_misc[3], rmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_min(&s_misc[4], cmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_max(&s_misc[5], cmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    tv = lane < NCH * 4 ? s_tot[lane] : 0;
    tincl = wave_scan_incl_i32(tv, lane);
    ntaps = __builtin_amdgcn_readlane(tincl, NCH * 4 - 1);
    char *taps = reinterpret_cast<char *>(tab + table_taps_off(K));
#pragma unroll 1
    for (int i = 0; i < nch; ++i) {
      const int base = __builtin_amdgcn_readlane(tincl - tv, i * 4 + wave);
      const unsigned b = (unsigned)(bits >> (8 * i)) & 0xffu;
      const int cb = __popc(b);
      int pos = base + wave_scan_incl_i32(cb, lane) - cb;
      const int e0 = (i * 256 + tid) * 8;
      if ((tid & 15) == 0) st32<WT>(tab + table_rowptr_off() + 16 * i + (tid >> 4), (unsigned)pos);   // a row = 16 pieces
      if (b == 0) continue;
      const uint4 v = p4[i * 256 + tid];
#pragma unroll 1
      for (int k = 0; k < nk; ++k)
        if (b & (1u << k)) {
          const unsigned rc = (unsigned)((((e0 + k) >> LK) << 8) | ((e0 + k) & (K - 1))), wb = weight(half_of(v, k));
          st64<WT>(taps + 8 * (size_t)pos, rc, wb);
          if (pos < CSTAGE) { s_rc[pos] = (unsigned short)rc; s_wb[pos] = wb; }
          ++pos;
        }
    }
    if (tid == 0) {
      st32<WT>(tab + table_rowptr_off() + K, (unsigned)ntaps);
      st32<WT>(tab + HDR_NTAPS, (unsigned)ntaps);
      st32<WT>(tab + HDR_RMIN, (unsigned)s_misc[2]); st32<WT>(tab + HDR_RMAX, (unsigned)s_misc[3]);
      st32<WT>(tab + HDR_CMIN, (unsigned)s_misc[4]); st32<WT>(tab + HDR_CMAX, (unsigned)s_misc[5]);
      st32<WT>(tab + HDR_K, (unsigned)(K | ((flags & COMPACT_LARGE_WINDOW) ? 1 << 16 : 0))); st32<WT>(tab + HDR_SUM, E::bits(total));
    }
    // the segmenter reads taps beyond the LDS stage back from the table: this workgroup's own stores, drained first
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_block();
    __syncthreads();
  }
  if (wave == 0 && !(flags & COMPACT_NO_SEGMENTS)) { segment_taps<WT, CSTAGE, true>(tab, K, flags, ntaps, s_rc, s_wb, lane, s_misc + 8); publish_first(); }
#undef DIB_CSTAMP
}

}  // namespace dib
