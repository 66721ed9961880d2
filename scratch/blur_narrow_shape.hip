// Round 2's first 128-wide tile shape (4-byte LDS words {P[j], P[j+64]}, 8 x ds_read_b32 per tap), removed from the
// product library once the quad shape (csrc/dib_blur.hip) had replaced it: 44-46 us against 40 us on the BASELINE batch, its
// tap phase ran at the LDS's rate.  This is the section of csrc/dib_blur.hip as it stood (not compiled on its own).

// =============================================================================================================
// Narrow shape: 128 x 32 tiles.  Lane l owns columns x0 + l and x0 + l + 64 (ONE packed register per row), the LDS
// word is 4 bytes {P[j], P[j+64]} and the window 48 rows x 96 words x 4 B = 18 KB: seven workgroups per CU instead of
// four.  The kernel is a closed system -- a CU's slots each run dispatch -> prologue -> fill -> taps -> store in
// sequence, and its time is (tiles per slot) x (latency of one workgroup) -- so slots are what buys throughput:
// measured on the BASELINE batch 3 / 4 slots gave 61 / 51 us with the 256-wide tile.  Costs: 1.25 x instead of
// 1.125 x halo columns, and twice the workgroups (their fixed cost is ~2.4 us each).
// =============================================================================================================
constexpr int NTILE_W = 128;
constexpr int NPITCH = WIN_PITCH * 4;         // bytes per LDS row (384)
constexpr int NLDS_BYTES = LROWS * NPITCH;    // 18,432 B
static_assert(NPITCH == 384, "the asm below hard-codes the LDS row pitch");

// The r8 tap loop on 4-byte words: the ltap word carries the byte offset of the 8-byte-word layout, so it is halved
// (s_bfe_u32: bits 15..1) -- one more scalar instruction per tap.  Buffers v[32:39] / v[40:47], address v48.
// Operands: %0-%7 accumulators, %8 byte offset of the next ltap, %9 taps left, %10 A, %11 B, %12 C, %13 temp, %14 ltaps, %15 lane base.
#define DIBN_MUL(b) "v_pk_mul_f16 v" #b ", %10, v" #b " op_sel:[1,0] op_sel_hi:[1,1]\n\t"
#define DIBN_ADD(b, i) "v_pk_add_f16 %" #i ", %" #i ", v" #b "\n\t"
#define DIBN_FMA(b, i) "v_pk_fma_f16 %" #i ", %10, v" #b ", %" #i " op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
#define DIBN_MADD_A DIBN_MUL(32) DIBN_MUL(33) DIBN_MUL(34) DIBN_MUL(35) DIBN_MUL(36) DIBN_MUL(37) DIBN_MUL(38) DIBN_MUL(39) \
  DIBN_ADD(32, 0) DIBN_ADD(33, 1) DIBN_ADD(34, 2) DIBN_ADD(35, 3) DIBN_ADD(36, 4) DIBN_ADD(37, 5) DIBN_ADD(38, 6) DIBN_ADD(39, 7)
#define DIBN_MADD_B DIBN_MUL(40) DIBN_MUL(41) DIBN_MUL(42) DIBN_MUL(43) DIBN_MUL(44) DIBN_MUL(45) DIBN_MUL(46) DIBN_MUL(47) \
  DIBN_ADD(40, 0) DIBN_ADD(41, 1) DIBN_ADD(42, 2) DIBN_ADD(43, 3) DIBN_ADD(44, 4) DIBN_ADD(45, 5) DIBN_ADD(46, 6) DIBN_ADD(47, 7)
#define DIBN_FMADD_A DIBN_FMA(32, 0) DIBN_FMA(33, 1) DIBN_FMA(34, 2) DIBN_FMA(35, 3) DIBN_FMA(36, 4) DIBN_FMA(37, 5) DIBN_FMA(38, 6) DIBN_FMA(39, 7)
#define DIBN_FMADD_B DIBN_FMA(40, 0) DIBN_FMA(41, 1) DIBN_FMA(42, 2) DIBN_FMA(43, 3) DIBN_FMA(44, 4) DIBN_FMA(45, 5) DIBN_FMA(46, 6) DIBN_FMA(47, 7)
#define DIBN_READ(base)                                                                                      \
  "s_bfe_u32 %13, %11, 0xf0001\n\tv_add_u32 v48, %13, %15\n\t"                                                \
  "ds_read_b32 v[" #base "], v48\n\tds_read_b32 v[" #base "+1], v48 offset:384\n\t"                           \
  "ds_read_b32 v[" #base "+2], v48 offset:768\n\tds_read_b32 v[" #base "+3], v48 offset:1152\n\t"             \
  "ds_read_b32 v[" #base "+4], v48 offset:1536\n\tds_read_b32 v[" #base "+5], v48 offset:1920\n\t"            \
  "ds_read_b32 v[" #base "+6], v48 offset:2304\n\tds_read_b32 v[" #base "+7], v48 offset:2688\n\t"
#define DIBN_NEXTTAP "s_load_dword %12, %14, %8\n\ts_add_u32 %8, %8, 4\n\t"
template <bool FUSED>
__device__ __forceinline__ void tap_loop_narrow(h2 (&acc)[8], unsigned long long ltaps, int t0, int n, unsigned lane_addr) {
  unsigned toff = (unsigned)__builtin_amdgcn_readfirstlane(t0 * 4), cnt = (unsigned)__builtin_amdgcn_readfirstlane(n - 1);
  unsigned sA, sB, sC, st;
  unsigned a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(unsigned, acc[i]);
#define DIB_RN_ASM(ARITH_A, ARITH_B) \
  asm volatile( \
      "s_load_dword %11, %14, %8\n\ts_add_u32 %8, %8, 4\n\t" DIBN_NEXTTAP \
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" DIBN_READ(32) \
      "Ldibn_loop%=:\n\t" \
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %10, %11\n\ts_mov_b32 %11, %12\n\t" \
      DIBN_READ(40) DIBN_NEXTTAP \
      ARITH_A \
      "s_sub_u32 %9, %9, 1\n\ts_cbranch_scc1 Ldibn_done%=\n\t" \
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %10, %11\n\ts_mov_b32 %11, %12\n\t" \
      DIBN_READ(32) DIBN_NEXTTAP \
      ARITH_B \
      "s_sub_u32 %9, %9, 1\n\ts_cbranch_scc0 Ldibn_loop%=\n\t" \
      "Ldibn_done%=:\n\t" \
      "s_waitcnt lgkmcnt(0)" \
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+s"(toff), "+s"(cnt), \
        "=&s"(sA), "=&s"(sB), "=&s"(sC), "=&s"(st) \
      : "s"(ltaps), "v"(lane_addr) \
      : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "scc", \
        "memory")
  if constexpr (FUSED) { DIB_RN_ASM(DIBN_FMADD_A, DIBN_FMADD_B); } else { DIB_RN_ASM(DIBN_MADD_A, DIBN_MADD_B); }
#undef DIB_RN_ASM
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = __builtin_bit_cast(h2, a[i]);
}

template <int ACC>
__device__ __forceinline__ void blur_narrow_tile_f16(const ImageDesc &d, const int *__restrict__ tab, int K, int ch, int tx, int ty,
                                                     unsigned *lds) {
#pragma clang fp contract(off)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = d.H, W = d.W, w2 = W * 2;
  const int mode = pad_mode_for(K, H, W);
  const int pb = K / 2 - 1, pa = K / 2;
  const int nsegs = tab[HDR_NSEGS];
  const uint4 *segs = reinterpret_cast<const uint4 *>(tab + table_segs_off(K));
  const unsigned long long la = (unsigned long long)(tab + table_ltaps_off(K));
  const unsigned long long ltaps = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(la >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((unsigned)la);
  const int x0 = tx * NTILE_W, y0 = ty * TH;
  const __amdgpu_buffer_rsrc_t in_rsrc = plane_rsrc(d.in, ch, H, W);
  h2 acc[R];
#pragma unroll
  for (int i = 0; i < R; ++i) acc[i] = h2{0, 0};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
  const unsigned lane_addr = lds0 + (unsigned)((wave * R) * NPITCH + lane * 4);
  const int qb = wave * G;
  const unsigned wp = lds0 + (unsigned)(qb * NPITCH + lane * 4);
  typedef __attribute__((address_space(3))) unsigned lds_u1;

  for (int sg = 0; sg < nsegs; ++sg) {
    const Window w = window_of(segs[sg]);
    // ---- fill: per LDS row the three values P[lane + 64k] ---------------------------------------------------------
    unsigned v[G][3], coff[3];
    int soff[G];
    unsigned zmask = 0;
    const int c_first = x0 + pb - w.cmax, r_first = y0 + pb - w.rl;
    const bool zero_mode = mode == PAD_ZERO;
    if (!zero_mode && c_first >= 0 && c_first + 63 + 128 <= W - 1) {
      const unsigned c0 = 2u * (unsigned)(c_first + lane);
#pragma unroll
      for (int k = 0; k < 3; ++k) coff[k] = c0 + 128u * k;
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
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
        const int sr = map_coord_sel(r_first + min(qb + g, nrows - 1), H, pa, pb, mode, zr);
        zmask |= zr ? 1u << (8 + g) : 0u;
        soff[g] = sr * w2;
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int so = __builtin_amdgcn_readfirstlane(soff[g]);
#pragma unroll
      for (int k = 0; k < 3; ++k) v[g][k] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(in_rsrc, coff[k], so, 0);
    }
    if (sg > 0) __syncthreads();  // every wave is done reading the previous window
    const bool second = lane < w.cmax - w.cmin;
    const bool masked = __builtin_amdgcn_ballot_w64(zmask != 0) != 0;   // PAD_ZERO images only
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (masked) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (((zmask >> k) & 1u) || ((zmask >> (8 + g)) & 1u)) v[g][k] = 0;
      }
      const unsigned w0 = v[g][0] | (v[g][1] << 16);
      *(lds_u1 *)(size_t)(wp + (unsigned)(g * NPITCH)) = w0;
      if (second) *(lds_u1 *)(size_t)(wp + (unsigned)(g * NPITCH + 256)) = __builtin_amdgcn_alignbit(v[g][2], w0, 16);
    }
    __syncthreads();
    tap_loop_narrow<ACC == DIB_ACC_FMA16>(acc, ltaps, w.t0, w.n, lane_addr);
  }
  // ---- store (see store_tile: out-of-range lanes get an out-of-range offset, rows below the image a null descriptor) ----
  {
    const unsigned long long pa2 = (unsigned long long)d.out + (unsigned long long)ch * H * W * 2ull;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pa2), hi = __builtin_amdgcn_readfirstlane((unsigned)(pa2 >> 32));
    void *plane = (void *)(((unsigned long long)hi << 32) | lo);
    const int xr = W - x0 - lane;
    const unsigned voff = 2u * (unsigned)(x0 + lane), oob = 0x7ffffff0u;
    const unsigned vo0 = xr > 0 ? voff : oob, vo1 = xr > 64 ? voff + 128u : oob;
    const int yb = y0 + wave * R;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(plane, 0, yb + i < H ? H * w2 : 0, 0x00020000);
      const int so = (yb + i) * w2;
      const unsigned a = __builtin_bit_cast(unsigned, acc[i]);
      __builtin_amdgcn_raw_buffer_store_b16((short)(a & 0xffffu), out_rsrc, vo0, so, 0);
      __builtin_amdgcn_raw_buffer_store_b16((short)(a >> 16), out_rsrc, vo1, so, 0);
    }
  }
}

