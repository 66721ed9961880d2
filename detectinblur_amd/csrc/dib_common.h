// Shared declarations for the gfx950 kernels behind include/dib.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dib.h"

namespace dib {

// ---- tap-table layout (see include/dib.h) ------------------------------------------------
constexpr int HDR_NTAPS = 0, HDR_RMIN = 1, HDR_RMAX = 2, HDR_CMIN = 3, HDR_CMAX = 4, HDR_K = 5,
              HDR_SUM = 6, HDR_FLAGS = 7, HDR_WORDS = 8;

// [7] = number of tap segments.  A segment is a run of consecutive (row-major) taps whose bounding
// box spans at most SEG_ROWS+1 PSF rows and SEG_COLS+1 PSF columns: the unit the tiled blur stages
// in LDS.  Entry = uint4 {first tap, end tap, (r_first << 8) | r_last, (cmin << 8) | cmax}.
constexpr int HDR_NSEGS = 7;
constexpr int SEG_ROWS = 12, SEG_COLS = 24;
// LDS window geometry of the tiled blur (dib_blur.hip), needed here because the compaction kernel
// pre-computes, per tap, the byte offset of its source word inside that window, once per window layout:
//   ltap   = ((r_last - r) * WIN_PITCH  + (cmax - c)) * 8   |   fp16 weight bits << 16     256- and 128-wide tiles
//   ltap_q = ((r_last - r) * QUAD_PITCH + (cmax - c)) * 8   |   fp16 weight bits << 16     128-wide "quad" tiles
constexpr int WIN_PITCH = 96;              // 8-byte words per LDS window row (64 + SEG_COLS are used)
constexpr int QUAD_PITCH = 32 + SEG_COLS;  // 8-byte elements {P[k], P[k+32], P[k+64], P[k+96]} per LDS window row (56)
static_assert(WIN_PITCH >= 64 + SEG_COLS, "window row too short for a segment");
// The LARGE window of the default ("quad") tiles: segments of up to 21 rows x 64 columns, 52 LDS rows of 96 elements = 39,936 B
// (4 workgroups per CU instead of 8).  For launches that leave most of the chip's slots empty anyway (evaluation at batch 1:
// 825 workgroups on 2,048 slots) and whose PSF is wide or tall.  Segments are runs of ROW-MAJOR consecutive taps (the order
// the reference accumulates in), so a run can only cross a row boundary when whole rows fit the window's columns: a
// full-exposure PSF 48 columns wide is 19 standard segments (two per row) and ONE large one; every segment is a window refill
// (~3 us at batch 1) that nothing overlaps.  A table is compacted for ONE geometry (tab[HDR_K] = K | geometry << 16); the blur
// is told which by its caller.
constexpr int SEG_ROWS_L = 20, SEG_COLS_L = 63, QUAD_PITCH_L = 32 + SEG_COLS_L + 1;   // 96 elements per LDS row
static_assert(QUAD_PITCH_L == 96, "the large window's fill writes elements lane and 64 + lane");
constexpr int COMPACT_NORMALIZE = 1, COMPACT_NO_SEGMENTS = 4, COMPACT_LARGE_WINDOW = 8, COMPACT_VRUNS = 16, COMPACT_DEBUG_SKIP = 0x100, COMPACT_DEBUG_NOSIGNAL = 0x200;   // flag bits of the compaction kernel
// Vertical-run groups (DIB_COMPACT_VRUNS -> DIB_ACC_FAST16, the tolerance mode that may reorder taps): inside a segment, taps of one
// PSF column in consecutive PSF rows form a run; a run is cut into groups of at most VRUN_MAX taps.  The n taps of a group share
// n + 3 window rows per lane (the lane's four output rows slide down the column) instead of 4 n: half the LDS reads of the tap
// loop.  The groups of the segment whose taps are [t0, t1) occupy records t0, t0 + 1, ... of the table's `vgroups` section, one
// uint4 each:
//   x = LDS byte offset of the NEXT group's first window row and column (low 16) | (its tap count - 1) << 16; 4 << 16 behind the last
//   y = w0 | w1 << 16,  z = w2 | w3 << 16     fp16 weights; tap j of the group reads window rows j .. j + 3 from the group's
//                                             offset: w0 belongs to the HIGHEST PSF row of the group
//   w = in the segment's FIRST record: its own offset | (its tap count - 1) << 16 (the form of x)
// so that a tap loop needs the record of group g alone to run group g and to fetch group g + 1's pixels.  A segment's groups are
// stored sorted by size, fours first: the loop runs one straight-line body per size and changes body at most three times.
// tab[HDR_K] bit 17 says that the section is valid (fp16 PSF, standard window, every segment's taps staged in LDS).
constexpr int VRUN_MAX = 4;
constexpr unsigned HDR_K_VRUNS = 1u << 17;

__host__ __device__ inline int table_rowptr_off() { return HDR_WORDS; }
__host__ __device__ inline int table_taps_off(int K) { return (HDR_WORDS + K + 1 + 3) & ~3; }
__host__ __device__ inline int table_segs_off(int K) { return table_taps_off(K) + 2 * K * K; }
__host__ __device__ inline int table_ltaps_off(int K) { return table_segs_off(K) + 4 * K * K; }
__host__ __device__ inline int table_ltaps_q_off(int K) { return table_ltaps_off(K) + K * K + 8; }
__host__ __device__ inline int table_vgroups_off(int K) { return (table_ltaps_q_off(K) + K * K + 8 + 3) & ~3; }   // uint4 records: 16-byte aligned
__host__ __device__ inline int table_words(int K) { return table_vgroups_off(K) + 4 * K * K + 16; }

// ---- padding modes of manual_blur (models/blur_functions.py:28-31, :55-58) ---------------
enum PadMode { PAD_REFLECT = 0, PAD_ZERO = 1, PAD_REPLICATE = 2 };

__host__ __device__ inline int pad_mode_for(int K, int H, int W) {
  if (K > 129) return PAD_REPLICATE;
  return (H < 64 || W < 64) ? PAD_ZERO : PAD_REFLECT;
}

// Maps a virtual (un-padded) coordinate s in [-pa, n-1+pb] to a source index in [0, n).
// s == -pa is the circular-wrap row/column of torch.roll on the padded image:
// it reads padded index n+K-2, i.e. virtual coordinate n+pb.  `zero` reports a zero-fill read.
// Coordinates past n-1+pb (lanes outside the image) are clamped; their results are never stored.
__device__ inline int map_coord(int s, int n, int pa, int pb, int mode, bool &zero) {
  if (s == -pa) s = n + pb;
  zero = false;
  if (mode == PAD_REFLECT) {
    if (s < 0) s = -s;
    if (s > n - 1) s = 2 * (n - 1) - s;
  } else if (mode == PAD_ZERO) {
    zero = (s < 0) || (s > n - 1);
  }
  return min(max(s, 0), n - 1);
}

void set_error(const char *fmt, ...);
// dib_compact.hip: the launch behind dib_psf_compact_list; any_order = hipExtAnyOrderLaunch (see dib_step.hip)
int compact_launch(const void *const *ptrs, int dtype, int B, int K, int normalize, int *tables, hipStream_t s, bool any_order);
// dib_blur.hip: compaction + blur as ONE launch (fp16 PSFs and images, K = 128, default tiles, at most MAX_BATCH of each).
// Returns 1 when the arguments are outside what that launch serves (nothing launched: the caller takes the two-launch path).
int blur_step_fused_launch(const void *const *psf_ptrs, int num_psfs, int normalize, const void *const *in_dev, void *const *out_dev,
                           const int *C, const int *H, const int *W, const int *table_index, int B, int acc_mode, int *tables,
                           unsigned *sync, unsigned *rec, unsigned target, hipStream_t s);

// dib_blur.hip: the device's status word (pinned host memory a kernel writes a code to instead of trapping).  Looks at the current
// device's word WITHOUT synchronising; DIB_OK when it is clear, else the error to return (text set, word cleared; a hand-off
// timeout also takes the blur step's single launch out of service on that device).
int consume_device_status(const char *who);
// number of hand-off timeouts consumed on `dev` so far: dib_step.hip clears a slot's hand-off words again when this has moved since
// it last did (a launch that gave up may have left its counter short of the value the host expects)
unsigned handoff_generation(int dev);

#define DIB_HIP_CHECK(expr)                                                          \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      dib::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return DIB_EHIP;                                                               \
    }                                                                                \
  } while (0)

// Per-launch image descriptors travel by value in the kernel argument buffer (no H2D copy,
// no host sync).  MAX_BATCH images per launch; larger batches are split by the host wrapper.
constexpr int MAX_BATCH = 32;

struct ImageDesc {
  const void *in;
  void *out;
  int C, H, W;
  int table;       // index into the table array
  int tile_begin;  // first flattened tile id of this image in the launch
  int tiles_x;     // tiles per row (128 or 256 px wide, by the shape of the launch)
  int tiles_y;     // 32-row tiles per channel
  // floor(2^32 / d) + 1 for d = tiles_x * tiles_y and d = tiles_x (0 when d == 1): the workgroup's tile index is split into
  // (channel, tile row, tile column) with two multiplies instead of two ~25-instruction scalar divisions per wave
  unsigned inv_per_ch, inv_tiles_x;
  const int *tab;  // this image's tap table (tables + table * table_words(K)), formed on the host
};

// n / d for n * d < 2^32, inv = floor(2^32 / d) + 1 (exact: the error term n * (inv * d - 2^32) / (d * 2^32) stays below 1 / d)
__host__ __device__ inline unsigned magic_inverse(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / d) + 1u; }
__device__ __forceinline__ int magic_div(int n, unsigned inv) { return inv ? (int)__umulhi((unsigned)n, inv) : n; }

// PSF pointers travel by value in the kernel-argument buffer: a batch whose PSFs live in separate
// tensors (the reference's `psfs_GPU` list, engine.py:84) needs no torch.stack copy.
struct PsfPtrs { const void *p[MAX_BATCH]; };

// The blur step's single launch (dib_blur.hip: blur_step_f16_kernel; dib_step.hip owns the buffers).  `sync`: STEP_REPLICAS
// copies of one monotonic counter, each on a 128-byte line of its own; every compacting workgroup adds 1 to every copy once its
// table is written through to memory, a blur workgroup reads the tables once its copy has reached `target` (the value the
// counter has when all of THIS launch's n_psf tables are done; compared as a signed difference, so it may wrap).
constexpr int STEP_REPLICAS = 32, STEP_REPLICA_WORDS = 32, STEP_REC_WORDS = 32;
struct StepSync {
  unsigned *sync;     // STEP_REPLICAS x STEP_REPLICA_WORDS words
  unsigned target;
  int n_psf;          // PSFs (= compacting workgroups) of this launch
  int ncx;            // grid columns in front of the blur's: n_psf rounded up to a multiple of 8
  int flags;          // the compaction's flag word (COMPACT_NORMALIZE | ...)
  int row;            // grid x extent (= ncx + the blur's columns): a block's linear index is y * row + x
  int *tables;        // n_psf tables, table_words(K) apart
  // First-segment records: per PSF STEP_REPLICAS copies, each on a 128-byte line of its own, of two 8-byte words {data, tag}:
  // {end of the first segment | segments << 16, tag}, {(r_first << 8 | r_last) << 16 | cmin << 8 | cmax, tag}; tag = `target`.
  // Written (sc1) as soon as the PSF's segments are final, long before its offsets are: a blur workgroup that finds both tags
  // fills its first window from these words and looks at the counter only in front of its first tap loop.
  unsigned *rec;      // MAX_BATCH x STEP_REPLICAS x STEP_REC_WORDS words
  // Polls (sc1 load + s_sleep 2, ~1 us each) a blur workgroup spends on a hand-off word before it gives up: it then writes
  // DIB_STATUS_HANDOFF into the device's status word and ends instead of hanging or trapping (dib_blur.hip: dib_status_word).
  unsigned poll_budget;
};

// Flat grid of a RAGGED batch on the default tiles (dib_blur.hip: blur_quad_f16_kernel<.., FLAT = true>): the 2-D grid (band
// entry x image) of a batch whose images differ in size is a third empty workgroups, and the dispatcher's round robin over the
// CUs then leaves them 4-8 working ones each; a 1-D grid of exactly the working workgroups, heaviest image first, gives every
// CU one workgroup of every 256 consecutive ones.  Workgroup b works for XCD list x = b & 7, entry t = b >> 3; list x is band
// x of image 0, then band x of image 1, ...: begin[x][k] = entry at which image k starts (INT_MAX for k >= n), begin[x][15] =
// the list's length.  One 64-byte scalar load per workgroup; at most FLAT_MAX images per launch.
constexpr int FLAT_MAX = 15;
struct FlatBands { int begin[8][16]; unsigned rev_mask; };     // rev_mask bit r: full stride r (32 workgroups) of every list is walked backwards

struct BlurBatch {
  ImageDesc img[MAX_BATCH];
  int tile_begin[MAX_BATCH + 1];  // copy of img[i].tile_begin (+ total), contiguous for the image lookup
  int n;
  int total_tiles;
  int xcd_bands;  // tile order of the tiled kernel: 1 = per-XCD bands of every image, 0 = flat
};

}  // namespace dib
