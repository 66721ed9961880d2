// One C entry point per blur step: tap compaction + sparse correlation of a whole batch behind ONE call
// (reference models/blur_functions.py:92-100, `blur_image_list`: `psf / psf.sum()` + `manual_blur` per image).
//
// Why it exists: one call instead of two for the reference's one function, and the place where overlapping the compaction
// (8 workgroups, a 5-6 us latency chain that depends on nothing but the PSFs) with the PREVIOUS step's blur was tried:
//   * side stream + events (compaction on a library stream, hipEventRecord / hipStreamWaitEvent around it): 47.4 us per
//     step against 44.3 for the plain serial pair -- on this runtime every event call costs the host 4-5 us and the
//     cross-queue dependency ~3 us of GPU time (scratch/t_step_cost.py, profiles/r4_step_cost.txt);
//   * DIB_STEP_PSFS_COMPLETE -> the compaction launched with hipExtAnyOrderLaunch (no barrier bit in front of it, so the
//     command processor could start it while the kernel queued before it on the same stream still runs; the step's own blur
//     is an ordinary launch and waits for everything in front of it).  This is what the flag does today.  The HIP runtime
//     bundled with torch 2.10+rocm7.0 keeps the barrier anyway: 44.3 us with and without the flag.  The flag stays as a
//     hint that costs nothing and is correct by construction (below); a runtime that honours it gets the overlap.
//
// dib_blur_step_ws is the allocation-free form: tables and hand-off words live in a workspace the caller owns.
// Without one, tables live in TWO buffers per (device, stream) owned by this file (the one exception to include/dib.h's "no entry
// point allocates device memory"; grown to the largest batch seen, freed by dib_blur_step_release; at most STEP_MAX_STREAMS
// streams per device keep buffers -- a further stream takes over the least recently used one's, behind a device
// synchronisation, so a process that cycles through short-lived streams does not leak), used alternately:
// compact(n+1) writes the buffer blur(n-1) read, and blur(n-1) is complete before blur(n) -- which precedes compact(n+1) in
// the queue -- starts.  While the stream is being captured into a HIP graph the buffers are NOT used (a replay would
// write them behind this bookkeeping's back): the caller passes its own table buffer and both launches are ordinary.
#include "dib_common.h"
#include <mutex>
#include <unordered_map>

using namespace dib;

namespace {
struct Slot {
  int *buf = nullptr;
  size_t bytes = 0;        // of the tables; behind them (256-byte aligned) sit the single launch's counter replicas
  unsigned *sync = nullptr;
  unsigned target = 0;     // value the counter replicas reach when the slot's last single launch has compacted all its PSFs
  unsigned generation = 0; // handoff_generation(device) when the hand-off words were last cleared
};
constexpr size_t COUNTER_BYTES = (size_t)STEP_REPLICAS * STEP_REPLICA_WORDS * sizeof(unsigned);
constexpr size_t SYNC_BYTES = COUNTER_BYTES + (size_t)MAX_BATCH * STEP_REPLICAS * STEP_REC_WORDS * sizeof(unsigned);   // + the first-segment records

struct StepState {
  Slot slot[2];
  unsigned next = 0;
  unsigned long long last_use = 0;
};
constexpr size_t STEP_MAX_STREAMS = 8;     // per device
unsigned long long g_use = 0;

std::mutex g_step_mutex;
// keyed by (device, stream handle).  A handle reused by a later hipStreamCreate keeps its buffers; steps still in flight on
// a destroyed stream are the caller's bug, as with any other buffer.
std::unordered_map<unsigned long long, StepState> g_step;

unsigned long long key_of(int dev, hipStream_t s) { return ((unsigned long long)(uintptr_t)s << 6) ^ (unsigned long long)dev; }

// The state of (dev, s); a stream seen for the first time while the device already has STEP_MAX_STREAMS of them takes the place of
// the least recently used one.  That one's last step may still be running -- or its stream may be gone -- so the device is
// synchronised before its buffers are freed (rare: only a process that keeps opening new streams gets here).
int state_of(int dev, hipStream_t s, StepState **out) {
  const unsigned long long key = key_of(dev, s);
  auto it = g_step.find(key);
  if (it == g_step.end()) {
    size_t n = 0;
    auto lru = g_step.end();
    for (auto jt = g_step.begin(); jt != g_step.end(); ++jt) {
      if ((int)(jt->first & 63) != (dev & 63)) continue;
      ++n;
      if (lru == g_step.end() || jt->second.last_use < lru->second.last_use) lru = jt;
    }
    if (n >= STEP_MAX_STREAMS) {
      DIB_HIP_CHECK(hipDeviceSynchronize());
      for (Slot &sl : lru->second.slot)
        if (sl.buf) DIB_HIP_CHECK(hipFree(sl.buf));
      g_step.erase(lru);
    }
    it = g_step.emplace(key, StepState()).first;
  }
  it->second.last_use = ++g_use;
  *out = &it->second;
  return DIB_OK;
}
}  // namespace

extern "C" size_t dib_blur_step_workspace_bytes(int K, int num_psfs) {
  const size_t need = dib_tap_tables_bytes(K, num_psfs);
  if (need == 0) return 0;
  return ((SYNC_BYTES + 255) & ~(size_t)255) + need;
}

extern "C" int dib_blur_step(const void *const *psf_ptrs, int psf_dtype, int num_psfs, int K, int normalize,
                             const void *const *in_dev, void *const *out_dev, const int *C, const int *H, const int *W,
                             const int *table_index, int B, int dtype, int acc_mode, void *tables_dev, int flags,
                             void *stream) {
  if (num_psfs <= 0 || !psf_ptrs) { set_error("dib_blur_step: no PSFs"); return DIB_EINVAL; }
  if (K != 128 && K != 256) { set_error("dib_blur_step: K must be 128 or 256, got %d", K); return DIB_EINVAL; }
  if (psf_dtype != DIB_F16 && psf_dtype != DIB_F32) { set_error("dib_blur_step: unknown PSF dtype %d", psf_dtype); return DIB_EINVAL; }
  if (flags & ~(DIB_STEP_PSFS_COMPLETE | DIB_STEP_LARGE_WINDOW)) { set_error("dib_blur_step: unknown flags 0x%x", flags); return DIB_EINVAL; }
  if (flags & DIB_STEP_LARGE_WINDOW) {     // compact for and blur with the large LDS window
    normalize = (normalize ? 1 : 0) | DIB_COMPACT_LARGE_WINDOW;
    acc_mode |= DIB_WINDOW_LARGE;
  }
  if (acc_mode == DIB_ACC_FAST16) normalize = (normalize ? 1 : 0) | DIB_COMPACT_VRUNS;    // that mode walks the tables' vertical-run groups (always two launches)
  // what an earlier launch on this device left in the status word (a hand-off that timed out, a table of the wrong geometry):
  // reported here, once, without any synchronisation
  if (int rc = consume_device_status("dib_blur_step")) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  DIB_HIP_CHECK(hipStreamIsCapturing(s, &cap));
  if (tables_dev || cap != hipStreamCaptureStatusNone) {
    if (!tables_dev) {
      set_error("dib_blur_step: the stream is being captured into a graph: pass a table buffer of dib_tap_tables_bytes(K, num_psfs) bytes");
      return DIB_ECAPTURE;
    }
    // caller-owned tables: two ordinary launches in stream order
    if (int rc = dib_psf_compact_list(psf_ptrs, psf_dtype, num_psfs, K, normalize, tables_dev, stream)) return rc;
    return dib_sparse_blur(in_dev, out_dev, C, H, W, table_index, B, dtype, tables_dev, num_psfs, K, acc_mode, stream);
  }
  int dev = 0;
  DIB_HIP_CHECK(hipGetDevice(&dev));
  if (s) {   // the table buffers are allocated on the CURRENT device: a stream of another one would launch on memory it cannot reach
    hipDevice_t sdev = dev;
    if (hipStreamGetDevice(s, &sdev) == hipSuccess && (int)sdev != dev) {
      set_error("dib_blur_step: `stream` belongs to device %d, the current device is %d (hipSetDevice first)", (int)sdev, dev);
      return DIB_EINVAL;
    }
  }
  std::lock_guard<std::mutex> lock(g_step_mutex);
  StepState *stp = nullptr;
  if (int rc = state_of(dev, s, &stp)) return rc;
  StepState &st = *stp;
  Slot &sl = st.slot[st.next & 1];
  const size_t need = dib_tap_tables_bytes(K, num_psfs);
  if (sl.bytes < need) {
    if (sl.buf) {   // rare (a larger batch than any before on this stream): nobody may still read the old buffer
      DIB_HIP_CHECK(hipStreamSynchronize(s));
      DIB_HIP_CHECK(hipFree(sl.buf));
      sl.buf = nullptr; sl.bytes = 0;
    }
    const size_t tab_bytes = (need + 255) & ~(size_t)255;
    DIB_HIP_CHECK(hipMalloc((void **)&sl.buf, tab_bytes + SYNC_BYTES));
    sl.bytes = need;
    sl.sync = (unsigned *)((char *)sl.buf + tab_bytes);
    sl.target = 0;
    sl.generation = handoff_generation(dev);
    DIB_HIP_CHECK(hipMemsetAsync(sl.sync, 0, SYNC_BYTES, s));      // stream-ordered in front of the first launch that counts
  } else if (sl.generation != handoff_generation(dev)) {
    // a single launch on this device gave up since: its compacting workgroups may never have counted.  Start the slot's
    // hand-off words over (stream-ordered behind everything that could still touch them).
    sl.target = 0;
    sl.generation = handoff_generation(dev);
    DIB_HIP_CHECK(hipMemsetAsync(sl.sync, 0, SYNC_BYTES, s));
  }
  // One launch for the whole step where the shapes allow it (dib_blur.hip: blur_step_f16_kernel); else compaction + blur.
  if (K == 128 && psf_dtype == DIB_F16 && dtype == DIB_F16 && !(flags & DIB_STEP_LARGE_WINDOW)) {
    const int rc = blur_step_fused_launch(psf_ptrs, num_psfs, normalize, in_dev, out_dev, C, H, W, table_index, B, acc_mode, sl.buf, sl.sync,
                                          (unsigned *)((char *)sl.sync + COUNTER_BYTES), sl.target + (unsigned)num_psfs, s);
    if (rc <= 0) {
      if (rc == DIB_OK) { sl.target += (unsigned)num_psfs; ++st.next; }
      return rc;
    }
  }
  if (int rc = compact_launch(psf_ptrs, psf_dtype, num_psfs, K, normalize, sl.buf, s, (flags & DIB_STEP_PSFS_COMPLETE) != 0)) return rc;
  ++st.next;   // from here on the buffer counts as in use by this step, whatever the blur returns
  return dib_sparse_blur(in_dev, out_dev, C, H, W, table_index, B, dtype, sl.buf, num_psfs, K, acc_mode, stream);
}

extern "C" int dib_blur_step_release(void) {
  int dev = 0;
  DIB_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_step_mutex);
  DIB_HIP_CHECK(hipDeviceSynchronize());
  for (auto it = g_step.begin(); it != g_step.end();) {
    if ((int)(it->first & 63) != (dev & 63)) { ++it; continue; }
    for (Slot &sl : it->second.slot)
      if (sl.buf) DIB_HIP_CHECK(hipFree(sl.buf));
    it = g_step.erase(it);
  }
  return DIB_OK;
}

// The same call with its nine host arrays packed into two (a ctypes caller pays ~1.2 us per array it builds):
//   ptrs = psf_ptrs[num_psfs] | in_dev[B] | out_dev[B];   ints = C[B] | H[B] | W[B] | table_index[B]
extern "C" int dib_blur_step_packed(const void *const *ptrs, const int *ints, int psf_dtype, int num_psfs, int K, int normalize, int B,
                                    int dtype, int acc_mode, void *tables_dev, int flags, void *stream) {
  if (!ptrs || !ints || num_psfs < 0 || B < 0) { set_error("dib_blur_step_packed: null pointer or negative count"); return DIB_EINVAL; }
  return dib_blur_step(ptrs, psf_dtype, num_psfs, K, normalize, ptrs + num_psfs, (void *const *)(ptrs + num_psfs + B), ints, ints + B, ints + 2 * B,
                       ints + 3 * B, B, dtype, acc_mode, tables_dev, flags, stream);
}

// dib_blur_step on a workspace the CALLER owns (dib_blur_step_workspace_bytes(K, num_psfs) bytes of device memory, 256-byte
// aligned): no allocation, no library state, one buffer -- every launch is an ordinary stream-ordered one, so step n + 1's
// compaction cannot start before step n's blur has read its tables.  `ws_state` (host, in / out) is the workspace's hand-off
// state: 0 before the workspace's first use (that call clears the hand-off words with one hipMemsetAsync), updated by every call
// (bit 32: cleared; low word: compactions counted so far); one per workspace, and the steps on one workspace go to one stream
// at a time.  Under graph capture: two ordinary launches.
extern "C" int dib_blur_step_ws(const void *const *psf_ptrs, int psf_dtype, int num_psfs, int K, int normalize,
                                const void *const *in_dev, void *const *out_dev, const int *C, const int *H, const int *W,
                                const int *table_index, int B, int dtype, int acc_mode, void *workspace_dev, size_t workspace_bytes,
                                unsigned long long *ws_state, int flags, void *stream) {
  if (num_psfs <= 0 || !psf_ptrs) { set_error("dib_blur_step_ws: no PSFs"); return DIB_EINVAL; }
  if (K != 128 && K != 256) { set_error("dib_blur_step_ws: K must be 128 or 256, got %d", K); return DIB_EINVAL; }
  if (psf_dtype != DIB_F16 && psf_dtype != DIB_F32) { set_error("dib_blur_step_ws: unknown PSF dtype %d", psf_dtype); return DIB_EINVAL; }
  if (flags & ~(DIB_STEP_PSFS_COMPLETE | DIB_STEP_LARGE_WINDOW)) { set_error("dib_blur_step_ws: unknown flags 0x%x", flags); return DIB_EINVAL; }
  if (!workspace_dev || ((uintptr_t)workspace_dev & 255) || !ws_state || workspace_bytes < dib_blur_step_workspace_bytes(K, num_psfs)) {
    set_error("dib_blur_step_ws: the workspace must be 256-byte aligned device memory of at least dib_blur_step_workspace_bytes(K, num_psfs) = %zu bytes, with its host state word",
              dib_blur_step_workspace_bytes(K, num_psfs));
    return DIB_EINVAL;
  }
  if (int rc = consume_device_status("dib_blur_step_ws")) return rc;
  if (flags & DIB_STEP_LARGE_WINDOW) {
    normalize = (normalize ? 1 : 0) | DIB_COMPACT_LARGE_WINDOW;
    acc_mode |= DIB_WINDOW_LARGE;
  }
  if (acc_mode == DIB_ACC_FAST16) normalize = (normalize ? 1 : 0) | DIB_COMPACT_VRUNS;
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  DIB_HIP_CHECK(hipStreamIsCapturing(s, &cap));
  // layout: the hand-off words first (a fixed size), the tables behind them
  constexpr size_t SYNC_ALIGNED = (SYNC_BYTES + 255) & ~(size_t)255;
  int *tables = (int *)((char *)workspace_dev + SYNC_ALIGNED);
  if (cap == hipStreamCaptureStatusNone && K == 128 && psf_dtype == DIB_F16 && dtype == DIB_F16 && !(flags & DIB_STEP_LARGE_WINDOW)) {
    unsigned *sync = (unsigned *)workspace_dev;
    if (!(*ws_state >> 32)) {
      DIB_HIP_CHECK(hipMemsetAsync(sync, 0, SYNC_BYTES, s));
      *ws_state = 1ull << 32;
    }
    const unsigned counter = (unsigned)*ws_state;
    const int rc = blur_step_fused_launch(psf_ptrs, num_psfs, normalize, in_dev, out_dev, C, H, W, table_index, B, acc_mode, tables, sync,
                                          (unsigned *)((char *)sync + COUNTER_BYTES), counter + (unsigned)num_psfs, s);
    if (rc <= 0) {
      if (rc == DIB_OK) *ws_state = (1ull << 32) | (unsigned)(counter + (unsigned)num_psfs);
      return rc;
    }
  }
  if (int rc = dib_psf_compact_list(psf_ptrs, psf_dtype, num_psfs, K, normalize, tables, stream)) return rc;
  return dib_sparse_blur(in_dev, out_dev, C, H, W, table_index, B, dtype, tables, num_psfs, K, acc_mode, stream);
}
