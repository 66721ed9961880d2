// Is v_mad_f16 / v_mac_f16 (gfx950 still has them) the reference's arithmetic -- RN16(RN16(w * p) + acc), subnormals kept -- in ONE
// instruction?  Compares them bit for bit with v_mul_f16 + v_add_f16 over (1) ALL 2^32 (p, w) pairs against a few accumulators,
// (2) 2^32 random (p, w, acc) triples.  Prints the number of differing results by class.
//   hipcc -O3 --offload-arch=gfx950 scratch/ubench/ub_mad.hip -o scratch/ubench/ub_mad && scratch/ubench/ub_mad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned ref_muladd(unsigned p, unsigned w, unsigned c) {
  unsigned t, r;
  asm volatile("v_mul_f16 %0, %1, %2" : "=v"(t) : "v"(p), "v"(w));
  asm volatile("v_add_f16 %0, %1, %2" : "=v"(r) : "v"(t), "v"(c));
  return r & 0xffffu;
}
__device__ __forceinline__ unsigned mad(unsigned p, unsigned w, unsigned c) {
  unsigned r;
  asm volatile("v_mad_f16 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(w), "v"(c));
  return r & 0xffffu;
}
__device__ __forceinline__ unsigned mad_legacy(unsigned p, unsigned w, unsigned c) {
  unsigned r;
  asm volatile("v_mad_legacy_f16 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(w), "v"(c));
  return r & 0xffffu;
}
__device__ __forceinline__ unsigned mac(unsigned p, unsigned w, unsigned c) {
  unsigned r = c;
  asm volatile("v_mac_f16 %0, %1, %2" : "+v"(r) : "v"(p), "v"(w));
  return r & 0xffffu;
}
__device__ __forceinline__ unsigned mad_hi(unsigned p, unsigned w, unsigned c) {     // operands and result in the HIGH halves (op_sel)
  unsigned r = 0;
  asm volatile("v_mad_f16 %0, %1, %2, %3 op_sel:[1,1,1,1]" : "+v"(r) : "v"(p << 16), "v"(w << 16), "v"(c << 16));
  return r >> 16;
}
__device__ __forceinline__ unsigned mad_sw(unsigned p, unsigned w, unsigned c) {     // weight from a scalar register's high half
  unsigned r;
  const unsigned ws = __builtin_amdgcn_readfirstlane(w << 16);
  asm volatile("v_mad_f16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "s"(ws), "v"(p), "v"(c));
  return r & 0xffffu;
}
__device__ __forceinline__ bool is_nan(unsigned h) { return (h & 0x7c00u) == 0x7c00u && (h & 0x3ffu); }
__device__ __forceinline__ bool is_sub(unsigned h) { return (h & 0x7c00u) == 0 && (h & 0x3ffu); }
__device__ __forceinline__ bool same(unsigned a, unsigned b) { return a == b || (is_nan(a) && is_nan(b)); }

// counters: [variant 0..3 = mad, mad_legacy, mac, mad_hi][class 0 = all, 1 = a subnormal among inputs / product / result]
__global__ void k_pairs(unsigned long long *cnt, unsigned acc_bits) {
  const unsigned p = blockIdx.x * 256 + threadIdx.x;       // 65536 threads: one p each
  unsigned long long d[5][2] = {};
  for (unsigned w = 0; w < 65536; ++w) {
    const unsigned r = ref_muladd(p, w, acc_bits);
    unsigned t;
    asm volatile("v_mul_f16 %0, %1, %2" : "=v"(t) : "v"(p), "v"(w));
    const bool sub = is_sub(p) || is_sub(w) || is_sub(acc_bits) || is_sub(t & 0xffffu) || is_sub(r);
    const unsigned v[4] = {mad(p, w, acc_bits), mad_legacy(p, w, acc_bits), mac(p, w, acc_bits), mad_hi(p, w, acc_bits)};
    for (int i = 0; i < 4; ++i)
      if (!same(v[i], r)) { d[i][0]++; if (sub) d[i][1]++; }
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) if (d[i][j]) atomicAdd(&cnt[i * 2 + j], d[i][j]);
}
__global__ void k_random(unsigned long long *cnt, unsigned seed, unsigned *first) {
  unsigned long long s = (unsigned long long)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + seed;
  unsigned long long d[5][2] = {};
  for (int it = 0; it < 16384; ++it) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const unsigned p = (unsigned)s & 0xffffu, w = (unsigned)(s >> 16) & 0xffffu, c = (unsigned)(s >> 32) & 0xffffu;
    const unsigned r = ref_muladd(p, w, c);
    unsigned t;
    asm volatile("v_mul_f16 %0, %1, %2" : "=v"(t) : "v"(p), "v"(w));
    const bool sub = is_sub(p) || is_sub(w) || is_sub(c) || is_sub(t & 0xffffu) || is_sub(r);
    const unsigned v[4] = {mad(p, w, c), mad_legacy(p, w, c), mac(p, w, c), mad_hi(p, w, c)};
    for (int i = 0; i < 4; ++i)
      if (!same(v[i], r)) {
        d[i][0]++; if (sub) d[i][1]++;
        if (i == 0 && !sub && atomicAdd(first, 1u) < 8) printf("mad differs without subnormals: p %04x w %04x c %04x: mul+add %04x mad %04x (product %04x)\n", p, w, c, r, v[0], t & 0xffffu);
      }
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) if (d[i][j]) atomicAdd(&cnt[i * 2 + j], d[i][j]);
}
// the blur's operand form: weight from a scalar register (wave-uniform), pixel and accumulator per lane
__global__ void k_scalar_w(unsigned long long *cnt, unsigned w) {
  const unsigned p = blockIdx.x * 256 + threadIdx.x;
  unsigned long long d = 0;
  for (unsigned c = 0; c < 65536; c += 7) { if (!same(mad_sw(p, w, c), ref_muladd(p, w, c))) d++; }
  if (d) atomicAdd(&cnt[0], d);
}

int main() {
  unsigned long long *cnt; unsigned *first;
  CK(hipMalloc(&cnt, 16 * 8)); CK(hipMalloc(&first, 4));
  const char *names[4] = {"v_mad_f16", "v_mad_legacy_f16", "v_mac_f16", "v_mad_f16 op_sel hi"};
  const unsigned accs[] = {0x0000, 0x8000, 0x3c00, 0x3555, 0x0001, 0x03ff, 0x0400, 0x1234, 0xb7ff, 0x7bff};
  unsigned long long h[16];
  for (unsigned a : accs) {
    CK(hipMemset(cnt, 0, 16 * 8));
    k_pairs<<<256, 256>>>(cnt, a);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(h, cnt, 16 * 8, hipMemcpyDeviceToHost));
    printf("all 2^32 (p, w) pairs, acc %04x:", a);
    for (int i = 0; i < 4; ++i) printf("  %s %llu differ (%llu with a subnormal involved)", names[i], h[2 * i], h[2 * i + 1]);
    printf("\n");
  }
  CK(hipMemset(cnt, 0, 16 * 8)); CK(hipMemset(first, 0, 4));
  k_random<<<1024, 256>>>(cnt, 12345u, first);
  CK(hipDeviceSynchronize()); CK(hipMemcpy(h, cnt, 16 * 8, hipMemcpyDeviceToHost));
  printf("2^32 random (p, w, acc) triples:");
  for (int i = 0; i < 4; ++i) printf("  %s %llu differ (%llu with a subnormal involved)", names[i], h[2 * i], h[2 * i + 1]);
  printf("\n");
  unsigned long long tot = 0;
  for (unsigned w : {0x2400u, 0x1c00u, 0x0c00u, 0x0155u, 0x3bffu, 0x8400u}) {
    CK(hipMemset(cnt, 0, 16 * 8));
    k_scalar_w<<<256, 256>>>(cnt, w);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
    tot += h[0];
  }
  printf("weight from a scalar register's high half (6 weights x 65536 p x 9363 acc): %llu differ\n", tot);
  return 0;
}
