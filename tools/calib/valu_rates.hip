// round 6: issue cost of the integer VALU instructions the tick kernel's classification round is made of, on gfx950 —
// one wave per SIMD slot (1 024 waves x 4 per SIMD would hide nothing: the chains are independent), 256 instructions of one kind
// back to back, 8 192 times; cycles per instruction from s_memtime (100 MHz constant clock is NOT what s_memtime counts on gfx9: it
// counts the shader clock's REFCLK-independent counter; we report the RATIO to v_add_u32, which is what matters).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int K>
__global__ void k(unsigned long long* out, unsigned* sink, unsigned a0, unsigned b0) {
  unsigned a = a0 + threadIdx.x, b = b0 ^ threadIdx.x, c = a * 3u, d = b + 7u;
  unsigned long long x = ((unsigned long long)a << 32) | b, y = ((unsigned long long)c << 32) | d;
  unsigned long long m0 = __builtin_amdgcn_readfirstlane(a0) * 0x9E3779B97F4A7C15ull, m1 = ~m0;
  unsigned s0 = __builtin_amdgcn_readfirstlane(a0), s1 = __builtin_amdgcn_readfirstlane(b0);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 8192; ++it) {
    if (K == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
    if (K == 1) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
    if (K == 2) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y) : "v"(c), "v"(d) : "vcc");) }
    if (K == 3) { REP64(asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(x), "v"(y) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(y), "v"(x) : "vcc");) }
    if (K == 4) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(c), "v"(d) : "vcc");) }
    if (K == 5) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 4, %1" : "+v"(x) : "v"(y)); asm volatile("v_lshl_add_u64 %0, %0, 4, %1" : "+v"(y) : "v"(x));) }
    if (K == 6) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c) : "v"(d));) }
    if (K == 7) { REP64(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(a));) }
    if (K == 8) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c) : "v"(d) : "vcc");) }
    if (K == 9) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
    if (K == 10) { REP64(asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x)); asm volatile("v_lshlrev_b64 %0, 5, %0" : "+v"(y));) }
    if (K == 11) { REP64(asm volatile("v_cmp_eq_u64 vcc, %0, %1" : : "v"(x), "v"(y) : "vcc"); asm volatile("v_cmp_eq_u64 vcc, %0, %1" : : "v"(y), "v"(x) : "vcc");) }
    if (K == 12) { REP64(asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a)); asm volatile("v_bfe_u32 %0, %0, 2, 11" : "+v"(c));) }
    if (K == 14) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c) : "v"(d));) }
    if (K == 15) { REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(m0)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(c) : "v"(d), "s"(m1));) }
    if (K == 16) { REP64(asm volatile("v_add_u32 %0, %0, %1\n s_nop 0" : "+v"(a) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1\n s_nop 0" : "+v"(c) : "v"(d));) }
    if (K == 17) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c) : "v"(d) : "vcc");) }
    if (K == 18) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b), "v"(c), "v"(d) : "vcc");) }
    if (K == 19) { REP64(asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(c) : "v"(d), "v"(a));) }
    if (K == 20) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "s"(s0)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "s"(s1));) }
    if (K == 21) { REP64(asm volatile("s_and_b64 %0, %0, %1" : "+s"(m0) : "s"(m1)); asm volatile("s_or_b64 %0, %0, %1" : "+s"(m1) : "s"(m0));) }
    if (K == 22) { REP64(asm volatile("v_add_u32 %0, %0, %2\n s_and_b64 %1, %1, %3" : "+v"(a), "+s"(m0) : "v"(b), "s"(m1)); asm volatile("v_add_u32 %0, %0, %2\n s_or_b64 %1, %1, %3" : "+v"(c), "+s"(m1) : "v"(d), "s"(m0));) }
    if (K == 13) { REP64(asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (a + c + (unsigned)x + (unsigned)y + (unsigned)m0 + (unsigned)m1 == 0x12345u) *sink = 1;
}
template <int K> static double run(int waves_per_simd) {
  unsigned long long* out; unsigned* sink;
  const int blocks = 256 * 4 * waves_per_simd;
  hipMalloc(&out, blocks * 8); hipMalloc(&sink, 4);
  k<K><<<blocks, 64>>>(out, sink, 1, 2); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<K><<<blocks, 64>>>(out, sink, 1, 2); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out); hipFree(sink);
  return ms * 1e6 / (8192.0 * 128.0);  // ns per instruction per wave-slot set (all SIMDs in parallel)
}
int main() {
  const char* names[] = {"v_add_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_cmp_lt_u64", "v_cmp_lt_u32", "v_lshl_add_u64", "v_mul_u32_u24", "v_mad_u32_u24", "v_cndmask_b32", "v_mul_hi_u32", "v_lshlrev_b64", "v_cmp_eq_u64", "v_bfe_u32", "v_min/max_u32", "v_cndmask vcc (no nop)", "v_cndmask_e64 sgpr", "v_add + s_nop 0", "cmp;s_nop 1;cndmask (3)", "cmp;add;add;cndmask (4)", "v_bfi_b32", "v_add_u32 v,v,sgpr", "s_and/or_b64", "v_add + s_and (2)"};
  for (int w : {1, 4}) {
    double r[23] = {run<0>(w), run<1>(w), run<2>(w), run<3>(w), run<4>(w), run<5>(w), run<6>(w), run<7>(w), run<8>(w), run<9>(w), run<10>(w), run<11>(w), run<12>(w), run<13>(w), run<14>(w), run<15>(w), run<16>(w), run<17>(w), run<18>(w), run<19>(w), run<20>(w), run<21>(w), run<22>(w)};
    for (int i = 0; i < 23; ++i) printf("waves/SIMD %d  %-16s %7.3f ns per instruction (x %.2f of v_add_u32)\n", w, names[i], r[i] / w, r[i] / r[0]);
  }
  return 0;
}
