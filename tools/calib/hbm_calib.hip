// hbm_calib.hip — known-byte-count kernels in the tick kernel's access shapes, for calibrating rocprofv3's
// FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads 1/2 of a WIDE coalesced stream;
// other widths and WRITE_SIZE are uncalibrated).  Every kernel touches a 2 GiB region (8x the 256 MiB Infinity
// Cache) exactly once, so the bytes that must cross the HBM interface are known; `tools/calib/run_calib.sh`
// collects FETCH_SIZE and WRITE_SIZE per kernel in separate passes and tools/calib/calib_summary.py divides.
//
//   rd16_s16   16 B per lane, lanes 16 B apart  (row groups, sort keys: dense 1 KiB per wave)
//   rd16_s32   16 B per lane, lanes 32 B apart  (head of a 32-byte view entry / ring bucket, same slot in every lane)
//   rd16_s64   16 B per lane, lanes 64 B apart  (first record of an inbox cell)
//   rd64_cell  4 x 16 B per lane, lanes 64 B apart (a whole inbox cell per lane: 4 KiB contiguous per wave)
//   rd16_rand  16 B per lane at a random 32-byte entry (heads of different slots per lane)
//   rd4_rand   4 B per lane at a random word of a 4 MiB table (slot map: cache resident, should read ~0)
//   wr16_s16   16 B per lane, dense
//   wr64_quad  one 64-byte cell per quad of lanes at a random cell (the cooperative scatter store)
//   wr16_rand4 four 16 B stores per lane into one random 64-byte cell (the scatter before it was made cooperative)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 v4u __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ static inline u64 mix64(u64 z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// a bijection on [0, 2^bits): random but hits every index exactly once
__device__ static inline u64 perm(u64 x, u32 bits) {
  u64 mask = (1ull << bits) - 1;
  x = (x * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask;
  x ^= x >> (bits / 2);
  x = (x * 0xBF58476D1CE4E5B9ull + 0x1CE4E5B9ull) & mask;
  x ^= x >> (bits / 2);
  x = (x * 0x94D049BB133111EBull + 0x133111EBull) & mask;
  return x;
}

__global__ void rd16(const v4u* __restrict__ p, u64 n_lanes, u32 stride16, u32* sink) {
  u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  v4u v = p[i * stride16];
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = 1;
}
__global__ void rd64_cell(const v4u* __restrict__ p, u64 n_lanes, u32* sink) {
  u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  v4u a = p[i * 4], b = p[i * 4 + 1], c = p[i * 4 + 2], d = p[i * 4 + 3];
  if ((a.x ^ b.y ^ c.z ^ d.w) == 0x12345u) *sink = 1;
}
__global__ void rd16_rand(const v4u* __restrict__ p, u64 n_lanes, u32 bits, u32* sink) {  // entries of 32 B, 2^bits of them
  u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  v4u v = p[perm(i, bits) * 2];
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = 1;
}
__global__ void rd4_rand(const u32* __restrict__ p, u64 n_lanes, u32 words_mask, u32* sink) {
  u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  u32 v = p[(u32)mix64(i) & words_mask];
  if (v == 0x12345u) *sink = 1;
}
__global__ void wr16(v4u* __restrict__ p, u64 n_lanes) {
  u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  v4u v = {(u32)i, 1u, 2u, 3u};
  p[i] = v;
}
__global__ void wr64_quad(v4u* __restrict__ p, u64 n_lanes, u32 bits) {  // 2^bits cells of 64 B, one per quad
  u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  u64 cell = perm(i >> 2, bits);
  v4u v = {(u32)i, 1u, 2u, 3u};
  p[cell * 4 + (i & 3)] = v;
}
__global__ void wr16_rand4(v4u* __restrict__ p, u64 n_lanes, u32 bits) {  // one 64-byte cell per lane, four stores
  u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  u64 cell = perm(i, bits);
  v4u v = {(u32)i, 1u, 2u, 3u};
  p[cell * 4] = v; p[cell * 4 + 1] = v; p[cell * 4 + 2] = v; p[cell * 4 + 3] = v;
}
__global__ void flush_rd(const v4u* __restrict__ p, u64 n, u32* sink) {  // evict: stream another region through the caches
  u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  if (i >= n) return;
  v4u v = p[i];
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = 1;
}

int main(int argc, char** argv) {
  const u64 REGION = 2ull << 30;  // bytes each kernel covers
  int reps = argc > 1 ? atoi(argv[1]) : 3;
  void *buf = nullptr, *evict = nullptr, *tab = nullptr;
  u32* sink = nullptr;
  CK(hipMalloc(&buf, REGION));
  CK(hipMalloc(&evict, 1ull << 30));
  CK(hipMalloc(&tab, 4u << 20));
  CK(hipMalloc((void**)&sink, 4));
  CK(hipMemset(buf, 0x5A, REGION));
  CK(hipMemset(evict, 0x33, 1ull << 30));
  CK(hipMemset(tab, 0x11, 4u << 20));
  CK(hipMemset(sink, 0, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int B = 256;
  auto grid = [&](u64 n) { return (unsigned)((n + B - 1) / B); };
  printf("{\"region_bytes\": %llu, \"kernels\": [\n", (unsigned long long)REGION);
  bool first = true;
  auto report = [&](const char* name, u64 useful_rd, u64 line_rd, u64 useful_wr, float ms) {
    printf("%s {\"name\": \"%s\", \"useful_read_bytes\": %llu, \"line64_read_bytes\": %llu, \"write_bytes\": %llu, \"ms\": %.4f}", first ? "" : ",\n",
           name, (unsigned long long)useful_rd, (unsigned long long)line_rd, (unsigned long long)useful_wr, ms);
    first = false;
  };
#define RUN(name, useful_rd, line_rd, useful_wr, launch)                                     \
  for (int r = 0; r < reps; ++r) {                                                           \
    flush_rd<<<grid((1ull << 30) / 16), B>>>((const v4u*)evict, (1ull << 30) / 16, sink);    \
    CK(hipEventRecord(e0));                                                                  \
    launch;                                                                                  \
    CK(hipEventRecord(e1));                                                                  \
    CK(hipEventSynchronize(e1));                                                             \
    float ms = 0;                                                                            \
    CK(hipEventElapsedTime(&ms, e0, e1));                                                    \
    if (r == reps - 1) report(name, useful_rd, line_rd, useful_wr, ms);                      \
  }
  u64 n;
  n = REGION / 16; RUN("rd16_s16", n * 16, n * 16, 0, (rd16<<<grid(n), B>>>((const v4u*)buf, n, 1, sink)));
  n = REGION / 32; RUN("rd16_s32", n * 16, n * 32, 0, (rd16<<<grid(n), B>>>((const v4u*)buf, n, 2, sink)));
  n = REGION / 64; RUN("rd16_s64", n * 16, n * 64, 0, (rd16<<<grid(n), B>>>((const v4u*)buf, n, 4, sink)));
  n = REGION / 64; RUN("rd64_cell", n * 64, n * 64, 0, (rd64_cell<<<grid(n), B>>>((const v4u*)buf, n, sink)));
  n = REGION / 32; RUN("rd16_rand", n * 16, n * 32, 0, (rd16_rand<<<grid(n), B>>>((const v4u*)buf, n, 26, sink)));  // 2^26 entries of 32 B = 2 GiB
  n = REGION / 32; RUN("rd4_rand", 0, 0, 0, (rd4_rand<<<grid(n), B>>>((const u32*)tab, n, (1u << 20) - 1, sink)));
  n = REGION / 16; RUN("wr16_s16", 0, 0, n * 16, (wr16<<<grid(n), B>>>((v4u*)buf, n)));
  n = REGION / 16; RUN("wr64_quad", 0, 0, n * 16, (wr64_quad<<<grid(n), B>>>((v4u*)buf, n, 25)));   // 2^25 cells of 64 B = 2 GiB
  n = REGION / 64; RUN("wr16_rand4", 0, 0, n * 64, (wr16_rand4<<<grid(n), B>>>((v4u*)buf, n, 25)));
  printf("\n]}\n");
  return 0;
}
