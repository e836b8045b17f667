// Microbenchmark (not product code): does a wave that prefetches its NEXT board into LDS (global_load_lds, no
// registers held) while it computes the current one hide the step kernel's load round trip?
// Per board: header 96 B + planes 1,312 B + armies 896 B read; header + 780 B of planes + armies + 832 B of masks
// written (non-temporal); in between a dependent chain of `work` vector instructions standing in for the turn.
//   A: one board per wave, loads -> compute -> stores (the round-2 step kernel's shape)
//   B: persistent waves (grid = resident workgroups), board k+1 streams into LDS while board k is computed
//   hipcc --offload-arch=gfx950 -O3 -o prefetch_pipeline prefetch_pipeline.hip && ./prefetch_pipeline
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct Args { uint32_t *hdr, *rows, *army, *mask; int n, work; };
typedef __attribute__((address_space(3))) uint32_t* lds_ptr;
typedef __attribute__((address_space(1))) const uint32_t* gptr;

__device__ __forceinline__ void st(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ uint32_t fake_turn(uint32_t h, const uint32_t (&p)[7], const uint32_t (&q)[4], int work) {
  uint32_t x = h;
  for (int k2 = 0; k2 < 7; ++k2) x ^= p[k2];
  for (int j = 0; j < 4; ++j) x += q[j];
  for (int i = 0; i < work; ++i) x = x * 0x9E3779B1u + (x >> 7);  // 3 dependent vector instructions per round (one quarter rate)
  return x;
}

__device__ __forceinline__ void store_board(const Args& a, int env, int lane, uint32_t h, const uint32_t (&p)[7], const uint32_t (&q)[4], uint32_t x) {
  const int r = lane >> 4, i = lane & 15;
  uint32_t* hdr_o = a.hdr + (size_t)env * 24;
  uint32_t* rows_o = a.rows + (size_t)env * 328;
  uint32_t* army_o = a.army + (size_t)env * 224;
  uint32_t* mask = a.mask + (size_t)env * 208;
  if (lane < 24) st(hdr_o + lane, h + 1u);
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2)
    if (i < 13 && 4 * k2 + r < 15) st(rows_o + (4 * k2 + r) * 13 + i, p[k2] ^ x);
#pragma unroll
  for (int j = 0; j < 4; ++j) if (64 * j + lane < 224) st(army_o + 64 * j + lane, q[j] + x);
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) if (lane < 52) st(mask + 52 * k2 + lane, p[k2] + q[k2]);
}

__global__ __launch_bounds__(256, 8) void plain(Args a) {
  const int env = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int lane = threadIdx.x & 63;
  if (env >= a.n) return;
  const uint32_t* hdr = a.hdr + (size_t)env * 24;
  const uint32_t* rows = a.rows + (size_t)env * 328;
  const uint32_t* army = a.army + (size_t)env * 224;
  const int r = lane >> 4, i = lane & 15;
  uint32_t p[7], q[4];
  const uint32_t h = lane < 24 ? hdr[lane] : 0u;
#pragma unroll
  for (int k2 = 0; k2 < 7; ++k2) p[k2] = (i < 13 && 4 * k2 + r < 25) ? rows[(4 * k2 + r) * 13 + i] : 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = (64 * j + lane < 224) ? army[64 * j + lane] : 0u;
  const uint32_t x = fake_turn(h, p, q, a.work);
  store_board(a, env, lane, h, p, q, x);
}

constexpr int STAGE_DW = 32 + 328 + 224;  // header (padded) | planes | armies

__device__ __forceinline__ void prefetch(const Args& a, int env, uint32_t* stage, int lane) {
  const uint32_t* hdr = a.hdr + (size_t)env * 24;
  const uint32_t* rows = a.rows + (size_t)env * 328;
  const uint32_t* army = a.army + (size_t)env * 224;
  if (lane < 6) __builtin_amdgcn_global_load_lds((gptr)(hdr + lane * 4), (lds_ptr)stage, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr)(rows + lane * 4), (lds_ptr)(stage + 32), 16, 0, 0);
  if (lane < 18) __builtin_amdgcn_global_load_lds((gptr)(rows + 256 + lane * 4), (lds_ptr)(stage + 32 + 256), 16, 0, 0);
  if (lane < 56) __builtin_amdgcn_global_load_lds((gptr)(army + lane * 4), (lds_ptr)(stage + 32 + 328), 16, 0, 0);
}

__global__ __launch_bounds__(256, 8) void piped(Args a) {
  __shared__ uint32_t stage_all[4][STAGE_DW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nw = (int)gridDim.x * 4;
  int env = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
  if (env >= a.n) return;
  uint32_t* stage = stage_all[wave];
  const int r = lane >> 4, i = lane & 15;
  prefetch(a, env, stage, lane);
  for (; env < a.n; env += nw) {
    uint32_t p[7], q[4];
    const uint32_t h = lane < 24 ? stage[lane] : 0u;
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) p[k2] = (i < 13 && 4 * k2 + r < 25) ? stage[32 + (4 * k2 + r) * 13 + i] : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = (64 * j + lane < 224) ? stage[32 + 328 + 64 * j + lane] : 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage is free again
    if (env + nw < a.n) prefetch(a, env + nw, stage, lane);
    const uint32_t x = fake_turn(h, p, q, a.work);
    store_board(a, env, lane, h, p, q, x);
  }
}

// C: one board per wave like A, but the loads go through LDS (4 global_load_lds instructions, then ds_reads)
__global__ __launch_bounds__(256, 8) void staged(Args a) {
  __shared__ uint32_t stage_all[4][STAGE_DW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int env = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
  if (env >= a.n) return;
  uint32_t* stage = stage_all[wave];
  const int r = lane >> 4, i = lane & 15;
  prefetch(a, env, stage, lane);
  uint32_t p[7], q[4];
  const uint32_t h = lane < 24 ? stage[lane] : 0u;
#pragma unroll
  for (int k2 = 0; k2 < 7; ++k2) p[k2] = (i < 13 && 4 * k2 + r < 25) ? stage[32 + (4 * k2 + r) * 13 + i] : 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = (64 * j + lane < 224) ? stage[32 + 328 + 64 * j + lane] : 0u;
  const uint32_t x = fake_turn(h, p, q, a.work);
  store_board(a, env, lane, h, p, q, x);
}

template <typename K>
void run(K kern, const Args& a, int grid, const char* what) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double us = best / 20 * 1e3;
  printf("%-64s work %4d: %7.1f us  %5.0f GB/s\n", what, a.work, us, a.n * 4896.0 / us / 1e3);
}

int main() {
  const int n = 262144;
  Args a; a.n = n;
  uint32_t** bufs[] = {&a.hdr, &a.rows, &a.army, &a.mask};
  const size_t sz[] = {96, 1312, 896, 832};
  for (int b = 0; b < 4; ++b) { (void)hipMalloc(bufs[b], n * sz[b]); (void)hipMemset(*bufs[b], 0, n * sz[b]); }
  for (int work : {0, 100}) {
    a.work = work;
    run(plain, a, n / 4, "A one board per wave");
    run(staged, a, n / 4, "C one board per wave, loads through LDS (4 global_load_lds)");
    run(piped, a, 2048, "B persistent, next board prefetched into LDS (2048 workgroups)");
    run(piped, a, 4096, "B persistent, prefetched (4096 workgroups)");
    run(piped, a, 16384, "B 4 boards per wave, prefetched (16384 workgroups)");
  }
  return 0;
}
