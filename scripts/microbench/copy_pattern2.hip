// Microbenchmark (not product code): the round-2 step kernel's HBM access pattern without its compute.
// One wavefront per env: header 96 B (read + write), planes 25 x 52 B read as 7 packed-register loads (lane 16r+i =
// plane 4k+r, dword i) and 15 x 52 B written back as 4 stores, armies as u16 pairs (224 dwords read + written),
// legal masks 4 x 208 B written.  Flags try the same bytes out of place (ping-pong buffers) and with plain stores.
//   hipcc --offload-arch=gfx950 -O3 -o copy_pattern2 copy_pattern2.hip && ./copy_pattern2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct Args { uint32_t *hdr, *rows, *army, *mask, *hdr2, *rows2, *army2, *zeros; int* err; int n; };

template <bool NT> __device__ __forceinline__ void st(uint32_t* p, uint32_t v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// F bits: 1 out of place (stores go to the second buffer set), 2 non-temporal stores, 4 no mask stores, 8 loads only,
// 128 planes loaded the way the step kernel does (3 packed + 13 row-replicated loads, idle lanes pointed at a zero block),
// 256 + four header fields through the scalar cache, 512 + one int per env (err), 1024 + 9 KB of LDS per workgroup
// 2048 XCD-aware env order (blocks b, b+8, ... walk one contiguous eighth), 4096 mask stores the way the step kernel
// issues them (one direction of all four players per instruction: four 52-byte pieces 208 bytes apart),
// 8192 header and planes blocks padded to whole 128-byte lines
// 16 stores only, 32 skip the stores of 2 of the 4 plane registers (quiet turn), 64 army stores only for lanes < 16 (partial lines)
template <int F, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k(Args a) {
  // 16384: blocks walk the envs with a large odd stride (concurrent workgroups far apart in memory);
  // 32768: neighbouring blocks 64 envs apart inside 1,024-env tiles (the 8 XCDs interleave at 16-block granularity)
  int blk = (F & 2048) ? (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if constexpr ((F & 16384) != 0) blk = (int)(((unsigned long long)blockIdx.x * 40503ull) % gridDim.x);
  if constexpr ((F & 32768) != 0) blk = (int)((blockIdx.x & ~255u) | ((blockIdx.x & 15u) << 4) | ((blockIdx.x >> 4) & 15u));
  const int env = __builtin_amdgcn_readfirstlane(blk * 4 + (int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (env >= a.n) return;
  constexpr bool NT = (F & 2) != 0;
  constexpr int HS = (F & 8192) ? 32 : 24, RS = (F & 8192) ? 352 : 328;
  const uint32_t* hdr = a.hdr + (size_t)env * HS;
  const uint32_t* rows = a.rows + (size_t)env * RS;
  const uint32_t* army = a.army + (size_t)env * 224;
  uint32_t* hdr_o = ((F & 1) ? a.hdr2 : a.hdr) + (size_t)env * HS;
  uint32_t* rows_o = ((F & 1) ? a.rows2 : a.rows) + (size_t)env * RS;
  uint32_t* army_o = ((F & 1) ? a.army2 : a.army) + (size_t)env * 224;
  uint32_t* mask = a.mask + (size_t)env * 208;
  const int r = lane >> 4, i = lane & 15;
  uint32_t h = 0, p[7], q[4];
  __shared__ uint32_t lds_pad[(F & 1024) ? 2304 : 1];
  if constexpr ((F & 1024) != 0) lds_pad[threadIdx.x] = threadIdx.x;
  if constexpr (!(F & 16)) {
    h = lane < 24 ? hdr[lane] : 0u;
    if constexpr ((F & 128) != 0) {
      const bool in = i < 13;
      const uint32_t* gs = in ? rows + i : a.zeros;
      const uint32_t* gp = in ? rows + r * 13 + i : a.zeros;
      p[0] = gp[0 * 13]; p[1] = gp[4 * 13]; p[2] = gp[8 * 13];
      uint32_t sh[13];
#pragma unroll
      for (int k2 = 0; k2 < 13; ++k2) sh[k2] = gs[(12 + k2) * 13];
      p[3] = sh[0] ^ sh[1] ^ sh[2]; p[4] = sh[3] + sh[4] + sh[5]; p[5] = sh[6] ^ sh[7] ^ sh[8]; p[6] = sh[9] + sh[10] + sh[11] + sh[12];
    } else {
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) p[k2] = (i < 13 && 4 * k2 + r < 25) ? rows[(4 * k2 + r) * 13 + i] : 0u;
    }
    if constexpr ((F & 256) != 0) {
      typedef const __attribute__((address_space(4))) uint32_t* kptr;
      kptr kk = (kptr)hdr;
      h += kk[0] + kk[1] + kk[2] + kk[20];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = (64 * j + lane < 224) ? army[64 * j + lane] : 0u;
  } else {
    h = lane;
    for (int k2 = 0; k2 < 7; ++k2) p[k2] = lane * k2;
    for (int j = 0; j < 4; ++j) q[j] = lane + j;
  }
  if constexpr (F & 8) {
    uint32_t acc = h;
    for (int k2 = 0; k2 < 7; ++k2) acc += p[k2];
    for (int j = 0; j < 4; ++j) acc += q[j];
    if (acc == 0x12345678u) a.hdr2[0] = acc;
  } else {
    if constexpr ((F & 1024) != 0) h += lds_pad[(threadIdx.x + 64) & 255];
    if constexpr ((F & 512) != 0) if (lane == 0) a.err[env] = (int)h;
    if (lane < 24) st<NT>(hdr_o + lane, h + 1u);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
      if ((F & 32) && (k2 == 0 || k2 == 2)) continue;
      if (i < 13 && 4 * k2 + r < 15) st<NT>(rows_o + (4 * k2 + r) * 13 + i, p[k2] ^ p[6]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (64 * j + lane < 224 && (!(F & 64) || (lane & 15) < 4)) st<NT>(army_o + 64 * j + lane, q[j] + 1u);
    if constexpr (!(F & 4)) {
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        if constexpr ((F & 4096) != 0) { if (i < 13) st<NT>(mask + r * 52 + k2 * 13 + i, p[k2] + q[k2]); }
        else if (lane < 52) st<NT>(mask + 52 * k2 + lane, p[k2] + q[k2]);
      }
    }
  }
}

// the same bytes as 16-byte-per-lane accesses of consecutive lanes (LOADS: also the loads; else the loads as in k<2>)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// cache-policy bits on a 16-byte store: 0 nt (the builtin), 1 none, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 nt sc0 sc1, 6 sc0
template <int POL> __device__ __forceinline__ void stx4(u32x4* p, u32x4 v) {
  if constexpr (POL == 0) __builtin_nontemporal_store(v, p);
  else if constexpr (POL == 1) *p = v;
  else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  else if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off nt sc1" :: "v"(p), "v"(v) : "memory");
  else if constexpr (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
template <bool LOADS, int POL = 0, bool OUT = false>
__global__ __launch_bounds__(256, 8) void wide(Args a) {
  const int env = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int lane = threadIdx.x & 63;
  if (env >= a.n) return;
  u32x4* hdr = reinterpret_cast<u32x4*>(a.hdr + (size_t)env * 24);
  u32x4* rows = reinterpret_cast<u32x4*>(a.rows + (size_t)env * 328);
  u32x4* army = reinterpret_cast<u32x4*>(a.army + (size_t)env * 224);
  u32x4* mask = reinterpret_cast<u32x4*>(a.mask + (size_t)env * 208);
  u32x4 h = {0, 0, 0, 0}, r0 = h, r1 = h, q = h;
  if constexpr (LOADS) {
    if (lane < 6) h = hdr[lane];
    r0 = rows[lane];
    if (lane < 18) r1 = rows[64 + lane];
    if (lane < 56) q = army[lane];
  } else {
    const uint32_t* hd = a.hdr + (size_t)env * 24;
    const uint32_t* rw = a.rows + (size_t)env * 328;
    const uint32_t* ar = a.army + (size_t)env * 224;
    const int r = lane >> 4, i = lane & 15;
    h.x = lane < 24 ? hd[lane] : 0u;
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) r0[k2 & 3] += (i < 13 && 4 * k2 + r < 25) ? rw[(4 * k2 + r) * 13 + i] : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = (64 * j + lane < 224) ? ar[64 * j + lane] : 0u;
    r1 = r0;
  }
  h.x += 1u; r0.y ^= r1.x; q.z += 1u;
  if constexpr (LOADS) { if (lane < 6) __builtin_nontemporal_store(h, hdr + lane); }
  else if (lane < 24) __builtin_nontemporal_store(h.x, (OUT ? a.hdr2 : a.hdr) + (size_t)env * 24 + lane);
  if constexpr (OUT) {  // ping-pong: the state goes to the second buffer set
    hdr = reinterpret_cast<u32x4*>(a.hdr2 + (size_t)env * 24);
    rows = reinterpret_cast<u32x4*>(a.rows2 + (size_t)env * 328);
    army = reinterpret_cast<u32x4*>(a.army2 + (size_t)env * 224);
  }
  if (lane < 48) stx4<POL>(rows + lane, r0);
  if (lane < 3) __builtin_nontemporal_store(r1.x, (OUT ? a.rows2 : a.rows) + (size_t)env * 328 + 192 + lane);
  if (lane < 56) stx4<POL>(army + lane, q);
  if (lane < 52) stx4<POL>(mask + lane, r0 + q);
}
template <bool LOADS, int POL = 0, bool OUT = false>
void run_wide(const Args& a, const char* what) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    for (int it = 0; it < 40; ++it) hipLaunchKernelGGL((wide<LOADS, POL, OUT>), dim3(a.n / 4), dim3(256), 0, 0, a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double us = best / 40 * 1e3;
  printf("%-66s %7.1f us  %5.0f GB/s  (%4.0f B read + %4.0f B written per env)\n", what, us, a.n * 4896.0 / us / 1e3, 2292.0, 2604.0);
}

template <int F, int WAVES>
void run(const Args& a, const char* what) {
  double rd = 96 + 1300 + 896, wr = 96 + 780 + 896 + 832;
  if (F & 4) wr -= 832;
  if (F & 32) wr -= 416;
  if (F & 64) wr -= 896 * 0.75;
  if (F & 8) wr = 0;
  if (F & 16) rd = 0;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    for (int it = 0; it < 40; ++it) hipLaunchKernelGGL((k<F, WAVES>), dim3(a.n / 4), dim3(256), 0, 0, a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double us = best / 40 * 1e3;
  printf("%-66s %7.1f us  %5.0f GB/s  (%4.0f B read + %4.0f B written per env)\n", what, us, a.n * (rd + wr) / us / 1e3, rd, wr);
}

int main() {
  const int n = 262144;
  Args a; a.n = n;
  uint32_t** bufs[] = {&a.hdr, &a.rows, &a.army, &a.mask, &a.hdr2, &a.rows2, &a.army2, &a.zeros, (uint32_t**)&a.err};
  const size_t sz[] = {128, 1408, 896, 832, 128, 1408, 896, 1, 4};
  for (int b = 0; b < 9; ++b) { (void)hipMalloc(bufs[b], n * sz[b]); (void)hipMemset(*bufs[b], 0, n * sz[b]); }
  run<2, 8>(a, "in place, non-temporal stores (the step kernel's pattern)");
  run_wide<false>(a, "stores as 16-byte-per-lane chunks (the step kernel now), dword loads");
  run_wide<true>(a, "loads AND stores as 16-byte-per-lane chunks");
  run_wide<false, 1>(a, "wide stores, no policy bits");
  run_wide<false, 2>(a, "wide stores, sc1");
  run_wide<false, 3>(a, "wide stores, sc0 sc1");
  run_wide<false, 4>(a, "wide stores, nt sc1");
  run_wide<false, 5>(a, "wide stores, nt sc0 sc1");
  run_wide<false, 6>(a, "wide stores, sc0");
  run_wide<false, 4, true>(a, "wide stores, nt sc1, OUT OF PLACE (ping-pong)");
  run_wide<false, 4>(a, "wide stores, nt sc1 (again)");
  run<2 | 128, 8>(a, "  + planes as 3 packed + 13 replicated loads, zero block");
  run<2 | 128 | 256, 8>(a, "  + header fields through the scalar cache");
  run<2 | 128 | 256 | 512, 8>(a, "  + err store");
  run<2 | 128 | 256 | 512 | 1024, 8>(a, "  + 9 KB of LDS per workgroup");
  run<2 | 4 | 128 | 256 | 512 | 1024, 8>(a, "  ... without mask stores (the frozen-board step)");
  run<2 | 2048, 8>(a, "the first line, XCD-aware env order");
  run<2 | 128 | 256 | 512 | 1024 | 2048, 8>(a, "step-kernel-like, XCD-aware env order");
  run<2 | 4096, 8>(a, "the first line, mask stores in four 52-byte pieces per instruction");
  run<2 | 2048 | 4096, 8>(a, "  ... with XCD-aware env order");
  run<2 | 8192, 8>(a, "the first line, header and planes blocks padded to 128-byte lines");
  run<2 | 2048 | 8192, 8>(a, "  ... with XCD-aware env order");
  run<2 | 16384, 8>(a, "the first line, blocks walk the envs with a large odd stride");
  run<2 | 32768, 8>(a, "the first line, 16x16 transposed block order inside 1,024-env tiles");
  run<0, 8>(a, "in place, plain stores");
  run<3, 8>(a, "out of place (ping-pong), non-temporal stores");
  run<1, 8>(a, "out of place, plain stores");
  run<2, 4>(a, "in place, nt, 4 waves/SIMD");
  run<2, 2>(a, "in place, nt, 2 waves/SIMD");
  run<2 | 8, 8>(a, "loads only");
  run<2 | 16, 8>(a, "stores only, nt");
  run<16, 8>(a, "stores only, plain");
  run<2 | 4, 8>(a, "in place, nt, no mask stores");
  run<2 | 32, 8>(a, "in place, nt, two plane registers not stored (quiet turn)");
  run<2 | 64, 8>(a, "in place, nt, army stores on a quarter of the lanes (partial lines)");
  return 0;
}
