// Hardware question (gfx950): does a raw buffer load's range check include the scalar offset?
// SRD: base = p, num_records = 52 bytes; lane l loads voffset = 4l with soffset = 52k.
// Prints, per k, how many of lanes 0..12 returned the right dword and how many of lanes 13..63 returned 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* p, uint32_t* out, int plane_bytes) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(p, (short)0, 52, 0x00020000);
  const int lane = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q * 64 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, plane_bytes * q, 0);
}
int main() {
  uint32_t h[256], *d, *o, r[256];
  for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
  (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&o, sizeof r);
  (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 52);
  (void)hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  for (int q = 0; q < 4; ++q) {
    int ok = 0, zero = 0;
    for (int l = 0; l < 13; ++l) ok += r[q * 64 + l] == 1000u + 13 * q + l;
    for (int l = 13; l < 64; ++l) zero += r[q * 64 + l] == 0u;
    printf("soffset %3d: %2d/13 in-range lanes correct, %2d/51 out-of-range lanes zero (lane 13 -> %u)\n", 52 * q, ok, zero, r[q * 64 + 13]);
  }
  return 0;
}
