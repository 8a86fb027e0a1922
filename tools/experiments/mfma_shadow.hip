// mfma_shadow.hip — how much VALU / LDS work fits "under" v_mfma_f32_16x16x32_f16 on gfx950?
//   (1) one wave: every MFMA followed by K independent v_add_u32 (K = 0..6)
//   (2) two waves per SIMD: one issues only MFMAs, the other only v_add_u32
//   (3) one wave: every MFMA followed by K ds_read_b32 / ds_read2_b32 / ds_read_b64 (conflict-free)
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_shadow.hip -o /tmp/ms && /tmp/ms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int ITERS = 4096;

// MODE 0: K VALU per MFMA in the same wave.  MODE 1: waves 0-3 MFMA only, waves 4-7 VALU only (K per "slot").
// MODE 2/3/4: K ds_read_b32 / ds_read2_b32 / ds_read_b64 per MFMA in the same wave.  MF: with / without the MFMAs.
template <int MODE, int K, int MF>
__global__ __launch_bounds__(512) void shadow(float* sink) {
  __shared__ uint64_t lds[4096];
  const int tid = threadIdx.x, wave = tid >> 6;
  lds[tid] = tid;
  lds[tid + 512] = tid;
  __syncthreads();
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(tid * 0.001f + e); b[e] = (_Float16)(e - tid * 0.002f); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  uint32_t v[8];
  for (int i = 0; i < 8; ++i) v[i] = tid + i;
  uint32_t addr = (tid & 63) * 8;  // conflict-free for b32 and b64
  uint64_t q = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  if (MODE == 1 && wave >= 4) {  // VALU-only waves (waves 0-3 and 4-7 of a workgroup each cover the 4 SIMDs once)
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int k = 0; k < 4 * K; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(addr));
    }
  } else {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (MF) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
        if (MODE == 0) {
#pragma unroll
          for (int k = 0; k < K; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(addr));
        } else if (MODE >= 2) {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            if (MODE == 2) { uint32_t r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(addr)); v[k & 7] ^= r; }
            if (MODE == 3) { uint64_t r; asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(r) : "v"(addr)); q ^= r; }
            if (MODE == 4) { uint64_t r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(addr)); q ^= r; }
          }
        }
      }
      if (MODE >= 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
  }
#endif
  float s = (float)q;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  sink[blockIdx.x * 512 + tid] = s;
}

template <int MODE, int K, int MF>
static double run(float* sink, int waves_per_simd) {
  const int threads = 64 * 4 * waves_per_simd;  // one workgroup per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {  // the first timed region after idle runs at ramping clocks: keep the best
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((shadow<MODE, K, MF>), dim3(256), dim3(threads), 0, 0, sink);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((shadow<MODE, K, MF>), dim3(256), dim3(threads), 0, 0, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms / 10 * 1e3 < best) best = ms / 10 * 1e3;
  }
  return best;  // us per launch
}

int main() {
  float* sink; hipMalloc(&sink, 256 * 512 * 4);
  for (int w = 0; w < 300; ++w) hipLaunchKernelGGL((shadow<0, 6, 1>), dim3(256), dim3(512), 0, 0, sink);  // warm clocks
  hipDeviceSynchronize();
  const double mf = 4.0 * ITERS;  // MFMAs per wave
  printf("cycles per MFMA slot at 2.4 GHz nominal (us * 2400 / MFMAs per wave)\n");
#define ROW(MODE, K, WPS, MF) { double us = run<MODE, K, MF>(sink, WPS); printf("  mode %d K=%d waves/SIMD=%d mfma=%d : %8.1f us  %6.2f cyc/slot\n", MODE, K, WPS, MF, us, us * 2400 / mf); }
  printf("(1) one wave per SIMD, K v_add_u32 after every MFMA\n");
  ROW(0, 0, 1, 1) ROW(0, 1, 1, 1) ROW(0, 2, 1, 1) ROW(0, 3, 1, 1) ROW(0, 4, 1, 1) ROW(0, 6, 1, 1)
  printf("    the same VALU work without the MFMAs\n");
  ROW(0, 3, 1, 0) ROW(0, 6, 1, 0)
  printf("    two waves per SIMD, both doing MFMA + K VALU\n");
  ROW(0, 0, 2, 1) ROW(0, 3, 2, 1) ROW(0, 6, 2, 1)
  printf("(2) two waves per SIMD: one MFMA only, the other K v_add_u32 per slot\n");
  ROW(1, 0, 2, 1) ROW(1, 3, 2, 1) ROW(1, 4, 2, 1) ROW(1, 6, 2, 1)
  printf("(3) one wave per SIMD, K LDS reads after every MFMA (+ s_waitcnt lgkmcnt(0))\n");
  ROW(2, 1, 1, 1) ROW(2, 2, 1, 1) ROW(3, 1, 1, 1) ROW(3, 2, 1, 1) ROW(4, 1, 1, 1) ROW(4, 2, 1, 1)
  printf("    two waves per SIMD\n");
  ROW(2, 2, 2, 1) ROW(3, 2, 2, 1) ROW(4, 2, 2, 1)
  printf("    LDS reads only (no MFMA), two waves per SIMD\n");
  ROW(2, 2, 2, 0) ROW(3, 2, 2, 0) ROW(4, 2, 2, 0)
  return 0;
}
