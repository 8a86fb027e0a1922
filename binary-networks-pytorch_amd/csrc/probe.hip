// probe.hip — integer-ALU roofline calibration for the bconv main loop.
// Runs the exact instruction pair of the hot loop (v_bitop3_b32 with a scalar weight
// operand + accumulating v_bcnt_u32_b32) from registers only, on every CU, and reports
// sustained 32-bit lane-ops/s.  bench.py prints this next to the theoretical peak
// (CUs x 4 SIMD x 32 lanes x clock) so the roofline denominator is evidenced, not assumed.
#include "bnn_dev.h"

namespace bnn {

constexpr int kProbeRegs = 16;

__global__ __launch_bounds__(256) void probe_int_alu_kernel(int iters, uint32_t seed,
                                                            uint32_t* __restrict__ sink) {
  uint32_t p[kProbeRegs], m[kProbeRegs];
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < kProbeRegs; ++i) {
    p[i] = (t * 2654435761u) ^ (0x9e3779b9u * (i + 1));
    m[i] = ~p[i] & ((t + i) * 40503u);
  }
  uint32_t w = seed;  // wave-uniform -> lives in an SGPR, like a streamed weight word
  int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < kProbeRegs; i += 4) {
      a0 += __builtin_popcount(disagree(w, m[i], p[i]));
      a1 += __builtin_popcount(disagree(w, m[i + 1], p[i + 1]));
      a2 += __builtin_popcount(disagree(w, m[i + 2], p[i + 2]));
      a3 += __builtin_popcount(disagree(w, m[i + 3], p[i + 3]));
    }
    w = w * 1664525u + 1013904223u;
  }
  sink[t] = (uint32_t)(a0 + a1 + a2 + a3);
}

int launch_probe_int_alu(int iters, double* lane_ops_per_s, double* elapsed_ms, hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return BNN_HIP_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return BNN_HIP_ERR_NO_DEVICE;
  const int blocks = prop.multiProcessorCount * 8;  // 32 waves per CU = 8 per SIMD
  uint32_t* sink = nullptr;
  if (hipMalloc(&sink, (size_t)blocks * 256 * sizeof(uint32_t)) != hipSuccess) return BNN_HIP_ERR_LAUNCH;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe_int_alu_kernel, dim3(blocks), dim3(256), 0, s, 8, 1u, sink);  // warm-up
  (void)hipEventRecord(e0, s);
  hipLaunchKernelGGL(probe_int_alu_kernel, dim3(blocks), dim3(256), 0, s, iters, 12345u, sink);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const int st = hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  const double ops = 2.0 * kProbeRegs * (double)iters * (double)blocks * 256.0;
  *lane_ops_per_s = ops / (ms * 1e-3);
  if (elapsed_ms) *elapsed_ms = ms;
  return st;
}

}  // namespace bnn
