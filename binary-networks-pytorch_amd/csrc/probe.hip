// probe.hip — integer-ALU roofline calibration for the bconv main loop.
// Register-only instruction streams (inline asm, so the compiler can neither fold nor
// re-associate them) run on every CU at full occupancy; the sustained 32-bit lane-ops/s is
// what bench.py prints next to the theoretical peak, so the roofline denominator is
// evidenced, not assumed.
//
//   mode 0  v_bitop3_b32 (scalar weight operand) + v_bcnt_u32_b32   <- the hot loop's pair
//   mode 1  v_xor_b32 (scalar operand)           + v_bcnt_u32_b32   <- plain XNOR pair
//   mode 2  v_bcnt_u32_b32 only
//   mode 3  v_bitop3_b32 only
//   mode 4  v_xor_b32 only
//   mode 5  v_fma_f32 only (reference point: the guide's 2-cycle wave64 FMA)
//   mode 6  v_add_u32 only
//   mode 7  v_and_b32 with VGPR-only operands
//   mode 8  v_and_b32 (VGPR-only) + v_bcnt_u32_b32
//   mode 9  v_xor_b32 with VGPR-only operands
//   mode 10 v_and_b32 (scalar operand) + v_bcnt_u32_b32           <- the non-negative kernels' pair
#include "bnn_dev.h"

namespace bnn {

constexpr int kProbeRegs = 16;

template <int MODE>
__global__ __launch_bounds__(256) void probe_kernel(int iters, uint32_t seed,
                                                    uint32_t* __restrict__ sink) {
  uint32_t p[kProbeRegs], m[kProbeRegs], a[8], t[8];
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < kProbeRegs; ++i) {
    p[i] = (tid * 2654435761u) ^ (0x9e3779b9u * (i + 1));
    m[i] = ~p[i] & ((tid + i) * 40503u);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = i; t[i] = tid + i; }
  uint32_t w = seed;  // wave-uniform -> SGPR, like a streamed weight word
#if defined(__HIP_DEVICE_COMPILE__)  // gfx950 asm: the x86 host pass must not see the constraints
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < kProbeRegs; ++i) {
      const int k = i & 7;
      if (MODE == 0) {
        asm volatile("v_bitop3_b32 %0, %2, %3, %4 bitop3:0xe4\n\tv_bcnt_u32_b32 %1, %0, %1"
                     : "=&v"(t[k]), "+v"(a[k]) : "v"(m[i]), "v"(p[i]), "s"(w));
      } else if (MODE == 1) {
        asm volatile("v_xor_b32 %0, %3, %2\n\tv_bcnt_u32_b32 %1, %0, %1"
                     : "=&v"(t[k]), "+v"(a[k]) : "v"(p[i]), "s"(w));
      } else if (MODE == 2) {
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(p[i]));
      } else if (MODE == 3) {
        asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(t[k]) : "v"(m[i]), "v"(p[i]), "s"(w));
      } else if (MODE == 4) {
        asm volatile("v_xor_b32 %0, %2, %1" : "=v"(t[k]) : "v"(p[i]), "s"(w));
      } else if (MODE == 5) {
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(p[i]), "v"(m[i]));
      } else if (MODE == 6) {
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[k]) : "v"(p[i]));
      } else if (MODE == 7) {
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(t[k]) : "v"(p[i]), "v"(m[(i + 3) & 15]));
      } else if (MODE == 8) {
        asm volatile("v_and_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1"
                     : "=&v"(t[k]), "+v"(a[k]) : "v"(p[i]), "v"(m[(i + 3) & 15]));
      } else if (MODE == 9) {
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[k]) : "v"(p[i]), "v"(m[(i + 3) & 15]));
      } else {
        asm volatile("v_and_b32 %0, %3, %2\n\tv_bcnt_u32_b32 %1, %0, %1"
                     : "=&v"(t[k]), "+v"(a[k]) : "v"(p[i]), "s"(w));
      }
    }
    w = w * 1664525u + 1013904223u;
  }
#endif
  uint32_t r = w;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i] + t[i];
  sink[tid] = r;
}

template <int MODE>
static void launch_mode(int blocks, int iters, uint32_t seed, uint32_t* sink, hipStream_t s) {
  hipLaunchKernelGGL(probe_kernel<MODE>, dim3(blocks), dim3(256), 0, s, iters, seed, sink);
}

static void launch_any(int mode, int blocks, int iters, uint32_t seed, uint32_t* sink, hipStream_t s) {
  switch (mode) {
    case 0: launch_mode<0>(blocks, iters, seed, sink, s); break;
    case 1: launch_mode<1>(blocks, iters, seed, sink, s); break;
    case 2: launch_mode<2>(blocks, iters, seed, sink, s); break;
    case 3: launch_mode<3>(blocks, iters, seed, sink, s); break;
    case 4: launch_mode<4>(blocks, iters, seed, sink, s); break;
    case 5: launch_mode<5>(blocks, iters, seed, sink, s); break;
    case 6: launch_mode<6>(blocks, iters, seed, sink, s); break;
    case 7: launch_mode<7>(blocks, iters, seed, sink, s); break;
    case 8: launch_mode<8>(blocks, iters, seed, sink, s); break;
    case 9: launch_mode<9>(blocks, iters, seed, sink, s); break;
    default: launch_mode<10>(blocks, iters, seed, sink, s); break;
  }
}

// ops per lane per loop step for each mode
static int ops_per_step(int mode) { return (mode == 0 || mode == 1 || mode == 8 || mode == 10) ? 2 : 1; }

int launch_probe_int_alu(int mode, int iters, double* lane_ops_per_s, double* elapsed_ms, hipStream_t s) {
  if (mode < 0 || mode > 10) return BNN_HIP_ERR_INVALID_ARG;
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
  launch_any(mode, blocks, 8, 1u, sink, s);  // warm-up
  (void)hipEventRecord(e0, s);
  launch_any(mode, blocks, iters, 12345u, sink, s);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const int st = hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  const double ops = (double)ops_per_step(mode) * kProbeRegs * (double)iters * (double)blocks * 256.0;
  *lane_ops_per_s = ops / (ms * 1e-3);
  if (elapsed_ms) *elapsed_ms = ms;
  return st;
}

// Engine (shader) clock right now: one wave reads the shader-clock counter (s_memtime) and the constant-rate
// counter (s_memrealtime, hipDeviceAttributeWallClockRate kHz) around a dependent ALU chain; their ratio is the
// clock the SIMDs ran at.  bench.py calls it straight after a timed region, on the same stream: DVFS moves on a
// millisecond scale, so the ~30 us sample is the clock the preceding kernels had.
__global__ __launch_bounds__(64) void clock_probe_kernel(int iters, unsigned long long* out) {
  uint32_t a = threadIdx.x;
  const unsigned long long c0 = clock64(), r0 = wall_clock64();
#if defined(__HIP_DEVICE_COMPILE__)
  for (int i = 0; i < iters; ++i) asm volatile("v_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0" : "+v"(a));
#endif
  const unsigned long long c1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = a; }
}

int launch_probe_clock(int spin_iters, double* shader_mhz, double* elapsed_us, hipStream_t s) {
  int dev = 0, wall_khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return BNN_HIP_ERR_NO_DEVICE;
  if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || wall_khz <= 0)
    return BNN_HIP_ERR_UNSUPPORTED;
  unsigned long long* out = nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&out), 3 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess)
    return BNN_HIP_ERR_LAUNCH;
  out[0] = out[1] = 0;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s, spin_iters, out);
  int st = (hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess) ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
  if (st == BNN_HIP_OK && out[1] == 0) st = BNN_HIP_ERR_UNSUPPORTED;
  if (st == BNN_HIP_OK) {
    *shader_mhz = (double)out[0] / (double)out[1] * (double)wall_khz * 1e-3;
    if (elapsed_us) *elapsed_us = (double)out[1] / (double)wall_khz * 1e3;
  }
  (void)hipHostFree(out);
  return st;
}

}  // namespace bnn
