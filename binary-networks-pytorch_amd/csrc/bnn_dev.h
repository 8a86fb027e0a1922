// bnn_dev.h — device-side helpers shared by the gfx950 kernels.
// Hand-written for CDNA4 (wave64, v_bitop3_b32, v_bcnt_u32_b32); no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/bnn_hip.h"

namespace bnn {

constexpr int kWave = 64;          // CDNA wavefront
// hipFuncAttributeMaxDynamicSharedMemorySize for every kernel that can ask for more than 64 KB of dynamic LDS: ONE
// constant (the CU's whole LDS), never a launch's own size — a per-launch value set from two threads could be lowered
// between one thread's set and its launch.
constexpr int kMaxDynamicLds = 160 * 1024;
constexpr int kOCB = BNN_HIP_OCB;  // output channels per weight block

// Disagreement word of 32 ternary activations against 32 binary weights.
//   w bit = 1 (weight +1): disagree where the activation is negative  -> take M
//   w bit = 0 (weight -1): disagree where the activation is positive  -> take P
// The expression is matched by hipcc to ONE v_bitop3_b32 (truth table 0xe4) on gfx950.
__device__ __forceinline__ uint32_t disagree(uint32_t w, uint32_t m, uint32_t p) {
  return (w & m) | (~w & p);
}

// v_cmp_class_f32 masks: bit0 sNaN,1 qNaN,2 -inf,3 -normal,4 -denorm,5 -0,6 +0,7 +denorm,8 +normal,9 +inf.
// Using the class test (not a float compare) makes the planes independent of the
// kernel's denormal mode and maps NaN to "zero", exactly like torch.sign on CPU.
constexpr int kClassPos = 0x380;  // +denorm | +normal | +inf
constexpr int kClassNeg = 0x01C;  // -inf | -normal | -denorm

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ bool is_pos(float x) { return __builtin_amdgcn_classf(x, kClassPos); }
__device__ __forceinline__ bool is_neg(float x) { return __builtin_amdgcn_classf(x, kClassNeg); }
#else  // host pass of hipcc only parses these; the amdgcn builtin does not exist for x86
__device__ __forceinline__ bool is_pos(float) { return false; }
__device__ __forceinline__ bool is_neg(float) { return false; }
#endif

template <int N>
struct WordVec;
template <>
struct WordVec<1> { using type = uint32_t; };
template <>
struct WordVec<2> { using type = uint2; };
template <>
struct WordVec<4> { using type = uint4; };

// Everything one binary-convolution launch needs (device pointers + geometry + epilogue).
struct ConvP {
  const uint32_t* P;
  const uint32_t* M;
  const uint32_t* W;
  const uint32_t* Z;
  // epilogue (see bnn_hip_epilogue in include/bnn_hip.h)
  const float* alpha;
  const float* bias;
  const float* scale;
  const float* bn_a;
  const float* bn_b;
  const float* prelu;
  const float* res;
  void* out;       // fp32 NCHW, or int32 NCHW when raw
  uint32_t* outP;  // packed sign planes of the output, or null
  uint32_t* outM;
  // generalised epilogue (BNN_HIP_EPI_*): next layer's BN before packing, residual after the
  // activation, packing the pre-residual value, channel slice of a wider output tensor
  const float* pack_a;
  const float* pack_b;
  const int32_t* thr;  // sign thresholds {T, flip word of the block} per channel (BN + ReLU -> packed-only epilogue), or null
  int eflags;
  int c_off, c_tot;
  bool raw;
  bool relu;
  int N, H, Wd, Ho, Wo, O;
  int KH, KW, sh, sw, ph, pw, dh, dw;
  int cw32, cwc, nchunk;
  int npix;  // N*Ho*Wo
  int C;     // input channels
  // shortcut convolution folded into this one (bconv_core.h ShortcutArgs): 1x1 / stride 1 over ds_C channels at the
  // OUTPUT resolution, P plane only; null = none
  const uint32_t* ds_P;
  const uint32_t* ds_W;
  const float* ds_alpha;
  const float* ds_a;
  const float* ds_b;
  int ds_C;
  int ds_inH, ds_inW;  // 0: ds_P is at the output resolution; else ds_P is un-pooled [N, C/64, ds_inH, ds_inW] and the kernel
                   // ORs the 2 x 2 window itself (ceil(ds_inH / 2) == Ho, ceil(ds_inW / 2) == Wo)
};

// Compute units of the current device (for grids of persistent workgroups).  Cached per device in atomics: a
// launch must not pay hipGetDeviceProperties (a slow host call in eager mode), and the C-ABI promises re-entrancy.
inline int current_device_cus() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  const bool cacheable = dev >= 0 && dev < 64;
  if (cacheable) {
    const int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
  if (cacheable) cache[dev].store(cus, std::memory_order_relaxed);
  return cus;
}

// host-side launchers (one per .hip file); return a bnn_hip_status
int choose_cwc(int cw32, int KH, int KW);
int launch_sign_thresholds(const float* alpha, const float* bias, const float* scale, const float* bn_a,
                           const float* bn_b, int O, int kmax, int32_t* thr, hipStream_t stream);
int launch_pack_act(const float* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M,
                    hipStream_t stream);
int launch_pack_act_f16(const void* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M,
                        hipStream_t stream);
int launch_bn_act_pack(const float* x, int N, int C, int H, int W, const float* bn_a, const float* bn_b,
                       int relu, uint64_t* P, uint64_t* M, hipStream_t stream);
int launch_avgpool_pack(const float* x, int N, int C, int H, int W, int k, uint64_t* P, uint64_t* M,
                        hipStream_t stream);
int launch_avgpool2_bn_pack2(const float* x, int N, int C, int H, int W, const float* a1, const float* b1, int relu1,
                             uint64_t* P1, uint64_t* M1, const float* a2, const float* b2, int relu2, uint64_t* P2,
                             uint64_t* M2, float* out, hipStream_t stream);
int launch_orpool_packed(const uint64_t* P, int N, int C, int H, int W, int k, uint64_t* outP, uint64_t* outM,
                         hipStream_t stream);
int launch_bn_relu_maxpool_pack(const float* x, int N, int C, int H, int W, const float* bn_a,
                                const float* bn_b, int relu, int k, int stride, int pad, float* out,
                                uint64_t* P, uint64_t* M, hipStream_t stream);
int launch_stem(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
                int W, int flags, float* out, uint64_t* P, uint64_t* M, hipStream_t stream);
// csrc/legacy/ (libbnn_hip_legacy.so, test-only): the round-2 stem kernel and the LDS-staged weight tile
int launch_stem_split(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
                      int W, int half, float* out, uint64_t* P, uint64_t* M, hipStream_t stream);
int launch_bconv_lds(const ConvP& p, hipStream_t s);
int launch_stem_rows(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
                      int W, int half, float* out, uint64_t* P, uint64_t* M, hipStream_t stream);
int launch_stem_conv(const float* x, const float* w, int N, int H, int W, int half, float* out, hipStream_t stream);
int launch_stem_rows_aff(const float* x, const float* w, const float* bn_a, const float* bn_b, const float* pk_a,
                         const float* pk_b, int N, int H, int W, int half, float* out, uint64_t* P, uint64_t* M,
                         hipStream_t stream);
// stem_wgrad.hip: weight gradient of the stem convolution (fp32 MFMA; partial slabs in `work`, reduced in index order)
bool stem_wgrad_supported(int H, int W);
size_t stem_wgrad_workspace_bytes(int N, int H, int W);
int launch_stem_wgrad(const float* x, const float* dy, int N, int H, int W, float* work, float* dw, hipStream_t stream);
int launch_avgpool2x2_bwd(const float* gy, int N, int C, int Ho, int Wo, float* gx, hipStream_t s);
int launch_avgpool_fc(const float* x, const float* wt, const float* bias, float* out, int N, int C, int HW,
                      int O, hipStream_t stream);
size_t avgpool_fc_workspace_bytes(int N, int C);
bool avgpool_fc_ws_supported(int C, int HW);
int launch_avgpool_fc_ws(const float* x, const float* wt, const float* bias, float* out, float* ws, int N, int C, int HW,
                         int O, hipStream_t stream);
size_t grad_weight_pack_bytes(int O, int C, int ks);
int launch_grad_pack_weight(const float* what, int O, int C, int ks, void* packed, float* alpha, hipStream_t s);
int launch_dgrad(const float* g, const float* alpha, const void* packed, const void* xin, int x_planes, float* gx, int N,
                 int O, int C, int H, int W, int ks, int stride, hipStream_t s);
int grad_wgrad_splits(int N, int O, int C, int ks);
int launch_wgrad(const float* g, const void* xin, const void* xin2, int x_planes, float* part, int splits, int N, int O,
                 int C, int H, int W, int ks, int stride, hipStream_t s);
int launch_pack_ste(const float* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M, uint64_t* T,
                    hipStream_t stream);
int launch_pack_weight(const float* w, int O, int C, int KH, int KW, int center, int compute_alpha,
                       const bnn_hip_wlayout& L, uint32_t* wbits, uint32_t* wnz, float* alpha,
                       int32_t* zero_flag, hipStream_t stream);
bool ds_fold_applies(const ConvP& p, int flags);
int launch_bconv(const ConvP& p, int flags, hipStream_t s);
// bconv_fly.hip: the whole layer in one launch, activations (fp32, or fp16 when x_half) binarised on the fly into LDS.
// p.P / p.M are unused; p.alpha / bias / scale / out as for launch_bconv.  `plan` may be null (default plan).
bool fly_supported(const ConvP& p);
int fly_default_plan(const ConvP& p, int flags, bnn_hip_fly_plan* plan);
int launch_bconv_fly(const ConvP& p, const void* x, int x_half, int flags, const bnn_hip_fly_plan* plan, hipStream_t s);
// hblock.hip: the hierarchical block in one launch
bool hblock_supported(const bnn_hip_hblock_desc* d);
int hblock_layout(int C_in, int planes, bnn_hip_hblock_layout* L);
int launch_hblock_pack_weights(int C_in, int planes, const uint32_t* const w[3], uint32_t* dst, hipStream_t s);
int launch_hblock(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc, const float* res,
                  float* out, uint64_t* outP, hipStream_t stream);
int launch_hblock_pool(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc, const float* Kp,
                       const float* res, uint64_t* outP1, uint64_t* outP2, uint64_t* outM2, hipStream_t stream);
bool hblock_pool_supported(const bnn_hip_hblock_desc* d);
int launch_hblock_ds(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc,
                     const uint64_t* dsP, const uint64_t* dsM, const uint32_t* dsW, const float* dsA, float* out,
                     uint64_t* outP, hipStream_t stream);
bool hblock_ds_supported(const bnn_hip_hblock_desc* d);
int launch_hblock_ds_pack_weights(int C_in, int planes, const uint32_t* w, uint32_t* dst, hipStream_t s);
// hblock_cl.hip: the same block for 14 x 14 / 7 x 7 images, lanes = output channels
bool hblock_cl_supported(const bnn_hip_hblock_desc* d);
int launch_hblock_cl_pack_weights(int C_in, int planes, const uint32_t* const w[3], uint32_t* dst, hipStream_t s);
int launch_hblock_cl(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc, const float* res,
                     float* out, uint64_t* outP, hipStream_t stream);
bool hblock_cl_ds_supported(const bnn_hip_hblock_desc* d);
int launch_hblock_cl_ds(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc,
                        const uint64_t* dsP, const uint64_t* dsM, const uint32_t* dsW, const float* dsA, float* out,
                        uint64_t* outP, hipStream_t stream);
// xnor_train.hip: XNORWeightBinarizer under autograd, value and backward
int launch_xnor_grad_pack(const float* w, int O, int C, int ks, int center, int compute_alpha, void* packed, float* alpha,
                          hipStream_t s);
int launch_xnor_what(const float* w, int O, int C, int taps, int center, int compute_alpha, float* what, float* alpha,
                     hipStream_t s);
int launch_xnor_weight_bwd(const float* w, const float* dwhat, int splits, int O, int C, int taps, int center,
                           int compute_alpha, float* dw, hipStream_t s);
// bn_train.hip: training-mode BatchNorm (+ residual) (+ ReLU), forward and backward
int bn_train_splits(int N, int C, int HW);
int launch_bn_stats(const float* x, int N, int C, int HW, int splits, double* partial, hipStream_t s);
int launch_bn_apply(const float* x, const double* partial, int splits, const float* gamma, const float* beta,
                    const float* res, int relu, float* y, int N, int C, int HW, float eps, float momentum, float* rm,
                    float* rv, float* mean_out, float* invstd_out, float* work, hipStream_t s);
int launch_bn_act(const float* x, const float* scale, const float* shift, const float* res, int relu, float* y, int N,
                  int C, int HW, hipStream_t s);
int launch_bn_bwd_reduce(const float* gy, const float* y, const float* x, const float* mean, const float* invstd, int N,
                         int C, int HW, int splits, double* partial, hipStream_t s);
int launch_bn_bwd_dx(const float* gy, const float* y, const float* x, const float* mean, const float* invstd,
                     const float* gamma, const double* partial, int splits, float* dx, float* dres, float* dgamma,
                     float* dbeta, int N, int C, int HW, float* work, hipStream_t s);
int launch_bn_relu_pool_fwd(const float* x, const double* partial, int splits, const float* gamma, const float* beta,
                            float* p, unsigned char* code, int N, int C, int H, int W, float eps, float momentum, float* rm,
                            float* rv, float* mean_out, float* invstd_out, float* work, hipStream_t s);
int launch_bn_relu_pool_bwd(const float* gy, const float* p, const unsigned char* code, const float* x, const float* mean,
                            const float* invstd, const float* gamma, int N, int C, int H, int W, int splits,
                            double* partial, float* work, float* dx, float* dgamma, float* dbeta, hipStream_t s);
int launch_probe_int_alu(int mode, int iters, double* lane_ops_per_s, double* elapsed_ms, hipStream_t s);
int launch_probe_clock(int spin_iters, double* shader_mhz, double* elapsed_us, hipStream_t s);

}  // namespace bnn
