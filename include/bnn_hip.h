/*
 * bnn_hip.h — C-ABI of the MI355X (gfx950) binary-convolution inference path.
 *
 * This is the drop-in boundary for ONE hot path of 1adrianb/binary-networks-pytorch
 * (package `bnn`): the fake-binarized Conv2d / Linear forward
 *
 *     out = post( conv2d( sign(x), sign(W) * mean|W| , bias ), x )
 *
 * Reference call sites that each entry point replaces (paths relative to the
 * reference repository root):
 *
 *   bnn_hip_pack_act_f32      <- bnn/ops.py:63-66,151-152   SignActivation.forward / BasicInputBinarizer.forward
 *   bnn_hip_pack_weight_f32   <- bnn/ops.py:116-140         XNORWeightBinarizer._compute_alpha / .forward
 *   bnn_hip_bconv2d           <- bnn/layers/conv.py:90-97   Conv2d.forward  (nn.Conv2d._conv_forward + post-process)
 *   bnn_hip_bconv2d_direct    <- bnn/layers/conv.py:90-97   the same forward from the FLOAT input, sign() on the fly
 *   bnn_hip_blinear           <- bnn/layers/linear.py:22-27 Linear.forward  (F.linear + post-process)
 *   post_scale argument       <- bnn/ops.py:200-202         BasicScaleBinarizer.forward (out.mul_(alpha))
 *   (Identity post-process    <- bnn/bconfig.py:6-8 : pass post_scale = NULL)
 *
 * The reference has no FFI of its own (pure Python on top of PyTorch), so these
 * are the functions a ctypes / cgo / JNI stub binds; see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns an int status: 0 = ok, negative = bnn_hip_status
 *   - no function throws, aborts, allocates device memory or synchronises the device
 *   - all pointers are DEVICE pointers on the *current* HIP device unless stated
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream)
 *   - functions are re-entrant and keep no global mutable state except a
 *     monotonically increasing launch counter (bnn_hip_launch_count)
 *
 * Data formats
 *   Activations are ternary {-1,0,+1} because the reference's sign(0) == 0 and its
 *   zero padding is applied AFTER binarisation (bnn/layers/conv.py:91-92).  They are
 *   stored as two bit-planes, channel-group planar ("N C/64 H W" of uint64 words):
 *       P[n][g][y][x]  bit b set  <=>  x[n][64*g + b][y][x] > 0
 *       M[n][g][y][x]  bit b set  <=>  x[n][64*g + b][y][x] < 0
 *   with g < cw64 = ceil(C/64); pad bits are 0 in both planes.  (Consecutive pixels of one
 *   group are contiguous, so a wavefront whose lanes are pixels reads 512 contiguous bytes
 *   per load for any C; a pixel-major layout wastes 3/4 of every cache line at C = 512.)
 *   NaN -> neither plane (torch.sign(nan) == 0), denormals keep their sign.
 *
 *   Weights are one bit-plane (+1 -> 1, -1 -> 0) plus a non-zero mask, in the
 *   kernel-facing layout  wbits[ob][chunk][j][tap][cwc]  (uint32 words) where
 *       ob    = o / 32, j = o % 32           (output channels padded to a multiple of 32)
 *       tap   = ky*KW + kx
 *       chunk = which group of `cwc` consecutive 32-channel words of the input channels
 *   (cwc, nchunk) are a pure function of (C, KH, KW): bnn_hip_weight_layout().
 *
 *   Integer dot product per output element (exact):
 *       D   = popcount( (W & M) | (~W & P) )       disagreeing non-zero positions
 *       dot = popcount(P | M) - 2*D                 (= sum of sign(x)*sign(w)), over the window
 *   and the float result  out = fmaf(alpha[o], (float)dot, bias[o]) [* post_scale[o]].
 */
#ifndef BNN_HIP_H_
#define BNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI 15 (round 6): + bnn_hip_hblock_{supported,layout_of,pack_weights,forward} (the hierarchical block in one launch);
 * + bnn_hip_avgpool2_bn_pack2_f32 (the pool in front of a pre-activation stage + both sign planes it feeds).
 * ABI 14 (round 5): + bnn_hip_stem7x7_wgrad_f32 / bnn_hip_stem7x7_wgrad_workspace_bytes (weight gradient of the stem
 * convolution: the training backward of that layer); + bnn_hip_avgpool2x2_backward_f32, bnn_hip_xnor_grad_pack_weight_f32; + bnn_hip_avgpool_fc_ws_f32 / bnn_hip_avgpool_fc_workspace_bytes (the head as two streaming launches
 * through a workspace); + bnn_hip_stem7x7_conv_f32 (the stem's convolution alone: the training forward); the table of bnn_hip_sign_thresholds_f32 holds FOUR words per channel (was two) and kmax < 2^20.
 * ABI 13 (round 4): + bnn_hip_bn_act_f32 (eval-mode BatchNorm + residual + ReLU tail of the per-layer path);
 * bnn_hip_xnor_weight_backward_f32 takes `splits` partial slabs.
 * ABI 12 (round 4): + bnn_hip_probe_clock; + the training-side entry points bnn_hip_pack_act_ste_f32,
 * bnn_hip_bconv_grad_{input,weight}_packed_f32 (3-bit saved state), bnn_hip_bn_train_{workspace_bytes,forward,backward}_f32,
 * bnn_hip_bn_relu_maxpool_train_{forward,backward}_f32, bnn_hip_xnor_weight_{forward,backward}_f32;
 * - BNN_HIP_STEM_STAGED and BNN_HIP_FLAG_WEIGHTS_LDS (those kernels are test-only now: csrc/legacy/);
 * stem tensors capped at the 32-bit buffer-descriptor range; size arithmetic of all validators saturating.          */
#define BNN_HIP_ABI_VERSION 15
#define BNN_HIP_OCB 32 /* output channels per weight block (padding granularity of O) */

typedef enum bnn_hip_status {
  BNN_HIP_OK = 0,
  BNN_HIP_ERR_INVALID_ARG = -1,  /* null pointer, non-positive size, misaligned buffer        */
  BNN_HIP_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels implement                   */
  BNN_HIP_ERR_LAUNCH = -3,       /* hipGetLastError() != hipSuccess after a launch             */
  BNN_HIP_ERR_TOO_LARGE = -4,    /* a tensor exceeds one launch's addressing: 2^30 fp32 elements (4 GiB) for the
                                    convolution entry points, 2^31 elements elsewhere — split the batch */
  BNN_HIP_ERR_NO_DEVICE = -5     /* no HIP device / wrong architecture                         */
} bnn_hip_status;

/* Geometry of one binary convolution (mirrors torch.nn.Conv2d hyper-parameters that
 * bnn.layers.Conv2d forwards unchanged: bnn/layers/conv.py:68-83). groups == 1 only. */
typedef struct bnn_hip_conv_desc {
  int32_t N, C, H, W;          /* input  [N,C,H,W]                        */
  int32_t O, KH, KW;           /* weight [O,C,KH,KW]                      */
  int32_t stride_h, stride_w;
  int32_t pad_h, pad_w;        /* zero padding, applied after sign()      */
  int32_t dil_h, dil_w;
  int32_t flags;               /* BNN_HIP_FLAG_*                          */
} bnn_hip_conv_desc;

#define BNN_HIP_FLAG_FORCE_GENERIC 1 /* use the shape-generic kernel even if a tiled one exists */
#define BNN_HIP_FLAG_WEIGHT_ZEROS 2  /* some sign(W) == 0: honour the wnz mask (slower kernel)  */
#define BNN_HIP_FLAG_WEIGHTS_SGPR 4  /* tiled kernel: force the scalar-cache weight stream       */
/* bit 8 (BNN_HIP_FLAG_WEIGHTS_LDS until ABI 11: the LDS-staged weight tile, 2x slower than the scalar-cache stream on
 * every shape) is ignored since ABI 12; that kernel is a test-only cross-check now (csrc/legacy/).            */
#define BNN_HIP_FLAG_ACT_NONNEG 32   /* caller vouches that the M plane is ALL ZERO (activations
                                        out of a ReLU / max-pool of a ReLU are {0,+1}): 3x3 kernels
                                        then keep only the P plane in registers.  M must still be a
                                        valid pointer.  A wrong promise gives wrong results.      */
#define BNN_HIP_FLAG_THROUGHPUT 64   /* other work shares the GPU (several batches in flight): prefer fewer,
                                        longer waves — the multi-chunk 3x3 kernels then do not split a
                                        32-channel block over two waves (same results; ResNet-18 b256:
                                        -3 % with one batch in flight, +2 % with two)              */

/* Everything that happens to the integer dot after the popcount loop, fused into the conv
 * kernel so that activations can stay bit-packed between binary layers (callers:
 * bnn/models/layers/res_block.py:40-56, conv -> BN -> ReLU -> conv -> BN -> +identity -> ReLU):
 *     y = fmaf(alpha[o], dot, bias[o])                 XNOR scale + conv bias   (required)
 *     y = y * post_scale[o]                            BasicScaleBinarizer      (optional)
 *     y = fmaf(y, bn_scale[o], bn_shift[o])            eval-mode BatchNorm      (optional)
 *     y = y + residual[n,o,y,x]                        shortcut                 (optional)
 *     y = relu(y)  |  y = y >= 0 ? y : prelu[o]*y      activation               (optional)
 * Outputs: out_f32 (fp32 NCHW) and/or out_P/out_M (sign(y) as bit planes, format above, with
 * C := O; every word is written, pad bits 0).  At least one output is required.
 *
 * Pre-activation dataflows (bnn/models/layers/res_block.py:121-167 PreBasicBlock,
 * hierarchical_block.py:38-60 HBlock) order things differently; `flags` and the trailing fields
 * cover them without leaving the kernel:
 *     BNN_HIP_EPI_RES_AFTER_ACT    the residual is added AFTER the activation   (y = act(conv); y += id)
 *     p = value that is binarised for the next binary layer:
 *         default                      p = y (the stored value)
 *         BNN_HIP_EPI_PACK_BEFORE_RES  p = the value before a RES_AFTER_ACT residual was added
 *     pack_scale/pack_shift != NULL    p = fmaf(p, pack_scale[o], pack_shift[o])   (the NEXT layer's BatchNorm)
 *     BNN_HIP_EPI_PACK_RELU            planes of sign(relu(p)):  P = (p > 0), M = 0
 *     out_c_total != 0                 out_f32 and residual are [N,out_c_total,Ho,Wo] tensors and this
 *                                      convolution owns channels [out_c_offset, out_c_offset + O)
 *                                      (torch.cat of HBlock written in place).                     */
#define BNN_HIP_EPI_RES_AFTER_ACT 1
#define BNN_HIP_EPI_PACK_BEFORE_RES 2
#define BNN_HIP_EPI_PACK_RELU 4
typedef struct bnn_hip_epilogue {
  const float* alpha;      /* [o_pad]                                   */
  const float* bias;       /* [O] or NULL                               */
  const float* post_scale; /* [O] or NULL                               */
  const float* bn_scale;   /* [O] or NULL  gamma / sqrt(var + eps)      */
  const float* bn_shift;   /* [O] or NULL  beta - mean * bn_scale       */
  const float* residual;   /* [N,O,Ho,Wo] or NULL                       */
  const float* prelu;      /* [O] or NULL                               */
  int32_t relu;            /* non-zero: clamp at 0 after the residual   */
  int32_t flags;           /* BNN_HIP_EPI_*                             */
  float* out_f32;          /* [N,O,Ho,Wo] or NULL                       */
  uint64_t* out_P;         /* [N,ceil(O/64),Ho,Wo] or NULL              */
  uint64_t* out_M;
  const float* pack_scale; /* [O] or NULL (both or neither)             */
  const float* pack_shift;
  int32_t out_c_offset;    /* see above; 0/0 = plain [N,O,Ho,Wo]        */
  int32_t out_c_total;
  const int32_t* sign_thresholds; /* NULL, or [32*ceil(O/32)][4] from bnn_hip_sign_thresholds_f32 for THIS alpha / bn_scale /
                                     bn_shift: used when the epilogue is exactly BN + ReLU -> planes only (no bias,
                                     scale, residual, fp32 output): the sign bit then comes from an integer compare
                                     of the dot — same bits as the float path, fewer instructions.  Ignored
                                     otherwise.                                                          */
  /* ABI 11 — the shortcut branch of a down-sampling residual block folded into the block's second convolution
   * (bnn/models/layers/res_block.py:31-37,52-54: AvgPool2d(2) -> Conv2d 1x1 -> BatchNorm, added before the last ReLU):
   *     residual[n,o,y,x] = fmaf(fmaf(sc_alpha[o], dot1x1(sc_P[n,:,y,x], sc_wbits[o]), 0), sc_bn_scale[o], sc_bn_shift[o])
   * is computed inside the kernel with the float operations of the stand-alone 1x1 convolution + BN epilogue (same
   * bits), instead of being written to and read back from an fp32 tensor.  sc_P: sign plane P of the 1x1 conv's
   * input at the OUTPUT resolution, [N, ceil(sc_C/64), Ho, Wo] uint64, non-negative activations (its M plane is all
   * zero and not passed); sc_wbits / sc_alpha: bnn_hip_pack_weight_f32 of the [O, sc_C, 1, 1] weight.  Either all five
   * pointers or none (then sc_C is ignored); not together with `residual`.  Supported where
   * bnn_hip_shortcut_fold_supported() says so; BNN_HIP_ERR_UNSUPPORTED otherwise.
   * ABI 12 — sc_in_hw != 0: sc_P is the UN-POOLED plane of the block's input, [N, ceil(sc_C/64), H, W] with
   * sc_in_hw = (H << 16) | W, ceil(H/2) == Ho, ceil(W/2) == Wo: the kernel ORs the 2 x 2 window of every output pixel
   * itself (sign of AvgPool2d(2, ceil_mode=True, count_include_pad=False) of non-negative values) — no
   * bnn_hip_orpool_packed launch, same bits.  0 (the field was `reserved` until ABI 11): sc_P at the output resolution. */
  const uint64_t* sc_P;
  const uint32_t* sc_wbits;
  const float* sc_alpha;
  const float* sc_bn_scale;
  const float* sc_bn_shift;
  int32_t sc_C;
  int32_t sc_in_hw;
} bnn_hip_epilogue;

/* Per channel the integer dots (|dot| <= kmax = C*KH*KW) whose epilogue value
 *   fmaf(fmaf(alpha, dot, bias) [* post_scale], bn_scale, bn_shift)   is > 0.
 * Every step is monotone in dot, so that set is one-sided:  bit = (dot >= T) XOR flip.
 * `thresholds` holds 4 * 32 * ceil(O / 32) int32 (whole 32-channel blocks; pad channels are written as "never").
 * thresholds[4o] = T; thresholds[4o+1] = for EVEN o the flip bits of o's 32-channel block as one word (bit k = channel
 * 32*(o/32)+k), for ODD o the parities of the block's T (bit k = T of channel k is odd); thresholds[4o+2], [4o+3] = the
 * comparands of the kernels'
 * two-instruction form of the same test on the agreement / disagreement count (ceil(T/2) + 2^20 and
 * max(floor(-T/2) + 1 + 2^20, 0): csrc/bconv_core.h midt2_shift_in).  Found by bisection with the conv epilogue's
 * own float operations.  Re-derive when any input changes.  kmax < 2^20.
 * (ABI 14: four words per channel; ABI 9-13: {T, flip}; up to ABI 8 the table held {lo, span} of an interval.)     */
int bnn_hip_sign_thresholds_f32(const float* alpha, const float* bias, const float* post_scale,
                                const float* bn_scale, const float* bn_shift, int O, int kmax,
                                int32_t* thresholds, void* stream);

typedef struct bnn_hip_wlayout {
  int32_t cw32;     /* 32-bit words per pixel per plane (= 2*ceil(C/64))          */
  int32_t cwc;      /* words per chunk                                            */
  int32_t nchunk;   /* cw32 / cwc                                                 */
  int32_t taps;     /* KH*KW                                                      */
  int32_t o_pad;    /* O rounded up to a multiple of BNN_HIP_OCB                  */
  int32_t reserved;
  int64_t n_words;  /* uint32 words in wbits (and in wnz): o_pad*taps*cw32        */
} bnn_hip_wlayout;

typedef struct bnn_hip_devinfo {
  char name[64];
  char arch[32];          /* e.g. "gfx950:sramecc+:xnack-"                        */
  int32_t compute_units;
  int32_t clock_khz;      /* max engine clock                                     */
  int32_t mem_clock_khz;
  int32_t mem_bus_bits;
  int32_t wavefront;
  int32_t lds_bytes_per_block;
  int64_t total_mem_bytes;
  int32_t l2_bytes;
  int32_t reserved;
} bnn_hip_devinfo;

int bnn_hip_abi_version(void);
const char* bnn_hip_status_string(int status);
/* number of kernel launches issued through this library since load (host counter) */
uint64_t bnn_hip_launch_count(void);

/* HOST: fill *out for HIP device `device`.                                        */
int bnn_hip_device_info(int device, bnn_hip_devinfo* out);

/* HOST: uint64 words per pixel per activation plane for C channels: ceil(C/64).  */
int bnn_hip_act_words(int C);
/* HOST: weight layout for a [O,C,KH,KW] weight.                                   */
int bnn_hip_weight_layout(int O, int C, int KH, int KW, bnn_hip_wlayout* out);

/* sign(x) as two bit planes.  x: float32 NCHW contiguous.  P, M: [N,ceil(C/64),H,W]
 * uint64 each, 16-byte aligned.                                                    */
int bnn_hip_pack_act_f32(const float* x, int N, int C, int H, int W,
                         uint64_t* P, uint64_t* M, void* stream);

/* Same planes from an IEEE half-precision tensor (a `.half()` model: Tensor.sign() of fp16 values,
 * bnn/ops.py:66).  x: [N,C,H,W] of 16-bit floats, 2-byte aligned.  sign() is exact in any precision, so
 * the planes equal those of the widened fp32 tensor bit for bit.                                  */
int bnn_hip_pack_act_f16(const void* x, int N, int C, int H, int W,
                         uint64_t* P, uint64_t* M, void* stream);

/* AvgPool2d(kernel=k, stride=k, ceil_mode=True, count_include_pad=False) followed by
 * sign(): the shortcut branch of a down-sampling residual stage
 * (bnn/models/resnet.py:128-133: AvgPool2d -> binary conv1x1 -> BN).  Output planes have
 * ceil(H/k) x ceil(W/k) pixels.                                                    */
int bnn_hip_avgpool_pack_f32(const float* x, int N, int C, int H, int W, int k,
                             uint64_t* P, uint64_t* M, void* stream);

/* AvgPool2d(2, 2) (even H, W; bnn_amd/models/resnet.py: the pool in front of a hierarchical-block stage) + the sign planes
 * of up to two BatchNorm branches of the pooled tensor in one pass:  t = window sum (row by row) / 4 as ATen computes it,
 * P1/M1 = sign(act1(fmaf(t, a1, b1))), P2/M2 = sign(act2(fmaf(t, a2, b2))) (relu: M = 0).  a2 == NULL: one branch.
 * out_f32: NULL, or the pooled tensor [N, C, H/2, W/2] when somebody needs it.                                     */
int bnn_hip_avgpool2_bn_pack2_f32(const float* x, int N, int C, int H, int W, const float* a1, const float* b1, int relu1,
                                  uint64_t* P1, uint64_t* M1, const float* a2, const float* b2, int relu2, uint64_t* P2,
                                  uint64_t* M2, float* out_f32, void* stream);

/* The same shortcut input from the SIGN PLANES of a NON-NEGATIVE tensor (a ReLU output: M == 0): the average of
 * non-negative values is positive iff one of them is, so sign(AvgPool_k(x)) is the OR of the P plane over each
 * k x k window (ceil mode: windows are clipped at the border) — 2 bits per element read instead of 32.
 * P: [N,ceil(C/64),H,W]; out_P / out_M: [N,ceil(C/64),ceil(H/k),ceil(W/k)], out_M is written as 0.
 * Exact for finite inputs; the caller vouches for x >= 0 (as with BNN_HIP_FLAG_ACT_NONNEG).            */
int bnn_hip_orpool_packed(const uint64_t* P, int N, int C, int H, int W, int k,
                          uint64_t* out_P, uint64_t* out_M, void* stream);

/* BatchNorm(eval) -> [ReLU] -> sign() of an fp32 NCHW tensor in one pass: the input binarisation of a
 * pre-activation block (res_block.py:148 `conv1(bn1(x))`, hierarchical_block.py:39 `conv1(act1(bn1(x)))`).
 *   v = fmaf(x, bn_scale[c], bn_shift[c])  (both NULL: v = x);  relu != 0: planes of sign(max(v, 0)).  */
int bnn_hip_bn_act_pack_f32(const float* x, int N, int C, int H, int W, const float* bn_scale,
                            const float* bn_shift, int relu, uint64_t* P, uint64_t* M, void* stream);

/* Tail of the real-valued stem, one pass over the stem conv's fp32 NCHW output
 * (bnn/models/resnet.py:150-153: bn1 -> relu -> maxpool, then the sign() of the first binary
 * conv):  v = fma(x, bn_scale[c], bn_shift[c]) (both NULL = no BN);  m = max over the k x k
 * window with stride/pad of nn.MaxPool2d;  y = relu ? max(m,0) : m.
 * Outputs (either may be NULL): out_f32 [N,C,Ho,Wo] and sign(y) as planes P, M.     */
int bnn_hip_bn_relu_maxpool_pack_f32(const float* x, int N, int C, int H, int W,
                                     const float* bn_scale, const float* bn_shift, int relu,
                                     int k, int stride, int pad,
                                     float* out_f32, uint64_t* P, uint64_t* M, void* stream);

/* The whole real-valued stem of the reference's ResNets as one MFMA kernel
 * (bnn/models/resnet.py:93-96,150-153): conv 7x7 / stride 2 / pad 3, 3 -> 64 channels, no bias
 * (w: float32 [64,3,7,7]) -> y*bn_scale[c] + bn_shift[c] -> ReLU -> MaxPool 3x3 / 2 / 1.
 * x: float32 [N,3,H,W].  Outputs (either may be NULL): out_f32 [N,64,Hp,Wp] and its sign planes
 * P, M ([N,1,Hp,Wp] uint64), Hp = ((H-1)/2 + 1 - 1)/2 + 1 (56 for H = 224).
 * flags = 0: fp32 operands split into fp16 hi+lo halves (22 mantissa bits), products as
 *            hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation — error
 *            ~3e-7 relative, the rounding class of an fp32 convolution, at 3/16 of the matrix time;
 * flags = BNN_HIP_STEM_EXACT_FP32: v_mfma_f32_16x16x4_f32, bit-for-bit an fp32 fmaf chain.     */
#define BNN_HIP_STEM_EXACT_FP32 1
/* Plain fp16 operands, one MFMA per product, fp32 accumulation (the "fp16 MFMA stem" of BASELINE config 5):
 * ~5e-4 relative error instead of ~3e-7, one third of the matrix work.  Opt-in; not with EXACT_FP32.        */
#define BNN_HIP_STEM_FP16 4
/* (flag 8, BNN_HIP_STEM_STAGED until ABI 11 — the round-2 kernel of the same arithmetic — is rejected since ABI 12:
 * that kernel is a test-only cross-check now, csrc/legacy/stem_split.hip in libbnn_hip_legacy.so.)
 * Every tensor of a launch stays below 2^32 - 512 bytes (32-bit buffer descriptors): BNN_HIP_ERR_TOO_LARGE above
 * that — split the batch (~7100 images of 224 x 224 per launch).                                                    */
int bnn_hip_stem7x7_bn_relu_pool_pack_f32(const float* x, const float* w,
                                          const float* bn_scale, const float* bn_shift,
                                          int N, int H, int W, int flags,
                                          float* out_f32, uint64_t* P, uint64_t* M, void* stream);

/* ABI 14 — the stem's convolution ALONE, for the training forward (bnn/models/resnet.py:150: `x = self.conv1(x)`; the
 * BatchNorm of a training step needs the raw conv output for its batch statistics): conv 7x7 / stride 2 / pad 3,
 * 3 -> 64 channels, no bias.  x: float32 [N,3,H,W], w: float32 [64,3,7,7], out: float32 [N,64,Hc,Wc] with
 * Hc = (H - 1) / 2 + 1.  The same MFMA stream as the fused stem (fp32 operands as fp16 hi + lo, three products, fp32
 * accumulation; flags = BNN_HIP_STEM_FP16: plain fp16 operands): the values the fused kernel normalises and pools, bit
 * for bit.  Same size limits as the fused stem.                                                                     */
/* The same kernel with the sign planes taken behind a per-channel affine of its output (ABI 15):
 *     P = fmaf(y, pack_scale[c], pack_shift[c]) > 0,  M = 0          (y: the pooled ReLU output, written to out_f32 unchanged)
 * — what the first binary layer of a pre-activation network reads (HBlock: bn1 -> relu -> conv1, hierarchical_block.py:39):
 * the packing pass over the fp32 tensor (bnn_hip_bn_act_pack_f32, relu = 1) disappears, same bits.  flags: 0 or
 * BNN_HIP_STEM_FP16.                                                                                              */
int bnn_hip_stem7x7_bn_relu_pool_pack_affine_f32(const float* x, const float* w, const float* bn_scale,
                                                 const float* bn_shift, const float* pack_scale, const float* pack_shift,
                                                 int N, int H, int W, int flags, float* out_f32, uint64_t* P, uint64_t* M,
                                                 void* stream);
int bnn_hip_stem7x7_conv_f32(const float* x, const float* w, int N, int H, int W, int flags, float* out, void* stream);

/* ABI 14 — the weight gradient of that convolution: the training backward of bnn/models/resnet.py:150 (the input is data,
 * so this is the layer's whole backward; torch: aten::convolution_backward with output_mask = {0, 1, 0}).
 *   dw[o][c][ky][kx] = sum over n, y, x of dy[n][o][y][x] * x[n][c][2y + ky - 3][2x + kx - 3]        (zero padding)
 * x: float32 [N,3,H,W], dy: float32 [N,64,Hc,Wc] (Hc = (H - 1) / 2 + 1), dw: float32 [64,3,7,7] (overwritten).
 * fp32 products and fp32 accumulation on the matrix cores (v_mfma_f32_16x16x4_f32), per-workgroup partial sums in
 * `workspace` added in index order in fp64: deterministic, no atomics.  Two launches.  workspace: at least
 * bnn_hip_stem7x7_wgrad_workspace_bytes(N, H, W) bytes (0 = this shape is not supported: image rows wider than ~950
 * pixels do not fit the kernel's LDS patch, BNN_HIP_ERR_UNSUPPORTED — the caller keeps its own backward).          */
size_t bnn_hip_stem7x7_wgrad_workspace_bytes(int N, int H, int W);
int bnn_hip_stem7x7_wgrad_f32(const float* x, const float* dy, int N, int H, int W, float* workspace,
                              size_t workspace_bytes, float* dw, void* stream);

/* ABI 14 — backward of the shortcut's AvgPool2d(2, 2) on an even-sized map (bnn/models/resnet.py:128-133 in a training
 * step): gx[n, c, 2y + a, 2x + b] = gy[n, c, y, x] / 4.  gy: float32 [N,C,Ho,Wo], gx: float32 [N,C,2 Ho,2 Wo]
 * (overwritten).  One streaming launch.                                                                              */
int bnn_hip_avgpool2x2_backward_f32(const float* gy, int N, int C, int Ho, int Wo, float* gx, void* stream);

/* The real-valued head of the reference's ResNets in one kernel (bnn/models/resnet.py:160-164:
 * avgpool -> flatten -> fc):  out[n,o] = bias[o] + sum_c w_t[c,o] * mean_hw x[n,c,hw].
 * x: float32 [N,C,HW] (an NCHW tensor with HW = H*W), w_t: the Linear weight TRANSPOSED to [C,O]
 * (so that a wavefront reads contiguous bytes; nn.Linear stores [O,C]), bias: [O] or NULL,
 * out: float32 [N,O].  fp32 fmaf accumulation in index order; C*16 bytes of LDS (C <= 10240).       */
int bnn_hip_avgpool_fc_f32(const float* x, int N, int C, int HW, const float* w_t, const float* bias,
                           int O, float* out, void* stream);
/* ABI 14 — the same head as two launches through a caller-provided workspace (the library allocates nothing):
 * a streaming average-pool kernel (coalesced loads, the same index-order fp32 sums) into the workspace, then the
 * product as 16-image x 64-output tiles with the k range split over eight waves whose partial sums are added in
 * segment order.  workspace: bnn_hip_avgpool_fc_workspace_bytes(N, C) bytes, 16-byte aligned, contents undefined
 * before and after.  Shapes the two-launch form does not cover (C * 64 bytes of LDS above the CU's 160 KB) run the
 * one-kernel form.  Per-image results do not depend on N or on the image's position in the batch.
 * ResNet-18, batch 256: see profiles/ (round 5) — the one-kernel form is 28 us, latency-bound.                     */
size_t bnn_hip_avgpool_fc_workspace_bytes(int N, int C);
int bnn_hip_avgpool_fc_ws_f32(const float* x, int N, int C, int HW, const float* w_t, const float* bias,
                              int O, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* XNOR-Net weight binarisation.  w: float32 [O,C,KH,KW] contiguous.
 *   center        != 0: subtract the mean over C per (o,ky,kx) first  (ops.py:130-132)
 *   compute_alpha != 0: alpha[o] = mean |w[o]| (after centering)      (ops.py:116-127)
 *                 == 0: alpha[o] = 1
 * Outputs: wbits, wnz (bnn_hip_weight_layout().n_words uint32 each), alpha[o_pad]
 * (pad channels get alpha 0), *zero_flag (int32, device) is OR-ed with 1 when some
 * in-range sign(w) == 0 (zero or NaN weight) — the caller clears it beforehand.    */
int bnn_hip_pack_weight_f32(const float* w, int O, int C, int KH, int KW,
                            int center, int compute_alpha,
                            uint32_t* wbits, uint32_t* wnz, float* alpha,
                            int32_t* zero_flag, void* stream);

/* ---- training: gradients of  out = conv2d(sign(x), What), What = sign(Wc) * alpha, for the
 * 3x3 / stride 1 or 2 / padding 1 layer and the 1x1 / stride 1 / padding 0 layer, dilation 1
 * (bnn/layers/conv.py:90-97 under autograd; input STE bnn/ops.py:68-73).  `ksize` is 3 or 1, p = ksize / 2.
 * Each is a GEMM with one real operand (the incoming gradient g, split into fp16 hi + lo) and one operand that is
 * exactly {-1,0,+1}: two v_mfma_f32_16x16x32_f16 per product, fp32 accumulation (fp32-convolution rounding class).
 * x: the layer's fp32 input [N,C,H,W], W <= 64; g: float32 [N,O,Hg,Wg] with Hg = (H-1)/stride + 1 (same for W).
 * Anything else (other kernel sizes, a strided 1x1, W > 64): BNN_HIP_ERR_UNSUPPORTED — the caller keeps its
 * fp32 path for those.
 *
 *   bnn_hip_grad_pack_weight_f32   What [O,C,k,k] (= +-alpha[o] or 0) -> alpha[o] = max|What[o]| and sign(What) as
 *                                  fp16 MFMA fragments in `packed` (bnn_hip_grad_weight_pack_bytes(O, C, k) bytes,
 *                                  16-byte aligned); once per optimiser step.
 *   bnn_hip_bconv_grad_input_f32   gx[n,c,y,x] = 1[|x| < 1] * sum_{o,ky,kx} g[n,o,(y+p-ky)/s,(x+p-kx)/s] * What[o,c,ky,kx]
 *                                  (terms whose (y+p-ky, x+p-kx) is not a multiple of the stride s do not exist)
 *   bnn_hip_bconv_grad_weight_f32  partial[s,o,c,ky,kx] = sum over the images of split s and all pixels of
 *                                  g[n,o,y,x] * sign(x)[n,c,s*y+ky-p,s*x+kx-p];  dL/dWhat = sum_s partial[s]
 *                                  (s < splits = bnn_hip_bconv_grad_weight_splits(N, O, C, k); the caller adds
 *                                  the `splits` slabs — a deterministic reduction instead of float atomics).       */
size_t bnn_hip_grad_weight_pack_bytes(int O, int C, int ksize);
int bnn_hip_grad_pack_weight_f32(const float* w_hat, int O, int C, int ksize, void* packed, float* alpha,
                                 void* stream);
int bnn_hip_bconv_grad_input_f32(const float* g, const float* alpha, const void* packed, const float* x, float* gx,
                                 int N, int O, int C, int H, int W, int ksize, int stride, void* stream);
int bnn_hip_bconv_grad_weight_splits(int N, int O, int C, int ksize);
int bnn_hip_bconv_grad_weight_f32(const float* g, const float* x, float* partial, int splits,
                                  int N, int O, int C, int H, int W, int ksize, int stride, void* stream);

/* ABI 12: the same two gradients from 3 BITS per input element instead of the fp32 tensor — what a training step has
 * to keep of x between forward and backward (the reference's autograd keeps the fp32 x and the fp32 sign(x),
 * bnn/ops.py:63-73).  bnn_hip_pack_act_ste_f32 writes, in one pass over x, the sign planes P = x > 0, M = x < 0 (the
 * format of bnn_hip_pack_act_f32) and the straight-through mask T = |x| < 1 (NaN -> 0); the *_packed_* kernels read T
 * (input gradient) resp. P and M (weight gradient).  Results are bit-identical to the fp32-x entry points.            */
int bnn_hip_pack_act_ste_f32(const float* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M, uint64_t* T,
                             void* stream);
int bnn_hip_bconv_grad_input_packed_f32(const float* g, const float* alpha, const void* packed_w, const uint64_t* T,
                                        float* gx, int N, int O, int C, int H, int W, int ksize, int stride,
                                        void* stream);
int bnn_hip_bconv_grad_weight_packed_f32(const float* g, const uint64_t* P, const uint64_t* M, float* partial, int splits,
                                         int N, int O, int C, int H, int W, int ksize, int stride, void* stream);

/* Binary convolution on packed operands.  out: float32 [N,O,Ho,Wo] contiguous.
 *   out[n,o,y,x] = fmaf(alpha[o], dot, bias ? bias[o] : 0) * (post_scale ? post_scale[o] : 1)
 * wnz may be NULL unless BNN_HIP_FLAG_WEIGHT_ZEROS is set.                         */
int bnn_hip_bconv2d(const bnn_hip_conv_desc* d,
                    const uint64_t* P, const uint64_t* M,
                    const uint32_t* wbits, const uint32_t* wnz,
                    const float* alpha, const float* bias, const float* post_scale,
                    float* out, void* stream);

/* Binary convolution with the fused epilogue described at bnn_hip_epilogue.        */
int bnn_hip_bconv2d_fused(const bnn_hip_conv_desc* d,
                          const uint64_t* P, const uint64_t* M,
                          const uint32_t* wbits, const uint32_t* wnz,
                          const bnn_hip_epilogue* epi, void* stream);

/* HOST: 1 when bnn_hip_bconv2d_fused() takes a folded shortcut convolution of sc_C input channels (bnn_hip_epilogue
 * sc_*) for this convolution — 3x3 tiled kernels on non-negative activations (BNN_HIP_FLAG_ACT_NONNEG) without zero
 * weights, epilogue = BatchNorm + shortcut + ReLU -> fp32 + sign planes, sc_C in {64, 128, 256}; 0 otherwise.       */
int bnn_hip_shortcut_fold_supported(const bnn_hip_conv_desc* d, int sc_C);

/* Same traversal, raw integer result: dot[n,o,y,x] (int32) — the bit-exact target
 * of the popcount path against the emulated-integer oracle.                        */
int bnn_hip_bconv2d_dot(const bnn_hip_conv_desc* d,
                        const uint64_t* P, const uint64_t* M,
                        const uint32_t* wbits, const uint32_t* wnz,
                        int32_t* dot, void* stream);

/* Binary fully-connected layer: x packed as [B][ceil(F/64)] planes (pack_act with
 * H=W=1, i.e. [B][ceil(F/64)] words), weight packed with KH=KW=1.  out: float32 [B,O].                         */
int bnn_hip_blinear(int B, int F, int O,
                    const uint64_t* P, const uint64_t* M,
                    const uint32_t* wbits, const uint32_t* wnz, int weight_zeros,
                    const float* alpha, const float* bias, const float* post_scale,
                    float* out, void* stream);

/* ---- The LAYER in one launch (ABI 10): bnn.layers.Conv2d.forward (bnn/layers/conv.py:90-97) from the float
 * activation tensor to the float output, sign(x) (bnn/ops.py:63-66,151-152) computed ON THE FLY inside the
 * convolution kernel: a workgroup binarises its band of the input (whole images, or a run of output rows of one
 * image with its halo) straight into LDS as two bit planes and convolves it from there — no packed copy of the
 * activations in HBM, no workspace, the input is read once.  Same integers, same float epilogue, same bits as
 * bnn_hip_pack_act_f32 + bnn_hip_bconv2d.
 *   x: [N,C,H,W] contiguous, x_dtype = BNN_HIP_DTYPE_F32 (4-byte aligned) or BNN_HIP_DTYPE_F16 (a `.half()` model:
 *      the planes are those of the exactly widened values; the output stays fp32, the caller rounds it once);
 *      N*C*H*W < 2^30 elements per launch.
 *   plan: NULL = bnn_hip_bconv2d_direct_plan()'s choice.  A caller's plan (tests, tuning) supplies
 *      images_per_band / rows_per_band / waves / blocks_per_unit / pack_ahead / fine_head / fine_tail /
 *      producers;
 *      lds_bytes and n_bands are outputs.
 * Returns BNN_HIP_ERR_UNSUPPORTED when not even one output row of one image with its halo fits into the 160 KiB
 * of LDS of a CU (or an index factor leaves the 24-bit multiplies of the tiled kernels): use
 * bnn_hip_pack_act_f32 + bnn_hip_bconv2d (what bnn_hip_bconv2d_f32 does by itself) for those.                   */
#define BNN_HIP_DTYPE_F32 0
#define BNN_HIP_DTYPE_F16 1
typedef struct bnn_hip_fly_plan {
  int32_t images_per_band;  /* whole images per workgroup (>= 1); > 1 only with rows_per_band == Ho            */
  int32_t rows_per_band;    /* output rows per workgroup: Ho = whole images                                     */
  int32_t waves;            /* wavefronts per workgroup, 1..16                                                  */
  int32_t blocks_per_unit;  /* 32-channel output blocks per (coarse) work unit = per load of the field: 1, 2 or 4 */
  int32_t pack_ahead;       /* pixel groups (64 pixels) the binarisation is kept in front of the convolution;
                               < 0 = the planner's default                                                      */
  int32_t fine_head;        /* pixel groups at the start / end of a band that are split into single-pass units    */
  int32_t fine_tail;        /* (8 output channels of one block each); < 0 = the planner's default                 */
  int32_t producers;        /* wavefronts per workgroup that binarise the band front to back before they join the
                               convolution (0: every wave packs what it needs itself); < 0 = the planner's default */
  int32_t lds_bytes;        /* out: dynamic LDS per workgroup                                                   */
  int32_t n_bands;          /* out: workgroups of the launch                                                    */
} bnn_hip_fly_plan;
/* HOST: the plan bnn_hip_bconv2d_direct(plan = NULL) uses for this geometry. */
int bnn_hip_bconv2d_direct_plan(const bnn_hip_conv_desc* d, bnn_hip_fly_plan* plan);
int bnn_hip_bconv2d_direct(const bnn_hip_conv_desc* d, const void* x, int x_dtype,
                           const uint32_t* wbits, const uint32_t* wnz,
                           const float* alpha, const float* bias, const float* post_scale,
                           float* out, const bnn_hip_fly_plan* plan, void* stream);

/* ---- The hierarchical block in one launch (ABI 15): bnn.models.layers.HBlock.forward
 * (bnn/models/layers/hierarchical_block.py:38-60) behind its first BatchNorm + activation:
 *     o1 = conv1(sign(act1(bn1(x))));  o2 = conv2(sign(act2(bn2(o1))));  o3 = conv3(sign(act3(bn3(o2))))
 *     y  = cat(o1, o2, o3) + residual
 * three 3x3 / stride 1 / padding 1 binary convolutions (C_in -> C/2 -> C/4 -> C/4, C = `planes`) with ReLU
 * activations: every sign plane between them is non-negative (P plane only) and stays in LDS; a workgroup owns whole
 * images or a band of rows of one image with its halo (csrc/hblock.hip).  Bit-identical to three bnn_hip_bconv2d_fused
 * launches with BNN_HIP_EPI_RES_AFTER_ACT | PACK_BEFORE_RES | PACK_RELU.
 *   in_P:     the block's input planes, sign(act1(bn1(x))): [N, ceil(C_in/64), H, W] uint64, P plane (bnn_hip_bn_act_pack_f32
 *             with relu = 1, or the out_P of the previous block's launch)
 *   weights:  bnn_hip_hblock_pack_weights() of the three standard weight packs (no zero weights; bias-free convolutions)
 *   consts:   fp32, offsets from bnn_hip_hblock_layout_of(): alpha of conv1 / conv2 / conv3 | folded bn2 scale, shift |
 *             folded bn3 scale, shift | the NEXT block's folded bn1 scale, shift over all C channels (read only when out_P != NULL)
 *   residual: fp32 [N, C, H, W] (the block input, or its shortcut branch);  out: fp32 [N, C, H, W], must not alias residual
 *   out_P:    NULL, or [N, C/64, H, W] uint64: sign(relu(fmaf(y, next_a, next_b))) — the next block's in_P
 * planes % 64 == 0; C_in <= 32 or C_in % 64 == 0; N * planes < 2^23; N * planes * H * W < 2^30.                     */
/* flags bit of bnn_hip_hblock_desc: the SMALL-image form of the kernel (csrc/hblock_cl.hip: 14 x 14 and 7 x 7 images, planes a
 * multiple of 256, C_in a multiple of 64): a wave's lanes are 64 output channels instead of 64 pixels — no idle lanes on 196- or
 * 49-pixel images, taps in the left / right padding not computed.  `weights` must then come from
 * bnn_hip_hblock_pack_weights_cl (same size, another order); consts and results are the same, bit for bit.        */
#define BNN_HIP_HBLOCK_CHANNEL_LANES 128
typedef struct bnn_hip_hblock_desc {
  int32_t N, C_in, H, W;
  int32_t planes;
  int32_t flags;            /* BNN_HIP_FLAG_THROUGHPUT: other work shares the GPU — whole-image regions preferred;
                               BNN_HIP_HBLOCK_CHANNEL_LANES (below)                                                   */
  int32_t rows_per_band;    /* 0 = the planner's choice; else rows of one image per workgroup                         */
  int32_t images_per_band;  /* 0 = the planner's choice; > 1 only with rows_per_band == 0 or H                         */
  int32_t waves;            /* 0 = 16; wavefronts per workgroup, 1..16                                                */
  int32_t reserved;
} bnn_hip_hblock_desc;
typedef struct bnn_hip_hblock_layout {
  int64_t weight_words;     /* uint32 words of the weight buffer (64-byte aligned)                                    */
  int64_t w_off[3];         /* word offset of conv1 / conv2 / conv3 in it                                             */
  int64_t const_floats;     /* floats of the constants buffer                                                         */
  int64_t alpha_off[3];     /* float offsets: alpha[O_k] of conv k                                                    */
  int64_t pack_a_off[2], pack_b_off[2];  /* folded bn2 (C/2 values each) and bn3 (C/4)                                */
  int64_t next_a_off, next_b_off;        /* the next block's folded bn1 (C values each)                               */
} bnn_hip_hblock_layout;
/* HOST: 1 when bnn_hip_hblock_forward covers this geometry on the current device, else 0. */
int bnn_hip_hblock_supported(const bnn_hip_hblock_desc* d);
/* HOST: buffer sizes and offsets for a block of this shape. */
int bnn_hip_hblock_layout_of(int C_in, int planes, bnn_hip_hblock_layout* out);
/* wbits1..3: bnn_hip_pack_weight_f32 packs of [C/2, C_in, 3, 3], [C/4, C/2, 3, 3], [C/4, C/4, 3, 3]. */
int bnn_hip_hblock_pack_weights(int C_in, int planes, const uint32_t* wbits1, const uint32_t* wbits2,
                                const uint32_t* wbits3, uint32_t* weights, void* stream);
int bnn_hip_hblock_pack_weights_cl(int C_in, int planes, const uint32_t* wbits1, const uint32_t* wbits2,
                                   const uint32_t* wbits3, uint32_t* weights, void* stream);
int bnn_hip_hblock_forward(const bnn_hip_hblock_desc* d, const uint64_t* in_P, const uint32_t* weights,
                           const float* consts, const float* residual, float* out, uint64_t* out_P, void* stream);

/* The last block of a stage in front of `AvgPool2d(2, 2)` and a block with a shortcut convolution (ABI 15;
 * bnn/models/resnet.py: the stages of the hierarchical-block network, hierarchical_block.py:30-39): the block's output is
 * only ever pooled and binarised — by the next block's bn1 -> ReLU and by its shortcut's BatchNorm — so this launch writes
 * those planes at half resolution and NO fp32 tensor:
 *     t      = AvgPool2d(2, 2)(y)                     (((y00 + y01) + y10) + y11) / 4, as ATen sums a window
 *     out_P1 = fmaf(t, a1[c], b1[c]) > 0              (the minus plane of a ReLU'd tensor is zero: not written)
 *     out_P2 = fmaf(t, a2[c], b2[c]) > 0,  out_M2 = ... < 0
 * pool_consts = [a1/4 | b1 | a2/4 | b2 | -a2/4 | -b2 | 0 | 0], `planes` floats each, 32-byte aligned (the four lanes of a
 * window each test one plane's bit as fmaf(window sum, a, b) > 0: the caller checks that a/4 is exact, i.e. that a is not
 * within two binades of the smallest normal).  C_in == planes, even H and W, the widths of
 * bnn_hip_hblock_pool_supported; the planes are [N, planes / 64, H / 2, W / 2].  Same bits as bnn_hip_hblock_forward +
 * bnn_hip_avgpool2_bn_pack2_f32.  */
int bnn_hip_hblock_pool_supported(const bnn_hip_hblock_desc* d);
int bnn_hip_hblock_pool_forward(const bnn_hip_hblock_desc* d, const uint64_t* in_P, const uint32_t* weights,
                                const float* consts, const float* pool_consts, const float* residual, uint64_t* out_P1,
                                uint64_t* out_P2, uint64_t* out_M2, void* stream);

/* The FIRST block of a stage with its shortcut convolution inside the launch (ABI 15; hierarchical_block.py:30-36: the
 * shortcut of a block that changes width is BatchNorm -> sign -> binary conv1x1 of the block's input).  Instead of a residual
 * tensor the launch takes the two sign planes of that binarisation (sc_P / sc_M: [N, C_in / 64, H, W], what
 * bnn_hip_hblock_pool_forward or bnn_hip_avgpool2_bn_pack2_f32 wrote), the 1 x 1 weights as [planes][C_in / 32] words
 * (bnn_hip_hblock_pack_shortcut_weights of the standard pack; no zero weights, no bias) and their alpha [planes], and
 * computes  shortcut = fmaf(alpha[c], dot, 0)  per pass — the integer and the one rounding of bnn_hip_bconv2d — so the
 * shortcut launch and its fp32 tensor disappear.  planes == 2 * C_in, C_in in {64, 128} — with
 * BNN_HIP_HBLOCK_CHANNEL_LANES (14 x 14 / 7 x 7 images, weights of bnn_hip_hblock_pack_weights_cl) C_in in {128, 256};
 * out_P is required (a next block in the stage); same bits as bnn_hip_bconv2d + bnn_hip_hblock_forward.  */
int bnn_hip_hblock_shortcut_supported(const bnn_hip_hblock_desc* d);
int bnn_hip_hblock_pack_shortcut_weights(int C_in, int planes, const uint32_t* wbits, uint32_t* weights, void* stream);
int bnn_hip_hblock_shortcut_forward(const bnn_hip_hblock_desc* d, const uint64_t* in_P, const uint32_t* weights,
                                    const float* consts, const uint64_t* sc_P, const uint64_t* sc_M,
                                    const uint32_t* sc_weights, const float* sc_alpha, float* out, uint64_t* out_P,
                                    void* stream);

/* fp32 NCHW in -> fp32 NCHW out.  ONE launch (bnn_hip_bconv2d_direct) wherever that path applies:
 * bnn_hip_conv_workspace_bytes(d) is then 0 and `workspace` may be NULL.  For the remaining geometries (see above)
 * it is pack_act + conv through `workspace` (bnn_hip_conv_workspace_bytes(d) bytes, 16-byte aligned).          */
size_t bnn_hip_conv_workspace_bytes(const bnn_hip_conv_desc* d);
int bnn_hip_bconv2d_f32(const bnn_hip_conv_desc* d, const float* x,
                        const uint32_t* wbits, const uint32_t* wnz,
                        const float* alpha, const float* bias, const float* post_scale,
                        float* out, void* workspace, void* stream);

/* XNORWeightBinarizer.forward under autograd (bnn/ops.py:129-140 with the STE of bnn/ops.py:68-73), value and backward
 * as one kernel each (torch: ~14 small kernels per layer and step).  w, what, dwhat, dw: fp32 [O, C, KH, KW].
 *   forward :  what = sign(Wc) * alpha[o],  Wc = w - mean over C (center),  alpha = mean |Wc| (compute_alpha) or 1;
 *              alpha (may be NULL) receives alpha[O] (fp64 sums in a fixed order; bnn_hip_pack_weight_f32's alpha to the
 *              rounding of that sum).
 *   backward:  dw from dwhat = dL/dwhat:  dWc = dwhat * alpha * 1[|Wc| < 1] + sign(Wc) * sum(dwhat * sign(Wc)) / (C KH KW),
 *              dw = dWc - mean over C of dWc when centred.  dwhat: `splits` >= 1 slabs [splits][O][C][KH][KW] that are
 *              added here in slab order (ABI 13: the split-K partial sums of bnn_hip_bconv_grad_weight_f32 go in as they
 *              are — no separate reduction pass); splits == 1: the gradient itself.                                    */
int bnn_hip_xnor_weight_forward_f32(const float* w, int O, int C, int KH, int KW, int center, int compute_alpha,
                                    float* what, float* alpha, void* stream);
int bnn_hip_xnor_weight_backward_f32(const float* w, const float* dwhat, int splits, int O, int C, int KH, int KW,
                                     int center, int compute_alpha, float* dw, void* stream);

/* ABI 14 — what bnn_hip_bconv_grad_input_f32 needs of the weights, straight from W in ONE launch: the fragments and alpha of
 * bnn_hip_grad_pack_weight_f32(What) for What = XNORWeightBinarizer(W) (bnn/ops.py:129-140) — the same bytes as
 * bnn_hip_xnor_weight_forward_f32 followed by bnn_hip_grad_pack_weight_f32 (three launches), without the fp32 What.
 * w: float32 [O,C,k,k], k = 3 or 1; packed: bnn_hip_grad_weight_pack_bytes(O, C, k) bytes, 16-byte aligned (written
 * completely: no clearing needed); alpha: float32 [O].                                                                  */
int bnn_hip_xnor_grad_pack_weight_f32(const float* w, int O, int C, int ksize, int center, int compute_alpha, void* packed,
                                      float* alpha, void* stream);

/* Training-mode BatchNorm2d fused with what follows it in the reference's residual blocks:
 *     y = relu?( batch_norm_train(x) (+ residual) )          bnn/models/layers/res_block.py:40-56, resnet.py:150-153
 * x, y, residual: fp32 [N, C, HW] (NCHW with HW = H * W).  forward: per-channel batch statistics (fp64 accumulation,
 * deterministic), y, save_mean / save_invstd [C] for the backward, and — when running_mean / running_var are given —
 * their momentum update (unbiased variance), exactly torch.nn.BatchNorm2d's training semantics to fp32 rounding.
 * backward: g = gy * 1[y > 0] when `y` is given (a fused ReLU; pass NULL otherwise); dgamma = sum g * xhat, dbeta = sum g,
 * dx = gamma * invstd * (g - dbeta / m - xhat * dgamma / m), and dres = g (the residual branch's gradient) when dres
 * is given.  gamma / beta may be NULL (1 / 0).  `workspace`: bnn_hip_bn_train_workspace_bytes(N, C, HW) bytes,
 * 8-byte aligned, not shared between concurrent calls.  Three launches each, no synchronisation.                     */
size_t bnn_hip_bn_train_workspace_bytes(int N, int C, int HW);
int bnn_hip_bn_train_forward_f32(const float* x, int N, int C, int HW, const float* gamma, const float* beta,
                                 const float* residual, int relu, float eps, float momentum, float* running_mean,
                                 float* running_var, float* y, float* save_mean, float* save_invstd, void* workspace,
                                 void* stream);
int bnn_hip_bn_train_backward_f32(const float* gy, const float* y, const float* x, const float* save_mean,
                                  const float* save_invstd, const float* gamma, int N, int C, int HW, float* dx,
                                  float* dres, float* dgamma, float* dbeta, void* workspace, void* stream);

/* Eval-mode BatchNorm2d with folded constants, fused with what follows it in the reference's residual blocks:
 *     y = relu?( fmaf(x, scale[c], shift[c]) (+ residual) )     bnn/models/layers/res_block.py:40-56 under .eval()
 * x, y, residual: fp32 [N, C, HW]; scale, shift: fp32 [C] (scale = weight / sqrt(running_var + eps), shift = bias -
 * running_mean * scale, rounded as the caller wishes — the host side rounds them as ATen's CPU kernel does).  The same
 * float operations in the same order as the BN / residual / ReLU epilogue of bnn_hip_bconv2d_fused, so a network run
 * layer by layer and the fused executor agree bit for bit.  ReLU keeps NaN (torch.relu).  One launch.  y must not
 * alias x or residual.  (ABI 13)                                                                                      */
int bnn_hip_bn_act_f32(const float* x, int N, int C, int HW, const float* scale, const float* shift, const float* residual,
                       int relu, float* y, void* stream);

/* The stem tail in training mode:  maxpool3x3/2/1( relu( batch_norm_train(x) ) )   (bnn/models/resnet.py:150-153) without
 * ever writing the normalised tensor: `pooled` [N, C, Hp, Wp] (Hp = (H - 1) / 2 + 1) and ONE BYTE per pooled output,
 * `code` = 3 * dy + dx of the winning window position, are all the forward leaves behind (the library: the fp32
 * BatchNorm output, the ReLU output and int64 pooling indices).  The backward routes gy through the codes inside the
 * BatchNorm reductions (a winner of value 0 passes nothing: ReLU).  Workspace: bnn_hip_bn_train_workspace_bytes(N, C,
 * H * W).  Same statistics semantics as bnn_hip_bn_train_forward_f32.                                                   */
int bnn_hip_bn_relu_maxpool_train_forward_f32(const float* x, int N, int C, int H, int W, const float* gamma,
                                              const float* beta, float eps, float momentum, float* running_mean,
                                              float* running_var, float* pooled, uint8_t* code, float* save_mean,
                                              float* save_invstd, void* workspace, void* stream);
int bnn_hip_bn_relu_maxpool_train_backward_f32(const float* gy, const float* pooled, const uint8_t* code, const float* x,
                                               const float* save_mean, const float* save_invstd, const float* gamma,
                                               int N, int C, int H, int W, float* dx, float* dgamma, float* dbeta,
                                               void* workspace, void* stream);

/* Roofline calibration: runs a register-only instruction stream on every CU at full
 * occupancy and reports the sustained 32-bit lane-ops/s.  mode: 0 = v_bitop3_b32 +
 * v_bcnt_u32_b32 (the hot loop's pair), 1 = v_xor_b32 + v_bcnt_u32_b32, 2 = bcnt only,
 * 3 = bitop3 only, 4 = xor only, 5 = v_fma_f32 only, 6 = v_add_u32 only, 7 = v_and_b32 with
 * VGPR-only operands, 8 = that + bcnt, 9 = v_xor_b32 VGPR-only, 10 = v_and_b32 (scalar operand) + bcnt.
 * Synchronous (hipEvents on `stream`, temporary hipMalloc); not part of the inference path. */
int bnn_hip_probe_int_alu(int mode, int iters, double* lane_ops_per_s, double* elapsed_ms,
                          void* stream);

/* The engine (shader) clock at this moment, MHz: one wave reads the shader-clock counter (s_memtime) and the
 * constant-rate counter (s_memrealtime) around `spin_iters` dependent ALU steps (10000 ~ 20 us).  Issued on `stream`
 * right behind a timed region it reports the clock those kernels ran at (DVFS moves on a millisecond scale) — the
 * figure bench.py prints next to every sustained number.  Synchronous; not part of the inference path.            */
int bnn_hip_probe_clock(int spin_iters, double* shader_mhz, double* elapsed_us, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BNN_HIP_H_ */
