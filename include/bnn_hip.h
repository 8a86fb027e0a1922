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
 *   stored as two bit-planes, pixel-major ("NHWC of words"):
 *       P[n][y][x][cw64]  bit c%64 of word c/64 set  <=>  x[n][c][y][x] > 0
 *       M[n][y][x][cw64]  bit c%64 of word c/64 set  <=>  x[n][c][y][x] < 0
 *   with cw64 = ceil(C/64) uint64 words per pixel; pad bits are 0 in both planes.
 *   NaN -> neither plane (torch.sign(nan) == 0), denormals keep their sign.
 *   nzc[n][y][x] (uint16) = popcount(P|M) over the pixel = number of non-zero channels.
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
 *       dot = nzc_window - 2*D                      (= sum of sign(x)*sign(w))
 *   and the float result  out = fmaf(alpha[o], (float)dot, bias[o]) [* post_scale[o]].
 */
#ifndef BNN_HIP_H_
#define BNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BNN_HIP_ABI_VERSION 1
#define BNN_HIP_OCB 32 /* output channels per weight block (padding granularity of O) */

typedef enum bnn_hip_status {
  BNN_HIP_OK = 0,
  BNN_HIP_ERR_INVALID_ARG = -1,  /* null pointer, non-positive size, misaligned buffer        */
  BNN_HIP_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels implement                   */
  BNN_HIP_ERR_LAUNCH = -3,       /* hipGetLastError() != hipSuccess after a launch             */
  BNN_HIP_ERR_TOO_LARGE = -4,    /* a tensor exceeds the 2^31-element addressing of one launch */
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

/* sign(x) as two bit planes.  x: float32 NCHW contiguous.  P, M: N*H*W*ceil(C/64)
 * uint64 each, 16-byte aligned.  nzc: N*H*W uint16 (C <= 65535).                  */
int bnn_hip_pack_act_f32(const float* x, int N, int C, int H, int W,
                         uint64_t* P, uint64_t* M, uint16_t* nzc, void* stream);

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

/* Binary convolution on packed operands.  out: float32 [N,O,Ho,Wo] contiguous.
 *   out[n,o,y,x] = fmaf(alpha[o], dot, bias ? bias[o] : 0) * (post_scale ? post_scale[o] : 1)
 * wnz may be NULL unless BNN_HIP_FLAG_WEIGHT_ZEROS is set.                         */
int bnn_hip_bconv2d(const bnn_hip_conv_desc* d,
                    const uint64_t* P, const uint64_t* M, const uint16_t* nzc,
                    const uint32_t* wbits, const uint32_t* wnz,
                    const float* alpha, const float* bias, const float* post_scale,
                    float* out, void* stream);

/* Same traversal, raw integer result: dot[n,o,y,x] (int32) — the bit-exact target
 * of the popcount path against the emulated-integer oracle.                        */
int bnn_hip_bconv2d_dot(const bnn_hip_conv_desc* d,
                        const uint64_t* P, const uint64_t* M, const uint16_t* nzc,
                        const uint32_t* wbits, const uint32_t* wnz,
                        int32_t* dot, void* stream);

/* Binary fully-connected layer: x packed as [B][ceil(F/64)] planes (pack_act with
 * H=W=1), weight packed with KH=KW=1.  out: float32 [B,O].                         */
int bnn_hip_blinear(int B, int F, int O,
                    const uint64_t* P, const uint64_t* M, const uint16_t* nzc,
                    const uint32_t* wbits, const uint32_t* wnz, int weight_zeros,
                    const float* alpha, const float* bias, const float* post_scale,
                    float* out, void* stream);

/* Convenience: fp32 NCHW in -> fp32 NCHW out in one call (pack + conv).  `workspace`
 * must hold bnn_hip_conv_workspace_bytes(d) bytes, 16-byte aligned.                */
size_t bnn_hip_conv_workspace_bytes(const bnn_hip_conv_desc* d);
int bnn_hip_bconv2d_f32(const bnn_hip_conv_desc* d, const float* x,
                        const uint32_t* wbits, const uint32_t* wnz,
                        const float* alpha, const float* bias, const float* post_scale,
                        float* out, void* workspace, void* stream);

/* Roofline calibration: runs a register-only v_bitop3_b32 + v_bcnt_u32_b32 chain on
 * every CU and reports the sustained 32-bit lane-ops/s (two ops per loop step).
 * Synchronous (uses hipEvents on `stream`); not part of the inference path.        */
int bnn_hip_probe_int_alu(int iters, double* lane_ops_per_s, double* elapsed_ms, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BNN_HIP_H_ */
