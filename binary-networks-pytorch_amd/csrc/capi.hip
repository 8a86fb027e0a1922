// capi.hip — the extern "C" boundary declared in include/bnn_hip.h.
// Argument validation, geometry, workspace carving and launch bookkeeping live here;
// kernels live in pack_act.hip / pack_weight.hip / bconv.hip.
#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

#include "bnn_dev.h"

namespace {

std::atomic<uint64_t> g_launches{0};

// roctx ranges around every launching entry point (the reference's users profile with named ranges):
// off unless BNN_HIP_ROCTX=1 is in the environment when the first entry point runs.  The marker library is dlopen'ed —
// libbnn_hip.so has no link-time dependency on a profiler — and a box without it silently runs without ranges.
// rocprofv3 --marker-trace shows one range per C-ABI call, named after it.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* e = std::getenv("BNN_HIP_ROCTX");
    if (!e || e[0] != '1') return;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
        push = nullptr; pop = nullptr;
      }
    }
  }
};
const Roctx& roctx() { static const Roctx r; return r; }
struct Range {
  bool on;
  explicit Range(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
  ~Range() { if (on) roctx().pop(); }
  Range(const Range&) = delete;
  Range& operator=(const Range&) = delete;
};
#define BNN_RANGE() Range bnn_range_(__func__)

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr long long kMaxElems = (1LL << 31) - 1;
// the convolution kernels address every tensor as uniform base + 32-bit BYTE offset per lane: fp32 / int32 tensors
// of one launch stay below 2^30 elements, packed planes (uint64 per 64 channels per pixel) below 2^29 words
constexpr long long kMaxConvElems = (1LL << 30) - 1;
constexpr long long kMaxPlaneWords = (1LL << 29) - 1;
// tensors addressed through a sized 32-bit buffer descriptor (stem input / output, one-launch layer input): the
// out-of-range marker offsets the kernels use for dead lanes (0xFFFFFFF0 + small immediates) must stay out of range
constexpr long long kMaxDescBytes = 0xFFFFFE00LL;

// Size arithmetic on caller-supplied 32-bit integers: products of up to five of them.  Computed in 64 bits with
// saturation (all factors are checked positive first), so that a hostile descriptor cannot wrap a limit check
// (found by tools/fuzz/capi_fuzz.hip under UBSan: N * H * W * words overflowed `long long` for N = H = W = 2^24).
constexpr long long kSat = 1LL << 62;
inline long long mulc(long long a, long long b) { return (a <= 0 || b <= 0) ? 0 : (a > kSat / b ? kSat : a * b); }
inline long long mulc(long long a, long long b, long long c) { return mulc(mulc(a, b), c); }
inline long long mulc(long long a, long long b, long long c, long long d) { return mulc(mulc(a, b, c), d); }

// output extent of a convolution dimension, or 0 when it is empty / does not fit an int
int out_dim(int in, int k, int s, int p, int d) {
  const long long span = (long long)in + 2LL * p - (long long)d * (k - 1) - 1;
  if (span < 0) return 0;
  const long long o = span / s + 1;
  return o > 0x7fffffffLL ? 0 : (int)o;
}

int check_desc(const bnn_hip_conv_desc* d, int* Ho, int* Wo) {
  if (!d) return BNN_HIP_ERR_INVALID_ARG;
  if (d->N <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0 || d->O <= 0 || d->KH <= 0 || d->KW <= 0)
    return BNN_HIP_ERR_INVALID_ARG;
  if (d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0 || d->dil_h <= 0 ||
      d->dil_w <= 0)
    return BNN_HIP_ERR_INVALID_ARG;
  const int ho = out_dim(d->H, d->KH, d->stride_h, d->pad_h, d->dil_h);
  const int wo = out_dim(d->W, d->KW, d->stride_w, d->pad_w, d->dil_w);
  if (ho <= 0 || wo <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(d->N, d->H, d->W, ((long long)d->C + 63) / 64) > kMaxPlaneWords) return BNN_HIP_ERR_TOO_LARGE;
  if (mulc(d->N, ho, wo, ((long long)d->O + 63) / 64) > kMaxPlaneWords) return BNN_HIP_ERR_TOO_LARGE;
  if (mulc(d->N, d->O, ho, wo) > kMaxConvElems) return BNN_HIP_ERR_TOO_LARGE;
  // weight words of the layer (o_pad x taps x cw32) and the receptive field must be addressable too
  if (mulc(((long long)d->O + 31) / 32 * 32, d->KH, d->KW, 2 * (((long long)d->C + 63) / 64)) > kMaxElems)
    return BNN_HIP_ERR_TOO_LARGE;
  *Ho = ho;
  *Wo = wo;
  return BNN_HIP_OK;
}

// Shared tail of every conv entry point: fills the geometry part of `p` and launches.
int run_conv(const bnn_hip_conv_desc* d, const uint64_t* P, const uint64_t* M, const uint32_t* wbits,
             const uint32_t* wnz, bnn::ConvP p, void* stream) {
  int Ho = 0, Wo = 0;
  const int st = check_desc(d, &Ho, &Wo);
  if (st != BNN_HIP_OK) return st;
  if (!P || !M || !wbits) return BNN_HIP_ERR_INVALID_ARG;
  if (!p.out && !(p.outP && p.outM)) return BNN_HIP_ERR_INVALID_ARG;
  if (!p.raw && !p.alpha) return BNN_HIP_ERR_INVALID_ARG;
  if ((p.bn_a == nullptr) != (p.bn_b == nullptr)) return BNN_HIP_ERR_INVALID_ARG;
  if ((p.outP == nullptr) != (p.outM == nullptr)) return BNN_HIP_ERR_INVALID_ARG;
  if ((p.pack_a == nullptr) != (p.pack_b == nullptr)) return BNN_HIP_ERR_INVALID_ARG;
  if (p.c_tot == 0) { p.c_off = 0; p.c_tot = d->O; }
  if (p.c_off < 0 || p.c_tot <= 0 || (long long)p.c_off + d->O > p.c_tot) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(d->N, p.c_tot, Ho, Wo) > kMaxConvElems) return BNN_HIP_ERR_TOO_LARGE;
  if ((d->flags & BNN_HIP_FLAG_WEIGHT_ZEROS) && !wnz) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(P, 16) || !aligned(M, 16) || !aligned(wbits, 16)) return BNN_HIP_ERR_INVALID_ARG;
  if (p.outP && (!aligned(p.outP, 8) || !aligned(p.outM, 8))) return BNN_HIP_ERR_INVALID_ARG;
  bnn_hip_wlayout L;
  bnn_hip_weight_layout(d->O, d->C, d->KH, d->KW, &L);
  p.P = reinterpret_cast<const uint32_t*>(P);
  p.M = reinterpret_cast<const uint32_t*>(M);
  p.W = wbits;
  p.Z = wnz;
  p.N = d->N; p.H = d->H; p.Wd = d->W; p.Ho = Ho; p.Wo = Wo; p.O = d->O;
  p.KH = d->KH; p.KW = d->KW; p.sh = d->stride_h; p.sw = d->stride_w;
  p.ph = d->pad_h; p.pw = d->pad_w; p.dh = d->dil_h; p.dw = d->dil_w;
  p.cw32 = L.cw32; p.cwc = L.cwc; p.nchunk = L.nchunk;
  p.npix = d->N * Ho * Wo;
  p.C = d->C;
  if (p.ds_P) {  // folded shortcut convolution (bnn_hip_epilogue sc_*)
    if (!p.ds_W || !p.ds_alpha || !p.ds_a || !p.ds_b || p.res || p.ds_C <= 0) return BNN_HIP_ERR_INVALID_ARG;
    if (!aligned(p.ds_P, 8) || !aligned(p.ds_W, 16)) return BNN_HIP_ERR_INVALID_ARG;
    if (!bnn::ds_fold_applies(p, d->flags)) return BNN_HIP_ERR_UNSUPPORTED;
    if (p.ds_inW > 0 && ((p.ds_inH + 1) / 2 != Ho || (p.ds_inW + 1) / 2 != Wo)) return BNN_HIP_ERR_INVALID_ARG;
    if (p.ds_inW > 0 && mulc(d->N, ((long long)p.ds_C + 63) / 64, p.ds_inH, p.ds_inW) > kMaxPlaneWords) return BNN_HIP_ERR_TOO_LARGE;
  }
  g_launches.fetch_add(1, std::memory_order_relaxed);
  Range range(p.raw ? "bnn_hip_bconv2d_dot" : p.ds_P ? "bnn_hip_bconv2d_fused+shortcut"
              : (p.bn_a || p.res || p.outP || p.relu || p.prelu) ? "bnn_hip_bconv2d_fused" : "bnn_hip_bconv2d");
  return bnn::launch_bconv(p, d->flags, static_cast<hipStream_t>(stream));
}

bnn::ConvP empty_convp() {
  bnn::ConvP p;
  std::memset(&p, 0, sizeof(p));
  return p;
}

}  // namespace

extern "C" {

int bnn_hip_abi_version(void) { return BNN_HIP_ABI_VERSION; }

const char* bnn_hip_status_string(int status) {
  switch (status) {
    case BNN_HIP_OK: return "ok";
    case BNN_HIP_ERR_INVALID_ARG: return "invalid argument (null/misaligned pointer or non-positive size)";
    case BNN_HIP_ERR_UNSUPPORTED: return "unsupported shape";
    case BNN_HIP_ERR_LAUNCH: return "HIP kernel launch failed";
    case BNN_HIP_ERR_TOO_LARGE: return "tensor exceeds 2^31-1 elements per launch; split the batch";
    case BNN_HIP_ERR_NO_DEVICE: return "no usable HIP device";
    default: return "unknown status";
  }
}

uint64_t bnn_hip_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

int bnn_hip_device_info(int device, bnn_hip_devinfo* out) {
  if (!out) return BNN_HIP_ERR_INVALID_ARG;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return BNN_HIP_ERR_NO_DEVICE;
  std::memset(out, 0, sizeof(*out));
  std::strncpy(out->name, prop.name, sizeof(out->name) - 1);
  std::strncpy(out->arch, prop.gcnArchName, sizeof(out->arch) - 1);
  out->compute_units = prop.multiProcessorCount;
  out->clock_khz = prop.clockRate;
  out->mem_clock_khz = prop.memoryClockRate;
  out->mem_bus_bits = prop.memoryBusWidth;
  out->wavefront = prop.warpSize;
  out->lds_bytes_per_block = (int32_t)prop.sharedMemPerBlock;
  out->total_mem_bytes = (int64_t)prop.totalGlobalMem;
  out->l2_bytes = prop.l2CacheSize;
  return BNN_HIP_OK;
}

int bnn_hip_act_words(int C) { return C > 0 ? (int)(((long long)C + 63) / 64) : BNN_HIP_ERR_INVALID_ARG; }

int bnn_hip_weight_layout(int O, int C, int KH, int KW, bnn_hip_wlayout* out) {
  if (!out || O <= 0 || C <= 0 || KH <= 0 || KW <= 0) return BNN_HIP_ERR_INVALID_ARG;
  const long long cw32 = 2 * (((long long)C + 63) / 64), o_pad = ((long long)O + BNN_HIP_OCB - 1) / BNN_HIP_OCB * BNN_HIP_OCB;
  const long long taps = mulc(KH, KW);
  // the packed weight (one bit per weight) addresses words with 31-bit indices; taps and the padded channel count
  // are ints of the layout struct
  if (taps > 0x7fffffffLL || o_pad > 0x7fffffffLL || mulc(o_pad, taps, cw32) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  out->cw32 = (int32_t)cw32;
  out->cwc = bnn::choose_cwc(out->cw32, KH, KW);
  out->nchunk = out->cw32 / out->cwc;
  out->taps = (int32_t)taps;
  out->o_pad = (int32_t)o_pad;
  out->reserved = 0;
  out->n_words = o_pad * taps * cw32;
  return BNN_HIP_OK;
}

int bnn_hip_pack_act_f32(const float* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M,
                         void* stream) {
  if (!x || !P || !M || N <= 0 || C <= 0 || H <= 0 || W <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, H, W) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (((long long)C + 63) / 64 > 65535) return BNN_HIP_ERR_UNSUPPORTED;  // grid.y limit
  if (!aligned(P, 8) || !aligned(M, 8) || !aligned(x, 4)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_pack_act(x, N, C, H, W, P, M, static_cast<hipStream_t>(stream));
}

int bnn_hip_pack_act_f16(const void* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M,
                         void* stream) {
  if (!x || !P || !M || N <= 0 || C <= 0 || H <= 0 || W <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, H, W) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (((long long)C + 63) / 64 > 65535) return BNN_HIP_ERR_UNSUPPORTED;  // grid.y limit
  if (!aligned(P, 8) || !aligned(M, 8) || !aligned(x, 2)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_pack_act_f16(x, N, C, H, W, P, M, static_cast<hipStream_t>(stream));
}

int bnn_hip_bn_act_pack_f32(const float* x, int N, int C, int H, int W, const float* bn_scale,
                            const float* bn_shift, int relu, uint64_t* P, uint64_t* M, void* stream) {
  if (!x || !P || !M || N <= 0 || C <= 0 || H <= 0 || W <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if ((bn_scale == nullptr) != (bn_shift == nullptr)) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, H, W) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (((long long)C + 63) / 64 > 65535) return BNN_HIP_ERR_UNSUPPORTED;  // grid.y limit
  if (!aligned(P, 8) || !aligned(M, 8) || !aligned(x, 4)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_bn_act_pack(x, N, C, H, W, bn_scale, bn_shift, relu, P, M,
                                 static_cast<hipStream_t>(stream));
}

int bnn_hip_avgpool_pack_f32(const float* x, int N, int C, int H, int W, int k, uint64_t* P,
                             uint64_t* M, void* stream) {
  if (!x || !P || !M || N <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, H, W) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (2 * (((long long)C + 63) / 64) > 65535) return BNN_HIP_ERR_UNSUPPORTED;
  if (!aligned(P, 8) || !aligned(M, 8)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_avgpool_pack(x, N, C, H, W, k, P, M, static_cast<hipStream_t>(stream));
}

int bnn_hip_avgpool2_bn_pack2_f32(const float* x, int N, int C, int H, int W, const float* a1, const float* b1, int relu1,
                                  uint64_t* P1, uint64_t* M1, const float* a2, const float* b2, int relu2, uint64_t* P2,
                                  uint64_t* M2, float* out_f32, void* stream) {
  if (!x || !a1 || !b1 || !P1 || !M1 || N <= 0 || C <= 0 || H <= 0 || W <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (H % 2 || W % 2) return BNN_HIP_ERR_UNSUPPORTED;
  if ((a2 == nullptr) != (b2 == nullptr) || (a2 && (!P2 || !M2))) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, C, H, W) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (2 * (((long long)C + 63) / 64) > 65535) return BNN_HIP_ERR_UNSUPPORTED;
  if (!aligned(x, 8) || !aligned(P1, 8) || !aligned(M1, 8) || (a2 && (!aligned(P2, 8) || !aligned(M2, 8))) ||
      (out_f32 && !aligned(out_f32, 4)))
    return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_avgpool2_bn_pack2(x, N, C, H, W, a1, b1, relu1, P1, M1, a2, b2, relu2, P2, M2, out_f32,
                                       static_cast<hipStream_t>(stream));
}

int bnn_hip_orpool_packed(const uint64_t* P, int N, int C, int H, int W, int k, uint64_t* out_P,
                          uint64_t* out_M, void* stream) {
  if (!P || !out_P || !out_M || N <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, ((long long)C + 63) / 64, H, W) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (!aligned(P, 8) || !aligned(out_P, 8) || !aligned(out_M, 8)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_orpool_packed(P, N, C, H, W, k, out_P, out_M, static_cast<hipStream_t>(stream));
}

int bnn_hip_bn_relu_maxpool_pack_f32(const float* x, int N, int C, int H, int W,
                                     const float* bn_scale, const float* bn_shift, int relu, int k,
                                     int stride, int pad, float* out_f32, uint64_t* P, uint64_t* M,
                                     void* stream) {
  if (!x || N <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0 || pad < 0)
    return BNN_HIP_ERR_INVALID_ARG;
  if (!out_f32 && !P) return BNN_HIP_ERR_INVALID_ARG;
  if ((P == nullptr) != (M == nullptr)) return BNN_HIP_ERR_INVALID_ARG;
  if ((bn_scale == nullptr) != (bn_shift == nullptr)) return BNN_HIP_ERR_INVALID_ARG;
  if (2LL * pad > k || (long long)H + 2LL * pad < k || (long long)W + 2LL * pad < k) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, C, H, W) > 4 * kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (P && (!aligned(P, 8) || !aligned(M, 8))) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_bn_relu_maxpool_pack(x, N, C, H, W, bn_scale, bn_shift, relu, k, stride, pad,
                                          out_f32, P, M, static_cast<hipStream_t>(stream));
}

int bnn_hip_stem7x7_bn_relu_pool_pack_f32(const float* x, const float* w, const float* bn_scale,
                                          const float* bn_shift, int N, int H, int W, int flags,
                                          float* out_f32, uint64_t* P, uint64_t* M, void* stream) {
  if (!x || !w || !bn_scale || !bn_shift || N <= 0 || H <= 0 || W <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (!out_f32 && !P) return BNN_HIP_ERR_INVALID_ARG;
  if (flags & ~(BNN_HIP_STEM_EXACT_FP32 | BNN_HIP_STEM_FP16)) return BNN_HIP_ERR_INVALID_ARG;
  if ((flags & BNN_HIP_STEM_EXACT_FP32) && (flags & BNN_HIP_STEM_FP16)) return BNN_HIP_ERR_INVALID_ARG;
  if ((P == nullptr) != (M == nullptr)) return BNN_HIP_ERR_INVALID_ARG;
  if (P && (!aligned(P, 8) || !aligned(M, 8))) return BNN_HIP_ERR_INVALID_ARG;
  // the stem kernels address x and out through 32-bit buffer descriptors (range-checked loads are the zero padding,
  // out-of-range stores are dropped): every tensor of a launch stays below 2^32 - 16 bytes (4 GiB: ~7100 images of
  // 224 x 224 in, ~5300 out) — larger batches are split by the caller (hipops.stem7x7 does)
  {
    const long long hc = (H - 1) / 2 + 1, wc = (W - 1) / 2 + 1, hp = (hc - 1) / 2 + 1, wp = (wc - 1) / 2 + 1;
    if (mulc(N, 12, H, W) > kMaxDescBytes || mulc(N, 256, hp, wp) > kMaxDescBytes)
      return BNN_HIP_ERR_TOO_LARGE;
  }
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_stem(x, w, bn_scale, bn_shift, N, H, W, flags, out_f32, P, M,
                          static_cast<hipStream_t>(stream));
}

int bnn_hip_stem7x7_bn_relu_pool_pack_affine_f32(const float* x, const float* w, const float* bn_scale,
                                                 const float* bn_shift, const float* pack_scale, const float* pack_shift,
                                                 int N, int H, int W, int flags, float* out_f32, uint64_t* P, uint64_t* M,
                                                 void* stream) {
  if (!x || !w || !bn_scale || !bn_shift || !pack_scale || !pack_shift || !P || !M || N <= 0 || H <= 0 || W <= 0)
    return BNN_HIP_ERR_INVALID_ARG;
  if (flags & ~BNN_HIP_STEM_FP16) return BNN_HIP_ERR_INVALID_ARG;   // (the exact-fp32 stem has no such variant)
  if (!aligned(P, 8) || !aligned(M, 8)) return BNN_HIP_ERR_INVALID_ARG;
  {
    const long long hc = (H - 1) / 2 + 1, wc = (W - 1) / 2 + 1, hp = (hc - 1) / 2 + 1, wp = (wc - 1) / 2 + 1;
    if (mulc(N, 12, H, W) > kMaxDescBytes || mulc(N, 256, hp, wp) > kMaxDescBytes) return BNN_HIP_ERR_TOO_LARGE;
  }
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_stem_rows_aff(x, w, bn_scale, bn_shift, pack_scale, pack_shift, N, H, W, (flags & BNN_HIP_STEM_FP16) != 0,
                                   out_f32, P, M, static_cast<hipStream_t>(stream));
}

int bnn_hip_stem7x7_conv_f32(const float* x, const float* w, int N, int H, int W, int flags, float* out,
                             void* stream) {
  if (!x || !w || !out || N <= 0 || H <= 0 || W <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (flags & ~BNN_HIP_STEM_FP16) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(x, 4) || !aligned(w, 4) || !aligned(out, 4)) return BNN_HIP_ERR_INVALID_ARG;
  {  // 32-bit buffer descriptors, as in the fused stem: input and the [N, 64, Hc, Wc] output below 2^32 - 512 bytes
    const long long hc = (H - 1) / 2 + 1, wc = (W - 1) / 2 + 1;
    if (mulc(N, 12, H, W) > kMaxDescBytes || mulc(N, 256, hc, wc) > kMaxDescBytes) return BNN_HIP_ERR_TOO_LARGE;
  }
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_stem_conv(x, w, N, H, W, (flags & BNN_HIP_STEM_FP16) != 0, out, static_cast<hipStream_t>(stream));
}

int bnn_hip_avgpool2x2_backward_f32(const float* gy, int N, int C, int Ho, int Wo, float* gx, void* stream) {
  if (!gy || !gx || N <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(gy, 4) || !aligned(gx, 4)) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, C, Ho, Wo) > kMaxElems / 4) return BNN_HIP_ERR_TOO_LARGE;   // gx holds four times as many elements
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_avgpool2x2_bwd(gy, N, C, Ho, Wo, gx, static_cast<hipStream_t>(stream));
}

size_t bnn_hip_stem7x7_wgrad_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  const long long hc = (H - 1) / 2 + 1, wc = (W - 1) / 2 + 1;
  if (mulc(N, 3, H, W) > kMaxElems || mulc(N, 64, hc, wc) > kMaxElems) return 0;
  return bnn::stem_wgrad_workspace_bytes(N, H, W);
}

int bnn_hip_stem7x7_wgrad_f32(const float* x, const float* dy, int N, int H, int W, float* workspace,
                              size_t workspace_bytes, float* dw, void* stream) {
  if (!x || !dy || !dw || !workspace || N <= 0 || H <= 0 || W <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(x, 4) || !aligned(dy, 4) || !aligned(dw, 4) || !aligned(workspace, 4)) return BNN_HIP_ERR_INVALID_ARG;
  {
    const long long hc = (H - 1) / 2 + 1, wc = (W - 1) / 2 + 1;
    if (mulc(N, 3, H, W) > kMaxElems || mulc(N, 64, hc, wc) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  }
  if (!bnn::stem_wgrad_supported(H, W)) return BNN_HIP_ERR_UNSUPPORTED;
  if (workspace_bytes < bnn::stem_wgrad_workspace_bytes(N, H, W)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(2, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_stem_wgrad(x, dy, N, H, W, workspace, dw, static_cast<hipStream_t>(stream));
}

int bnn_hip_avgpool_fc_f32(const float* x, int N, int C, int HW, const float* w_t, const float* bias, int O,
                           float* out, void* stream) {
  if (!x || !w_t || !out || N <= 0 || C <= 0 || HW <= 0 || O <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(x, 4) || !aligned(w_t, 4) || !aligned(out, 4)) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, C, HW) > 4 * kMaxElems || mulc(N, O) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_avgpool_fc(x, w_t, bias, out, N, C, HW, O, static_cast<hipStream_t>(stream));
}

size_t bnn_hip_avgpool_fc_workspace_bytes(int N, int C) {
  if (N <= 0 || C <= 0) return 0;
  return bnn::avgpool_fc_workspace_bytes(N, C);
}

int bnn_hip_avgpool_fc_ws_f32(const float* x, int N, int C, int HW, const float* w_t, const float* bias, int O,
                              float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !w_t || !out || N <= 0 || C <= 0 || HW <= 0 || O <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(x, 4) || !aligned(w_t, 4) || !aligned(out, 4)) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, C, HW) > 4 * kMaxElems || mulc(N, O) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (!bnn::avgpool_fc_ws_supported(C, HW)) {  // the one-kernel form needs no workspace
    g_launches.fetch_add(1, std::memory_order_relaxed);
    BNN_RANGE();
    return bnn::launch_avgpool_fc(x, w_t, bias, out, N, C, HW, O, static_cast<hipStream_t>(stream));
  }
  if (!workspace || !aligned(workspace, 16) || workspace_bytes < bnn::avgpool_fc_workspace_bytes(N, C))
    return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(2, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_avgpool_fc_ws(x, w_t, bias, out, static_cast<float*>(workspace), N, C, HW, O,
                                   static_cast<hipStream_t>(stream));
}

static bool grad_ks_ok(int ksize) { return ksize == 1 || ksize == 3; }

size_t bnn_hip_grad_weight_pack_bytes(int O, int C, int ksize) {
  return (O > 0 && C > 0 && grad_ks_ok(ksize)) ? bnn::grad_weight_pack_bytes(O, C, ksize) : 0;
}

int bnn_hip_grad_pack_weight_f32(const float* w_hat, int O, int C, int ksize, void* packed, float* alpha,
                                 void* stream) {
  if (!w_hat || !packed || !alpha || O <= 0 || C <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (!grad_ks_ok(ksize)) return BNN_HIP_ERR_UNSUPPORTED;
  if (!aligned(packed, 16)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(2, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_grad_pack_weight(w_hat, O, C, ksize, packed, alpha, static_cast<hipStream_t>(stream));
}

static int check_grad_shape(int N, int O, int C, int H, int W, int ksize, int stride) {
  if (N <= 0 || O <= 0 || C <= 0 || H <= 0 || W <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (!grad_ks_ok(ksize) || (stride != 1 && stride != 2) || (ksize == 1 && stride != 1)) return BNN_HIP_ERR_UNSUPPORTED;
  if (W > 64) return BNN_HIP_ERR_UNSUPPORTED;
  if (mulc(N, O, H, W) > kMaxElems || mulc(N, C, H, W) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  return BNN_HIP_OK;
}

int bnn_hip_bconv_grad_input_f32(const float* g, const float* alpha, const void* packed, const float* x, float* gx,
                                 int N, int O, int C, int H, int W, int ksize, int stride, void* stream) {
  if (!g || !alpha || !packed || !x || !gx) return BNN_HIP_ERR_INVALID_ARG;
  const int st = check_grad_shape(N, O, C, H, W, ksize, stride);
  if (st != BNN_HIP_OK) return st;
  if (!aligned(packed, 16) || !aligned(x, 4) || !aligned(g, 4) || !aligned(gx, 4)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_dgrad(g, alpha, packed, x, 0, gx, N, O, C, H, W, ksize, stride, static_cast<hipStream_t>(stream));
}

int bnn_hip_bconv_grad_input_packed_f32(const float* g, const float* alpha, const void* packed, const uint64_t* T,
                                        float* gx, int N, int O, int C, int H, int W, int ksize, int stride,
                                        void* stream) {
  if (!g || !alpha || !packed || !T || !gx) return BNN_HIP_ERR_INVALID_ARG;
  const int st = check_grad_shape(N, O, C, H, W, ksize, stride);
  if (st != BNN_HIP_OK) return st;
  if (!aligned(packed, 16) || !aligned(T, 8) || !aligned(g, 4) || !aligned(gx, 4)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_dgrad(g, alpha, packed, T, 1, gx, N, O, C, H, W, ksize, stride, static_cast<hipStream_t>(stream));
}

int bnn_hip_pack_act_ste_f32(const float* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M, uint64_t* T,
                             void* stream) {
  if (!x || !P || !M || !T || N <= 0 || C <= 0 || H <= 0 || W <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, H, W) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (((long long)C + 63) / 64 > 65535) return BNN_HIP_ERR_UNSUPPORTED;  // grid.y limit
  if (!aligned(P, 8) || !aligned(M, 8) || !aligned(T, 8) || !aligned(x, 4)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_pack_ste(x, N, C, H, W, P, M, T, static_cast<hipStream_t>(stream));
}

int bnn_hip_bconv_grad_weight_splits(int N, int O, int C, int ksize) {
  return (N > 0 && O > 0 && C > 0 && grad_ks_ok(ksize)) ? bnn::grad_wgrad_splits(N, O, C, ksize)
                                                         : BNN_HIP_ERR_INVALID_ARG;
}

int bnn_hip_bconv_grad_weight_f32(const float* g, const float* x, float* partial, int splits, int N, int O, int C,
                                  int H, int W, int ksize, int stride, void* stream) {
  if (!g || !x || !partial || !aligned(g, 4) || !aligned(x, 4) || !aligned(partial, 4)) return BNN_HIP_ERR_INVALID_ARG;
  const int st = check_grad_shape(N, O, C, H, W, ksize, stride);
  if (st != BNN_HIP_OK) return st;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_wgrad(g, x, nullptr, 0, partial, splits, N, O, C, H, W, ksize, stride,
                           static_cast<hipStream_t>(stream));
}

int bnn_hip_bconv_grad_weight_packed_f32(const float* g, const uint64_t* P, const uint64_t* M, float* partial, int splits,
                                         int N, int O, int C, int H, int W, int ksize, int stride, void* stream) {
  if (!g || !P || !M || !partial || !aligned(P, 8) || !aligned(M, 8) || !aligned(g, 4) || !aligned(partial, 4))
    return BNN_HIP_ERR_INVALID_ARG;
  const int st = check_grad_shape(N, O, C, H, W, ksize, stride);
  if (st != BNN_HIP_OK) return st;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_wgrad(g, P, M, 1, partial, splits, N, O, C, H, W, ksize, stride, static_cast<hipStream_t>(stream));
}

int bnn_hip_pack_weight_f32(const float* w, int O, int C, int KH, int KW, int center,
                            int compute_alpha, uint32_t* wbits, uint32_t* wnz, float* alpha,
                            int32_t* zero_flag, void* stream) {
  if (!w || !wbits || !wnz || !alpha || !zero_flag) return BNN_HIP_ERR_INVALID_ARG;
  bnn_hip_wlayout L;
  const int st = bnn_hip_weight_layout(O, C, KH, KW, &L);
  if (st != BNN_HIP_OK) return st;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_pack_weight(w, O, C, KH, KW, center, compute_alpha, L, wbits, wnz, alpha,
                                 zero_flag, static_cast<hipStream_t>(stream));
}

int bnn_hip_bconv2d(const bnn_hip_conv_desc* d, const uint64_t* P, const uint64_t* M,
                    const uint32_t* wbits, const uint32_t* wnz, const float* alpha,
                    const float* bias, const float* post_scale, float* out, void* stream) {
  bnn::ConvP p = empty_convp();
  p.alpha = alpha; p.bias = bias; p.scale = post_scale; p.out = out;
  return run_conv(d, P, M, wbits, wnz, p, stream);
}

int bnn_hip_bconv2d_fused(const bnn_hip_conv_desc* d, const uint64_t* P, const uint64_t* M,
                          const uint32_t* wbits, const uint32_t* wnz, const bnn_hip_epilogue* e,
                          void* stream) {
  if (!e) return BNN_HIP_ERR_INVALID_ARG;
  bnn::ConvP p = empty_convp();
  p.alpha = e->alpha; p.bias = e->bias; p.scale = e->post_scale;
  p.bn_a = e->bn_scale; p.bn_b = e->bn_shift; p.res = e->residual; p.prelu = e->prelu;
  p.relu = e->relu != 0;
  p.out = e->out_f32;
  p.outP = reinterpret_cast<uint32_t*>(e->out_P);
  p.outM = reinterpret_cast<uint32_t*>(e->out_M);
  p.pack_a = e->pack_scale; p.pack_b = e->pack_shift;
  p.thr = e->sign_thresholds;
  if (p.thr && !aligned(p.thr, 4)) return BNN_HIP_ERR_INVALID_ARG;
  p.eflags = e->flags;
  p.c_off = e->out_c_offset; p.c_tot = e->out_c_total;
  if (e->sc_P || e->sc_wbits || e->sc_alpha || e->sc_bn_scale || e->sc_bn_shift) {
    if (!e->sc_P) return BNN_HIP_ERR_INVALID_ARG;
    p.ds_P = reinterpret_cast<const uint32_t*>(e->sc_P);
    p.ds_W = e->sc_wbits; p.ds_alpha = e->sc_alpha; p.ds_a = e->sc_bn_scale; p.ds_b = e->sc_bn_shift;
    p.ds_C = e->sc_C;
    if (e->sc_in_hw != 0) {      // un-pooled shortcut plane: (H << 16) | W of it; the kernel ORs the 2 x 2 windows
      p.ds_inH = (int)((uint32_t)e->sc_in_hw >> 16);
      p.ds_inW = (int)((uint32_t)e->sc_in_hw & 0xFFFFu);
      if (p.ds_inH <= 0 || p.ds_inW <= 0) return BNN_HIP_ERR_INVALID_ARG;
    }
  }
  return run_conv(d, P, M, wbits, wnz, p, stream);
}

int bnn_hip_shortcut_fold_supported(const bnn_hip_conv_desc* d, int sc_C) {
  int Ho = 0, Wo = 0;
  if (check_desc(d, &Ho, &Wo) != BNN_HIP_OK) return 0;
  bnn::ConvP p = empty_convp();
  bnn_hip_wlayout L;
  bnn_hip_weight_layout(d->O, d->C, d->KH, d->KW, &L);
  p.N = d->N; p.H = d->H; p.Wd = d->W; p.Ho = Ho; p.Wo = Wo; p.O = d->O; p.C = d->C;
  p.KH = d->KH; p.KW = d->KW; p.sh = d->stride_h; p.sw = d->stride_w;
  p.ph = d->pad_h; p.pw = d->pad_w; p.dh = d->dil_h; p.dw = d->dil_w;
  p.cw32 = L.cw32; p.cwc = L.cwc; p.nchunk = L.nchunk;
  p.npix = d->N * Ho * Wo;
  p.c_off = 0; p.c_tot = d->O;
  // the epilogue the fold belongs to: BatchNorm + shortcut + ReLU -> fp32 + sign planes (any non-null pointers)
  const float* f = reinterpret_cast<const float*>(&p);
  p.alpha = f; p.bn_a = f; p.bn_b = f; p.relu = true; p.out = &p;
  p.outP = reinterpret_cast<uint32_t*>(&p); p.outM = reinterpret_cast<uint32_t*>(&p);
  p.ds_C = sc_C;
  return bnn::ds_fold_applies(p, d->flags) ? 1 : 0;
}

int bnn_hip_sign_thresholds_f32(const float* alpha, const float* bias, const float* post_scale, const float* bn_scale,
                                const float* bn_shift, int O, int kmax, int32_t* thresholds, void* stream) {
  // (kmax < 2^20: the two-instruction form of the test biases its counts by 2^20 > kmax / 2, bconv_core.h kMidtBias)
  if (!alpha || !thresholds || O <= 0 || kmax <= 0 || kmax >= (1 << 20)) return BNN_HIP_ERR_INVALID_ARG;
  if ((bn_scale == nullptr) != (bn_shift == nullptr)) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(thresholds, 4)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_sign_thresholds(alpha, bias, post_scale, bn_scale, bn_shift, O, kmax, thresholds,
                                     static_cast<hipStream_t>(stream));
}

int bnn_hip_bconv2d_dot(const bnn_hip_conv_desc* d, const uint64_t* P, const uint64_t* M,
                        const uint32_t* wbits, const uint32_t* wnz, int32_t* dot, void* stream) {
  bnn::ConvP p = empty_convp();
  p.raw = true; p.out = dot;
  return run_conv(d, P, M, wbits, wnz, p, stream);
}

int bnn_hip_blinear(int B, int F, int O, const uint64_t* P, const uint64_t* M, const uint32_t* wbits,
                    const uint32_t* wnz, int weight_zeros, const float* alpha, const float* bias,
                    const float* post_scale, float* out, void* stream) {
  bnn_hip_conv_desc d;
  std::memset(&d, 0, sizeof(d));
  d.N = B; d.C = F; d.H = 1; d.W = 1; d.O = O; d.KH = 1; d.KW = 1;
  d.stride_h = d.stride_w = 1; d.dil_h = d.dil_w = 1;
  d.flags = weight_zeros ? BNN_HIP_FLAG_WEIGHT_ZEROS : 0;
  return bnn_hip_bconv2d(&d, P, M, wbits, wnz, alpha, bias, post_scale, out, stream);
}

// Geometry part of a ConvP for the one-launch layer (no packed operands).
static int fly_convp(const bnn_hip_conv_desc* d, bnn::ConvP* out) {
  int Ho = 0, Wo = 0;
  const int st = check_desc(d, &Ho, &Wo);
  if (st != BNN_HIP_OK) return st;
  if (mulc(d->N, d->C, d->H, d->W) > kMaxConvElems) return BNN_HIP_ERR_TOO_LARGE;
  // the kernel reads x through a sized buffer descriptor and marks dead lanes with offset 0xFFFFFFF0 (+ soffset): a
  // tensor within a few elements of 2^32 bytes would bring the marker in range.  UNSUPPORTED, not TOO_LARGE: the
  // two-launch form (pack_act + conv) of the same layer has no such limit and callers fall back to it
  if (mulc(d->N, d->C, d->H, d->W) * 4 > kMaxDescBytes) return BNN_HIP_ERR_UNSUPPORTED;
  bnn::ConvP p = empty_convp();
  bnn_hip_wlayout L;
  bnn_hip_weight_layout(d->O, d->C, d->KH, d->KW, &L);
  p.N = d->N; p.H = d->H; p.Wd = d->W; p.Ho = Ho; p.Wo = Wo; p.O = d->O; p.C = d->C;
  p.KH = d->KH; p.KW = d->KW; p.sh = d->stride_h; p.sw = d->stride_w;
  p.ph = d->pad_h; p.pw = d->pad_w; p.dh = d->dil_h; p.dw = d->dil_w;
  p.cw32 = L.cw32; p.cwc = L.cwc; p.nchunk = L.nchunk;
  p.npix = d->N * Ho * Wo;
  p.c_off = 0; p.c_tot = d->O;
  *out = p;
  return BNN_HIP_OK;
}

int bnn_hip_bconv2d_direct_plan(const bnn_hip_conv_desc* d, bnn_hip_fly_plan* plan) {
  if (!plan) return BNN_HIP_ERR_INVALID_ARG;
  bnn::ConvP p;
  const int st = fly_convp(d, &p);
  if (st != BNN_HIP_OK) return st;
  return bnn::fly_default_plan(p, d->flags, plan);
}

int bnn_hip_bconv2d_direct(const bnn_hip_conv_desc* d, const void* x, int x_dtype, const uint32_t* wbits,
                           const uint32_t* wnz, const float* alpha, const float* bias, const float* post_scale,
                           float* out, const bnn_hip_fly_plan* plan, void* stream) {
  if (!d || !x || !wbits || !alpha || !out) return BNN_HIP_ERR_INVALID_ARG;
  if (x_dtype != BNN_HIP_DTYPE_F32 && x_dtype != BNN_HIP_DTYPE_F16) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(x, x_dtype == BNN_HIP_DTYPE_F16 ? 2 : 4) || !aligned(wbits, 16) || !aligned(out, 4))
    return BNN_HIP_ERR_INVALID_ARG;
  if ((d->flags & BNN_HIP_FLAG_WEIGHT_ZEROS) && !wnz) return BNN_HIP_ERR_INVALID_ARG;
  bnn::ConvP p;
  const int st = fly_convp(d, &p);
  if (st != BNN_HIP_OK) return st;
  p.W = wbits; p.Z = wnz; p.alpha = alpha; p.bias = bias; p.scale = post_scale; p.out = out;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_bconv_fly(p, x, x_dtype == BNN_HIP_DTYPE_F16, d->flags, plan, static_cast<hipStream_t>(stream));
}

static bool direct_applies(const bnn_hip_conv_desc* d) {
  bnn::ConvP p;
  return fly_convp(d, &p) == BNN_HIP_OK && bnn::fly_supported(p);
}

size_t bnn_hip_conv_workspace_bytes(const bnn_hip_conv_desc* d) {
  int Ho = 0, Wo = 0;
  if (check_desc(d, &Ho, &Wo) != BNN_HIP_OK) return 0;   // (bnn_hip_bconv2d_f32 rejects such a descriptor itself)
  if (direct_applies(d)) return 0;
  const size_t npix = (size_t)d->N * d->H * d->W;
  return 2 * align_up(npix * ((d->C + 63) / 64) * sizeof(uint64_t), 256);
}

int bnn_hip_bconv2d_f32(const bnn_hip_conv_desc* d, const float* x, const uint32_t* wbits,
                        const uint32_t* wnz, const float* alpha, const float* bias,
                        const float* post_scale, float* out, void* workspace, void* stream) {
  if (!d || !x) return BNN_HIP_ERR_INVALID_ARG;
  if (direct_applies(d))
    return bnn_hip_bconv2d_direct(d, x, BNN_HIP_DTYPE_F32, wbits, wnz, alpha, bias, post_scale, out, nullptr, stream);
  if (!workspace || !aligned(workspace, 16)) return BNN_HIP_ERR_INVALID_ARG;
  int Ho, Wo;
  int st = check_desc(d, &Ho, &Wo);
  if (st != BNN_HIP_OK) return st;
  const size_t npix = (size_t)d->N * d->H * d->W;
  const size_t plane = align_up(npix * ((d->C + 63) / 64) * sizeof(uint64_t), 256);
  char* ws = static_cast<char*>(workspace);
  uint64_t* P = reinterpret_cast<uint64_t*>(ws);
  uint64_t* M = reinterpret_cast<uint64_t*>(ws + plane);
  st = bnn_hip_pack_act_f32(x, d->N, d->C, d->H, d->W, P, M, stream);
  if (st != BNN_HIP_OK) return st;
  return bnn_hip_bconv2d(d, P, M, wbits, wnz, alpha, bias, post_scale, out, stream);
}

int bnn_hip_probe_int_alu(int mode, int iters, double* lane_ops_per_s, double* elapsed_ms,
                          void* stream) {
  if (iters <= 0 || !lane_ops_per_s) return BNN_HIP_ERR_INVALID_ARG;
  return bnn::launch_probe_int_alu(mode, iters, lane_ops_per_s, elapsed_ms,
                                   static_cast<hipStream_t>(stream));
}

// ---- XNORWeightBinarizer under autograd: csrc/xnor_train.hip
int bnn_hip_xnor_weight_forward_f32(const float* w, int O, int C, int KH, int KW, int center, int compute_alpha,
                                    float* what, float* alpha, void* stream) {
  if (!w || !what || O <= 0 || C <= 0 || KH <= 0 || KW <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(KH, KW) > 1024) return BNN_HIP_ERR_UNSUPPORTED;
  if (mulc(O, C, KH, KW) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (!aligned(w, 4) || !aligned(what, 4)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_xnor_what(w, O, C, KH * KW, center != 0, compute_alpha != 0, what, alpha, static_cast<hipStream_t>(stream));
}

int bnn_hip_xnor_grad_pack_weight_f32(const float* w, int O, int C, int ksize, int center, int compute_alpha, void* packed,
                                      float* alpha, void* stream) {
  if (!w || !packed || !alpha || O <= 0 || C <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (!grad_ks_ok(ksize)) return BNN_HIP_ERR_UNSUPPORTED;
  if (mulc(O, C, ksize, ksize) > kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (!aligned(w, 4) || !aligned(packed, 16) || !aligned(alpha, 4)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_xnor_grad_pack(w, O, C, ksize, center != 0, compute_alpha != 0, packed, alpha,
                                    static_cast<hipStream_t>(stream));
}

int bnn_hip_xnor_weight_backward_f32(const float* w, const float* dwhat, int splits, int O, int C, int KH, int KW,
                                     int center, int compute_alpha, float* dw, void* stream) {
  if (!w || !dwhat || !dw || O <= 0 || C <= 0 || KH <= 0 || KW <= 0 || splits <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(KH, KW) > 1024) return BNN_HIP_ERR_UNSUPPORTED;
  if (mulc(O, C, KH, KW) > kMaxElems || mulc(mulc(O, C, KH, KW), splits) > 4 * kMaxElems) return BNN_HIP_ERR_TOO_LARGE;
  if (!aligned(w, 4) || !aligned(dwhat, 4) || !aligned(dw, 4)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_xnor_weight_bwd(w, dwhat, splits, O, C, KH * KW, center != 0, compute_alpha != 0, dw,
                                     static_cast<hipStream_t>(stream));
}

// ---- training-mode BatchNorm (+ residual) (+ ReLU): csrc/bn_train.hip
static int check_bn(int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (mulc(N, C, HW) > 4 * kMaxElems || mulc(N, HW) > kMaxElems || C > (1 << 20)) return BNN_HIP_ERR_TOO_LARGE;
  return BNN_HIP_OK;
}

size_t bnn_hip_bn_train_workspace_bytes(int N, int C, int HW) {
  if (check_bn(N, C, HW) != BNN_HIP_OK) return 0;
  // [C][splits][2] doubles of partial sums + 3 C floats of per-channel coefficients
  return align_up((size_t)C * bnn::bn_train_splits(N, C, HW) * 2 * sizeof(double), 256) + (size_t)3 * C * sizeof(float);
}

int bnn_hip_bn_train_forward_f32(const float* x, int N, int C, int HW, const float* gamma, const float* beta,
                                 const float* residual, int relu, float eps, float momentum, float* running_mean,
                                 float* running_var, float* y, float* save_mean, float* save_invstd, void* workspace,
                                 void* stream) {
  if (!x || !y || !save_mean || !save_invstd || !workspace) return BNN_HIP_ERR_INVALID_ARG;
  const int st = check_bn(N, C, HW);
  if (st != BNN_HIP_OK) return st;
  if ((running_mean == nullptr) != (running_var == nullptr) || !(eps >= 0.0f)) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(x, 4) || !aligned(y, 4) || (residual && !aligned(residual, 4)) || !aligned(workspace, 8))
    return BNN_HIP_ERR_INVALID_ARG;
  const int S = bnn::bn_train_splits(N, C, HW);
  double* partial = static_cast<double*>(workspace);
  float* work = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)C * S * 2 * sizeof(double), 256));
  g_launches.fetch_add(3, std::memory_order_relaxed);
  BNN_RANGE();
  const int st2 = bnn::launch_bn_stats(x, N, C, HW, S, partial, static_cast<hipStream_t>(stream));
  if (st2 != BNN_HIP_OK) return st2;
  return bnn::launch_bn_apply(x, partial, S, gamma, beta, residual, relu, y, N, C, HW, eps, momentum, running_mean,
                              running_var, save_mean, save_invstd, work, static_cast<hipStream_t>(stream));
}

int bnn_hip_bn_act_f32(const float* x, int N, int C, int HW, const float* scale, const float* shift, const float* residual,
                       int relu, float* y, void* stream) {
  if (!x || !y || !scale || !shift) return BNN_HIP_ERR_INVALID_ARG;
  const int st = check_bn(N, C, HW);
  if (st != BNN_HIP_OK) return st;
  if (!aligned(x, 4) || !aligned(y, 4) || !aligned(scale, 4) || !aligned(shift, 4) || (residual && !aligned(residual, 4)))
    return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_bn_act(x, scale, shift, residual, relu, y, N, C, HW, static_cast<hipStream_t>(stream));
}

int bnn_hip_bn_train_backward_f32(const float* gy, const float* y, const float* x, const float* save_mean,
                                  const float* save_invstd, const float* gamma, int N, int C, int HW, float* dx,
                                  float* dres, float* dgamma, float* dbeta, void* workspace, void* stream) {
  if (!gy || !x || !save_mean || !save_invstd || !dx || !workspace) return BNN_HIP_ERR_INVALID_ARG;
  const int st = check_bn(N, C, HW);
  if (st != BNN_HIP_OK) return st;
  if (!aligned(gy, 4) || !aligned(x, 4) || !aligned(dx, 4) || (y && !aligned(y, 4)) || (dres && !aligned(dres, 4)) ||
      !aligned(workspace, 8))
    return BNN_HIP_ERR_INVALID_ARG;
  const int S = bnn::bn_train_splits(N, C, HW);
  double* partial = static_cast<double*>(workspace);
  float* work = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)C * S * 2 * sizeof(double), 256));
  g_launches.fetch_add(3, std::memory_order_relaxed);
  BNN_RANGE();
  const int st2 = bnn::launch_bn_bwd_reduce(gy, y, x, save_mean, save_invstd, N, C, HW, S, partial,
                                            static_cast<hipStream_t>(stream));
  if (st2 != BNN_HIP_OK) return st2;
  return bnn::launch_bn_bwd_dx(gy, y, x, save_mean, save_invstd, gamma, partial, S, dx, dres, dgamma, dbeta, N, C, HW, work,
                               static_cast<hipStream_t>(stream));
}

int bnn_hip_bn_relu_maxpool_train_forward_f32(const float* x, int N, int C, int H, int W, const float* gamma,
                                              const float* beta, float eps, float momentum, float* running_mean,
                                              float* running_var, float* pooled, uint8_t* code, float* save_mean,
                                              float* save_invstd, void* workspace, void* stream) {
  if (!x || !pooled || !code || !save_mean || !save_invstd || !workspace || H <= 0 || W <= 0) return BNN_HIP_ERR_INVALID_ARG;
  const int st = check_bn(N, C, (int)(mulc(H, W) > 0x7fffffffLL ? 0 : (long long)H * W));
  if (st != BNN_HIP_OK) return st;
  if ((running_mean == nullptr) != (running_var == nullptr) || !(eps >= 0.0f)) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(x, 4) || !aligned(pooled, 4) || !aligned(workspace, 8)) return BNN_HIP_ERR_INVALID_ARG;
  const int HW = H * W, S = bnn::bn_train_splits(N, C, HW);
  double* partial = static_cast<double*>(workspace);
  float* work = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)C * S * 2 * sizeof(double), 256));
  g_launches.fetch_add(3, std::memory_order_relaxed);
  BNN_RANGE();
  const int st2 = bnn::launch_bn_stats(x, N, C, HW, S, partial, static_cast<hipStream_t>(stream));
  if (st2 != BNN_HIP_OK) return st2;
  return bnn::launch_bn_relu_pool_fwd(x, partial, S, gamma, beta, pooled, code, N, C, H, W, eps, momentum, running_mean,
                                      running_var, save_mean, save_invstd, work, static_cast<hipStream_t>(stream));
}

int bnn_hip_bn_relu_maxpool_train_backward_f32(const float* gy, const float* pooled, const uint8_t* code, const float* x,
                                               const float* save_mean, const float* save_invstd, const float* gamma,
                                               int N, int C, int H, int W, float* dx, float* dgamma, float* dbeta,
                                               void* workspace, void* stream) {
  if (!gy || !pooled || !code || !x || !save_mean || !save_invstd || !dx || !workspace || H <= 0 || W <= 0)
    return BNN_HIP_ERR_INVALID_ARG;
  const int st = check_bn(N, C, (int)(mulc(H, W) > 0x7fffffffLL ? 0 : (long long)H * W));
  if (st != BNN_HIP_OK) return st;
  if (!aligned(gy, 4) || !aligned(pooled, 4) || !aligned(x, 4) || !aligned(dx, 4) || !aligned(workspace, 8))
    return BNN_HIP_ERR_INVALID_ARG;
  const int HW = H * W, S = bnn::bn_train_splits(N, C, HW);
  double* partial = static_cast<double*>(workspace);
  float* work = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)C * S * 2 * sizeof(double), 256));
  g_launches.fetch_add(3, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_bn_relu_pool_bwd(gy, pooled, code, x, save_mean, save_invstd, gamma, N, C, H, W, S, partial, work, dx,
                                      dgamma, dbeta, static_cast<hipStream_t>(stream));
}

// ---- the hierarchical block in one launch (csrc/hblock.hip) ----
static int check_hblock(const bnn_hip_hblock_desc* d) {
  if (!d) return BNN_HIP_ERR_INVALID_ARG;
  if (d->N <= 0 || d->C_in <= 0 || d->H <= 0 || d->W <= 0 || d->planes <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (d->rows_per_band < 0 || d->images_per_band < 0 || d->waves < 0 || d->waves > 16) return BNN_HIP_ERR_INVALID_ARG;
  if (d->flags & ~(BNN_HIP_FLAG_THROUGHPUT | BNN_HIP_HBLOCK_CHANNEL_LANES)) return BNN_HIP_ERR_INVALID_ARG;
  // the kernel's index arithmetic: 24-bit multiplies, 32-bit byte offsets into the fp32 tensors
  const long long lim = 1ll << 23;
  if (mulc(d->N, d->planes) >= lim || mulc(d->H, d->W) >= lim || mulc(d->H + 8, d->W + 2) >= lim) return BNN_HIP_ERR_TOO_LARGE;
  if (mulc(d->N, d->planes, d->H, d->W) > kMaxConvElems) return BNN_HIP_ERR_TOO_LARGE;
  if (mulc(d->N, ((long long)d->C_in + 63) / 64, d->H, d->W) > kMaxPlaneWords) return BNN_HIP_ERR_TOO_LARGE;
  return BNN_HIP_OK;
}

int bnn_hip_hblock_supported(const bnn_hip_hblock_desc* d) {
  if (check_hblock(d) != BNN_HIP_OK) return 0;
  if (d->flags & BNN_HIP_HBLOCK_CHANNEL_LANES) return bnn::hblock_cl_supported(d) ? 1 : 0;
  return bnn::hblock_supported(d) ? 1 : 0;
}

int bnn_hip_hblock_layout_of(int C_in, int planes, bnn_hip_hblock_layout* out) {
  if (!out || C_in <= 0 || planes <= 0) return BNN_HIP_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  return bnn::hblock_layout(C_in, planes, out);
}

int bnn_hip_hblock_pack_weights(int C_in, int planes, const uint32_t* wbits1, const uint32_t* wbits2,
                                const uint32_t* wbits3, uint32_t* weights, void* stream) {
  if (!wbits1 || !wbits2 || !wbits3 || !weights || C_in <= 0 || planes <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(wbits1, 4) || !aligned(wbits2, 4) || !aligned(wbits3, 4) || !aligned(weights, 64)) return BNN_HIP_ERR_INVALID_ARG;
  const uint32_t* const w[3] = {wbits1, wbits2, wbits3};
  g_launches.fetch_add(3, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_hblock_pack_weights(C_in, planes, w, weights, static_cast<hipStream_t>(stream));
}

int bnn_hip_hblock_pack_weights_cl(int C_in, int planes, const uint32_t* wbits1, const uint32_t* wbits2,
                                   const uint32_t* wbits3, uint32_t* weights, void* stream) {
  if (!wbits1 || !wbits2 || !wbits3 || !weights || C_in <= 0 || planes <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(wbits1, 4) || !aligned(wbits2, 4) || !aligned(wbits3, 4) || !aligned(weights, 64)) return BNN_HIP_ERR_INVALID_ARG;
  const uint32_t* const w[3] = {wbits1, wbits2, wbits3};
  g_launches.fetch_add(3, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_hblock_cl_pack_weights(C_in, planes, w, weights, static_cast<hipStream_t>(stream));
}

int bnn_hip_hblock_forward(const bnn_hip_hblock_desc* d, const uint64_t* in_P, const uint32_t* weights,
                           const float* consts, const float* residual, float* out, uint64_t* out_P, void* stream) {
  const int st = check_hblock(d);
  if (st != BNN_HIP_OK) return st;
  if (!in_P || !weights || !consts || !residual || !out || residual == out) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(in_P, 8) || !aligned(weights, 64) || !aligned(consts, 8) || !aligned(residual, 4) || !aligned(out, 4) ||
      (out_P && !aligned(out_P, 8)))
    return BNN_HIP_ERR_INVALID_ARG;
  if (d->flags & BNN_HIP_HBLOCK_CHANNEL_LANES) {
    if (!bnn::hblock_cl_supported(d)) return BNN_HIP_ERR_UNSUPPORTED;
    g_launches.fetch_add(1, std::memory_order_relaxed);
    Range range("bnn_hip_hblock_forward(channel lanes)");
    return bnn::launch_hblock_cl(d, in_P, weights, consts, residual, out, out_P, static_cast<hipStream_t>(stream));
  }
  if (!bnn::hblock_supported(d)) return BNN_HIP_ERR_UNSUPPORTED;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_hblock(d, in_P, weights, consts, residual, out, out_P, static_cast<hipStream_t>(stream));
}

int bnn_hip_hblock_pool_supported(const bnn_hip_hblock_desc* d) {
  if (check_hblock(d) != BNN_HIP_OK || (d->flags & BNN_HIP_HBLOCK_CHANNEL_LANES)) return 0;
  return bnn::hblock_pool_supported(d) ? 1 : 0;
}

int bnn_hip_hblock_pool_forward(const bnn_hip_hblock_desc* d, const uint64_t* in_P, const uint32_t* weights,
                                const float* consts, const float* pool_consts, const float* residual, uint64_t* out_P1,
                                uint64_t* out_P2, uint64_t* out_M2, void* stream) {
  const int st = check_hblock(d);
  if (st != BNN_HIP_OK) return st;
  if (d->flags & BNN_HIP_HBLOCK_CHANNEL_LANES) return BNN_HIP_ERR_INVALID_ARG;
  if (!in_P || !weights || !consts || !pool_consts || !residual || !out_P1 || !out_P2 || !out_M2) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(in_P, 8) || !aligned(weights, 64) || !aligned(consts, 8) || !aligned(pool_consts, 32) || !aligned(residual, 4) ||
      !aligned(out_P1, 8) || !aligned(out_P2, 8) || !aligned(out_M2, 8))
    return BNN_HIP_ERR_INVALID_ARG;
  if (!bnn::hblock_pool_supported(d)) return BNN_HIP_ERR_UNSUPPORTED;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_hblock_pool(d, in_P, weights, consts, pool_consts, residual, out_P1, out_P2, out_M2,
                                 static_cast<hipStream_t>(stream));
}

int bnn_hip_hblock_shortcut_supported(const bnn_hip_hblock_desc* d) {
  if (check_hblock(d) != BNN_HIP_OK) return 0;
  if (d->flags & BNN_HIP_HBLOCK_CHANNEL_LANES) return bnn::hblock_cl_ds_supported(d) ? 1 : 0;
  return bnn::hblock_ds_supported(d) ? 1 : 0;
}

int bnn_hip_hblock_pack_shortcut_weights(int C_in, int planes, const uint32_t* wbits, uint32_t* weights, void* stream) {
  if (!wbits || !weights || C_in <= 0 || planes <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(wbits, 4) || !aligned(weights, 32)) return BNN_HIP_ERR_INVALID_ARG;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_hblock_ds_pack_weights(C_in, planes, wbits, weights, static_cast<hipStream_t>(stream));
}

int bnn_hip_hblock_shortcut_forward(const bnn_hip_hblock_desc* d, const uint64_t* in_P, const uint32_t* weights,
                                    const float* consts, const uint64_t* sc_P, const uint64_t* sc_M,
                                    const uint32_t* sc_weights, const float* sc_alpha, float* out, uint64_t* out_P,
                                    void* stream) {
  const int st = check_hblock(d);
  if (st != BNN_HIP_OK) return st;
  if (!in_P || !weights || !consts || !sc_P || !sc_M || !sc_weights || !sc_alpha || !out || !out_P) return BNN_HIP_ERR_INVALID_ARG;
  if (!aligned(in_P, 8) || !aligned(weights, 64) || !aligned(consts, 8) || !aligned(sc_P, 8) || !aligned(sc_M, 8) ||
      !aligned(sc_weights, 32) || !aligned(sc_alpha, 32) || !aligned(out, 4) || !aligned(out_P, 8))
    return BNN_HIP_ERR_INVALID_ARG;
  if (d->flags & BNN_HIP_HBLOCK_CHANNEL_LANES) {   // (weights: those of bnn_hip_hblock_pack_weights_cl)
    if (!bnn::hblock_cl_ds_supported(d)) return BNN_HIP_ERR_UNSUPPORTED;
    g_launches.fetch_add(1, std::memory_order_relaxed);
    Range range("bnn_hip_hblock_shortcut_forward(channel lanes)");
    return bnn::launch_hblock_cl_ds(d, in_P, weights, consts, sc_P, sc_M, sc_weights, sc_alpha, out, out_P,
                                    static_cast<hipStream_t>(stream));
  }
  if (!bnn::hblock_ds_supported(d)) return BNN_HIP_ERR_UNSUPPORTED;
  g_launches.fetch_add(1, std::memory_order_relaxed);
  BNN_RANGE();
  return bnn::launch_hblock_ds(d, in_P, weights, consts, sc_P, sc_M, sc_weights, sc_alpha, out, out_P,
                               static_cast<hipStream_t>(stream));
}

int bnn_hip_probe_clock(int spin_iters, double* shader_mhz, double* elapsed_us, void* stream) {
  if (spin_iters <= 0 || spin_iters > (1 << 24) || !shader_mhz) return BNN_HIP_ERR_INVALID_ARG;
  return bnn::launch_probe_clock(spin_iters, shader_mhz, elapsed_us, static_cast<hipStream_t>(stream));
}

}  // extern "C"
