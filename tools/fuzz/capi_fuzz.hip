// capi_fuzz.hip — argument fuzz of the C-ABI's validators (csrc/capi.hip) under AddressSanitizer + UBSan, HOST ONLY.
//
// Built by `make -C binary-networks-pytorch_amd/csrc fuzz` (hipcc --cuda-host-only -fsanitize=address,undefined): capi.hip
// is compiled as it is, the kernels' launchers (bnn::launch_*) are replaced by the stubs below, which never touch a
// pointer and instead CHECK what the kernels rely on — the contract capi.hip has to enforce before it launches:
//   * every tensor of a conv launch below 2^30 fp32 elements / 2^29 plane words, stem and one-launch-layer tensors
//     below the 32-bit buffer-descriptor limit, alignments, consistent weight layout, positive geometry.
// The driver throws edge-value integers and null / misaligned / plausible pointers at every entry point: no call may
// crash, overflow a signed integer, read out of bounds, or reach a launcher with arguments that break the contract.
// Test infrastructure (tests/test_native_cpu.py runs it); nothing here ships in libbnn_hip.so.
#include <cinttypes>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../binary-networks-pytorch_amd/csrc/bnn_dev.h"

namespace {
unsigned long long g_reached = 0, g_calls = 0;
uint64_t g_rng = 0x9E3779B97F4A7C15ull;
uint64_t rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return g_rng; }

[[noreturn]] void broken(const char* what) {
  std::fprintf(stderr, "CONTRACT BROKEN: %s\n", what);
  std::abort();
}
#define REQUIRE(c) do { if (!(c)) broken(#c); } while (0)
bool al(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
constexpr long long kConvElems = (1LL << 30) - 1, kPlaneWords = (1LL << 29) - 1, kDesc = 0xFFFFFE00LL;
}  // namespace

namespace bnn {
int choose_cwc(int cw32, int KH, int KW) {  // same rule as csrc/bconv.hip
  if (KH == 3 && KW == 3) return (cw32 % 4 == 0) ? 4 : 2;
  if (KH == 1 && KW == 1) return cw32 % 16 == 0 ? 16 : cw32 % 8 == 0 ? 8 : cw32 % 4 == 0 ? 4 : 2;
  return 2;
}
static void check_convp(const ConvP& p) {
  REQUIRE(p.N > 0 && p.C > 0 && p.H > 0 && p.Wd > 0 && p.O > 0 && p.KH > 0 && p.KW > 0 && p.Ho > 0 && p.Wo > 0);
  REQUIRE(p.sh > 0 && p.sw > 0 && p.ph >= 0 && p.pw >= 0 && p.dh > 0 && p.dw > 0);
  REQUIRE(p.cw32 == 2 * ((p.C + 63) / 64) && p.cwc > 0 && p.cwc * p.nchunk == p.cw32);
  REQUIRE((long long)p.N * p.Ho * p.Wo == p.npix);
  const long long ctot = p.c_tot > 0 ? p.c_tot : p.O;
  REQUIRE(p.c_off >= 0 && p.c_off + p.O <= ctot);
  REQUIRE((long long)p.N * ctot * p.Ho * p.Wo <= kConvElems);
  REQUIRE((long long)p.N * p.Ho * p.Wo * ((p.O + 63) / 64) <= kPlaneWords);
}
int launch_bconv(const ConvP& p, int flags, hipStream_t) {
  ++g_reached;
  check_convp(p);
  REQUIRE((long long)p.N * p.H * p.Wd * ((p.C + 63) / 64) <= kPlaneWords);
  REQUIRE(p.P && p.M && p.W && al(p.P, 16) && al(p.M, 16) && al(p.W, 16));
  REQUIRE(p.out || (p.outP && p.outM));
  REQUIRE(p.raw || p.alpha);
  REQUIRE((p.bn_a == nullptr) == (p.bn_b == nullptr) && (p.outP == nullptr) == (p.outM == nullptr));
  REQUIRE((p.pack_a == nullptr) == (p.pack_b == nullptr));
  REQUIRE(!(flags & BNN_HIP_FLAG_WEIGHT_ZEROS) || p.Z);
  REQUIRE(!p.outP || (al(p.outP, 8) && al(p.outM, 8)));
  REQUIRE(!p.ds_P || (p.ds_W && p.ds_alpha && p.ds_a && p.ds_b && !p.res && p.ds_C > 0 && al(p.ds_P, 8) && al(p.ds_W, 16)));
  return BNN_HIP_OK;
}
bool ds_fold_applies(const ConvP& p, int flags) { return p.KH == 3 && p.KW == 3 && (flags & BNN_HIP_FLAG_ACT_NONNEG); }
bool fly_supported(const ConvP& p) { return p.KH * p.KW <= 49; }
int fly_default_plan(const ConvP& p, int, bnn_hip_fly_plan* plan) {
  check_convp(p);
  std::memset(plan, 0, sizeof(*plan));
  return fly_supported(p) ? BNN_HIP_OK : BNN_HIP_ERR_UNSUPPORTED;
}
int launch_bconv_fly(const ConvP& p, const void* x, int half, int flags, const bnn_hip_fly_plan*, hipStream_t) {
  ++g_reached;
  check_convp(p);
  REQUIRE(x && p.W && p.alpha && p.out && al(x, half ? 2 : 4) && al(p.W, 16) && al(p.out, 4));
  REQUIRE((long long)p.N * p.C * p.H * p.Wd * (half ? 2 : 4) <= kDesc);
  REQUIRE(!(flags & BNN_HIP_FLAG_WEIGHT_ZEROS) || p.Z);
  return BNN_HIP_OK;
}
static void check_planes(const void* x, int N, int C, int H, int W, const void* P, const void* M, int xal) {
  REQUIRE(x && P && M && N > 0 && C > 0 && H > 0 && W > 0 && al(x, xal) && al(P, 8) && al(M, 8));
  REQUIRE((long long)N * H * W <= (1LL << 31) - 1 && (C + 63) / 64 <= 65535);
}
int launch_pack_act(const float* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M, hipStream_t) {
  ++g_reached; check_planes(x, N, C, H, W, P, M, 4); return BNN_HIP_OK;
}
int launch_pack_act_f16(const void* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M, hipStream_t) {
  ++g_reached; check_planes(x, N, C, H, W, P, M, 2); return BNN_HIP_OK;
}
int launch_bn_act_pack(const float* x, int N, int C, int H, int W, const float* a, const float* b, int, uint64_t* P,
                       uint64_t* M, hipStream_t) {
  ++g_reached; check_planes(x, N, C, H, W, P, M, 4); REQUIRE((a == nullptr) == (b == nullptr)); return BNN_HIP_OK;
}
int launch_avgpool_pack(const float* x, int N, int C, int H, int W, int k, uint64_t* P, uint64_t* M, hipStream_t) {
  ++g_reached; REQUIRE(x && P && M && N > 0 && C > 0 && H > 0 && W > 0 && k > 0 && al(P, 8) && al(M, 8)); return BNN_HIP_OK;
}
int launch_avgpool2_bn_pack2(const float* x, int N, int C, int H, int W, const float* a1, const float* b1, int, uint64_t* P1,
                             uint64_t* M1, const float* a2, const float* b2, int, uint64_t* P2, uint64_t* M2, float* out,
                             hipStream_t) {
  ++g_reached;
  REQUIRE(x && a1 && b1 && P1 && M1 && N > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && al(x, 8));
  REQUIRE((a2 == nullptr) == (b2 == nullptr) && (!a2 || (P2 && M2 && al(P2, 8) && al(M2, 8))) && al(P1, 8) && al(M1, 8));
  REQUIRE((long long)N * C * H * W <= (1LL << 31) - 1 && (!out || al(out, 4)));
  return BNN_HIP_OK;
}
// csrc/hblock.hip: the shape rule and the launch contract of the one-launch hierarchical block
static bool stub_hb_shape(int C_in, int planes) {
  if (C_in <= 0 || C_in > (1 << 20) || planes < 64 || planes % 64 || planes > 4096) return false;
  const int cin[3] = {C_in, planes / 2, planes / 4};
  for (int k = 0; k < 3; ++k) {
    const int cw = (cin[k] + 31) / 32;
    if (cw == 3 || (cw > 4 && cw % 4)) return false;
  }
  const int cw0 = (C_in + 31) / 32;
  return cw0 == 1 ? C_in <= 32 : cw0 % 2 == 0;
}
bool hblock_supported(const bnn_hip_hblock_desc* d) {
  REQUIRE(d && d->N > 0 && d->C_in > 0 && d->H > 0 && d->W > 0 && d->planes > 0 && d->waves >= 0 && d->waves <= 16);
  REQUIRE((long long)d->N * d->planes < (1LL << 23) && (long long)d->H * d->W < (1LL << 23));
  return stub_hb_shape(d->C_in, d->planes) && (long long)(d->H + 6) * (d->W + 2) * ((d->C_in + 31) / 32) * 4 < 160 * 1024;
}
int hblock_layout(int C_in, int planes, bnn_hip_hblock_layout* L) {
  REQUIRE(L && C_in > 0 && planes > 0);
  if (!stub_hb_shape(C_in, planes)) return BNN_HIP_ERR_UNSUPPORTED;
  L->weight_words = 9LL * ((C_in + 31) / 32) * (planes / 2) + 9LL * ((planes / 2 + 31) / 32) * (planes / 4) +
                    9LL * ((planes / 4 + 31) / 32) * (planes / 4);
  L->const_floats = 4LL * planes;
  return BNN_HIP_OK;
}
bool hblock_cl_ds_supported(const bnn_hip_hblock_desc* d) {
  return d->planes == 2 * d->C_in && (d->C_in == 128 || d->C_in == 256) && d->H == d->W && (d->H == 7 || d->H == 14);
}
int launch_hblock_cl_ds(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc, const uint64_t* p,
                        const uint64_t* m, const uint32_t* w, const float* a, float* out, uint64_t* outP, hipStream_t) {
  ++g_reached;
  REQUIRE(d && inP && W && Kc && p && m && w && a && out && outP && al(w, 32) && al(a, 32) && al(outP, 8));
  REQUIRE(d->N > 0 && d->planes == 2 * d->C_in && d->H == d->W);
  return BNN_HIP_OK;
}
bool hblock_ds_supported(const bnn_hip_hblock_desc* d) {
  return d->planes == 2 * d->C_in && (d->C_in == 64 || d->C_in == 128) && (long long)d->H * d->W <= 4096;
}
int launch_hblock_ds(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc, const uint64_t* p,
                     const uint64_t* m, const uint32_t* w, const float* a, float* out, uint64_t* outP, hipStream_t) {
  ++g_reached;
  REQUIRE(d && inP && W && Kc && p && m && w && a && out && outP && al(w, 32) && al(a, 32) && al(outP, 8) && al(p, 8) && al(m, 8));
  REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->planes == 2 * d->C_in);
  return BNN_HIP_OK;
}
int launch_hblock_ds_pack_weights(int C_in, int planes, const uint32_t* w, uint32_t* dst, hipStream_t) {
  ++g_reached;
  REQUIRE(w && dst && C_in > 0 && planes > 0 && al(dst, 32));
  return (C_in % 64 || C_in > 4096 || planes > 4096) ? BNN_HIP_ERR_UNSUPPORTED : BNN_HIP_OK;
}
bool hblock_pool_supported(const bnn_hip_hblock_desc* d) {
  return d->C_in == d->planes && (d->planes == 64 || d->planes == 128 || d->planes == 256) && d->H % 2 == 0 && d->W % 2 == 0 &&
         d->rows_per_band % 2 == 0 && (long long)d->H * d->W <= 4096;
}
int launch_hblock_pool(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc, const float* Kp,
                       const float* res, uint64_t* o1, uint64_t* o2, uint64_t* o3, hipStream_t) {
  ++g_reached;
  REQUIRE(d && inP && W && Kc && Kp && res && o1 && o2 && o3 && al(Kp, 32) && al(o1, 8) && al(o2, 8) && al(o3, 8));
  REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0 && d->C_in == d->planes);
  return BNN_HIP_OK;
}
int launch_hblock_pack_weights(int C_in, int planes, const uint32_t* const w[3], uint32_t* dst, hipStream_t) {
  ++g_reached; REQUIRE(C_in > 0 && planes > 0 && w[0] && w[1] && w[2] && dst && al(dst, 64));
  return stub_hb_shape(C_in, planes) ? BNN_HIP_OK : BNN_HIP_ERR_UNSUPPORTED;
}
int launch_hblock(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc, const float* res,
                  float* out, uint64_t* outP, hipStream_t) {
  ++g_reached;
  REQUIRE(d && inP && W && Kc && res && out && res != out && al(inP, 8) && al(W, 64) && (!outP || al(outP, 8)));
  REQUIRE(stub_hb_shape(d->C_in, d->planes) && (long long)d->N * d->planes * d->H * d->W <= kConvElems);
  REQUIRE((long long)d->N * ((d->C_in + 63) / 64) * d->H * d->W <= kPlaneWords);
  return BNN_HIP_OK;
}
bool hblock_cl_supported(const bnn_hip_hblock_desc* d) {
  REQUIRE(d && d->N > 0 && d->C_in > 0 && d->planes > 0);
  return stub_hb_shape(d->C_in, d->planes) && d->planes % 256 == 0 && d->C_in % 64 == 0 && d->H == d->W && (d->H == 7 || d->H == 14);
}
int launch_hblock_cl_pack_weights(int C_in, int planes, const uint32_t* const w[3], uint32_t* dst, hipStream_t) {
  ++g_reached; REQUIRE(C_in > 0 && planes > 0 && w[0] && w[1] && w[2] && dst && al(dst, 64));
  return stub_hb_shape(C_in, planes) && planes % 256 == 0 ? BNN_HIP_OK : BNN_HIP_ERR_UNSUPPORTED;
}
int launch_hblock_cl(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc, const float* res,
                     float* out, uint64_t* outP, hipStream_t) {
  ++g_reached;
  REQUIRE(d && inP && W && Kc && res && out && res != out && al(inP, 8) && al(W, 64) && (!outP || al(outP, 8)));
  REQUIRE(d->planes % 256 == 0 && d->H == d->W && (d->H == 7 || d->H == 14));
  return BNN_HIP_OK;
}
int launch_orpool_packed(const uint64_t* P, int N, int C, int H, int W, int k, uint64_t* oP, uint64_t* oM, hipStream_t) {
  ++g_reached; REQUIRE(P && oP && oM && N > 0 && C > 0 && H > 0 && W > 0 && k > 0 && al(P, 8) && al(oP, 8) && al(oM, 8));
  return BNN_HIP_OK;
}
int launch_bn_relu_maxpool_pack(const float* x, int N, int C, int H, int W, const float* a, const float* b, int, int k,
                                int stride, int pad, float* out, uint64_t* P, uint64_t* M, hipStream_t) {
  ++g_reached;
  REQUIRE(x && N > 0 && C > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0 && (out || P));
  REQUIRE((P == nullptr) == (M == nullptr) && (a == nullptr) == (b == nullptr) && 2 * pad <= k);
  return BNN_HIP_OK;
}
int launch_stem(const float* x, const float* w, const float* a, const float* b, int N, int H, int W, int flags, float* out,
                uint64_t* P, uint64_t* M, hipStream_t) {
  ++g_reached;
  REQUIRE(x && w && a && b && N > 0 && H > 0 && W > 0 && (out || P) && (P == nullptr) == (M == nullptr));
  REQUIRE(!(flags & ~(BNN_HIP_STEM_EXACT_FP32 | BNN_HIP_STEM_FP16)));
  const long long hc = (H - 1) / 2 + 1, wc = (W - 1) / 2 + 1, hp = (hc - 1) / 2 + 1, wp = (wc - 1) / 2 + 1;
  REQUIRE((long long)N * 3 * H * W * 4 <= kDesc && (long long)N * 64 * hp * wp * 4 <= kDesc);
  REQUIRE(!P || (al(P, 8) && al(M, 8)));
  return BNN_HIP_OK;
}
int launch_stem_rows_aff(const float* x, const float* w, const float* a, const float* b, const float* pa, const float* pb,
                         int N, int H, int W, int, float*, uint64_t* P, uint64_t* M, hipStream_t) {
  ++g_reached;
  REQUIRE(x && w && a && b && pa && pb && P && M && N > 0 && H > 0 && W > 0 && al(P, 8) && al(M, 8));
  const long long hc = (H - 1) / 2 + 1, wc = (W - 1) / 2 + 1, hp = (hc - 1) / 2 + 1, wp = (wc - 1) / 2 + 1;
  REQUIRE((long long)N * 3 * H * W * 4 <= kDesc && (long long)N * 64 * hp * wp * 4 <= kDesc);
  return BNN_HIP_OK;
}
int launch_stem_conv(const float* x, const float* w, int N, int H, int W, int, float* out, hipStream_t) {
  ++g_reached;
  REQUIRE(x && w && out && N > 0 && H > 0 && W > 0);
  const long long hc = (H - 1) / 2 + 1, wc = (W - 1) / 2 + 1;
  REQUIRE((long long)N * 3 * H * W * 4 <= kDesc && (long long)N * 64 * hc * wc * 4 <= kDesc);
  return BNN_HIP_OK;
}
// the stub mirrors the product's support rule and slab arithmetic (csrc/stem_wgrad.hip) with a fixed workgroup count
static size_t stub_wgrad_lds(int W) {
  const int wc = (W - 1) / 2 + 1, need = 2 * ((wc + 15) / 16 * 16) + 8;
  int s = (need + 63) / 64 * 64 + 32;
  if (s - 64 >= need) s -= 64;
  return (size_t)3 * 13 * s * 4;
}
bool stem_wgrad_supported(int H, int W) { return H > 0 && W > 0 && stub_wgrad_lds(W) <= 64 * 1024; }
size_t stem_wgrad_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || !stem_wgrad_supported(H, W)) return 0;
  const long long bands = (long long)N * (((H - 1) / 2 + 1 + 3) / 4);
  return (size_t)std::min<long long>(bands, 768) * 64 * 176 * 4;
}
int launch_stem_wgrad(const float* x, const float* dy, int N, int H, int W, float* work, float* dw, hipStream_t) {
  ++g_reached;
  REQUIRE(x && dy && work && dw && N > 0 && stem_wgrad_supported(H, W));
  const long long hc = (H - 1) / 2 + 1, wc = (W - 1) / 2 + 1;
  REQUIRE((long long)N * 3 * H * W <= 0x7fffffffLL && (long long)N * 64 * hc * wc <= 0x7fffffffLL);
  return BNN_HIP_OK;
}
int launch_avgpool2x2_bwd(const float* gy, int N, int C, int Ho, int Wo, float* gx, hipStream_t) {
  ++g_reached;
  REQUIRE(gy && gx && N > 0 && C > 0 && Ho > 0 && Wo > 0 && (long long)N * C * Ho * Wo * 4 <= 0x7fffffffLL);
  return BNN_HIP_OK;
}
int launch_avgpool_fc(const float* x, const float* wt, const float*, float* out, int N, int C, int HW, int O, hipStream_t) {
  ++g_reached; REQUIRE(x && wt && out && N > 0 && C > 0 && HW > 0 && O > 0); return BNN_HIP_OK;
}
size_t avgpool_fc_workspace_bytes(int N, int C) {
  REQUIRE(N > 0 && C > 0);
  const unsigned long long e = (((unsigned long long)N + 15) / 16) * (unsigned long long)C;
  return e > (1ull << 56) ? ~(size_t)0 : (size_t)(e * 64);
}
bool avgpool_fc_ws_supported(int C, int HW) { REQUIRE(C > 0 && HW > 0); return (size_t)C * 64 <= 160 * 1024 - 1024; }
int launch_avgpool_fc_ws(const float* x, const float* wt, const float*, float* out, float* ws, int N, int C, int HW, int O,
                         hipStream_t) {
  ++g_reached; REQUIRE(x && wt && out && ws && al(ws, 16) && N > 0 && C > 0 && HW > 0 && O > 0 && (size_t)C * 64 <= 160 * 1024 - 1024);
  return BNN_HIP_OK;
}
size_t grad_weight_pack_bytes(int O, int C, int ks) { REQUIRE(O > 0 && C > 0 && (ks == 1 || ks == 3)); return 16; }
int launch_grad_pack_weight(const float* w, int O, int C, int ks, void* packed, float* alpha, hipStream_t) {
  ++g_reached; REQUIRE(w && packed && alpha && O > 0 && C > 0 && (ks == 1 || ks == 3) && al(packed, 16)); return BNN_HIP_OK;
}
static void check_grad(int N, int O, int C, int H, int W, int ks, int stride) {
  REQUIRE(N > 0 && O > 0 && C > 0 && H > 0 && W > 0 && W <= 64 && (ks == 1 || ks == 3) && (stride == 1 || stride == 2));
  REQUIRE((long long)N * O * H * W <= (1LL << 31) - 1 && (long long)N * C * H * W <= (1LL << 31) - 1);
}
int launch_dgrad(const float* g, const float* alpha, const void* packed, const void* x, int planes, float* gx, int N, int O,
                 int C, int H, int W, int ks, int stride, hipStream_t) {
  ++g_reached; REQUIRE(g && alpha && packed && x && gx && al(x, planes ? 8 : 4)); check_grad(N, O, C, H, W, ks, stride);
  return BNN_HIP_OK;
}
int launch_pack_ste(const float* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M, uint64_t* T, hipStream_t) {
  ++g_reached; check_planes(x, N, C, H, W, P, M, 4); REQUIRE(T && al(T, 8)); return BNN_HIP_OK;
}
int grad_wgrad_splits(int N, int O, int C, int ks) { REQUIRE(N > 0 && O > 0 && C > 0 && (ks == 1 || ks == 3)); return 1; }
int launch_wgrad(const float* g, const void* x, const void* x2, int planes, float* part, int, int N, int O, int C, int H,
                 int W, int ks, int stride, hipStream_t) {
  ++g_reached; REQUIRE(g && x && part && (!planes || (x2 && al(x, 8) && al(x2, 8)))); check_grad(N, O, C, H, W, ks, stride);
  return BNN_HIP_OK;
}
int launch_pack_weight(const float* w, int O, int C, int KH, int KW, int, int, const bnn_hip_wlayout& L, uint32_t* wb,
                       uint32_t* wz, float* alpha, int32_t* flag, hipStream_t) {
  ++g_reached;
  REQUIRE(w && wb && wz && alpha && flag && O > 0 && C > 0 && KH > 0 && KW > 0);
  REQUIRE(L.cw32 == 2 * ((C + 63) / 64) && L.cwc * L.nchunk == L.cw32 && L.o_pad >= O && L.o_pad % 32 == 0);
  REQUIRE(L.n_words == (int64_t)L.o_pad * KH * KW * L.cw32);
  return BNN_HIP_OK;
}
int launch_sign_thresholds(const float* alpha, const float*, const float*, const float* a, const float* b, int O, int kmax,
                           int32_t* thr, hipStream_t) {
  ++g_reached; REQUIRE(alpha && thr && O > 0 && kmax > 0 && kmax < (1 << 20) && (a == nullptr) == (b == nullptr) && al(thr, 4));
  return BNN_HIP_OK;
}
int launch_xnor_grad_pack(const float* w, int O, int C, int ks, int, int, void* packed, float* alpha, hipStream_t) {
  ++g_reached; REQUIRE(w && packed && alpha && O > 0 && C > 0 && (ks == 1 || ks == 3)); return BNN_HIP_OK;
}
int launch_xnor_what(const float* w, int O, int C, int taps, int, int, float* what, float*, hipStream_t) {
  ++g_reached; REQUIRE(w && what && O > 0 && C > 0 && taps > 0 && taps <= 1024 && (long long)O * C * taps <= (1LL << 31) - 1);
  return BNN_HIP_OK;
}
int launch_xnor_weight_bwd(const float* w, const float* g, int splits, int O, int C, int taps, int, int, float* dw, hipStream_t) {
  ++g_reached; REQUIRE(w && g && dw && splits > 0 && O > 0 && C > 0 && taps > 0 && taps <= 1024 && (long long)O * C * taps <= (1LL << 31) - 1 &&
                       (long long)O * C * taps * splits <= 4 * ((1LL << 31) - 1));
  return BNN_HIP_OK;
}
int bn_train_splits(int N, int C, int) { int s = (1024 + C - 1) / C; s = s > 64 ? 64 : s; s = s > N ? N : s; return s < 1 ? 1 : s; }
static void check_bn_args(int N, int C, int HW) {
  REQUIRE(N > 0 && C > 0 && HW > 0 && (long long)N * C * HW <= 4 * ((1LL << 31) - 1));
}
int launch_bn_stats(const float* x, int N, int C, int HW, int S, double* partial, hipStream_t) {
  ++g_reached; check_bn_args(N, C, HW); REQUIRE(x && partial && S >= 1 && S <= N && al(partial, 8)); return BNN_HIP_OK;
}
int launch_bn_act(const float* x, const float* scale, const float* shift, const float* res, int, float* y, int N, int C,
                  int HW, hipStream_t) {
  ++g_reached; check_bn_args(N, C, HW);
  REQUIRE(x && scale && shift && y && al(x, 4) && al(y, 4) && al(scale, 4) && al(shift, 4) && al(res, 4));
  return BNN_HIP_OK;
}
int launch_bn_apply(const float* x, const double* partial, int S, const float*, const float*, const float*, int, float* y,
                    int N, int C, int HW, float eps, float, float* rm, float* rv, float* mo, float* io, float* work,
                    hipStream_t) {
  ++g_reached; check_bn_args(N, C, HW);
  REQUIRE(x && partial && y && mo && io && work && S >= 1 && eps >= 0.0f && (rm == nullptr) == (rv == nullptr) && al(work, 4));
  return BNN_HIP_OK;
}
int launch_bn_bwd_reduce(const float* gy, const float*, const float* x, const float* m, const float* is, int N, int C, int HW,
                         int S, double* partial, hipStream_t) {
  ++g_reached; check_bn_args(N, C, HW); REQUIRE(gy && x && m && is && partial && S >= 1); return BNN_HIP_OK;
}
int launch_bn_bwd_dx(const float* gy, const float*, const float* x, const float* m, const float* is, const float*,
                     const double* partial, int S, float* dx, float*, float*, float*, int N, int C, int HW, float* work,
                     hipStream_t) {
  ++g_reached; check_bn_args(N, C, HW); REQUIRE(gy && x && m && is && partial && dx && work && S >= 1); return BNN_HIP_OK;
}
int launch_bn_relu_pool_fwd(const float* x, const double* partial, int S, const float*, const float*, float* p, unsigned char* code,
                            int N, int C, int H, int W, float eps, float, float* rm, float* rv, float* mo, float* io, float* work,
                            hipStream_t) {
  ++g_reached; REQUIRE(H > 0 && W > 0 && (long long)H * W <= 0x7fffffffLL); check_bn_args(N, C, H * W);
  REQUIRE(x && partial && p && code && mo && io && work && S >= 1 && eps >= 0.0f && (rm == nullptr) == (rv == nullptr));
  return BNN_HIP_OK;
}
int launch_bn_relu_pool_bwd(const float* gy, const float* p, const unsigned char* code, const float* x, const float* m,
                            const float* is, const float*, int N, int C, int H, int W, int S, double* partial, float* work,
                            float* dx, float*, float*, hipStream_t) {
  ++g_reached; REQUIRE(H > 0 && W > 0 && (long long)H * W <= 0x7fffffffLL); check_bn_args(N, C, H * W);
  REQUIRE(gy && p && code && x && m && is && partial && work && dx && S >= 1);
  return BNN_HIP_OK;
}
int launch_probe_int_alu(int mode, int iters, double* r, double*, hipStream_t) { REQUIRE(iters > 0 && r); (void)mode; return BNN_HIP_OK; }
int launch_probe_clock(int it, double* mhz, double*, hipStream_t) { REQUIRE(it > 0 && mhz); return BNN_HIP_OK; }
}  // namespace bnn

namespace {
const int kInts[] = {INT_MIN, -65536, -1, 0, 1, 2, 3, 7, 8, 31, 32, 33, 63, 64, 65, 127, 128, 129, 224, 255, 256, 1000, 4096,
                     65535, 65536, (1 << 20), (1 << 23), (1 << 24) - 1, (1 << 24), (1 << 28), (1 << 30) - 1, (1 << 30), INT_MAX};
const int kSmall[] = {1, 1, 1, 2, 3, 3, 4, 7, 8, 14, 16, 28, 56, 64, 112, 128, 224, 256, 512};
int pick_int() {
  const uint64_t r = rnd();
  if (r % 4 == 0) return kInts[(r >> 8) % (sizeof(kInts) / sizeof(int))];
  if (r % 4 == 1) return (int)(r >> 33) % 3000 - 10;
  return kSmall[(r >> 8) % (sizeof(kSmall) / sizeof(int))];       // mostly plausible: reach the launchers
}
template <class T>
T* pick_ptr() {
  static const uintptr_t kPtrs[] = {0, 0x10000, 0x10000, 0x10000, 0x10001, 0x10002, 0x10004, 0x10008, 0x7fff0000fff0ull};
  return reinterpret_cast<T*>(kPtrs[rnd() % (sizeof(kPtrs) / sizeof(uintptr_t))]);
}
bnn_hip_conv_desc pick_desc() {
  bnn_hip_conv_desc d;
  int* f = reinterpret_cast<int*>(&d);
  for (size_t i = 0; i < sizeof(d) / sizeof(int); ++i) f[i] = pick_int();
  if (rnd() % 2) { d.KH = d.KW = (rnd() % 2) ? 3 : 1; d.stride_h = d.stride_w = 1 + (int)(rnd() % 2); d.dil_h = d.dil_w = 1;
                   d.pad_h = d.pad_w = d.KH / 2; }
  d.flags = (int)(rnd() % 128);
  return d;
}
bool status_ok(int st) { return st <= 0 && st >= -5; }
}  // namespace

int main(int argc, char** argv) {
  const long iters = argc > 1 ? std::atol(argv[1]) : 200000;
  if (argc > 2) g_rng ^= (uint64_t)std::atoll(argv[2]) * 0xD1342543DE82EF95ull + 1;
  void* stream = nullptr;
  for (long it = 0; it < iters; ++it) {
    ++g_calls;
    int st = 0;
    switch (rnd() % 44) {
      case 0: { bnn_hip_conv_desc d = pick_desc();
        st = bnn_hip_bconv2d(rnd() % 16 ? &d : nullptr, pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), pick_ptr<uint32_t>(),
                             pick_ptr<uint32_t>(), pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(), stream);
        break; }
      case 1: { bnn_hip_conv_desc d = pick_desc(); bnn_hip_epilogue e; std::memset(&e, 0, sizeof(e));
        e.alpha = pick_ptr<float>(); e.bias = pick_ptr<float>(); e.post_scale = pick_ptr<float>(); e.bn_scale = pick_ptr<float>();
        e.bn_shift = pick_ptr<float>(); e.residual = pick_ptr<float>(); e.prelu = pick_ptr<float>(); e.relu = pick_int();
        e.flags = (int)(rnd() % 8); e.out_f32 = pick_ptr<float>(); e.out_P = pick_ptr<uint64_t>(); e.out_M = pick_ptr<uint64_t>();
        e.pack_scale = pick_ptr<float>(); e.pack_shift = pick_ptr<float>(); e.out_c_offset = pick_int(); e.out_c_total = pick_int();
        e.sign_thresholds = pick_ptr<int32_t>();
        if (rnd() % 3 == 0) { e.sc_P = pick_ptr<uint64_t>(); e.sc_wbits = pick_ptr<uint32_t>(); e.sc_alpha = pick_ptr<float>();
                              e.sc_bn_scale = pick_ptr<float>(); e.sc_bn_shift = pick_ptr<float>(); e.sc_C = pick_int(); }
        st = bnn_hip_bconv2d_fused(&d, pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), pick_ptr<uint32_t>(), pick_ptr<uint32_t>(),
                                   rnd() % 16 ? &e : nullptr, stream);
        break; }
      case 2: { bnn_hip_conv_desc d = pick_desc();
        st = bnn_hip_bconv2d_dot(&d, pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), pick_ptr<uint32_t>(), pick_ptr<uint32_t>(),
                                 pick_ptr<int32_t>(), stream);
        break; }
      case 3: { bnn_hip_conv_desc d = pick_desc(); bnn_hip_fly_plan plan; std::memset(&plan, 0, sizeof(plan));
        st = bnn_hip_bconv2d_direct(&d, pick_ptr<float>(), (int)(rnd() % 3), pick_ptr<uint32_t>(), pick_ptr<uint32_t>(),
                                    pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(),
                                    rnd() % 2 ? &plan : nullptr, stream);
        break; }
      case 4: { bnn_hip_conv_desc d = pick_desc(); bnn_hip_fly_plan plan;
        st = bnn_hip_bconv2d_direct_plan(rnd() % 16 ? &d : nullptr, rnd() % 16 ? &plan : nullptr);
        break; }
      case 5: { bnn_hip_conv_desc d = pick_desc();
        const size_t ws = bnn_hip_conv_workspace_bytes(&d);
        if (ws > (size_t)1 << 40) broken("workspace size overflow");
        st = bnn_hip_bconv2d_f32(&d, pick_ptr<float>(), pick_ptr<uint32_t>(), pick_ptr<uint32_t>(), pick_ptr<float>(),
                                 pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(), pick_ptr<char>(), stream);
        break; }
      case 6: st = bnn_hip_pack_act_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_ptr<uint64_t>(),
                                        pick_ptr<uint64_t>(), stream); break;
      case 7: st = bnn_hip_pack_act_f16(pick_ptr<char>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_ptr<uint64_t>(),
                                        pick_ptr<uint64_t>(), stream); break;
      case 8: st = bnn_hip_bn_act_pack_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_ptr<float>(),
                                           pick_ptr<float>(), pick_int(), pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), stream); break;
      case 9: st = bnn_hip_avgpool_pack_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_int(),
                                            pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), stream); break;
      case 10: st = bnn_hip_orpool_packed(pick_ptr<uint64_t>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_int(),
                                          pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), stream); break;
      case 11: st = bnn_hip_bn_relu_maxpool_pack_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(),
                                                     pick_ptr<float>(), pick_ptr<float>(), pick_int(), pick_int(), pick_int(),
                                                     pick_int(), pick_ptr<float>(), pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), stream); break;
      case 12: st = bnn_hip_stem7x7_bn_relu_pool_pack_f32(pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(),
                                                          pick_int(), pick_int(), pick_int(), (int)(rnd() % 16), pick_ptr<float>(),
                                                          pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), stream); break;
      case 13: st = bnn_hip_avgpool_fc_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_ptr<float>(), pick_ptr<float>(),
                                           pick_int(), pick_ptr<float>(), stream); break;
      case 14: { bnn_hip_wlayout L;
        st = bnn_hip_weight_layout(pick_int(), pick_int(), pick_int(), pick_int(), rnd() % 16 ? &L : nullptr);
        if (st == BNN_HIP_OK && (L.n_words <= 0 || L.cwc * L.nchunk != L.cw32)) broken("weight layout");
        break; }
      case 15: st = bnn_hip_pack_weight_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_int(), pick_int(),
                                            pick_ptr<uint32_t>(), pick_ptr<uint32_t>(), pick_ptr<float>(), pick_ptr<int32_t>(), stream); break;
      case 16: st = bnn_hip_bconv_grad_input_f32(pick_ptr<float>(), pick_ptr<float>(), pick_ptr<char>(), pick_ptr<float>(),
                                                 pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_int(),
                                                 (int)(rnd() % 5), (int)(rnd() % 4), stream); break;
      case 17: st = bnn_hip_bconv_grad_weight_f32(pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(), pick_int(), pick_int(),
                                                  pick_int(), pick_int(), pick_int(), pick_int(), (int)(rnd() % 5), (int)(rnd() % 4), stream); break;
      case 18: st = bnn_hip_sign_thresholds_f32(pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(),
                                                pick_ptr<float>(), pick_int(), pick_int(), pick_ptr<int32_t>(), stream); break;
      case 19: st = bnn_hip_pack_act_ste_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_ptr<uint64_t>(),
                                             pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), stream); break;
      case 20: st = bnn_hip_bconv_grad_input_packed_f32(pick_ptr<float>(), pick_ptr<float>(), pick_ptr<char>(), pick_ptr<uint64_t>(),
                                                        pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_int(),
                                                        (int)(rnd() % 5), (int)(rnd() % 4), stream); break;
      case 21: st = bnn_hip_bconv_grad_weight_packed_f32(pick_ptr<float>(), pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), pick_ptr<float>(),
                                                         pick_int(), pick_int(), pick_int(), pick_int(), pick_int(), pick_int(),
                                                         (int)(rnd() % 5), (int)(rnd() % 4), stream); break;
      case 22: { const int N = pick_int(), C = pick_int(), HW = pick_int();
        if (bnn_hip_bn_train_workspace_bytes(N, C, HW) > ((size_t)1 << 40)) broken("bn workspace size");
        st = bnn_hip_bn_train_forward_f32(pick_ptr<float>(), N, C, HW, pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(),
                                          pick_int(), (float)(pick_int() % 3) * 1e-5f, 0.1f, pick_ptr<float>(), pick_ptr<float>(),
                                          pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(), pick_ptr<double>(), stream);
        break; }
      case 23: st = bnn_hip_bn_train_backward_f32(pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(),
                                                  pick_ptr<float>(), pick_ptr<float>(), pick_int(), pick_int(), pick_int(),
                                                  pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(),
                                                  pick_ptr<double>(), stream); break;
      case 24: st = bnn_hip_bn_relu_maxpool_train_forward_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(),
                                                              pick_ptr<float>(), pick_ptr<float>(), 1e-5f, 0.1f, pick_ptr<float>(),
                                                              pick_ptr<float>(), pick_ptr<float>(), pick_ptr<uint8_t>(),
                                                              pick_ptr<float>(), pick_ptr<float>(), pick_ptr<double>(), stream); break;
      case 25: st = bnn_hip_bn_relu_maxpool_train_backward_f32(pick_ptr<float>(), pick_ptr<float>(), pick_ptr<uint8_t>(),
                                                               pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(),
                                                               pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(),
                                                               pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(),
                                                               pick_ptr<double>(), stream); break;
      case 26: st = bnn_hip_xnor_weight_forward_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_int(),
                                                    pick_int(), pick_ptr<float>(), pick_ptr<float>(), stream); break;
      case 27: st = bnn_hip_xnor_weight_backward_f32(pick_ptr<float>(), pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(),
                                                     pick_int(), pick_int(), pick_int(), pick_ptr<float>(), stream); break;
      case 28: st = bnn_hip_bn_act_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_ptr<float>(), pick_ptr<float>(),
                                       pick_ptr<float>(), pick_int(), pick_ptr<float>(), stream); break;
      case 29: { const int n = pick_int(), c = pick_int();
        const size_t need = bnn_hip_avgpool_fc_workspace_bytes(n, c);
        st = bnn_hip_avgpool_fc_ws_f32(pick_ptr<float>(), n, c, pick_int(), pick_ptr<float>(), pick_ptr<float>(), pick_int(),
                                       pick_ptr<float>(), pick_ptr<float>(), rnd() % 4 ? need : (size_t)(rnd() % 4096), stream);
        break; }
      case 30: st = bnn_hip_stem7x7_conv_f32(pick_ptr<float>(), pick_ptr<float>(), pick_int(), pick_int(), pick_int(),
                                             (int)(rnd() % 8), pick_ptr<float>(), stream); break;
      case 31: { const int n = pick_int(), h = pick_int(), w = pick_int();
        const size_t need = bnn_hip_stem7x7_wgrad_workspace_bytes(n, h, w);
        st = bnn_hip_stem7x7_wgrad_f32(pick_ptr<float>(), pick_ptr<float>(), n, h, w, pick_ptr<float>(),
                                       rnd() % 4 ? need : (size_t)(rnd() % 4096), pick_ptr<float>(), stream);
        break; }
      case 32: st = bnn_hip_avgpool2x2_backward_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(),
                                                    pick_ptr<float>(), stream); break;
      case 34: st = bnn_hip_avgpool2_bn_pack2_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_ptr<float>(),
                                                  pick_ptr<float>(), pick_int(), pick_ptr<uint64_t>(), pick_ptr<uint64_t>(),
                                                  pick_ptr<float>(), pick_ptr<float>(), pick_int(), pick_ptr<uint64_t>(),
                                                  pick_ptr<uint64_t>(), pick_ptr<float>(), stream); break;
      case 35: { bnn_hip_hblock_desc d; int* f = reinterpret_cast<int*>(&d);
        for (size_t i = 0; i < sizeof(d) / sizeof(int); ++i) f[i] = pick_int();
        if (rnd() % 2) { d.planes = 64 << (rnd() % 4); d.C_in = rnd() % 2 ? d.planes : d.planes / 2; d.flags = (int)(rnd() % 2) * 64 + (int)(rnd() % 2) * 128; if (rnd() % 2) d.H = d.W = (rnd() % 2) ? 7 : 14;
                         d.rows_per_band = d.images_per_band = d.waves = 0; }
        (void)bnn_hip_hblock_supported(rnd() % 16 ? &d : nullptr);
        st = bnn_hip_hblock_forward(rnd() % 16 ? &d : nullptr, pick_ptr<uint64_t>(), pick_ptr<uint32_t>(), pick_ptr<float>(),
                                    pick_ptr<float>(), pick_ptr<float>(), pick_ptr<uint64_t>(), stream);
        break; }
      case 36: { bnn_hip_hblock_layout L;
        st = bnn_hip_hblock_layout_of(pick_int(), pick_int(), rnd() % 16 ? &L : nullptr);
        if (st == BNN_HIP_OK && (L.weight_words <= 0 || L.const_floats <= 0)) broken("hblock layout");
        break; }
      case 38: st = bnn_hip_hblock_pack_weights_cl(pick_int(), pick_int(), pick_ptr<uint32_t>(), pick_ptr<uint32_t>(), pick_ptr<uint32_t>(),
                                                   pick_ptr<uint32_t>(), stream); break;
      case 39: st = bnn_hip_stem7x7_bn_relu_pool_pack_affine_f32(pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(),
                                                                pick_ptr<float>(), pick_ptr<float>(), pick_int(), pick_int(), pick_int(),
                                                                pick_int() & 3, pick_ptr<float>(), pick_ptr<uint64_t>(),
                                                                pick_ptr<uint64_t>(), stream); break;
      case 40: { bnn_hip_hblock_desc d; int* f = reinterpret_cast<int*>(&d);
        for (size_t i = 0; i < sizeof(d) / sizeof(int); ++i) f[i] = pick_int();
        if (rnd() % 2) { d.planes = d.C_in = 64 << (rnd() % 3); d.flags = (int)(rnd() % 2) * 64; d.H = d.W = 2 * (1 + (int)(rnd() % 28));
                         d.rows_per_band = d.images_per_band = d.waves = 0; }
        (void)bnn_hip_hblock_pool_supported(rnd() % 16 ? &d : nullptr);
        st = bnn_hip_hblock_pool_forward(rnd() % 16 ? &d : nullptr, pick_ptr<uint64_t>(), pick_ptr<uint32_t>(), pick_ptr<float>(),
                                         pick_ptr<float>(), pick_ptr<float>(), pick_ptr<uint64_t>(), pick_ptr<uint64_t>(),
                                         pick_ptr<uint64_t>(), stream);
        break; }
      case 41: { bnn_hip_hblock_desc d; int* f = reinterpret_cast<int*>(&d);
        for (size_t i = 0; i < sizeof(d) / sizeof(int); ++i) f[i] = pick_int();
        if (rnd() % 2) { d.C_in = 64 << (rnd() % 3); d.planes = 2 * d.C_in; d.flags = (int)(rnd() % 2) * 64 + (int)(rnd() % 2) * 128; d.H = d.W = (rnd() % 2) ? (rnd() % 2 ? 7 : 14) : 1 + (int)(rnd() % 56);
                         d.rows_per_band = d.images_per_band = d.waves = 0; }
        (void)bnn_hip_hblock_shortcut_supported(rnd() % 16 ? &d : nullptr);
        st = bnn_hip_hblock_shortcut_forward(rnd() % 16 ? &d : nullptr, pick_ptr<uint64_t>(), pick_ptr<uint32_t>(), pick_ptr<float>(),
                                             pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), pick_ptr<uint32_t>(), pick_ptr<float>(),
                                             pick_ptr<float>(), pick_ptr<uint64_t>(), stream);
        break; }
      case 42: st = bnn_hip_hblock_pack_shortcut_weights(pick_int(), pick_int(), pick_ptr<uint32_t>(), pick_ptr<uint32_t>(), stream); break;
      case 37: st = bnn_hip_hblock_pack_weights(pick_int(), pick_int(), pick_ptr<uint32_t>(), pick_ptr<uint32_t>(), pick_ptr<uint32_t>(),
                                                pick_ptr<uint32_t>(), stream); break;
      case 33: st = bnn_hip_xnor_grad_pack_weight_f32(pick_ptr<float>(), pick_int(), pick_int(), pick_int(), pick_int(), pick_int(),
                                                      pick_ptr<float>(), pick_ptr<float>(), stream); break;
      default: { bnn_hip_conv_desc d = pick_desc();
        (void)bnn_hip_shortcut_fold_supported(rnd() % 16 ? &d : nullptr, pick_int());
        st = bnn_hip_blinear(pick_int(), pick_int(), pick_int(), pick_ptr<uint64_t>(), pick_ptr<uint64_t>(), pick_ptr<uint32_t>(),
                             pick_ptr<uint32_t>(), pick_int(), pick_ptr<float>(), pick_ptr<float>(), pick_ptr<float>(),
                             pick_ptr<float>(), stream);
        break; }
    }
    if (!status_ok(st)) { std::fprintf(stderr, "unknown status %d\n", st); return 2; }
  }
  std::printf("CAPI_FUZZ_OK calls=%llu reached_launchers=%llu\n", g_calls, g_reached);
  return g_reached > g_calls / 200 ? 0 : 3;     // the fuzz must actually get past the validators often enough
}
