// Shared host/device helpers for libnlt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include "../../include/nlt_hip.h"

#define NLT_CHECK_LAUNCH()                                  \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return NLT_ERR_LAUNCH; \
  } while (0)

static inline bool nlt_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// float32(float64(u) / 255.0) -- `_load_data`'s normalize_uint + astype(float32) (nlt/datasets/nlt.py:131-136) -- without a
// table or a division: q = u * r, q += fma(-255, q, u) * r with r = fl(1 / 255); equal for all 256 bytes
// (tests/test_front4_index_math.py).  Explicitly rounded operations: immune to -ffp-contract and `#pragma clang fp`.
__device__ __forceinline__ float u8_unit(unsigned u) {
  const float r = 1.0f / 255.0f;
  const float uf = (float)u;
  const float q = __fmul_rn(uf, r);
  return __fmaf_rn(__fmaf_rn(-255.0f, q, uf), r, q);
}

// Implicit-GEMM view of every conv family (SURVEY.md 8a a-C1..a-D3):
//   rows    = texels of the "GEMM grid" (gh x gw per frame): output texels for the conv
//             modes, INPUT texels for DECONV_K2S2 (each owns a 2x2 output block);
//   K       = taps x (source0 channels | source1 channels), tap-major;
//   columns = output channels (DECONV_K2S2: (a,b,o), 4*cout columns).
// Split-K launches (conv_mfma.hip): the first NLT_SPLITK_COUNTERS words of the workspace are the tiles' ticket counters (zero between
// launches), the partial tiles follow.
constexpr int NLT_SPLITK_COUNTERS = 16384;

struct ConvP {
  const float* src0; const float* src1;
  const float* wgt;      // Keras layout (direct) or packed fragments (mfma)
  const float* bias;
  const float* mask_src; // optional LeakyReLU-derivative mask source (backward-data)
  float* out;
  int n, h, w;           // input dims
  int c0, c1, ld0, ld1;
  int cout, ldo, ldm;
  int gh, gw;            // GEMM row grid per frame
  int oh, ow;            // output dims
  int M;                 // n*gh*gw
  int N;                 // GEMM columns
  int act, accumulate;
  float alpha;
  // backward-data only (nlt_conv_backward_data): output channels >= split_c are the OBSERVATION half of dfm[l] (one
  // observation per frame).  They are not stored to `out`: v (+ *split_d when split_partial) times the LeakyReLU
  // derivative taken from split_y goes to split_d -- the gradient w.r.t. the observation path's pre-activation
  // (what level_split_bwd_kernel computes in its own pass).  split_y / split_d [rows, split_c].
  int split_c, split_partial;
  float split_alpha;
  const float* split_y;
  float* split_d;
  // nlt_conv_forward_map only: a per-OUTPUT-texel bias map [bmap_frames, oh, ow, cout] added to the pre-activation next to
  // `bias` (the frame-independent half of a conv over concat(q, given map): nlt/models/nlt.py:172-174 with obs_override);
  // bmap_mod = oh * ow when one map serves every frame (output texel index modulo it), 0 when there is one per frame.
  const float* bmap;
  int bmap_mod;
};

template <int MODE> struct ConvTraits;
template <> struct ConvTraits<NLT_CONV1X1>     { static constexpr int TAPS = 1; };
template <> struct ConvTraits<NLT_CONV_K2S2>   { static constexpr int TAPS = 4; };
template <> struct ConvTraits<NLT_CONV_K2S1>   { static constexpr int TAPS = 4; };
template <> struct ConvTraits<NLT_DECONV_K2S2> { static constexpr int TAPS = 1; };
template <> struct ConvTraits<NLT_DECONV_K2S1> { static constexpr int TAPS = 4; };

// Input texel (linear index within the [n,h,w] grid, or -1 when the tap falls in the zero
// padding) read by tap t=(a,b)=(t>>1,t&1) of GEMM row (f, y, x).
template <int MODE>
__device__ __forceinline__ int conv_tap_texel(const ConvP& p, int f, int y, int x, int t) {
  const int a = t >> 1, b = t & 1;
  int iy, ix;
  if (MODE == NLT_CONV1X1 || MODE == NLT_DECONV_K2S2) { iy = y; ix = x; }
  else if (MODE == NLT_CONV_K2S2) { iy = 2 * y + a; ix = 2 * x + b; }
  else if (MODE == NLT_CONV_K2S1) { iy = y + a; ix = x + b; if (iy >= p.h || ix >= p.w) return -1; }
  else { iy = y - a; ix = x - b; if (iy < 0 || ix < 0) return -1; }
  return (f * p.h + iy) * p.w + ix;
}

static inline int nlt_fill_conv_params(ConvP& p, int mode, const float* src0, int ld0, int c0,
                                       const float* src1, int ld1, int c1, int n, int h, int w,
                                       const float* wgt, const float* bias, int cout, float* out,
                                       int ldo, int act, float alpha, const float* mask_src, int ldm,
                                       int accumulate) {
  if (!src0 || !wgt || !bias || !out) return NLT_ERR_BAD_ARG;
  if (n <= 0 || h <= 0 || w <= 0 || c0 <= 0 || c1 < 0 || cout <= 0) return NLT_ERR_BAD_ARG;
  if (c1 > 0 && !src1) return NLT_ERR_BAD_ARG;
  if (ld0 < c0 || (c1 > 0 && ld1 < c1) || ldo < cout) return NLT_ERR_BAD_ARG;
  if (mask_src && ldm < cout) return NLT_ERR_BAD_ARG;
  if (mode < NLT_CONV1X1 || mode > NLT_DECONV_K2S1) return NLT_ERR_BAD_ARG;
  if (mode == NLT_CONV_K2S2 && ((h | w) & 1)) return NLT_ERR_UNSUPPORTED;  // TF SAME would pad odd sizes
  p.src0 = src0; p.src1 = src1; p.wgt = wgt; p.bias = bias; p.mask_src = mask_src; p.out = out;
  p.n = n; p.h = h; p.w = w; p.c0 = c0; p.c1 = c1; p.ld0 = ld0; p.ld1 = ld1;
  p.cout = cout; p.ldo = ldo; p.ldm = ldm; p.act = act; p.accumulate = accumulate; p.alpha = alpha;
  p.split_c = 0; p.split_partial = 0; p.split_alpha = 0.f; p.split_y = nullptr; p.split_d = nullptr;
  p.bmap = nullptr; p.bmap_mod = 0;
  p.gh = h; p.gw = w; p.oh = h; p.ow = w; p.N = cout;
  if (mode == NLT_CONV_K2S2) { p.gh = p.oh = h / 2; p.gw = p.ow = w / 2; }
  if (mode == NLT_DECONV_K2S2) { p.oh = 2 * h; p.ow = 2 * w; p.N = 4 * cout; }
  const long long M = (long long)n * p.gh * p.gw;
  const long long in_elems = (long long)n * h * w * (long long)(ld0 > ld1 ? ld0 : ld1);
  const long long out_elems = (long long)n * p.oh * p.ow * (long long)ldo;
  if (M >= (1ll << 31) || in_elems >= (1ll << 31) || out_elems >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  p.M = (int)M;
  return NLT_OK;
}

// 256 zero bytes of device memory, allocated once per process (one process drives one GPU): operand source of the
// row-walking weight-gradient kernels past the end of a run.
static inline const float* nlt_zero_page() {
  static const float* z = [] {
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess) return static_cast<const float*>(nullptr);
    return static_cast<const float*>(p);
  }();
  return z;
}

// NLT_WGRAD_GENERIC=1: keep the index-deriving weight-gradient kernels for every shape (A/B runs, parity tests of both forms)
static inline bool nlt_wgrad_generic_only() {
  static const bool v = [] { const char* e = getenv("NLT_WGRAD_GENERIC"); return e && e[0] == '1'; }();
  return v;
}

// Entry points implemented per algorithm file.
int nlt_conv_direct_launch(int mode, const ConvP& p, hipStream_t s);
int nlt_conv_mfma_launch(int mode, const ConvP& p, int tile_hint, hipStream_t s, int ksplit = 1, float* ws = nullptr);
bool nlt_conv_mfma_supported(int mode, const ConvP& p);
