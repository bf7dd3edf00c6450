// The conv family on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulate) for the MIDDLE of the network
// (BASELINE config 5: "2048^2 UV bf16, MFMA channel-mix path"): encoder levels >= 3 of both paths and the expanding
// blocks that mirror them -- every layer whose input or output has >= 64 channels (nlt/networks/convnet.py:50-59,
// 67-76 on the channel schedule of nlt/util/net.py).  Activations are STORED as bf16 between these layers, weights are
// bf16 fragments, bias + LeakyReLU run in fp32 on the accumulator; the full- and half-resolution ends stay fp32
// (csrc/front4.hip, dec_block.hip, fused.hip).  On fp32 MFMA these layers are the matrix-bound 75 % of the forward; at
// 16x the matrix rate they leave that roof and become bound by their (halved) activation traffic.
//
// Structure = conv_mfma.hip's register-tiled implicit GEMM: D[cout][texel] = W^T * X^T, weights = A operand, texels = B
// operand, a lane holds 4 consecutive output channels of one texel.  K runs in chunks of 32 channels = ONE MFMA; lane
// group g owns channels 8g .. 8g+7 of the chunk, so both operands are one 16-byte load per lane (8 bf16).  A source may
// also be fp32 (the region's inputs come from the fp32 level-2 kernels): two 16-byte loads, rounded to bf16 (nearest
// even) in registers.  The output is bf16 (8-byte store per lane) or fp32 (the region's last layer).  No LDS, no barriers.
#include "nlt_common.h"
#include "pack_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned short u16;

__device__ __forceinline__ u16 f2bf(float f) {                        // round to nearest even (NaN kept quiet)
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }
__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float((unsigned)v << 16); }

__host__ __device__ inline int chunks32(int c) { return (c + 31) >> 5; }

struct BfP {
  const void* src0; const void* src1;       // NHWC, fp32 or bf16 (f0 / f1)
  const u16* wgt;                            // packed fragments
  const float* bias;
  void* out;                                 // fp32 or bf16 (fo)
  int f0, f1, fo;
  int n, h, w, c0, c1, ld0, ld1, cout, ldo;
  int gh, gw, oh, ow, M, N;
  int act; float alpha;
};

template <int MODE>
__device__ __forceinline__ int tap_texel(const BfP& p, int f, int y, int x, int t) {
  const int a = t >> 1, b = t & 1;
  int iy, ix;
  if (MODE == NLT_CONV1X1 || MODE == NLT_DECONV_K2S2) { iy = y; ix = x; }
  else if (MODE == NLT_CONV_K2S2) { iy = 2 * y + a; ix = 2 * x + b; }
  else if (MODE == NLT_CONV_K2S1) { iy = y + a; ix = x + b; if (iy >= p.h || ix >= p.w) return -1; }
  else { iy = y - a; ix = x - b; if (iy < 0 || ix < 0) return -1; }
  return (f * p.h + iy) * p.w + ix;
}

// packed: [tap][32-channel chunk of (c0 | c1)][column tile][lane 64][8]; Keras index through nlt_keras_widx (pack_common.h)
template <int MODE>
__global__ void pack_bf16_kernel(const float* __restrict__ wk, int c0, int c1, int cout, int N, int ntiles, long total,
                                 u16* __restrict__ wp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7, lane = (idx >> 3) & 63;
  const long tile = idx >> 9;
  const int nt = tile % ntiles, kc = tile / ntiles;
  const int ch0 = chunks32(c0), ch1 = chunks32(c1);
  const int t = kc / (ch0 + ch1), r = kc % (ch0 + ch1);
  const bool s = r >= ch0;
  const int cl = (s ? r - ch0 : r) * 32 + 8 * (lane >> 4) + e;
  const int ncol = nt * 16 + (lane & 15);
  float v = 0.f;
  if (cl < (s ? c1 : c0) && ncol < N) v = wk[nlt_keras_widx<MODE>(t, (s ? c0 : 0) + cl, ncol, c0 + c1, cout, cout, 0)];
  wp[idx] = f2bf(v);
}

template <int MODE, int RT, int CT>
__global__ __launch_bounds__(256) void conv_bf16_kernel(BfP p, int mtiles, int ngroups, int ntiles) {
  constexpr int TAPS = (MODE == NLT_CONV1X1 || MODE == NLT_DECONV_K2S2) ? 1 : 4;
  const int lane = threadIdx.x & 63;
  int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ng = wave % ngroups; wave /= ngroups;
  const int mt = wave;
  if (mt >= mtiles) return;
  const int px = lane & 15, g = lane >> 4;

  int rf[RT], ry[RT], rx[RT], rm[RT];
  bool rv[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int m = (mt * RT + rt) * 16 + px;
    rv[rt] = m < p.M;
    const int mc = rv[rt] ? m : p.M - 1;
    rm[rt] = mc;
    rx[rt] = mc % p.gw;
    ry[rt] = (mc / p.gw) % p.gh;
    rf[rt] = mc / (p.gw * p.gh);
  }
  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(p.wgt) + (size_t)(ng * CT) * 64 + lane;
  const size_t wstride = (size_t)ntiles * 64;
  const int ch0 = chunks32(p.c0), ch1 = chunks32(p.c1);
  const int cps = ch0 + ch1;
  const int total = TAPS * cps;

  bool tv[RT];
  size_t tx[RT];
  auto set_tap = [&](int t) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int tex = rv[rt] ? tap_texel<MODE>(p, rf[rt], ry[rt], rx[rt], t) : -1;
      tv[rt] = tex >= 0;
      tx[rt] = tv[rt] ? (size_t)tex : 0;
    }
  };
  // unconditional loads from clamped addresses, value selected afterwards (see conv_mfma.hip)
  auto load_frags = [&](int r, int kci, uint4 (&a)[CT], uint4 (&b)[RT]) {
    const bool s = r >= ch0;
    const int k0 = (s ? r - ch0 : r) << 5;
    const bool kin = (k0 + 8 * g) < (s ? p.c1 : p.c0);
    const int ch = kin ? k0 + 8 * g : 0;
    const void* src = s ? p.src1 : p.src0;
    const int ld = s ? p.ld1 : p.ld0;
    const bool f32 = s ? p.f1 : p.f0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      uint4 v;
      if (f32) {
        const float* q = static_cast<const float*>(src) + tx[rt] * ld + ch;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(q), hi = *reinterpret_cast<const f32x4*>(q + 4);
        v = make_uint4(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
      } else {
        v = *reinterpret_cast<const uint4*>(static_cast<const u16*>(src) + tx[rt] * ld + ch);
      }
      const bool ok = kin && tv[rt];
      b[rt] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) a[ct] = wp[(size_t)kci * wstride + ct * 64];
  };
  uint4 a_cur[CT], b_cur[RT], a_nxt[CT], b_nxt[RT];
  int t_n = 0, r_n = 0;
  set_tap(0);
  load_frags(0, 0, a_cur, b_cur);
  for (int kc = 0; kc < total; ++kc) {
    if (kc + 1 < total) {
      if (++r_n == cps) { r_n = 0; set_tap(++t_n); }
      load_frags(r_n, kc + 1, a_nxt, b_nxt);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_cur[ct]), __builtin_bit_cast(bf16x8, b_cur[rt]),
                                                               acc[rt][ct], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) a_cur[ct] = a_nxt[ct];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) b_cur[rt] = b_nxt[rt];
  }

  // epilogue: lane holds outputs [ncol, ncol + 4) of texel px for every (rt, ct)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int ncol = (ng * CT + ct) * 16 + g * 4;
    if (ncol >= p.N) continue;
    int oc = ncol, ab = 0;
    if (MODE == NLT_DECONV_K2S2) { ab = ncol / p.cout; oc = ncol - ab * p.cout; }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + oc);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (!rv[rt]) continue;
      int otex = rm[rt];
      if (MODE == NLT_DECONV_K2S2) otex = (rf[rt] * p.oh + 2 * ry[rt] + (ab >> 1)) * p.ow + 2 * rx[rt] + (ab & 1);
      f32x4 v = acc[rt][ct] + bv;
      if (p.act) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : p.alpha * v[j];
      }
      if (p.fo) *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + (size_t)otex * p.ldo + oc) = v;
      else *reinterpret_cast<uint2*>(static_cast<u16*>(p.out) + (size_t)otex * p.ldo + oc) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
    }
  }
}

// mean over the k observation maps of a level, bf16 in / bf16 out into a channel slice (stride ldo) of fm[l]
// (tf.reduce_mean of nlt/models/nlt.py:161-164 on the stored bf16 values: fp32 sum in observation order, * (1/k), rounded)
__global__ __launch_bounds__(256) void obs_mean_bf16_kernel(const u16* __restrict__ obs, int k, long hw, int c, long total8,
                                                            u16* __restrict__ out, int ldo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total8) return;
  const int c8 = c >> 3;
  const int q = i % c8;
  const long t = i / c8;                                               // texel over frames
  const long f = t / hw, tt = t - f * hw;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < k; ++j) {
    const uint4 v = *reinterpret_cast<const uint4*>(obs + (((f * k + j) * hw + tt) * c + 8 * q));
    const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[2 * e] += bf2f((u16)(w4[e] & 0xffffu)); s[2 * e + 1] += bf2f((u16)(w4[e] >> 16)); }
  }
  const float inv = 1.f / (float)k;
  *reinterpret_cast<uint4*>(out + t * ldo + 8 * q) =
      make_uint4(pack2(s[0] * inv, s[1] * inv), pack2(s[2] * inv, s[3] * inv), pack2(s[4] * inv, s[5] * inv), pack2(s[6] * inv, s[7] * inv));
}

template <int MODE, int RT, int CT>
int launch_tile(const BfP& p, hipStream_t s) {
  const int ntiles = (p.N + 15) >> 4;
  const int ngroups = ntiles / CT;
  const int mtiles = (p.M + 16 * RT - 1) / (16 * RT);
  const long waves = (long)mtiles * ngroups;
  hipLaunchKernelGGL((conv_bf16_kernel<MODE, RT, CT>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, p, mtiles, ngroups, ntiles);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

template <int MODE>
int launch_mode(const BfP& p, int tile_hint, hipStream_t s) {
  const int ntiles = (p.N + 15) >> 4;
  int RT = 0, CT = 0;
  if (tile_hint > 0) { RT = tile_hint >> 4; CT = tile_hint & 15; }
  else {                                      // largest wave tile that still gives the chip >= 2 waves per SIMD
    static const int cand[7][2] = {{4, 4}, {2, 4}, {4, 2}, {2, 2}, {1, 2}, {2, 1}, {1, 1}};
    RT = 1; CT = 1;
    for (int i = 0; i < 7; ++i) {
      const int r = cand[i][0], c = cand[i][1];
      if (ntiles % c) continue;
      const long waves = (long)((p.M + 16 * r - 1) / (16 * r)) * (ntiles / c);
      if (waves >= 2048) { RT = r; CT = c; break; }
    }
  }
  if (CT <= 0 || ntiles % CT) return NLT_ERR_UNSUPPORTED;
#define NLT_TILE(R, C) if (RT == R && CT == C) return launch_tile<MODE, R, C>(p, s);
  NLT_TILE(4, 4) NLT_TILE(2, 4) NLT_TILE(4, 2) NLT_TILE(2, 2) NLT_TILE(1, 2) NLT_TILE(2, 1) NLT_TILE(1, 1)
#undef NLT_TILE
  return NLT_ERR_UNSUPPORTED;
}

int taps_of(int mode) { return (mode == NLT_CONV1X1 || mode == NLT_DECONV_K2S2) ? 1 : 4; }

}  // namespace

extern "C" long nlt_conv_bf16_packed_elems(int mode, int c0, int c1, int cout) {
  if (mode < NLT_CONV1X1 || mode > NLT_DECONV_K2S1 || c0 <= 0 || c1 < 0 || cout <= 0) return -1;
  if ((c0 & 7) || (c1 & 7) || (cout & 3)) return -1;
  const int N = (mode == NLT_DECONV_K2S2) ? 4 * cout : cout;
  return (long)taps_of(mode) * (chunks32(c0) + chunks32(c1)) * ((N + 15) >> 4) * 512;
}

extern "C" int nlt_conv_bf16_pack(int mode, const float* w_keras, int c0, int c1, int cout, unsigned short* packed, void* stream) {
  const long total = nlt_conv_bf16_packed_elems(mode, c0, c1, cout);
  if (total <= 0) return NLT_ERR_UNSUPPORTED;
  if (!w_keras || !packed || !nlt_aligned16(packed)) return NLT_ERR_BAD_ARG;
  const int N = (mode == NLT_DECONV_K2S2) ? 4 * cout : cout;
  const int ntiles = (N + 15) >> 4;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)((total + 255) / 256);
#define NLT_PACK(MODE) hipLaunchKernelGGL(pack_bf16_kernel<MODE>, dim3(blocks), dim3(256), 0, s, w_keras, c0, c1, cout, N, ntiles, total, packed)
  switch (mode) {
    case NLT_CONV1X1: NLT_PACK(NLT_CONV1X1); break;
    case NLT_CONV_K2S2: NLT_PACK(NLT_CONV_K2S2); break;
    case NLT_CONV_K2S1: NLT_PACK(NLT_CONV_K2S1); break;
    case NLT_DECONV_K2S2: NLT_PACK(NLT_DECONV_K2S2); break;
    case NLT_DECONV_K2S1: NLT_PACK(NLT_DECONV_K2S1); break;
  }
#undef NLT_PACK
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_conv_bf16_forward(int mode, int tile_hint,
                                     const void* src0, int ld0, int c0, int src0_is_f32,
                                     const void* src1, int ld1, int c1, int src1_is_f32,
                                     int n, int h, int w, const unsigned short* w_packed, const float* bias,
                                     int cout, void* out, int ldo, int out_is_f32, int act, float alpha, void* stream) {
  if (!src0 || !w_packed || !bias || !out) return NLT_ERR_BAD_ARG;
  if (n <= 0 || h <= 0 || w <= 0 || c0 <= 0 || c1 < 0 || cout <= 0 || (c1 > 0 && !src1)) return NLT_ERR_BAD_ARG;
  if (nlt_conv_bf16_packed_elems(mode, c0, c1, cout) <= 0) return NLT_ERR_UNSUPPORTED;
  if (ld0 < c0 || (ld0 & 7) || (c1 > 0 && (ld1 < c1 || (ld1 & 7))) || ldo < cout || (ldo & 3)) return NLT_ERR_BAD_ARG;
  if (mode == NLT_CONV_K2S2 && ((h | w) & 1)) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(src0) || (c1 > 0 && !nlt_aligned16(src1)) || !nlt_aligned16(w_packed) || !nlt_aligned16(bias) || !nlt_aligned16(out))
    return NLT_ERR_BAD_ARG;
  BfP p;
  p.src0 = src0; p.src1 = src1; p.wgt = w_packed; p.bias = bias; p.out = out;
  p.f0 = src0_is_f32; p.f1 = src1_is_f32; p.fo = out_is_f32;
  p.n = n; p.h = h; p.w = w; p.c0 = c0; p.c1 = c1; p.ld0 = ld0; p.ld1 = ld1; p.cout = cout; p.ldo = ldo;
  p.act = act; p.alpha = alpha;
  p.gh = h; p.gw = w; p.oh = h; p.ow = w; p.N = cout;
  if (mode == NLT_CONV_K2S2) { p.gh = p.oh = h / 2; p.gw = p.ow = w / 2; }
  if (mode == NLT_DECONV_K2S2) { p.oh = 2 * h; p.ow = 2 * w; p.N = 4 * cout; }
  const long long M = (long long)n * p.gh * p.gw;
  const long long in_elems = (long long)n * h * w * (long long)(ld0 > ld1 ? ld0 : ld1);
  const long long out_elems = (long long)n * p.oh * p.ow * (long long)ldo;
  if (M >= (1ll << 31) || in_elems >= (1ll << 31) || out_elems >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  p.M = (int)M;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (mode) {
    case NLT_CONV1X1: return launch_mode<NLT_CONV1X1>(p, tile_hint, s);
    case NLT_CONV_K2S2: return launch_mode<NLT_CONV_K2S2>(p, tile_hint, s);
    case NLT_CONV_K2S1: return launch_mode<NLT_CONV_K2S1>(p, tile_hint, s);
    case NLT_DECONV_K2S2: return launch_mode<NLT_DECONV_K2S2>(p, tile_hint, s);
    case NLT_DECONV_K2S1: return launch_mode<NLT_DECONV_K2S1>(p, tile_hint, s);
  }
  return NLT_ERR_BAD_ARG;
}

extern "C" int nlt_obs_mean_bf16(const unsigned short* obs, int n, int k, long hw, int c, unsigned short* out, int ldo, void* stream) {
  if (!obs || !out || n <= 0 || k <= 0 || hw <= 0 || c <= 0) return NLT_ERR_BAD_ARG;
  if ((c & 7) || (ldo & 7) || ldo < c) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(obs) || !nlt_aligned16(out)) return NLT_ERR_BAD_ARG;
  const long total8 = (long)n * hw * (c >> 3);
  hipLaunchKernelGGL(obs_mean_bf16_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     obs, k, hw, c, total8, out, ldo);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}
