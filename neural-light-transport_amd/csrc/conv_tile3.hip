// `precision = f32x3`: the encoder's k2 convs (Conv2D k2s2 / k2s1, 'same') with fp32 operands SPLIT into three bf16 terms and
// multiplied on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulate) -- 16x the rate of v_mfma_f32_16x16x4_f32.
//
//   x = hi + mid + lo exactly (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 8 + 8 + 8 significand bits = fp32's 24;
//   both subtractions are exact in fp32), so  a * b = sum over the nine term products, each EXACT in fp32 (8 x 8 bits).
//   NPROD = 9: all nine (what is lost is fp32 accumulation rounding only, as in the native fp32 MFMA);
//   NPROD = 6: the three products of order 2^-24 relative (mid*lo, lo*mid, lo*lo) are dropped: one extra fp32-rounding-sized
//              error per product.  Per fp32-equivalent K = 32 block that is 9 / 6 bf16 MFMAs of ~16 cycles against 8 fp32
//              MFMAs of 32 cycles: 0.56x / 0.375x the matrix-pipe time.
//   Small terms are accumulated first (lo-order products, then the mixed ones, hi*hi last).
//
// Structure = conv_tile.hip (8 x 16 output texels x TN output channels per 256-thread workgroup, 16-channel input slabs through a
// double-buffered LDS stage, one barrier per stage, observation mean in registers).  What differs:
//   * a K block of the bf16 MFMA is 32 = (2 taps) x (16 channels): lane group kk = lane >> 4 reads tap kk >> 1, channel half
//     kk & 1 -- 8 consecutive channels = one 16-byte LDS read per term plane;
//   * the texel slab is split when it is written to LDS (once per element per workgroup; every element is then used by
//     2-4 taps x TN output channels), into three planes [term][channel half][texel slot][8 bf16]; plane strides are multiples
//     of 16 slots, so the 16 lanes a ds_read_b128 services together (4 + 4 + 8 lanes of two lane groups) hit 16 different slots;
//   * the weights are split at pack time (nlt_pack_conv_tile3_weights): [g][cc][tap pair][ct][term][lane][8 bf16].
#include "nlt_common.h"
#include "pack_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16;
constexpr int QS2 = 272, ODD2 = 132;        // k2s2 tap row: slots per (term, half) plane (2 x 128 + pad, = 0 mod 16); odd-x offset (= 4 mod 8)
constexpr int PL = 160;                      // k2s1 haloed tile: 9 x 17 = 153 texel slots, plane padded to 160

template <int MODE> struct T3;
template <> struct T3<NLT_CONV_K2S1> { static constexpr int PAIRS = 2, B_UNITS = 153 * 2, B_SLOTS = 6 * PL; };
template <> struct T3<NLT_CONV_K2S2> { static constexpr int PAIRS = 1, B_UNITS = 256 * 2, B_SLOTS = 6 * QS2; };

struct Tile3P {
  const float* src; const u16* packed; const float* bias;
  float* out; float* mean_out;
  int ld, cin, frames, kobs, h, w;
  int oh, ow, cout, ldo, ldm;
  int tiles_y, tiles_x, ncc;
  int act; float alpha;
};

__device__ __forceinline__ int xcd_tile3(int b, int nblocks) {
  return (nblocks & 7) ? b : (b & 7) * (nblocks >> 3) + (b >> 3);
}

// two floats -> their bf16 roundings (nearest even), packed (lo half = a)
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  // (a conversion the compiler can see -- one v_cvt_pk_bf16_f32 -- not inline asm: the hazard recognizer pads MFMA -> VALU
  // read / write distances only for instructions it knows, r05)
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}

// (a, b) -> packed (hi, mid, lo) terms
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk_bf16(sa, sb);
}

__host__ __device__ inline u16 bf16_rne_bits(float f) {
  union { float f; unsigned u; } x; x.f = f;
  x.u += 0x7fffu + ((x.u >> 16) & 1u);
  return (u16)(x.u >> 16);
}
__host__ __device__ inline float bf16_bits_float(u16 b) {
  union { float f; unsigned u; } x; x.u = (unsigned)b << 16;
  return x.f;
}

// packed weights: [g = cout / TN][cc = cin / 16][pair 2][ct TNT][term 3][lane 64][e 8] bf16; Keras (kh,kw,Cin,Cout), t = a*2+b
__global__ void pack_tile3_kernel(const float* __restrict__ wk, int cin, int cout, int tnt, long total, u16* __restrict__ wp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7, lane = (idx >> 3) & 63;
  long r = idx >> 9;
  const int term = r % 3; r /= 3;
  const int ct = r % tnt; r /= tnt;
  const int pair = r & 1; r >>= 1;
  const int ncc = cin >> 4;
  const int cc = r % ncc;
  const int g = r / ncc;
  const int kk = lane >> 4, i = lane & 15;
  const int t = pair * 2 + (kk >> 1);
  const int c = cc * 16 + (kk & 1) * 8 + e;
  const int o = (g * tnt + ct) * 16 + i;
  const float v = wk[((long)t * cin + c) * cout + o];
  const u16 hi = bf16_rne_bits(v);
  const float r1 = v - bf16_bits_float(hi);
  const u16 mid = bf16_rne_bits(r1);
  const float r2 = r1 - bf16_bits_float(mid);
  wp[idx] = term == 0 ? hi : (term == 1 ? mid : bf16_rne_bits(r2));
}

// KO = observation frames of a tile that share ONE pass over the layer's weights (r04).  The weights are the larger half of what a
// stage moves (k2s1 at 64 channels: 24.6 KB of term fragments against 9.8 KB of texels; every workgroup of the chip streams them
// from L2), and the counter passes of r04 show the operand path, not the matrix pipe, bounding these launches.  With KO > 1 the
// stage order is (channel slab s, observation io) instead of (observation, slab): the slab's weight fragments are staged once and
// stay in LDS for KO micro-stages, each of which brings only the texel slab of its observation; every observation has its own
// accumulators (KO x RT x CT), and the frames' epilogues run together after the last slab.  kobs = groups x KO; the mean over
// all of them stays in registers as before.  KO = 1 is the r03 order.
template <int MODE, int TNT, int NPROD, int KO>
__global__ __launch_bounds__(256, 2) void conv_tile3_kernel(Tile3P p) {
  using TT = T3<MODE>;
  constexpr int WN = TNT == 4 ? 2 : 1, WM = 4 / WN, RT = TH / WM, CT = 2;
  constexpr int A_SLOTS = TT::PAIRS * TNT * 3 * 64;                    // 16-byte slots
  constexpr int NA = (A_SLOTS + 255) / 256, NB = (TT::B_UNITS + 255) / 256;      // (k2s2, TN = 32: 384 slots -> the last pass is half full)
  __shared__ u32x4 lds[2 * A_SLOTS + 2 * TT::B_SLOTS];                 // [A of slab parity 0 | 1][B of micro-stage parity 0 | 1]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, j = lane & 15;
  const int wn = wave % WN, wm = wave / WN;
  int tile = xcd_tile3(blockIdx.x, gridDim.x);
  const int tx0 = (tile % p.tiles_x) * TW; tile /= p.tiles_x;
  const int ty0 = (tile % p.tiles_y) * TH;
  const int f = tile / p.tiles_y;
  const int g = blockIdx.y;
  const int spf = (MODE == NLT_CONV_K2S1 ? 1 : 2) * p.ncc;             // slabs (stages) per frame
  const int groups = p.kobs / KO;
  const int total = spf * p.kobs;                                      // micro-stages
  const long in_frame = (long)p.h * p.w;

  // B copy units: unit = (texel, channel half): 8 consecutive lanes take 8 consecutive texels of one half (conflict-free
  // 16-byte LDS stores; the two halves of a texel are its 64 contiguous bytes in global memory)
  int b_lds[NB], b_hf[NB]; long b_tex[NB]; bool b_ok[NB], b_st[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int u = tid + 256 * i;
    const int hf = (u >> 3) & 1, tx = (u >> 4) * 8 + (u & 7);
    b_hf[i] = hf;
    if (MODE == NLT_CONV_K2S1) {
      const int hy = tx / 17, hx = tx % 17;
      const int gy = ty0 + hy, gx = tx0 + hx;
      b_st[i] = tx < 153;
      b_ok[i] = b_st[i] && gy < p.h && gx < p.w;
      b_tex[i] = (long)gy * p.w + gx;
      b_lds[i] = hf * PL + tx;                                        // + term * 2 * PL
    } else {
      const int y = tx >> 5, xx = tx & 31;
      const int gy = 2 * (ty0 + y), gx = 2 * tx0 + xx;
      b_st[i] = tx < 256;
      b_ok[i] = b_st[i] && (ty0 + y) < p.oh && gx < p.w;
      b_tex[i] = (long)gy * p.w + gx;
      b_lds[i] = hf * QS2 + (xx & 1) * ODD2 + y * 16 + (xx >> 1);     // + term * 2 * QS2
    }
  }
  constexpr int BPL = MODE == NLT_CONV_K2S1 ? PL : QS2;

  u32x4 ra[NA];
  f32x4 rb[NB][2];
  // micro-stage (group gi, slab s, observation io): texel slab of frame gi * KO + io; weights of slab s (io == 0 only)
  auto load_a = [&](int s) {
    const int cc = MODE == NLT_CONV_K2S1 ? s : (s >> 1);
    const int a = MODE == NLT_CONV_K2S1 ? 0 : (s & 1);
    const u32x4* ap = reinterpret_cast<const u32x4*>(p.packed) + (((long)g * p.ncc + cc) * 2 + a) * (TNT * 3 * 64);
#pragma unroll
    for (int n = 0; n < NA; ++n) ra[n] = ap[(A_SLOTS % 256 == 0 || tid + 256 * n < A_SLOTS) ? tid + 256 * n : tid];
  };
  auto load_b = [&](int s, int frame) {
    const int cc = MODE == NLT_CONV_K2S1 ? s : (s >> 1);
    const int a = MODE == NLT_CONV_K2S1 ? 0 : (s & 1);
    const float* sp = p.src + ((long)(f * p.kobs + frame) * in_frame + (long)a * p.w) * p.ld + cc * 16;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const float* q8 = sp + (b_ok[n] ? b_tex[n] : 0) * p.ld + 8 * b_hf[n];
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(q8), v1 = *reinterpret_cast<const f32x4*>(q8 + 4);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      rb[n][0] = b_ok[n] ? v0 : z;
      rb[n][1] = b_ok[n] ? v1 : z;
    }
  };
  auto store_a = [&](int buf) {
    u32x4* base = lds + buf * A_SLOTS;
#pragma unroll
    for (int n = 0; n < NA; ++n)
      if (A_SLOTS % 256 == 0 || tid + 256 * n < A_SLOTS) base[tid + 256 * n] = ra[n];
  };
  auto store_b = [&](int buf) {
    u32x4* base = lds + 2 * A_SLOTS + buf * TT::B_SLOTS;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      if (!b_st[n]) continue;
      unsigned hi[4], mid[4], lo[4];
      split2(rb[n][0][0], rb[n][0][1], hi[0], mid[0], lo[0]);
      split2(rb[n][0][2], rb[n][0][3], hi[1], mid[1], lo[1]);
      split2(rb[n][1][0], rb[n][1][1], hi[2], mid[2], lo[2]);
      split2(rb[n][1][2], rb[n][1][3], hi[3], mid[3], lo[3]);
      u32x4* bb = base + b_lds[n];
      bb[0] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
      bb[2 * BPL] = (u32x4){mid[0], mid[1], mid[2], mid[3]};
      bb[4 * BPL] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
    }
  };

  f32x4 acc[KO][RT][CT], mean[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      mean[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int io = 0; io < KO; ++io) acc[io][rt][ct] = mean[rt][ct];
    }

  load_a(0);
  load_b(0, 0);
  store_a(0);
  store_b(0);
  __syncthreads();
  int q = 0;                                                            // micro-stage counter (its parity = B buffer)
  for (int gi = 0; gi < groups; ++gi) {
    for (int s = 0; s < spf; ++s) {
      const int sa = (gi * spf + s) & 1;                                // A buffer of this slab
#pragma unroll
      for (int io = 0; io < KO; ++io, ++q) {
        // next micro-stage: (gi, s, io + 1), else (gi, s + 1, 0) -- with the next slab's weights --, else (gi + 1, 0, 0)
        const bool more = q + 1 < total;
        const bool new_slab = io + 1 == KO;
        int ns = s, ngi = gi;
        if (new_slab) { if (++ns == spf) { ns = 0; ++ngi; } }
        if (more) {
          if (new_slab) load_a(ns);
          load_b(ns, ngi * KO + (new_slab ? 0 : io + 1));
        }
        const u32x4* A = lds + sa * A_SLOTS;
        const u32x4* B = lds + 2 * A_SLOTS + (q & 1) * TT::B_SLOTS;
#pragma unroll
        for (int pl = 0; pl < TT::PAIRS; ++pl) {
          bf16x8 bt[3][RT], at[3][CT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const int y = wm * RT + rt;
            const int slot = MODE == NLT_CONV_K2S1 ? (kk & 1) * PL + (y + pl) * 17 + j + (kk >> 1)
                                                   : (kk & 1) * QS2 + (kk >> 1) * ODD2 + y * 16 + j;
#pragma unroll
            for (int t = 0; t < 3; ++t) bt[t][rt] = __builtin_bit_cast(bf16x8, B[t * 2 * BPL + slot]);
          }
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int t = 0; t < 3; ++t)
              at[t][ct] = __builtin_bit_cast(bf16x8, A[((pl * TNT + wn * CT + ct) * 3 + t) * 64 + lane]);
          // (weight term, texel term), smallest products first; consecutive MFMAs go to different accumulators
          constexpr int ORDER9[9][2] = {{2, 2}, {2, 1}, {1, 2}, {2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
          for (int pi = 9 - NPROD; pi < 9; ++pi)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
              for (int ct = 0; ct < CT; ++ct)
                acc[io][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[ORDER9[pi][0]][ct], bt[ORDER9[pi][1]][rt], acc[io][rt][ct], 0, 0, 0);
        }
        if (s == spf - 1 && io == KO - 1) {                             // the group's KO observation frames are complete
#pragma unroll
          for (int e = 0; e < KO; ++e) {
            const int i = gi * KO + e;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              const int oc = (g * TNT + wn * CT + ct) * 16 + 4 * kk;
              const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + oc);
#pragma unroll
              for (int rt = 0; rt < RT; ++rt) {
                const int gy = ty0 + wm * RT + rt, gx = tx0 + j;
                f32x4 v = acc[e][rt][ct] + bv;
                if (p.act) {
#pragma unroll
                  for (int c4 = 0; c4 < 4; ++c4) v[c4] = v[c4] > 0.f ? v[c4] : p.alpha * v[c4];
                }
                acc[e][rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                mean[rt][ct] += v;
                if (gy < p.oh && gx < p.ow) {
                  const long ot = ((long)(f * p.kobs + i) * p.oh + gy) * p.ow + gx;
                  if (p.out) *reinterpret_cast<f32x4*>(p.out + ot * p.ldo + oc) = v;
                  if (p.mean_out && i == p.kobs - 1) {
                    const long mt = ((long)f * p.oh + gy) * p.ow + gx;
                    *reinterpret_cast<f32x4*>(p.mean_out + mt * p.ldm + oc) = mean[rt][ct] * (1.f / (float)p.kobs);
                  }
                }
              }
            }
          }
        }
        if (more) {
          if (new_slab) store_a(sa ^ 1);
          store_b((q + 1) & 1);
        }
        __syncthreads();
      }
    }
  }
}

template <int MODE, int TNT, int KO>
int launch3k(const Tile3P& p, int nprod, hipStream_t s) {
  const long tiles = (long)p.frames * p.tiles_y * p.tiles_x;
  const dim3 grid((unsigned)tiles, (unsigned)(p.cout / (16 * TNT)));
  if (nprod == 9) hipLaunchKernelGGL((conv_tile3_kernel<MODE, TNT, 9, KO>), grid, dim3(256), 0, s, p);
  else if (nprod == 6) hipLaunchKernelGGL((conv_tile3_kernel<MODE, TNT, 6, KO>), grid, dim3(256), 0, s, p);
  else if (KO > 1) return NLT_ERR_UNSUPPORTED;
  else if (nprod == 3) hipLaunchKernelGGL((conv_tile3_kernel<MODE, TNT, 3, 1>), grid, dim3(256), 0, s, p);   // hi*hi + hi*mid + mid*hi (~2^-16)
  else hipLaunchKernelGGL((conv_tile3_kernel<MODE, TNT, 1, 1>), grid, dim3(256), 0, s, p);                  // hi*hi: plain bf16 operands
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

// NLT_TILE3_KO: observation frames per pass over the weights (0 = the largest that fits the registers: 4 at 32 channels per
// workgroup, 2 at 64; 1 = the r03 order) -- A/B switch
template <int MODE, int TNT>
int launch3(const Tile3P& p, int nprod, hipStream_t s) {
  static const int want = [] { const char* e = getenv("NLT_TILE3_KO"); return e ? atoi(e) : 0; }();
  // (stride-1 conv at 64 channels per workgroup: two sets of 4 x 2 accumulators beside the three-term fragments of 4 rows spill)
  const int cap = want > 0 ? want : (TNT == 2 ? 4 : (MODE == NLT_CONV_K2S1 ? 1 : 2));
  if (nprod >= 6) {
    if (TNT == 2 && cap >= 4 && p.kobs % 4 == 0) return launch3k<MODE, TNT, (TNT == 2 ? 4 : 1)>(p, nprod, s);
    if (cap >= 2 && p.kobs % 2 == 0) return launch3k<MODE, TNT, 2>(p, nprod, s);
  }
  return launch3k<MODE, TNT, 1>(p, nprod, s);
}

}  // namespace

extern "C" long nlt_conv_tile3_packed_elems(int mode, int cin, int cout, int tn) {
  if ((mode != NLT_CONV_K2S1 && mode != NLT_CONV_K2S2) || cin <= 0 || cout <= 0) return -1;
  if ((cin & 15) || (tn != 32 && tn != 64) || cout % tn) return -1;
  return (long)12 * cin * cout;                                       // 4 taps x 3 terms
}

extern "C" int nlt_pack_conv_tile3_weights(int mode, const float* w_keras, int cin, int cout, int tn, unsigned short* packed,
                                           void* stream) {
  const long total = nlt_conv_tile3_packed_elems(mode, cin, cout, tn);
  if (total <= 0) return NLT_ERR_UNSUPPORTED;
  if (!w_keras || !packed || !nlt_aligned16(packed)) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(pack_tile3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w_keras, cin, cout, tn / 16, total, packed);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_conv_tile3_forward(int mode, int nprod, const float* src, int ld, int cin, int frames, int kobs, int h, int w,
                                      const unsigned short* packed, const float* bias, int cout, int tn,
                                      float* out, int ldo, float* mean_out, int ldm, int act, float alpha, void* stream) {
  if (!src || !packed || !bias || (!out && !mean_out) || (nprod != 1 && nprod != 3 && nprod != 6 && nprod != 9)) return NLT_ERR_BAD_ARG;
  if (frames <= 0 || kobs <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return NLT_ERR_BAD_ARG;
  if (nlt_conv_tile3_packed_elems(mode, cin, cout, tn) <= 0) return NLT_ERR_UNSUPPORTED;
  if (mode == NLT_CONV_K2S2 && ((h | w) & 1)) return NLT_ERR_UNSUPPORTED;
  if (ld < cin || (ld & 3) || (out && (ldo < cout || (ldo & 3))) || (mean_out && (ldm < cout || (ldm & 3)))) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(src) || !nlt_aligned16(packed) || !nlt_aligned16(bias) || (out && !nlt_aligned16(out)) ||
      (mean_out && !nlt_aligned16(mean_out))) return NLT_ERR_BAD_ARG;
  if ((long long)frames * kobs * h * w * (long long)(ld > ldo ? ld : ldo) >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  Tile3P p;
  p.src = src; p.packed = packed; p.bias = bias; p.out = out; p.mean_out = mean_out;
  p.ld = ld; p.cin = cin; p.frames = frames; p.kobs = kobs; p.h = h; p.w = w;
  p.oh = mode == NLT_CONV_K2S2 ? h / 2 : h; p.ow = mode == NLT_CONV_K2S2 ? w / 2 : w;
  p.cout = cout; p.ldo = ldo; p.ldm = ldm; p.ncc = cin / 16; p.act = act; p.alpha = alpha;
  p.tiles_y = (p.oh + TH - 1) / TH; p.tiles_x = (p.ow + TW - 1) / TW;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mode == NLT_CONV_K2S1) return tn == 64 ? launch3<NLT_CONV_K2S1, 4>(p, nprod, s) : launch3<NLT_CONV_K2S1, 2>(p, nprod, s);
  return tn == 64 ? launch3<NLT_CONV_K2S2, 4>(p, nprod, s) : launch3<NLT_CONV_K2S2, 2>(p, nprod, s);
}
