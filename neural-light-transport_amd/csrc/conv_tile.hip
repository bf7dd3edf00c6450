// LDS-tiled implicit-GEMM conv for the encoder's k2 convs (Conv2D k2s2 / k2s1, 'same') on the fp32 matrix
// cores (v_mfma_f32_16x16x4_f32, exact fp32) -- the path for the MFMA-bound middle of the network.
//
// One 256-thread workgroup owns an 8 x 16 tile of OUTPUT texels x TN output channels.  The K loop runs
// over 16-channel slabs of the input; per stage the workgroup copies, with 16-byte coalesced loads,
//   B: the slab of the (haloed) input tile   -- k2s1: 9 x 17 texels, shared by all 4 taps;
//                                               k2s2: one tap row, 8 x 32 texels, split by x parity
//   A: the slab's weight fragments for the stage's taps (pre-packed in fragment order)
// into a double-buffered LDS stage (registers -> LDS after the current stage's MFMAs, one barrier per stage)
// and every wave then feeds its RT x 2 MFMA tiles from LDS with ds_read_b128.  Layouts are planar by
// channel quad ([kk][texel][4]) so that the 16 lanes a ds_read_b128 services together hit 16 different
// 16-byte slots (no bank conflicts); weights = MFMA A operand, texels = B operand, so a lane ends up with 4
// consecutive output channels of one texel and stores them with one 16-byte NHWC store.
//
// Observation path: with kobs > 1 the workgroup runs the kobs observation frames of its tile back to back
// and keeps their mean in registers -> the interleaved [query | mean] slice of fm[l] is written here and
// the separate reduce_mean pass (nlt/models/nlt.py:161-164) disappears.
#include "nlt_common.h"
#include "pack_common.h"

namespace {

constexpr int TH = 8, TW = 16;
constexpr int QS2 = 272, ODD2 = 132;        // k2s2 slab, in 16-byte slots: channel-quad stride (2 x 128 + pad, = 0 mod 16), odd-x plane offset (= 4 mod 8)
constexpr int PL = 160;                      // k2s1 haloed tile: 9 x 17 = 153 texels, plane padded to 160 slots

template <int MODE> struct TileTraits;
template <> struct TileTraits<NLT_CONV_K2S1> { static constexpr int STAGE_TAPS = 4, B_UNITS = 153 * 4, B_FLOATS = 4 * PL * 4; };
template <> struct TileTraits<NLT_CONV_K2S2> { static constexpr int STAGE_TAPS = 2, B_UNITS = 256 * 4, B_FLOATS = 4 * QS2 * 4; };
// Conv2DTranspose k2s1 ('same': taps (y - a, x - b), zero above / left of the image) = the k2s1 tile with its halo on the top /
// left: origin (ty0 - 1, tx0 - 1), tap (a, b) reads slot (y + 1 - a, x + 1 - b).  Backward-data of every encoder stride-1 conv.
template <> struct TileTraits<NLT_DECONV_K2S1> { static constexpr int STAGE_TAPS = 4, B_UNITS = 153 * 4, B_FLOATS = 4 * PL * 4; };
// Conv2DTranspose k2s2 (each input texel owns a 2 x 2 output block) = a 1x1-conv-shaped GEMM over the 8 x 16 INPUT tile with
// N = 4 * cout columns (a, b, o) and a scatter store: no halo, no taps -- a stage is TWO 16-channel slabs (the "taps" of the
// stage loop), 128 slots per channel-quad plane.  Backward-data of every encoder stride-2 conv (with the level-split epilogue).
template <> struct TileTraits<NLT_DECONV_K2S2> { static constexpr int STAGE_TAPS = 2, B_UNITS = 128 * 4 * 2, B_FLOATS = 2 * 4 * 128 * 4; };

struct TileP {
  const float* src; const float* packed; const float* bias;
  float* out; float* mean_out;
  int ld, cin, frames, kobs, h, w;           // input dims
  int oh, ow, cout, ldo, ldm;
  int tiles_y, tiles_x, ncc;                 // ncc = cin / 16
  int act; float alpha;
  // backward-data epilogue (nlt_conv_tile_backward_data): v (+= out when accumulate) times LeakyReLU'(mask_src) (slope alpha)
  const float* mask_src; int ld_mask; int accumulate;
  // ... and the level-split part of it (ConvP::split_* in nlt_common.h), transposed k2s2 mode only
  int split_c, split_partial; float split_alpha; const float* split_y; float* split_d;
};

__device__ __forceinline__ int xcd_tile(int b, int nblocks) {
  return (nblocks & 7) ? b : (b & 7) * (nblocks >> 3) + (b >> 3);
}

// packed weights: [g = cout / TN][cc = cin / 16][tap 4][ct TNT][lane 64][s4 4]
__global__ void pack_tile_kernel(const float* __restrict__ wk, int cin, int cout, int tnt, int full, int lo, int transposed, long total,
                                 float* __restrict__ wp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  wp[idx] = transposed == 2 ? nlt_tile_fragment_d2(wk, idx, cin, cout, tnt, full, lo)
                            : nlt_tile_fragment(wk, idx, cin, cout, tnt, full, lo, transposed != 0);   // t = a*2+b
}

// Register budget: the forward modes keep the allocation they were tuned with (<= 225 VGPR + AGPR, 2 workgroups per CU; the
// mean-free instantiation and a forced 2-waves-per-SIMD allocation were measured on the whole forward pass: 3 workgroups per
// CU of the query-path launches slow the observation-path launch running beside them, +2 % per step).  The transposed
// (backward-data) modes are compiled for 2 waves per SIMD (the transposed k2s1 tile at 64
// channels otherwise allocates past 256 registers = one workgroup per CU).
#ifndef NLT_TILE_MEANFREE
#define NLT_TILE_MEANFREE 0
#endif
template <int MODE, int TNT, bool MEAN>
__global__ __launch_bounds__(256, (MODE == NLT_DECONV_K2S1 || MODE == NLT_DECONV_K2S2) ? 2 : 1) void conv_tile_kernel(TileP p) {
  using TT = TileTraits<MODE>;
  constexpr int WN = TNT >= 4 ? 2 : 1, WM = 4 / WN, RT = TH / WM, CT = TNT / WN;
  constexpr int A_FLOATS = TT::STAGE_TAPS * TNT * 256;
  constexpr int A_UNITS = A_FLOATS / 4, NA = A_UNITS / 256, NB = (TT::B_UNITS + 255) / 256;
  constexpr int STAGE = A_FLOATS + TT::B_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, j = lane & 15;
  const int wn = wave % WN, wm = wave / WN;
  int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int tx0 = (tile % p.tiles_x) * TW; tile /= p.tiles_x;
  const int ty0 = (tile % p.tiles_y) * TH;
  const int f = tile / p.tiles_y;
  const int g = blockIdx.y;
  constexpr bool K2S1 = MODE == NLT_CONV_K2S1 || MODE == NLT_DECONV_K2S1;   // one stage per 16-channel slab, all 4 taps
  constexpr bool TR = MODE == NLT_DECONV_K2S1;
  constexpr bool D2 = MODE == NLT_DECONV_K2S2;                               // two 16-channel slabs per stage, no taps
  const int stages_per_frame = D2 ? p.ncc / 2 : (K2S1 ? 1 : 2) * p.ncc;
  const int total_stages = stages_per_frame * p.kobs;
  const long in_frame = (long)p.h * p.w;

  // B-slab addressing of this thread's copy units (fixed across stages except for the channel slab / tap row).
  // Unit -> thread: 8 consecutive lanes copy the SAME channel quad of 8 consecutive texels (a 32-lane group covers
  // 8 texels x 4 quads = the same 512 contiguous-per-texel bytes whichever way its lanes are dealt), because a
  // ds_write_b128 is serviced 8 contiguous lanes at a time over 32 banks: 8 consecutive 16-byte slots of one plane are
  // conflict-free, whereas the 4 quads of a texel sit a whole plane apart = on the same banks (measured 4-way / 8-way
  // store conflicts, SQ_LDS_BANK_CONFLICT 2.9 / 12.4 cycles per DS instruction for k2s1 / k2s2).  k2s2: the odd-x plane
  // sits 132 slots after the even one (4 mod 8) so that the two parities of those 8 texels fall on different banks too.
  int b_lds[NB], b_q[NB]; long b_tex[NB]; bool b_ok[NB], b_st[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int u = tid + 256 * i;
    const int q = (u >> 3) & 3, tx = (u >> 5) * 8 + (u & 7);
    b_q[i] = q;
    if (D2) {
      const int texel = tx & 127, slab = tx >> 7;
      const int gy = ty0 + (texel >> 4), gx = tx0 + (texel & 15);
      b_st[i] = true;
      b_ok[i] = gy < p.h && gx < p.w;
      b_tex[i] = (long)gy * p.w + gx;
      b_lds[i] = ((slab * 4 + q) * 128 + texel) * 4;
      b_q[i] = 4 * slab + q;                                          // channel quad inside the stage's 32 channels
    } else if (K2S1) {
      const int hy = tx / 17, hx = tx % 17;
      const int gy = ty0 + hy - (TR ? 1 : 0), gx = tx0 + hx - (TR ? 1 : 0);
      b_st[i] = tx < 153;
      b_ok[i] = b_st[i] && gy >= 0 && gx >= 0 && gy < p.h && gx < p.w;   // beyond the image: TF's zero padding (bottom / right; transposed: top / left)
      b_tex[i] = (long)gy * p.w + gx;
      b_lds[i] = (q * PL + tx) * 4;
    } else {
      const int y = tx >> 5, xx = tx & 31;
      const int gy = 2 * (ty0 + y), gx = 2 * tx0 + xx;               // + a (tap row) per stage
      b_st[i] = true;
      b_ok[i] = (ty0 + y) < p.oh && gx < p.w;                      // h, w even: both tap rows / parities exist
      b_tex[i] = (long)gy * p.w + gx;
      b_lds[i] = (q * QS2 + (xx & 1) * ODD2 + y * 16 + (xx >> 1)) * 4;
    }
  }

  f32x4 ra[NA], rb[NB];
  auto load_stage = [&](int q) {
    const int i = q / stages_per_frame, s = q - i * stages_per_frame;
    const int cc = D2 ? 2 * s : (K2S1 ? s : (s >> 1));
    const int a = (K2S1 || D2) ? 0 : (s & 1);
    const float* ap = D2 ? p.packed + (((long)g * p.ncc + cc) * TNT) * 256
                         : p.packed + ((((long)g * p.ncc + cc) * 4 + 2 * a) * TNT) * 256;
#pragma unroll
    for (int n = 0; n < NA; ++n) ra[n] = *reinterpret_cast<const f32x4*>(ap + (tid + 256 * n) * 4);
    const float* sp = p.src + ((long)(f * p.kobs + i) * in_frame + (long)a * p.w) * p.ld + cc * 16;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sp + (b_ok[n] ? b_tex[n] : 0) * p.ld + 4 * b_q[n]);
      rb[n] = b_ok[n] ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_stage = [&](int buf) {
    float* base = lds + buf * STAGE;
#pragma unroll
    for (int n = 0; n < NA; ++n) *reinterpret_cast<f32x4*>(base + (tid + 256 * n) * 4) = ra[n];
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (b_st[n]) *reinterpret_cast<f32x4*>(base + A_FLOATS + b_lds[n]) = rb[n];
  };

  f32x4 acc[RT][CT], mean[MEAN ? RT : 1][MEAN ? CT : 1];                 // MEAN = false: kobs == 1 and no mean_out (query path, backward-data)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (MEAN) mean[rt][ct] = acc[rt][ct];
    }

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int q = 0; q < total_stages; ++q) {
    if (q + 1 < total_stages) load_stage(q + 1);
    const float* A = lds + (q & 1) * STAGE;
    const float* B = A + A_FLOATS;
    const int s = q % stages_per_frame;
#pragma unroll
    for (int tl = 0; tl < TT::STAGE_TAPS; ++tl) {
      f32x4 bf[RT], af[CT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int y = wm * RT + rt;
        const int off = D2 ? ((tl * 4 + kk) * 128 + y * 16 + j) * 4
                         : TR ? (kk * PL + (y + 1 - (tl >> 1)) * 17 + j + 1 - (tl & 1)) * 4
                         : K2S1 ? (kk * PL + (y + (tl >> 1)) * 17 + j + (tl & 1)) * 4
                                : (kk * QS2 + tl * ODD2 + y * 16 + j) * 4;
        bf[rt] = *reinterpret_cast<const f32x4*>(B + off);
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        af[ct] = *reinterpret_cast<const f32x4*>(A + ((tl * TNT + wn * CT + ct) * 64 + lane) * 4);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ct][s4], bf[rt][s4], acc[rt][ct], 0, 0, 0);
    }
    if (s == stages_per_frame - 1) {                                  // this observation frame is complete
      const int i = q / stages_per_frame;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        int oc = (g * TNT + wn * CT + ct) * 16 + 4 * kk, ab = 0;
        if (D2) { ab = oc / p.cout; oc -= ab * p.cout; }                 // column (a, b, o) of the transposed k2s2 GEMM
        const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + oc) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int gy = D2 ? 2 * (ty0 + wm * RT + rt) + (ab >> 1) : ty0 + wm * RT + rt;
          const int gx = D2 ? 2 * (tx0 + j) + (ab & 1) : tx0 + j;
          f32x4 v = acc[rt][ct] + bv;
          if (D2 && p.split_c && oc >= p.split_c) {                    // observation half of dfm[l]: finished dobs (see ConvP)
            if (gy < p.oh && gx < p.ow) {
              const long ot = ((long)f * p.oh + gy) * p.ow + gx;
              if (p.accumulate) v += *reinterpret_cast<const f32x4*>(p.out + ot * p.ldo + oc);
              const long at = ot * p.split_c + (oc - p.split_c);
              f32x4* dd = reinterpret_cast<f32x4*>(p.split_d + at);
              if (p.split_partial) v += *dd;
              const f32x4 mk = *reinterpret_cast<const f32x4*>(p.split_y + at);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= (mk[e] > 0.f) ? 1.f : p.split_alpha;
              *dd = v;
            }
            acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            continue;
          }
          if (p.mask_src || p.accumulate) {                            // backward-data epilogue
            if (gy < p.oh && gx < p.ow) {
              const long ot = ((long)(f * p.kobs + i) * p.oh + gy) * p.ow + gx;
              f32x4* o = reinterpret_cast<f32x4*>(p.out + ot * p.ldo + oc);
              if (p.accumulate) v += *o;
              if (p.mask_src) {
                const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask_src + ot * p.ld_mask + oc);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= (mk[e] > 0.f) ? 1.f : p.alpha;
              }
              *o = v;
            }
            acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            continue;
          }
          if (p.act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : p.alpha * v[e];
          }
          acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (MEAN) mean[rt][ct] += v;
          if (gy < p.oh && gx < p.ow) {
            const long ot = ((long)(f * p.kobs + i) * p.oh + gy) * p.ow + gx;
            if (p.out) *reinterpret_cast<f32x4*>(p.out + ot * p.ldo + oc) = v;
            if (MEAN && p.mean_out && i == p.kobs - 1) {
              const long mt = ((long)f * p.oh + gy) * p.ow + gx;
              *reinterpret_cast<f32x4*>(p.mean_out + mt * p.ldm + oc) = mean[rt][ct] * (1.f / (float)p.kobs);
            }
          }
        }
      }
    }
    if (q + 1 < total_stages) store_stage((q + 1) & 1);
    __syncthreads();
  }
}

template <int MODE, int TNT>
int launch(const TileP& p, hipStream_t s) {
  const long tiles = (long)p.frames * p.tiles_y * p.tiles_x;
  const int ncols = MODE == NLT_DECONV_K2S2 ? 4 * p.cout : p.cout;
  const dim3 grid((unsigned)tiles, (unsigned)(ncols / (16 * TNT)));
  constexpr bool FWD = MODE == NLT_CONV_K2S1 || MODE == NLT_CONV_K2S2;       // (the transposed modes are backward-data only: never a mean)
  if (FWD && (p.kobs > 1 || p.mean_out || !NLT_TILE_MEANFREE)) hipLaunchKernelGGL((conv_tile_kernel<MODE, TNT, FWD>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((conv_tile_kernel<MODE, TNT, false>), grid, dim3(256), 0, s, p);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

}  // namespace

extern "C" long nlt_conv_tile_packed_floats(int mode, int cin, int cout, int tn) {
  if (mode == NLT_DECONV_K2S2) {                                       // K = cin in pairs of 16-channel slabs; N = 4 * cout columns
    if (cin <= 0 || cout <= 0 || (cin & 31) || (cout & 15) || (tn != 32 && tn != 64) || (4 * cout) % tn) return -1;
    return (long)4 * cin * cout;
  }
  if ((mode != NLT_CONV_K2S1 && mode != NLT_CONV_K2S2 && mode != NLT_DECONV_K2S1) || cin <= 0 || cout <= 0) return -1;
  if ((cin & 15) || (tn != 32 && tn != 64) || cout % tn) return -1;
  return (long)4 * cin * cout;
}

static int pack_tile(int mode, const float* w_keras, int cin, int cout, int tn, int full, int lo, float* packed, void* stream) {
  const long total = nlt_conv_tile_packed_floats(mode, cin, cout, tn);
  if (total <= 0) return NLT_ERR_UNSUPPORTED;
  if (!w_keras || !packed || !nlt_aligned16(packed) || lo < 0 || lo + cout > full) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(pack_tile_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w_keras, cin, cout, tn / 16, full, lo, mode == NLT_DECONV_K2S2 ? 2 : (mode == NLT_DECONV_K2S1 ? 1 : 0), total, packed);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_pack_conv_tile_weights(int mode, const float* w_keras, int cin, int cout, int tn, float* packed,
                                          void* stream) {
  return pack_tile(mode, w_keras, cin, cout, tn, cout, 0, packed, stream);
}

extern "C" int nlt_pack_conv_tile_weights_adjoint(int adj_mode, const float* w_keras, int cpre, int cout, int tn, int full, int lo,
                                                  float* packed, void* stream) {
  return pack_tile(adj_mode, w_keras, cpre, cout, tn, full, lo, packed, stream);
}

static int tile_run(int mode, TileP& p, int tn, hipStream_t s) {
  if (mode == NLT_CONV_K2S1) return tn == 64 ? launch<NLT_CONV_K2S1, 4>(p, s) : launch<NLT_CONV_K2S1, 2>(p, s);
  if (mode == NLT_DECONV_K2S1) return tn == 64 ? launch<NLT_DECONV_K2S1, 4>(p, s) : launch<NLT_DECONV_K2S1, 2>(p, s);
  if (mode == NLT_DECONV_K2S2) return tn == 64 ? launch<NLT_DECONV_K2S2, 4>(p, s) : launch<NLT_DECONV_K2S2, 2>(p, s);
  return tn == 64 ? launch<NLT_CONV_K2S2, 4>(p, s) : launch<NLT_CONV_K2S2, 2>(p, s);
}

extern "C" int nlt_conv_tile_forward(int mode, const float* src, int ld, int cin, int frames, int kobs, int h, int w,
                                     const float* packed, const float* bias, int cout, int tn,
                                     float* out, int ldo, float* mean_out, int ldm, int act, float alpha, void* stream) {
  if (!src || !packed || !bias || (!out && !mean_out)) return NLT_ERR_BAD_ARG;
  if (frames <= 0 || kobs <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return NLT_ERR_BAD_ARG;
  if (nlt_conv_tile_packed_floats(mode, cin, cout, tn) <= 0) return NLT_ERR_UNSUPPORTED;
  if (mode == NLT_CONV_K2S2 && ((h | w) & 1)) return NLT_ERR_UNSUPPORTED;
  if (ld < cin || (ld & 3) || (out && (ldo < cout || (ldo & 3))) || (mean_out && (ldm < cout || (ldm & 3)))) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(src) || !nlt_aligned16(packed) || !nlt_aligned16(bias) || (out && !nlt_aligned16(out)) ||
      (mean_out && !nlt_aligned16(mean_out))) return NLT_ERR_BAD_ARG;
  if ((long long)frames * kobs * h * w * (long long)(ld > ldo ? ld : ldo) >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  TileP p;
  p.src = src; p.packed = packed; p.bias = bias; p.out = out; p.mean_out = mean_out;
  p.ld = ld; p.cin = cin; p.frames = frames; p.kobs = kobs; p.h = h; p.w = w;
  p.oh = mode == NLT_CONV_K2S2 ? h / 2 : h; p.ow = mode == NLT_CONV_K2S2 ? w / 2 : w;
  p.cout = cout; p.ldo = ldo; p.ldm = ldm; p.ncc = cin / 16; p.act = act; p.alpha = alpha;
  p.mask_src = nullptr; p.ld_mask = 0; p.accumulate = 0;
  p.split_c = 0; p.split_partial = 0; p.split_alpha = 0.f; p.split_y = nullptr; p.split_d = nullptr;
  p.tiles_y = (p.oh + TH - 1) / TH; p.tiles_x = (p.ow + TW - 1) / TW;
  return tile_run(mode, p, tn, static_cast<hipStream_t>(stream));
}

extern "C" int nlt_conv_tile_backward_data(int adj_mode, const float* dpre, int ldp, int cpre, int n, int h, int w,
                                           const float* packed, int cout, int tn, float* out, int ldo,
                                           const float* mask_src, int ldm, float mask_alpha, int accumulate,
                                           int split_c, const float* split_y, float* split_d, float split_alpha, int split_partial,
                                           void* stream) {
  if (!dpre || !packed || !out || n <= 0 || h <= 0 || w <= 0 || cpre <= 0 || cout <= 0) return NLT_ERR_BAD_ARG;
  if (nlt_conv_tile_packed_floats(adj_mode, cpre, cout, tn) <= 0) return NLT_ERR_UNSUPPORTED;
  if (adj_mode == NLT_CONV_K2S2 && ((h | w) & 1)) return NLT_ERR_UNSUPPORTED;
  if (ldp < cpre || (ldp & 3) || ldo < cout || (ldo & 3) || (mask_src && (ldm < cout || (ldm & 3)))) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(dpre) || !nlt_aligned16(packed) || !nlt_aligned16(out) || (mask_src && !nlt_aligned16(mask_src))) return NLT_ERR_BAD_ARG;
  if ((long long)n * h * w * (adj_mode == NLT_DECONV_K2S2 ? 4 : 1) * (long long)(ldp > ldo ? ldp : ldo) >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  TileP p;
  p.src = dpre; p.packed = packed; p.bias = nullptr; p.out = out; p.mean_out = nullptr;
  p.ld = ldp; p.cin = cpre; p.frames = n; p.kobs = 1; p.h = h; p.w = w;
  p.oh = adj_mode == NLT_CONV_K2S2 ? h / 2 : (adj_mode == NLT_DECONV_K2S2 ? 2 * h : h);
  p.ow = adj_mode == NLT_CONV_K2S2 ? w / 2 : (adj_mode == NLT_DECONV_K2S2 ? 2 * w : w);
  p.split_c = 0; p.split_partial = 0; p.split_alpha = 0.f; p.split_y = nullptr; p.split_d = nullptr;
  if (split_c) {
    if (adj_mode != NLT_DECONV_K2S2 || split_c < 0 || (split_c & 3) || split_c >= cout || !split_y || !split_d ||
        !nlt_aligned16(split_y) || !nlt_aligned16(split_d))
      return NLT_ERR_BAD_ARG;
    p.split_c = split_c; p.split_y = split_y; p.split_d = split_d; p.split_alpha = split_alpha; p.split_partial = split_partial;
  }
  p.cout = cout; p.ldo = ldo; p.ldm = 0; p.ld_mask = ldm; p.ncc = cpre / 16; p.act = 0; p.alpha = mask_alpha;
  p.mask_src = mask_src; p.accumulate = accumulate;
  // the transposed k2s2 mode tiles the INPUT grid (a workgroup's 8 x 16 input texels own their 16 x 32 outputs)
  const int gh = adj_mode == NLT_DECONV_K2S2 ? h : p.oh, gw = adj_mode == NLT_DECONV_K2S2 ? w : p.ow;
  p.tiles_y = (gh + TH - 1) / TH; p.tiles_x = (gw + TW - 1) / TW;
  return tile_run(adj_mode, p, tn, static_cast<hipStream_t>(stream));
}
