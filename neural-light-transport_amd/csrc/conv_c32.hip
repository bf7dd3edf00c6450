// Conv2D k2s1 'same' + bias + LeakyReLU (+ observation mean) for the NARROW levels (16 or 32 input channels, 32 output channels:
// level 2 of both paths at 1/4 resolution, where the texels are) on the fp32 matrix cores -- nlt/networks/elements.py:26-31 as
// Model._call runs it (nlt/models/nlt.py:154-166).
//
// conv_tile.hip streams a layer in 16-channel stages: weights and texel slab of a stage go global -> registers -> LDS while
// the previous stage's MFMAs run.  At 32 channels that is TWO stages of 64 MFMAs per wave and frame -- 1 us of matrix work to
// hide ~3 us of loaded HBM latency behind, the 16 KB of weights staged again for every observation frame, an epilogue every
// second stage (r04 trace: L2.o.s1 125 us = 2.4 TB/s of its 300 MB, 84 TF: neither roof).  Here a stage is a whole FRAME:
//   * the layer's weights (16 KB of fragments, nlt_pack_conv_tile_weights order) are copied to LDS ONCE per workgroup;
//   * the haloed 9 x 17 texel tile comes in with all its channels (8 channel-quad planes of 160 slots, the conv_tile layout:
//     conflict-free 16-byte stores and ds_read_b128), double-buffered: frame i + 1 is in flight under the 128 MFMAs per wave
//     (4 us with two workgroups per CU) and the epilogue of frame i; one barrier per frame;
//   * the workgroup walks the kobs observation frames of its tile and keeps their mean in registers, as conv_tile does.
#include "nlt_common.h"
#include "pack_common.h"

namespace {

constexpr int TH = 8, TW = 16, PL = 160;       // output tile, slots per channel-quad plane (9 x 17 = 153 texels)

struct C32P {
  const float* src; const float* packed; const float* bias;
  float* out; float* mean_out;
  int ld, frames, kobs, h, w;
  int ldo, ldm;
  int tiles_y, tiles_x;
  int act; float alpha;
};

// NCC = input channels / 16; 32 output channels (TNT = 2 column tiles): waves 4 (rows) x 1, RT = 2, CT = 2.
// r04_c: PERSISTENT -- two workgroups per CU walk the tiles (an XCD keeps a contiguous run), so the weights go to LDS once per
// workgroup instead of once per tile (2048 tiles at the bench shape: 4 rounds of prologues, each an exposed L2 + HBM round trip),
// and the (tile, frame) sequence is one software pipeline: the first frame of the next tile is in flight under the last frame
// of the current one.
template <int NCC, bool MEAN>
__global__ __launch_bounds__(256, 2) void conv_c32_kernel(C32P p, int ntiles) {
  constexpr int TNT = 2, RT = 2, CT = 2;
  constexpr int NQ = 4 * NCC;                                          // channel quads
  constexpr int A_FLOATS = NCC * 4 * TNT * 256;                        // [cc][tap][ct][lane][4]
  constexpr int B_FLOATS = NQ * PL * 4;
  constexpr int B_UNITS = 153 * NQ, NB = (B_UNITS + 255) / 256, NA = A_FLOATS / 4 / 256;
  __shared__ __attribute__((aligned(16))) float lds[A_FLOATS + 2 * B_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int kk = lane >> 4, j = lane & 15;
  const long in_frame = (long)p.h * p.w;
  // this workgroup's tiles: XCD x = blockIdx & 7 owns tiles [x * per, (x + 1) * per), its workgroups take them round-robin
  const int per = (ntiles + 7) >> 3;
  const int t_hi = min(((int)(blockIdx.x & 7) + 1) * per, ntiles);
  const int stride = gridDim.x >> 3;
  int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tile >= t_hi) return;

  // B copy units: 8 consecutive lanes copy the same channel quad of 8 consecutive texels (conflict-free ds_write_b128)
  int b_lds[NB], b_q[NB]; long b_tex[NB]; bool b_ok[NB], b_st[NB];
  int lf = 0;                                                          // frame group of the tile whose frames are being LOADED
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int u = tid + 256 * i;
    const int q = (u >> 3) % NQ, tx = (u / (8 * NQ)) * 8 + (u & 7);
    b_q[i] = q;
    b_st[i] = tx < 153;
    b_lds[i] = (q * PL + tx) * 4;
  }
  auto load_geom = [&](int t) {
    const int tx0 = (t % p.tiles_x) * TW; t /= p.tiles_x;
    const int ty0 = (t % p.tiles_y) * TH;
    lf = t / p.tiles_y;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int u = tid + 256 * i;
      const int tx = (u / (8 * NQ)) * 8 + (u & 7);
      const int hy = tx / 17, hx = tx % 17;
      const int gy = ty0 + hy, gx = tx0 + hx;
      b_ok[i] = b_st[i] && gy < p.h && gx < p.w;                       // beyond the image: TF's zero padding (bottom / right)
      b_tex[i] = (long)gy * p.w + gx;
    }
  };
  f32x4 rb[NB];
  auto load_frame = [&](int i) {                                       // frame i of the tile being loaded
    const float* sp = p.src + (long)(lf * p.kobs + i) * in_frame * p.ld;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sp + (b_ok[n] ? b_tex[n] : 0) * p.ld + 4 * b_q[n]);
      rb[n] = b_ok[n] ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_frame = [&](int buf) {
    float* base = lds + A_FLOATS + buf * B_FLOATS;
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (b_st[n]) *reinterpret_cast<f32x4*>(base + b_lds[n]) = rb[n];
  };

  // first frame requested, then the weights: once per workgroup
  load_geom(tile);
  load_frame(0);
#pragma unroll
  for (int n = 0; n < NA; ++n)
    *reinterpret_cast<f32x4*>(lds + (tid + 256 * n) * 4) = *reinterpret_cast<const f32x4*>(p.packed + (tid + 256 * n) * 4);
  store_frame(0);
  __syncthreads();

  f32x4 bv[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(p.bias + ct * 16 + 4 * kk);
  int par = 0;                                                         // B buffer of the frame being computed

  for (;;) {
    int tt = tile;
    const int tx0 = (tt % p.tiles_x) * TW; tt /= p.tiles_x;
    const int ty0 = (tt % p.tiles_y) * TH;
    const int f = tt / p.tiles_y;
    const int next = tile + stride;
    const bool has_next = next < t_hi;
    f32x4 acc[RT][CT], mean[MEAN ? RT : 1][MEAN ? CT : 1];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (MEAN) mean[rt][ct] = acc[rt][ct];
      }

    for (int i = 0; i < p.kobs; ++i) {
      const bool more = i + 1 < p.kobs || has_next;                      // a frame follows (of this tile or of the next)
      if (i + 1 < p.kobs) load_frame(i + 1);
      else if (has_next) { load_geom(next); load_frame(0); }
      const float* A = lds;
      const float* B = lds + A_FLOATS + par * B_FLOATS;
#pragma unroll
      for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
          f32x4 bf[RT], af[CT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const int y = wm * RT + rt;
            bf[rt] = *reinterpret_cast<const f32x4*>(B + ((cc * 4 + kk) * PL + (y + (tl >> 1)) * 17 + j + (tl & 1)) * 4);
          }
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) af[ct] = *reinterpret_cast<const f32x4*>(A + (((cc * 4 + tl) * TNT + ct) * 64 + lane) * 4);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
              for (int ct = 0; ct < CT; ++ct)
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ct][s4], bf[rt][s4], acc[rt][ct], 0, 0, 0);
        }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int oc = ct * 16 + 4 * kk;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int gy = ty0 + wm * RT + rt, gx = tx0 + j;
          f32x4 v = acc[rt][ct] + bv[ct];
          if (p.act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : p.alpha * v[e];
          }
          acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (MEAN) mean[rt][ct] += v;
          if (gy < p.h && gx < p.w) {
            const long ot = ((long)(f * p.kobs + i) * p.h + gy) * p.w + gx;
            if (p.out) *reinterpret_cast<f32x4*>(p.out + ot * p.ldo + oc) = v;
            if (MEAN && p.mean_out && i == p.kobs - 1) {
              const long mt = ((long)f * p.h + gy) * p.w + gx;
              *reinterpret_cast<f32x4*>(p.mean_out + mt * p.ldm + oc) = mean[rt][ct] * (1.f / (float)p.kobs);
            }
          }
        }
      }
      if (more) store_frame(par ^ 1);
      __syncthreads();
      par ^= 1;
    }
    if (!has_next) break;
    tile = next;
  }
}

template <int NCC>
int launch_c32(const C32P& p, hipStream_t s) {
  const long tiles = (long)p.frames * p.tiles_y * p.tiles_x;
  // two workgroups per CU (57 KB of LDS each); fewer when there are fewer tiles; a multiple of 8 (one run of tiles per XCD)
  long groups = ((tiles + 7) / 8);
  if (groups > 64) groups = 64;
  const dim3 grid((unsigned)(8 * groups));
  if (p.kobs > 1 || p.mean_out) hipLaunchKernelGGL((conv_c32_kernel<NCC, true>), grid, dim3(256), 0, s, p, (int)tiles);
  else hipLaunchKernelGGL((conv_c32_kernel<NCC, false>), grid, dim3(256), 0, s, p, (int)tiles);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

}  // namespace

extern "C" int nlt_conv_c32_supported(int mode, int cin, int cout) {
  return mode == NLT_CONV_K2S1 && (cin == 16 || cin == 32) && cout == 32;
}

extern "C" int nlt_conv_c32_forward(int mode, const float* src, int ld, int cin, int frames, int kobs, int h, int w,
                                    const float* packed, const float* bias, int cout,
                                    float* out, int ldo, float* mean_out, int ldm, int act, float alpha, void* stream) {
  if (!src || !packed || !bias || (!out && !mean_out)) return NLT_ERR_BAD_ARG;
  if (frames <= 0 || kobs <= 0 || h <= 0 || w <= 0) return NLT_ERR_BAD_ARG;
  if (!nlt_conv_c32_supported(mode, cin, cout)) return NLT_ERR_UNSUPPORTED;
  if (ld < cin || (ld & 3) || (out && (ldo < cout || (ldo & 3))) || (mean_out && (ldm < cout || (ldm & 3)))) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(src) || !nlt_aligned16(packed) || !nlt_aligned16(bias) || (out && !nlt_aligned16(out)) ||
      (mean_out && !nlt_aligned16(mean_out))) return NLT_ERR_BAD_ARG;
  if ((long long)frames * kobs * h * w * (long long)(ld > ldo ? ld : ldo) >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  C32P p;
  p.src = src; p.packed = packed; p.bias = bias; p.out = out; p.mean_out = mean_out;
  p.ld = ld; p.frames = frames; p.kobs = kobs; p.h = h; p.w = w; p.ldo = ldo; p.ldm = ldm; p.act = act; p.alpha = alpha;
  p.tiles_y = (h + TH - 1) / TH; p.tiles_x = (w + TW - 1) / TW;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return cin == 32 ? launch_c32<2>(p, s) : launch_c32<1>(p, s);
}
