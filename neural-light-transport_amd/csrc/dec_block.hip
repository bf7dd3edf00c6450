// One expanding block of the decoder in one launch (inference): Conv2DTranspose k2s2 (cx + cs -> C) + LeakyReLU,
// Conv2DTranspose k2s1 (C -> C) + LeakyReLU (nlt/networks/convnet.py:67-76 as Model._call runs it on the virtual
// concat [x | popped encoder map], nlt/models/nlt.py:182-195), for the blocks where the texels are: C = 8 (output at
// 1/2 resolution) and C = 16 (1/4).  The layer-by-layer plan writes the block's intermediate map to HBM and reads it
// back (33.5 + 33.5 MB at C = 8, 1024^2, 4 frames: more than the block's own output) and launches twice; here the
// intermediate lives in LDS only -- the generalisation of back_kernel (fused.hip) to C > 4, both convs on the MFMA.
//
// Workgroup = 8 x 16 tile of INPUT texels (16 x 32 output texels), 256 threads.
//   stage 1: k2s2 transposed conv on the haloed input tile (1 texel top / left: the stride-1 transposed conv reads
//            (y - a, x - b)).  GEMM rows = the 4C columns (a, b, o) of the Keras (2,2,C,cin) kernel = A operand straight
//            from the Keras array (row-major in cin, L1-resident); columns = 16 haloed texels = B operand, one 16-byte
//            NHWC load per lane per 16-channel chunk of the virtual concat (permuted K: the lane's 4 channels feed 4
//            k-steps).  Results (+ bias, LeakyReLU, zero outside the image) go to a full-resolution LDS tile, planar by
//            channel quad so that ds_read_b128 of 16 consecutive texels is conflict-free.
//   stage 2: k2s1 transposed conv from that tile: rows = C output channels (padded to 16 at C = 8), K = 4 taps x C in
//            16-wide slabs (C = 8: two taps per slab), columns = 16 consecutive output texels of a row.
#include "nlt_common.h"
#include <cstdlib>

namespace {

constexpr int TH = 8, TW = 16;
constexpr int HH = TH + 1, HW = TW + 1, HT = HH * HW, NT = (HT + 15) / 16;   // haloed input tile: 153 texels, 10 column tiles
constexpr int FH = 2 * TH + 1, FW = 2 * TW + 1;                               // full-resolution tile incl. top / left halo: 17 x 33
constexpr int FP = 576;                                                       // slots per channel-quad plane (561 -> 576)
static_assert(FH * FW <= FP, "the full-resolution tile fits its plane");

__device__ __forceinline__ int xcd_tile(int b, int nblocks) {
  return (nblocks & 7) ? b : (b & 7) * (nblocks >> 3) + (b >> 3);
}
__device__ __forceinline__ f32x4 lrelu4(f32x4 v, float alpha) {
  return (f32x4){v[0] > 0.f ? v[0] : alpha * v[0], v[1] > 0.f ? v[1] : alpha * v[1],
                 v[2] > 0.f ? v[2] : alpha * v[2], v[3] > 0.f ? v[3] : alpha * v[3]};
}

struct DecP {
  const float *x, *skip;            // [n,h,w,cx], [n,h,w,cs]
  const float *w2, *b2, *w1, *b1;   // Keras (2,2,C,cx+cs), (C), (2,2,C,C), (C)
  float* out;                       // [n,2h,2w,C]
  int h, w, cx, cs, tiles_y, tiles_x;
  float alpha;
  // nlt_dec_block_forward_map (the reference's inference mode, engine_infer.py): `skip` is the interleaved encoder map
  // [query 4C | given 4C] with per-texel stride lds, of which only the QUERY half is read; what the given half adds to the first
  // conv's pre-activation (+ its bias) arrives as bmap [1,2h,2w,C], shared by all frames
  int lds;
  const float* bmap;
};

template <int C>
__global__ __launch_bounds__(256) void dec_block_kernel(DecP p) {
  constexpr int MT = C / 4;                       // stage 1: 16-row tiles of the 4C columns (a, b, o)
  constexpr int NQ = C / 4;                       // channel quads of the intermediate map
  constexpr int NS = 4 * C / 16;                  // stage 2: 16-wide K slabs
  __shared__ __attribute__((aligned(16))) float tile[NQ * FP * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = lane >> 4, j = lane & 15;
  int t_ = xcd_tile(blockIdx.x, gridDim.x);
  const int tx0 = (t_ % p.tiles_x) * TW; t_ /= p.tiles_x;
  const int ty0 = (t_ % p.tiles_y) * TH;
  const int f = t_ / p.tiles_y;
  const int K = p.cx + p.cs;
  const long hw = (long)p.h * p.w;

  // ---- stage 1
  for (int mt = wave; mt < NT; mt += 4) {
    const int t = mt * 16 + j;
    const bool live = t < HT;
    const int hy = live ? t / HW : 0, hx = live ? t % HW : 0;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool inside = live && gy >= 0 && gx >= 0 && gy < p.h && gx < p.w;
    const long tex = (long)f * hw + (inside ? (long)gy * p.w + gx : 0);
    const float* xp = p.x + tex * p.cx;
    const float* sp = p.skip + tex * p.cs - p.cx;                       // indexed by the concat channel
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K loop over 16-channel chunks in groups of up to GR: a group's texel loads (HBM) are ALL requested first -- one 16-byte
    // load per lane and chunk, GR of them in flight per wave --, the weight fragments (MT 16-byte loads per chunk, L1 / L2
    // resident) one chunk ahead of the MFMAs that use them.  (K % 16 != 0: the last chunk is partial -> zeros.)
    constexpr int GR = 10;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto fetch_a = [&](int c0, f32x4 (&a)[MT]) {
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = c0 < K ? *reinterpret_cast<const f32x4*>(p.w2 + (long)(16 * m + j) * K + c0) : z4;
    };
    const int nchunks = (K + 15) >> 4;
    for (int g0 = 0; g0 < nchunks; g0 += GR) {
      f32x4 bq[GR];
#pragma unroll
      for (int i = 0; i < GR; ++i) {
        const int c0 = 16 * (g0 + i) + 4 * kk;
        bq[i] = (g0 + i < nchunks && c0 < K) ? *reinterpret_cast<const f32x4*>(c0 < p.cx ? xp + c0 : sp + c0) : z4;
      }
      f32x4 acur[MT], anxt[MT];
      fetch_a(16 * g0 + 4 * kk, acur);
#pragma unroll
      for (int i = 0; i < GR; ++i) {
        if (g0 + i < nchunks) {                                          // wave-uniform
          if (i + 1 < GR && g0 + i + 1 < nchunks) fetch_a(16 * (g0 + i + 1) + 4 * kk, anxt);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[m][s4], bq[i][s4], acc[m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < MT; ++m) acur[m] = anxt[m];
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int col = 16 * m + 4 * kk;                                  // this lane's 4 consecutive columns: (ab, o0 .. o0 + 3)
      const int ab = col / C, o0 = col % C;
      f32x4 v = lrelu4(acc[m] + *reinterpret_cast<const f32x4*>(p.b2 + o0), p.alpha);
      if (!inside) v = (f32x4){0.f, 0.f, 0.f, 0.f};                     // zero padding above / left of the image
      const int ly = 2 * hy + (ab >> 1) - 1, lx = 2 * hx + (ab & 1) - 1;
      if (live && ly >= 0 && lx >= 0) *reinterpret_cast<f32x4*>(tile + ((o0 >> 2) * FP + ly * FW + lx) * 4) = v;
    }
  }
  __syncthreads();

  // ---- stage 2: 16 output rows x 2 segments of 16 texels = 32 column tiles, 8 per wave, 4 accumulators at a time
  f32x4 a1[NS];                                                          // A fragments: row j = output channel, K slab s
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tap = C >= 16 ? s / (C / 16) : 2 * s + (kk >> 1);
    const int q = C >= 16 ? (s % (C / 16)) * 4 + kk : (kk & 1);
    a1[s] = j < C ? *reinterpret_cast<const f32x4*>(p.w1 + ((long)(tap * C + j) * C + 4 * q)) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int H2 = 2 * p.h, W2 = 2 * p.w;
  const f32x4 bias1 = 4 * kk < C ? *reinterpret_cast<const f32x4*>(p.b1 + 4 * kk) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int tap = C >= 16 ? s / (C / 16) : 2 * s + (kk >> 1);
      const int q = C >= 16 ? (s % (C / 16)) * 4 + kk : (kk & 1);
      f32x4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ct = wave * 8 + g * 4 + u;                             // column tile -> (output row, 16-texel segment)
        const int oy = ct >> 1, ox = (ct & 1) * 16 + j;
        b[u] = *reinterpret_cast<const f32x4*>(tile + (q * FP + (oy + 1 - (tap >> 1)) * FW + ox + 1 - (tap & 1)) * 4);
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s][s4], b[u][s4], acc[u], 0, 0, 0);
    }
    if (4 * kk < C) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ct = wave * 8 + g * 4 + u;
        const int y = 2 * ty0 + (ct >> 1), xg = 2 * tx0 + (ct & 1) * 16 + j;
        if (y < H2 && xg < W2)
          *reinterpret_cast<f32x4*>(p.out + (((long)f * H2 + y) * W2 + xg) * C + 4 * kk) = lrelu4(acc[u] + bias1, p.alpha);
      }
    }
  }
}

// r04: the same block for the widths the U-Net actually has at these two levels -- x = the previous block's 2C channels, skip =
// [query | mean observation] of the INPUT level = 8C channels, K = 10C -- reorganised around what the r04 trace showed:
// dec_block_kernel ran at 0.16-0.26 of BOTH roofs (L10 45 us for 14 us of MFMAs and 59 MB) because every wave walked its 2-3 column
// tiles one after the other, each starting with an exposed HBM round trip for the texels and an L2 round trip for the weight
// fragments, then one for the bias, with the 10 column tiles dealt 3/3/2/2 to the four waves, and with all 512 workgroups of the
// launch in lockstep.
//   * a wave owns HALF of the 4C output columns (row tiles mh * MH ..) of FIVE of the ten column tiles: 20 balanced units,
//     and its weight fragments (MH x NCH 16-byte registers) are fetched once, not per column tile;
//   * the texel loads run NPF column tiles ahead of the MFMAs (requested before the weights; one exposed round trip per wave);
//   * biases and the second conv's fragments are fetched in the prologue, before the barrier.
// The accumulation order of every output is the one of dec_block_kernel: results are bit-identical.
template <int C, bool OVR = false>
__global__ __launch_bounds__(256, C == 8 ? 4 : 2) void dec_block10_kernel(DecP p) {
  constexpr int MT = C / 4, MH = MT / 2;          // stage 1: row tiles of the 4C columns (a, b, o); per wave
  constexpr int K = (OVR ? 6 : 10) * C, NCH = K / 16;   // 16-channel chunks of the virtual concat [x 2C | skip 8C] (OVR: [x 2C | query 4C])
  constexpr int LDS_ = OVR ? 0 : 8 * C;           // (OVR: the skip map's per-texel stride is an argument)
  constexpr int NQ = C / 4;                       // channel quads of the intermediate map
  constexpr int NS = 4 * C / 16;                  // stage 2: 16-wide K slabs
  constexpr int NI = NT / 2;                      // column tiles per wave
  constexpr int NPF = C == 8 ? 3 : 2;             // column tiles whose texels are in flight (registers: NPF x NCH x 4)
  __shared__ __attribute__((aligned(16))) float tile[NQ * FP * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = lane >> 4, j = lane & 15;
  const int mh = wave & 1, cg = wave >> 1;
  int t_ = xcd_tile(blockIdx.x, gridDim.x);
  const int tx0 = (t_ % p.tiles_x) * TW; t_ /= p.tiles_x;
  const int ty0 = (t_ % p.tiles_y) * TH;
  const int f = t_ / p.tiles_y;
  const long hw = (long)p.h * p.w;
  const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};

  // texels of column tile cg + 2 i: one 16-byte NHWC load per lane and chunk (permuted K: the lane's 4 channels feed 4 k-steps)
  auto load_b = [&](int i, f32x4 (&dst)[NCH]) {
    const int t = (cg + 2 * i) * 16 + j;
    const bool live = t < HT;
    const int hy = live ? t / HW : 0, hx = live ? t % HW : 0;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool inside = live && gy >= 0 && gx >= 0 && gy < p.h && gx < p.w;
    const long tex = (long)f * hw + (inside ? (long)gy * p.w + gx : 0);
    const float* xp = p.x + tex * (2 * C);
    const float* sp = p.skip + tex * (OVR ? p.lds : LDS_) - 2 * C;        // indexed by the concat channel
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = 16 * ch + 4 * kk;                                    // (C = 8: chunk 0 is x for every lane, 2C = 16)
      dst[ch] = *reinterpret_cast<const f32x4*>(c0 < 2 * C ? xp + c0 : sp + c0);
    }
  };
  // OVR: the map values of the lane's columns (ab, o0 .. o0 + 3) at column tile i's texel: the accumulators' initial values
  const int o0_ = (4 * kk) % C;
  auto load_m = [&](int i, f32x4 (&dst)[MH]) {
    if constexpr (OVR) {
      const int t = (cg + 2 * i) * 16 + j;
      const bool live = t < HT;
      const int hy = live ? t / HW : 0, hx = live ? t % HW : 0;
      const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
      const bool inside = live && gy >= 0 && gx >= 0 && gy < p.h && gx < p.w;
#pragma unroll
      for (int m = 0; m < MH; ++m) {
        const int ab = (16 * (mh * MH + m) + 4 * kk) / C;
        const long at = inside ? ((long)(2 * gy + (ab >> 1)) * (2 * p.w) + 2 * gx + (ab & 1)) * C + o0_ : 0;
        dst[m] = *reinterpret_cast<const f32x4*>(p.bmap + at);
      }
    }
  };
  f32x4 bq[NPF][NCH];
  f32x4 mq[NPF][MH];
#pragma unroll
  for (int i = 0; i < NPF; ++i) { load_b(i, bq[i]); load_m(i, mq[i]); }
  f32x4 a[MH][NCH];
#pragma unroll
  for (int m = 0; m < MH; ++m)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
      a[m][ch] = *reinterpret_cast<const f32x4*>(p.w2 + (long)(16 * (mh * MH + m) + j) * K + 16 * ch + 4 * kk);
  // this lane's 4 consecutive columns of row tile m: col = 16 m + 4 kk = (ab, o0 .. o0 + 3); o0 does not depend on m
  const int o0 = (4 * kk) % C;
  const f32x4 bias2 = *reinterpret_cast<const f32x4*>(p.b2 + o0);
  f32x4 a1[NS];                                                          // stage 2's A fragments: row j = output channel, K slab s
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tap = C >= 16 ? s / (C / 16) : 2 * s + (kk >> 1);
    const int q = C >= 16 ? (s % (C / 16)) * 4 + kk : (kk & 1);
    a1[s] = j < C ? *reinterpret_cast<const f32x4*>(p.w1 + ((long)(tap * C + j) * C + 4 * q)) : z4;
  }
  const f32x4 bias1 = 4 * kk < C ? *reinterpret_cast<const f32x4*>(p.b1 + 4 * kk) : z4;

  // ---- stage 1
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    f32x4 acc[MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) acc[m] = OVR ? mq[i % NPF][m] : z4;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int m = 0; m < MH; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][ch][s4], bq[i % NPF][ch][s4], acc[m], 0, 0, 0);
    if (i + NPF < NI) { load_b(i + NPF, bq[i % NPF]); load_m(i + NPF, mq[i % NPF]); }
    const int t = (cg + 2 * i) * 16 + j;
    const bool live = t < HT;
    const int hy = live ? t / HW : 0, hx = live ? t % HW : 0;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool inside = live && gy >= 0 && gx >= 0 && gy < p.h && gx < p.w;
#pragma unroll
    for (int m = 0; m < MH; ++m) {
      const int col = 16 * (mh * MH + m) + 4 * kk;
      const int ab = col / C;
      f32x4 v = lrelu4(OVR ? acc[m] : acc[m] + bias2, p.alpha);           // (OVR: the bias is part of the map)
      if (!inside) v = z4;                                               // zero padding above / left of the image
      const int ly = 2 * hy + (ab >> 1) - 1, lx = 2 * hx + (ab & 1) - 1;
      if (live && ly >= 0 && lx >= 0) *reinterpret_cast<f32x4*>(tile + ((o0 >> 2) * FP + ly * FW + lx) * 4) = v;
    }
  }
  __syncthreads();

  // ---- stage 2: 16 output rows x 2 segments of 16 texels = 32 column tiles, 8 per wave, 4 accumulators at a time
  const int H2 = 2 * p.h, W2 = 2 * p.w;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = z4;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int tap = C >= 16 ? s / (C / 16) : 2 * s + (kk >> 1);
      const int q = C >= 16 ? (s % (C / 16)) * 4 + kk : (kk & 1);
      f32x4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ct = wave * 8 + g * 4 + u;                             // column tile -> (output row, 16-texel segment)
        const int oy = ct >> 1, ox = (ct & 1) * 16 + j;
        b[u] = *reinterpret_cast<const f32x4*>(tile + (q * FP + (oy + 1 - (tap >> 1)) * FW + ox + 1 - (tap & 1)) * 4);
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s][s4], b[u][s4], acc[u], 0, 0, 0);
    }
    if (4 * kk < C) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ct = wave * 8 + g * 4 + u;
        const int y = 2 * ty0 + (ct >> 1), xg = 2 * tx0 + (ct & 1) * 16 + j;
        if (y < H2 && xg < W2)
          *reinterpret_cast<f32x4*>(p.out + (((long)f * H2 + y) * W2 + xg) * C + 4 * kk) = lrelu4(acc[u] + bias1, p.alpha);
      }
    }
  }
}

}  // namespace

extern "C" int nlt_dec_block_forward(const float* x, int cx, const float* skip, int cs, int n, int h, int w,
                                     const float* w_s2, const float* b_s2, const float* w_s1, const float* b_s1,
                                     int c, float alpha, float* out, void* stream) {
  if (!x || !skip || !w_s2 || !b_s2 || !w_s1 || !b_s1 || !out) return NLT_ERR_BAD_ARG;
  if (n <= 0 || h <= 0 || w <= 0 || cx <= 0 || cs <= 0) return NLT_ERR_BAD_ARG;
  if ((c != 8 && c != 16) || (cx & 3) || (cs & 3)) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(x) || !nlt_aligned16(skip) || !nlt_aligned16(w_s2) || !nlt_aligned16(b_s2) || !nlt_aligned16(w_s1) ||
      !nlt_aligned16(b_s1) || !nlt_aligned16(out))
    return NLT_ERR_BAD_ARG;
  if ((long long)n * h * w * 4 * (long long)(c > cs ? c : cs) >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  DecP p;
  p.x = x; p.skip = skip; p.w2 = w_s2; p.b2 = b_s2; p.w1 = w_s1; p.b1 = b_s1; p.out = out;
  p.h = h; p.w = w; p.cx = cx; p.cs = cs; p.alpha = alpha; p.lds = cs; p.bmap = nullptr;
  p.tiles_y = (h + TH - 1) / TH; p.tiles_x = (w + TW - 1) / TW;
  const long blocks = (long)n * p.tiles_y * p.tiles_x;
  hipStream_t s = static_cast<hipStream_t>(stream);
  static const bool generic_only = getenv("NLT_DEC_GENERIC") && atoi(getenv("NLT_DEC_GENERIC"));
  const bool ten = cx == 2 * c && cs == 8 * c && !generic_only;       // the U-Net's widths at these levels: x = 2C, skip = [q | mean obs] of the input level = 8C
  if (c == 8) {
    if (ten) hipLaunchKernelGGL(dec_block10_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(dec_block_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, p);
  } else {
    if (ten) hipLaunchKernelGGL(dec_block10_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(dec_block_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, p);
  }
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_dec_block_forward_map(const float* x, const float* skip, int lds, int n, int h, int w,
                                         const float* w_s2q, const float* w_s1, const float* b_s1, int c, float alpha,
                                         const float* bias_map, float* out, void* stream) {
  if (!x || !skip || !w_s2q || !w_s1 || !b_s1 || !bias_map || !out) return NLT_ERR_BAD_ARG;
  if (n <= 0 || h <= 0 || w <= 0) return NLT_ERR_BAD_ARG;
  if (c != 8 && c != 16) return NLT_ERR_UNSUPPORTED;
  if (lds < 4 * c || (lds & 3)) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(x) || !nlt_aligned16(skip) || !nlt_aligned16(w_s2q) || !nlt_aligned16(w_s1) || !nlt_aligned16(b_s1) ||
      !nlt_aligned16(bias_map) || !nlt_aligned16(out))
    return NLT_ERR_BAD_ARG;
  if ((long long)n * h * w * 4 * (long long)(c > lds ? c : lds) >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  DecP p;
  p.x = x; p.skip = skip; p.w2 = w_s2q; p.b2 = b_s1; p.w1 = w_s1; p.b1 = b_s1; p.out = out;     // (b2 unused: the map carries it)
  p.h = h; p.w = w; p.cx = 2 * c; p.cs = 4 * c; p.alpha = alpha; p.lds = lds; p.bmap = bias_map;
  p.tiles_y = (h + TH - 1) / TH; p.tiles_x = (w + TW - 1) / TW;
  const long blocks = (long)n * p.tiles_y * p.tiles_x;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c == 8) hipLaunchKernelGGL((dec_block10_kernel<8, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((dec_block10_kernel<16, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}
