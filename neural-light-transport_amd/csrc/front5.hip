// Fourth-generation fused front kernel (inference, precision f32x3_9 / f32x3): what front4.hip computes -- layers 0-1 of both
// paths, both observation means and level 2's stride-2 convs from the raw texel buffers (nlt/models/nlt.py:95-96,141-180) --
// with the 16-channel stages on the BF16 matrix cores through the three-term split of conv_tile3.hip (x = hi + mid + lo exactly,
// 8 + 8 + 8 significand bits; every bf16 x bf16 product is exact in the fp32 accumulator; 9 products per fp32 product, or 6),
// and with NO tile handed from one stage to the next through LDS.
//
// Why (r04 measurements, profiles/README.md):
//   * front4 without any global load or store still takes 0.202 of its 0.264 ms: it is bound by what its waves ISSUE, not by
//     HBM.  v_mfma_f32_16x16x4_f32 runs at the vector rate and does not co-execute with VALU work (SQ_VALU_MFMA_COEXEC_CYCLES
//     = 0 for every fp32-MFMA kernel of the pass): per wave 20.2 k cycles of fp32 MFMAs and 8.4 k cycles of VALU add up, for
//     both waves of a SIMD (model 57 k cycles per pair of strips, measured 60 k).
//   * a first bf16 version that kept front4's LDS hand-offs (stage-1 tile and level-1 tile written as three term planes) was no
//     faster: eight waves moved 460 KB per observation through the CU's one LDS port, whose STORE path (2 cycles per source
//     dword per wave instruction) then set the pace -- the kernel with every MFMA and split removed still took 0.106 ms.
//
// The hand-offs are therefore done in REGISTERS.  An MFMA leaves D[4 (lane >> 4) + r][lane & 15] in register r: lane (kk, j)
// holds output channels 4 kk .. 4 kk + 3 of texel j.  That IS the next conv's B operand for a K block whose slot (kk, e) means
// "input channel 4 kk + e" -- provided lane j of the consumer wants the same texel.  So every stage computes its texels in the
// order the NEXT stage's taps need them (lane j = level-2 texel (Y, X) = (j >> 3, j & 7) of the strip's 2 x 8 level-2 tile):
//   stage 3 (level 2, k2s2) tap (a, b)      reads level-1 texel (2Y + a, 2X + b)        -> stage 2 produces the four tap tiles O[a][b]
//   stage 2 (k2s1) tap (a', b') of O[a][b]  reads stage-1 texel (2Y + a + a', 2X + b + b') -> stage 1 produces nine tiles T[u][v], u, v in 0..2
//   stage 1 (folded L0 + k2s2) tap kk       reads raw texel (4Y + 2u + kk / 2, 4X + 2v + kk % 2) from the staged raw tile (LDS).
// The nine stage-1 tiles cover the haloed 5 x 17 tile 1.7 times (27 instead of 18 small fp32 MFMAs per observation); in return
// an observation needs no LDS tile, no LDS write besides the raw staging, and no intra-wave synchronisation between stages.
// A bf16 K block of 32 = (2 taps) x (16 channels): slots 0-3 of lane group kk = channels 4 kk.. of the first tap's tile, slots
// 4-7 = the second tap's, i.e. two tiles' term registers side by side.
//
// Organisation: one WORKGROUP = the eight waves of a CU (two per SIMD, <= 256 registers each), persistent: every wave walks
// its own sequence of 4 x 16 level-1 strips (no workgroup barrier after the weights are staged), and
//   * level 2's weights (its three bf16 terms: 12 KB observation + 24 KB query) and all biases live ONCE per CU in LDS; the
//     stride-1 conv's term fragments and stage 1's fp32 weights stay in registers for all strips;
//   * the raw inputs of a strip are a sequence of staged items -- observation 0 .. k - 1, then the query inputs -- that simply
//     continues into the next strip: while item t is computed, item t + 1 is converted into LDS and item t + 2 is in flight in
//     registers, so a wave never waits for the first loads of a strip (front4: every strip starts with an exposed HBM round trip);
//   * the observations' raw sum (the query path's mean input) is accumulated in an LDS tile while the items are staged.
// Stage 1 (K = 12 / 32) stays on the fp32 MFMA.  Biases are the accumulators' initial values.
// Results differ from front4 by re-association only (two fp32-accurate evaluations: ~3e-7 rel-L2 per output tensor, the same
// distance as between any two summation orders; tests/test_gpu_front5.py).
#include "front_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SH = 4, SW = 16;             // level-1 strip of one wave
constexpr int XH = 10;                     // raw rows of a strip incl. halo
constexpr int R3 = 104;                    // floats per staged 3-channel raw row (34 texels = 102, 26 float4)
constexpr int R1 = 40;                     // floats per staged 1-channel raw row
constexpr int W_R3 = 0;                    // raw 3-channel tile: observation (nn_rgb - nn_base), then the query's base
constexpr int W_RC = W_R3 + XH * R3;       // raw cvis
constexpr int W_RL = W_RC + XH * R1;       // raw lvis
constexpr int W_RS = W_RL + XH * R1;       // sum of the strip's raw observation tiles
constexpr int W_WAVE = W_RS + XH * R3;     // 2880 floats = 11520 B per wave
constexpr int NWAVES = 8;
constexpr int SA_O = 0;                    // level-2 weights, 16-byte units: obs [ct 2][chunk 2 = tap row][term 3][lane 64]
constexpr int SA_Q = SA_O + 2 * 2 * 3 * 64;           // query [ct 2][chunk 4 = slab * 2 + tap row][term 3][lane 64]
constexpr int SA_UNITS = SA_Q + 2 * 4 * 3 * 64;       // 2304 units = 36864 B
constexpr int W_BIAS = SA_UNITS * 4;       // floats: [bq2 | bo2 | bq1 | bo1] (16 each), [bq3 | bo3] (32 each)
constexpr int W_WAVES = W_BIAS + 128;
constexpr int LDS_FLOATS = W_WAVES + NWAVES * W_WAVE;   // 129536 B
constexpr int B_Q2 = 0, B_O2 = 16, B_Q1 = 32, B_O1 = 48, B_Q3 = 64, B_O3 = 96;

struct Front5In {
  const void *base, *cvis, *lvis, *nn_rgb, *nn_base;   // float buffers, or the uint8 stores (base = diffuse, nn_rgb = rgb store)
  const int *ids, *nn_ids;                              // U8 only: frame of each sample [n], of each observation [n,k] (-1: zeros)
};

__device__ __forceinline__ f32x4 u8x4_unit5(unsigned v) {
  return (f32x4){u8_unit(v & 255u), u8_unit((v >> 8) & 255u), u8_unit((v >> 16) & 255u), u8_unit(v >> 24)};
}

__device__ __forceinline__ void wave_sync5() {      // orders this wave's LDS traffic for the compiler; no instruction
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// two floats -> their bf16 roundings (nearest even), packed (low half = a)
__device__ __forceinline__ unsigned cvt_pk_bf16_5(float a, float b) {
  // (a conversion the compiler can see -- one v_cvt_pk_bf16_f32 -- not inline asm: the hazard recognizer pads MFMA -> VALU
  // read / write distances only for instructions it knows, r05)
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}

// four floats -> the three bf16 terms of each, packed in element order: t[term] = 4 bf16
__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&t)[3]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    const unsigned h = cvt_pk_bf16_5(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    const unsigned m = cvt_pk_bf16_5(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    t[0][p] = h; t[1][p] = m; t[2][p] = cvt_pk_bf16_5(sa, sb);
  }
}

// K block of two taps: slots 0-3 = the first tile's four channels, 4-7 = the second's
__device__ __forceinline__ bf16x8 pair8(const u32x2 a, const u32x2 b) {
  return __builtin_bit_cast(bf16x8, (u32x4){a[0], a[1], b[0], b[1]});
}

__device__ __forceinline__ f32x4 lrelu5(f32x4 v, float alpha) {        // 0 <= alpha <= 1 (checked by the launcher)
  return (f32x4){fmaxf(v[0], alpha * v[0]), fmaxf(v[1], alpha * v[1]), fmaxf(v[2], alpha * v[2]), fmaxf(v[3], alpha * v[3])};
}

// staged item in flight: float inputs = ten 16-byte pieces per lane, uint8 stores = six 8-byte pieces
template <bool U8> struct Stage5;
template <> struct Stage5<false> { f32x4 v[10]; };
template <> struct Stage5<true> { uint2 v[6]; };

// (weight term, texel term) of product pi, smallest products first: (2,2) (2,1) (1,2) (2,0) (0,2) (1,1) (1,0) (0,1) (0,0), two bits
// each; NPROD = 6 starts at pi = 3 (drops the three of relative order 2^-24)
__device__ __forceinline__ constexpr int ord_w(int pi) { return (5274 >> (2 * pi)) & 3; }
__device__ __forceinline__ constexpr int ord_b(int pi) { return (17958 >> (2 * pi)) & 3; }

template <bool U8, int NPROD>
__global__ __launch_bounds__(512, 1) void front5_kernel(
    Front5In in, int k, int h, int w, int tiles_y, int tiles_x, int ntiles, const float* __restrict__ blob, int add_base,
    float alpha, float* __restrict__ fm1, float* __restrict__ skip3, const float* __restrict__ blob3,
    float* __restrict__ qtmp2, float* __restrict__ otmp2) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int kk = lane >> 4, j = lane & 15;
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- level 2's weights: fp32 fragments of blob3 -> three bf16 terms in LDS, once per workgroup.  Unit (ct, chunk, lane
  // (lk, li)), half eh: tap (chunk row, eh), input channels 4 lk .. 4 lk + 3 (of the chunk's slab), output channel 16 ct + li
  {
    char* sa = reinterpret_cast<char*>(lds);
#pragma unroll
    for (int u = tid; u < 1536; u += 512) {
      const bool obs = u < 512;
      const int v = obs ? u : u - 512;
      const int eh = v & 1, ln = (v >> 1) & 63, lk = ln >> 4, li = ln & 15;
      const int mq = obs ? (v >> 7) & 1 : (v >> 7) & 3, ct = obs ? v >> 8 : v >> 9;
      const int tap = 2 * (mq & 1) + eh;
      const float* src = blob3 + (obs ? OFF3_AO + ((ct * 4 + lk) * 64 + tap * 16 + li) * 4
                                      : OFF3_AQ + ((ct * 8 + (mq >> 1) * 4 + lk) * 64 + tap * 16 + li) * 4);
      u32x2 t3[3];
      split4(*reinterpret_cast<const f32x4*>(src), t3);
      const int unit = obs ? SA_O + ((ct * 2 + mq) * 3) * 64 + ln : SA_Q + ((ct * 4 + mq) * 3) * 64 + ln;
#pragma unroll
      for (int t = 0; t < 3; ++t) *reinterpret_cast<u32x2*>(sa + (unit + t * 64) * 16 + eh * 8) = t3[t];
    }
    if (tid < 64) lds[W_BIAS + tid] = blob[OFF_BQ2 + tid];               // the four level-1 biases are contiguous in the blob
    else if (tid < 128) lds[W_BIAS + tid] = blob3[OFF3_BQ + tid - 64];   // so are level 2's
  }
  __syncthreads();

  // ---- this wave's strips: XCD x = blockIdx & 7 owns a contiguous run of tiles (neighbours share halo lines in its L2);
  // its waves take them round-robin (slot = wave * workgroups-per-XCD + workgroup: consecutive slots = consecutive strips)
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per_xcd = (ntiles + 7) >> 3;
  const int t_lo = (blockIdx.x & 7) * per_xcd;
  const int t_hi = min(t_lo + per_xcd, ntiles);
  const int stride = (gridDim.x >> 3) * NWAVES;
  int tile = t_lo + wv * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);   // wave-major: a small input still puts a wave on every CU
  if (tile >= t_hi) return;

  float* const wl = lds + W_WAVES + wv * W_WAVE;
  const char* const sa = reinterpret_cast<const char*>(lds);
  const int h2 = h >> 1, w2 = w >> 1, h4 = h2 >> 1, w4 = w2 >> 1;
  const long hw = (long)h * w;
  const float* const bl = lds + W_BIAS + 4 * kk;                         // biases = the accumulators' initial values
  auto bias4 = [&](int off) { return *reinterpret_cast<const f32x4*>(bl + off); };

  // ---- staging geometry of the strip whose items are being LOADED (runs ahead of the strip being computed)
  constexpr int N3 = U8 ? 13 : 26, P3 = U8 ? 3 : 5, E3 = U8 ? 8 : 4;
  constexpr int N1 = U8 ? 5 : 9, P1 = U8 ? 1 : 2, E1 = U8 ? 8 : 4;
  unsigned g3[P3], g1[P1];                                               // element offset inside a frame (0: a piece outside the image)
  int lf = 0;                                                            // sample (frame of the batch) of that strip
  // (`opaque`: the lane-only parts of these index computations are loop invariants; hoisted out of the strip loop they would
  // occupy ~40 registers -- recomputing them per strip costs ~100 VALU instructions)
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  auto load_geom = [&](int t) {
    const int tx0 = (t % tiles_x) * SW; t /= tiles_x;
    const int ty0 = (t % tiles_y) * SH;
    lf = t / tiles_y;
    const int ln = opaque(lane);
#pragma unroll
    for (int p = 0; p < P3; ++p) {
      const int item = p * 64 + ln;
      const int r = item / N3, i = item - r * N3;
      const int gy = 2 * ty0 + r;
      const bool ok = item < XH * N3 && gy < h && 3 * (2 * tx0) + E3 * i < 3 * w;
      g3[p] = ok ? (unsigned)((gy * w + 2 * tx0) * 3 + E3 * i) : 0u;
    }
#pragma unroll
    for (int p = 0; p < P1; ++p) {
      const int item = p * 64 + ln;
      const int r = item / N1, i = item - r * N1;
      const int gy = 2 * ty0 + r;
      const bool ok = item < XH * N1 && gy < h && 2 * tx0 + E1 * i < w;
      g1[p] = ok ? (unsigned)(gy * w + 2 * tx0 + E1 * i) : 0u;
    }
  };
  Stage5<U8> st;
  // loads are unconditional (wave-uniform frame pointer + 32-bit lane offset); a piece outside the image reads the frame's
  // first bytes: it only reaches stage-1 texels outside the image, whose outputs are forced to zero
  auto ld3 = [&](const void* arr, long frame, int at) {
#pragma unroll
    for (int p = 0; p < P3; ++p) {
      if constexpr (U8) st.v[at + p] = *reinterpret_cast<const uint2*>(static_cast<const unsigned char*>(arr) + frame * hw * 3 + g3[p]);
      else st.v[at + p] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(arr) + frame * hw * 3 + g3[p]);
    }
  };
  auto ld1 = [&](const void* arr, long frame, int at) {
#pragma unroll
    for (int p = 0; p < P1; ++p) {
      if constexpr (U8) st.v[at + p] = *reinterpret_cast<const uint2*>(static_cast<const unsigned char*>(arr) + frame * hw + g1[p]);
      else st.v[at + p] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(arr) + frame * hw + g1[p]);
    }
  };
  auto load_obs = [&](int i) {                                           // observation i of the strip being loaded
    long fr;
    if constexpr (U8) fr = in.nn_ids[lf * k + i]; else fr = (long)lf * k + i;
    if (fr < 0) {                                                        // a missing neighbour (wave-uniform): zeros
#pragma unroll
      for (int p = 0; p < 2 * P3; ++p) {
        if constexpr (U8) st.v[p] = make_uint2(0u, 0u); else st.v[p] = zero4;
      }
      return;
    }
    ld3(in.nn_rgb, fr, 0);
    ld3(in.nn_base, fr, P3);
  };
  auto load_query = [&]() {
    long fr;
    if constexpr (U8) fr = in.ids[lf]; else fr = lf;
    ld3(in.base, fr, 0);
    ld1(in.cvis, fr, P3);
    ld1(in.lvis, fr, P3 + P1);
  };
  // registers -> LDS, natural row layout: 16-byte stores at consecutive addresses.  An observation (a - b) is also added to
  // the strip's raw sum (first = the strip's observation 0: it starts the sum)
  auto st3 = [&](bool obs, bool first) {
#pragma unroll
    for (int p = 0; p < P3; ++p) {
      const int item = p * 64 + lane;
      if ((p + 1) * 64 > XH * N3 && item >= XH * N3) continue;            // only the last pass has lanes without an item
      if constexpr (U8) {
        f32x4 lo = u8x4_unit5(st.v[p].x), hi = u8x4_unit5(st.v[p].y);
        if (obs) { lo -= u8x4_unit5(st.v[P3 + p].x); hi -= u8x4_unit5(st.v[P3 + p].y); }
        *reinterpret_cast<f32x4*>(wl + W_R3 + item * 8) = lo;
        *reinterpret_cast<f32x4*>(wl + W_R3 + item * 8 + 4) = hi;
        if (obs) {
          f32x4* s = reinterpret_cast<f32x4*>(wl + W_RS + item * 8);
          if (first) { s[0] = lo; s[1] = hi; } else { s[0] += lo; s[1] += hi; }
        }
      } else {
        const f32x4 d = obs ? st.v[p] - st.v[P3 + p] : st.v[p];
        *reinterpret_cast<f32x4*>(wl + W_R3 + item * 4) = d;
        if (obs) {
          f32x4* s = reinterpret_cast<f32x4*>(wl + W_RS + item * 4);
          if (first) *s = d; else *s += d;
        }
      }
    }
  };
  auto st1 = [&](int dst, int at) {
    const int ln = opaque(lane);
#pragma unroll
    for (int p = 0; p < P1; ++p) {
      const int item = p * 64 + ln;
      if ((p + 1) * 64 > XH * N1 && item >= XH * N1) continue;
      const int r = item / N1, i = item - r * N1;
      if constexpr (U8) {
        *reinterpret_cast<f32x4*>(wl + dst + r * R1 + 8 * i) = u8x4_unit5(st.v[at + p].x);
        *reinterpret_cast<f32x4*>(wl + dst + r * R1 + 8 * i + 4) = u8x4_unit5(st.v[at + p].y);
      } else {
        *reinterpret_cast<f32x4*>(wl + dst + r * R1 + 4 * i) = st.v[at + p];
      }
    }
  };
  auto store_obs = [&](bool first) { st3(true, first); };
  auto store_query = [&]() { st3(false, false); st1(W_RC, P3); st1(W_RL, P3 + P1); };

  // ---- weights held in registers for every strip
  float ao2[3], aq2[8];
#pragma unroll
  for (int m = 0; m < 3; ++m) ao2[m] = blob[OFF_AO2 + m * 64 + lane];
#pragma unroll
  for (int m = 0; m < 8; ++m) aq2[m] = blob[OFF_AQ2 + m * 64 + lane];
  // stride-1 convs (2,2,16,16): chunk = tap row a'; slots 0-3 = tap (a', 0), channels 4 kk .., slots 4-7 = tap (a', 1)
  auto a1_load = [&](int base_off, f32x4 (&r)[4]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int eh = 0; eh < 2; ++eh) r[2 * a + eh] = *reinterpret_cast<const f32x4*>(blob + base_off + (((2 * a + eh) * 64 + lane) * 4));
  };
  auto a1_split = [&](const f32x4 (&r)[4], bf16x8 (&a)[2][3]) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      u32x2 ta[3], tc[3];
      split4(r[2 * m], ta);
      split4(r[2 * m + 1], tc);
#pragma unroll
      for (int t = 0; t < 3; ++t) a[m][t] = pair8(ta[t], tc[t]);
    }
  };
  bf16x8 a1o[2][3];
  {
    f32x4 r[4];
    a1_load(OFF_AO1, r);
    a1_split(r, a1o);
  }
  const float s0b = blob[OFF_BSK], s1b = blob[OFF_BSK + 1], s2b = blob[OFF_BSK + 2];
  const float inv_k = 1.f / (float)k;

  // ---- lane constants: lane (kk, j = (Y, X)); raw texel of stage-1 tap kk for tile (u, v): (4Y + 2u + kk / 2, 4X + 2v + kk % 2)
  const int Y = j >> 3, X = j & 7;
  const int rd3 = (4 * Y + (kk >> 1)) * R3 + (4 * X + (kk & 1)) * 3;     // + u * 2 * R3 + v * 6 (+ channel)
  const int rd1 = (4 * Y + (kk >> 1)) * R1 + 4 * X + (kk & 1);           // + u * 2 * R1 + v * 2

  // stage 2 as a stream over the stage-1 rows u = 0..2: row u's two K blocks (tiles (u,0)|(u,1) and (u,1)|(u,2)) feed tap row 0
  // of O[u][b] and tap row 1 of O[u - 1][b]
  auto stage2_row = [&](int u, const bf16x8 (&a)[2][3], const bf16x8 (&p)[2][3], f32x4 (&O)[2][2]) {
#pragma unroll
    for (int pi = 9 - NPROD; pi < 9; ++pi)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (u < 2) O[u < 2 ? u : 0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][ord_w(pi)], p[b][ord_b(pi)], O[u < 2 ? u : 0][b], 0, 0, 0);
        if (u > 0) O[u > 0 ? u - 1 : 0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][ord_w(pi)], p[b][ord_b(pi)], O[u > 0 ? u - 1 : 0][b], 0, 0, 0);
      }
  };
  // stage 3: level 2's stride-2 conv; chunk = tap row a: K block = tiles (a,0)|(a,1); weights from the workgroup's LDS copy
  auto stage3 = [&](int unit0, int nchunk, int chunk0, const f32x4 (&v)[2][2], f32x4 (&acc)[2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      u32x2 t0[3], t1[3];
      split4(v[a][0], t0);
      split4(v[a][1], t1);
      bf16x8 afr[2][3];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int t = 0; t < 3; ++t)
          afr[ct][t] = *reinterpret_cast<const bf16x8*>(sa + (unit0 + ((ct * nchunk + chunk0 + a) * 3 + t) * 64 + lane) * 16);
#pragma unroll
      for (int pi = 9 - NPROD; pi < 9; ++pi)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ct][ord_w(pi)], pair8(t0[ord_b(pi)], t1[ord_b(pi)]), acc[ct], 0, 0, 0);
    }
  };

  // ---- prologue: item 0 of the first strip into LDS, item 1 in flight
  load_geom(tile);
  load_obs(0);
  store_obs(true);
  if (k > 1) load_obs(1); else load_query();
  wave_sync5();

  for (;;) {
    // ---- geometry of the strip being computed
    int tt = tile;
    const int tx0 = (tt % tiles_x) * SW; tt /= tiles_x;
    const int ty0 = (tt % tiles_y) * SH;
    const int f = tt / tiles_y;
    const int next = tile + stride;
    const bool has_next = next < t_hi;
    const bool interior = ty0 + SH + 1 <= h2 && tx0 + SW + 1 <= w2;    // the haloed tile lies inside the image (wave-uniform)
    const int lim_r = h2 - ty0 - 2 * Y, lim_c = w2 - tx0 - 2 * X;      // stage-1 / level-1 texel (2Y + u, 2X + v) exists iff u < lim_r, v < lim_c
    const int gy2 = (ty0 >> 1) + Y, gx2 = (tx0 >> 1) + X;
    const bool in2 = gy2 < h4 && gx2 < w4;
    const long tex2 = (long)gy2 * w4 + gx2;

    f32x4 mean[2][2] = {{zero4, zero4}, {zero4, zero4}};

    for (int i = 0; i < k; ++i) {
      f32x4 O[2][2];
      {
        const f32x4 b1 = bias4(B_O1);
        O[0][0] = O[0][1] = O[1][0] = O[1][1] = b1;
      }
      const f32x4 b2 = bias4(B_O2);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        // ---- stage 1 (fp32 MFMA): folded L0 + L1 stride-2 conv at the three tiles of row u
        f32x4 T[3] = {b2, b2, b2};
        float d[3][3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const float* s = wl + W_R3 + rd3 + u * 2 * R3 + v * 6;
          d[v][0] = s[0]; d[v][1] = s[1]; d[v][2] = s[2];
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int v = 0; v < 3; ++v) T[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(ao2[m], d[v][m], T[v], 0, 0, 0);
        u32x2 tt3[3][3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          f32x4 x = lrelu5(T[v], alpha);
          if (!interior && !(u < lim_r && v < lim_c)) x = zero4;         // beyond the image: the stride-1 conv's zero padding
          split4(x, tt3[v]);
        }
        bf16x8 p[2][3];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int t = 0; t < 3; ++t) p[b][t] = pair8(tt3[b][t], tt3[b + 1][t]);
        // ---- stage 2
        stage2_row(u, a1o, p, O);
      }
      wave_sync5();                                                      // every lane has its raw values: the raw tile is free
      // item i + 1 goes to LDS, item i + 2 is requested
      if (i + 1 < k) store_obs(false); else store_query();
      if (i + 2 < k) load_obs(i + 2);
      else if (i + 2 == k) load_query();
      else if (has_next) { load_geom(next); load_obs(0); }
      // ---- stage 3: level 2's stride-2 conv of this observation's level-1 strip
      f32x4 o1[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          o1[a][b] = lrelu5(O[a][b], alpha);
          mean[a][b] += o1[a][b];
        }
      {
        f32x4 a3[2] = {bias4(B_O3), bias4(B_O3 + 16)};
        stage3(SA_O, 2, 0, o1, a3);
        if (in2) {
          float* o = otmp2 + (((long)f * k + i) * h4 * w4 + tex2) * 32 + 4 * kk;
          *reinterpret_cast<f32x4*>(o) = lrelu5(a3[0], alpha);
          *reinterpret_cast<f32x4*>(o + 16) = lrelu5(a3[1], alpha);
        }
      }
      wave_sync5();                                                      // the staged item is visible to the next stage 1
    }

    // ---- query path
    {
      float wsk[24];                                                     // wave-uniform (scalar loads)
#pragma unroll
      for (int rr = 0; rr < 24; ++rr) wsk[rr] = blob[OFF_WSK + rr];
      // the stride-1 conv's fp32 fragments (L2), requested before stage 1 and split into terms behind it
      f32x4 a1r[4];
      a1_load(OFF_AQ1, a1r);
      bf16x8 a1q[2][3];
      f32x4 O[2][2];
      {
        const f32x4 b1 = bias4(B_Q1);
        O[0][0] = O[0][1] = O[1][0] = O[1][1] = b1;
      }
      const f32x4 b2 = bias4(B_Q2);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        // stage 1 (fp32 MFMA, 8 per tile): raw = (base r g b, cvis, lvis, mean raw observation r g b)
        f32x4 T[3] = {b2, b2, b2};
        float raw[3][8];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const float* s = wl + W_R3 + rd3 + u * 2 * R3 + v * 6;
          const float* m = wl + W_RS + rd3 + u * 2 * R3 + v * 6;
          raw[v][0] = s[0]; raw[v][1] = s[1]; raw[v][2] = s[2];
          raw[v][3] = wl[W_RC + rd1 + u * 2 * R1 + v * 2]; raw[v][4] = wl[W_RL + rd1 + u * 2 * R1 + v * 2];
          raw[v][5] = m[0] * inv_k; raw[v][6] = m[1] * inv_k; raw[v][7] = m[2] * inv_k;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int v = 0; v < 3; ++v) T[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq2[m], raw[v][m], T[v], 0, 0, 0);
        u32x2 tt3[3][3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          f32x4 x = lrelu5(T[v], alpha);
          const bool inside = interior || (u < lim_r && v < lim_c);
          if (!inside) x = zero4;
          split4(x, tt3[v]);
          if (u < 2 && v < 2 && inside) {                                // the head's share of the L0 features (+ base): tiles (0..1, 0..1)
            float s0 = s0b, s1 = s1b, s2 = s2b;                          // own every raw texel of the strip exactly once
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              s0 = fmaf(raw[v][rr], wsk[rr * 3], s0);
              s1 = fmaf(raw[v][rr], wsk[rr * 3 + 1], s1);
              s2 = fmaf(raw[v][rr], wsk[rr * 3 + 2], s2);
            }
            if (add_base) { s0 += raw[v][0]; s1 += raw[v][1]; s2 += raw[v][2]; }
            float* sk = skip3 + ((long)f * hw + (long)(2 * (ty0 + 2 * Y + u) + (kk >> 1)) * w + 2 * (tx0 + 2 * X + v) + (kk & 1)) * 3;
            sk[0] = s0; sk[1] = s1; sk[2] = s2;
          }
        }
        if (u == 0) a1_split(a1r, a1q);
        bf16x8 p[2][3];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int t = 0; t < 3; ++t) p[b][t] = pair8(tt3[b][t], tt3[b + 1][t]);
        stage2_row(u, a1q, p, O);
      }
      wave_sync5();                                                      // the raw tiles are free
      if (has_next) {                                                    // next strip: observation 0 to LDS, its item 1 requested
        store_obs(true);
        if (k > 1) load_obs(1); else load_query();
      }
      f32x4 qv[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          qv[a][b] = lrelu5(O[a][b], alpha);
          mean[a][b] *= inv_k;
          if (a < lim_r && b < lim_c) {                                  // level-1 texel (2Y + a, 2X + b)
            float* o = fm1 + (((long)f * h2 + ty0 + 2 * Y + a) * w2 + tx0 + 2 * X + b) * 32 + 4 * kk;
            *reinterpret_cast<f32x4*>(o) = qv[a][b];
            *reinterpret_cast<f32x4*>(o + 16) = mean[a][b];
          }
        }
      // stage 3, query (2,2,32,32): slab 0 = q1, slab 1 = mean o1, accumulated in this order
      f32x4 a3[2] = {bias4(B_Q3), bias4(B_Q3 + 16)};
      stage3(SA_Q, 4, 0, qv, a3);
      stage3(SA_Q, 4, 2, mean, a3);
      if (in2) {
        float* o = qtmp2 + ((long)f * h4 * w4 + tex2) * 32 + 4 * kk;
        *reinterpret_cast<f32x4*>(o) = lrelu5(a3[0], alpha);
        *reinterpret_cast<f32x4*>(o + 16) = lrelu5(a3[1], alpha);
      }
      wave_sync5();
    }
    if (!has_next) break;
    tile = next;
  }
}

template <bool U8>
int front5_launch(const Front5In& in, int n, int k, int h, int w, const float* packed, const float* packed_l2, int add_base,
                  float alpha, float* fm1, float* skip3, float* qtmp2, float* otmp2, int products, void* stream) {
  if (!in.base || !in.cvis || !in.lvis || !in.nn_rgb || !in.nn_base || !packed || !packed_l2 || !fm1 || !skip3 || !qtmp2 || !otmp2)
    return NLT_ERR_BAD_ARG;
  if (U8 && (!in.ids || !in.nn_ids)) return NLT_ERR_BAD_ARG;
  if (n <= 0 || k <= 0 || h <= 0 || w <= 0) return NLT_ERR_BAD_ARG;
  if (products != 0 && products != 6 && products != 9) return NLT_ERR_BAD_ARG;
  if ((h | w) & 3) return NLT_ERR_UNSUPPORTED;                         // level 2 halves the half-resolution grid again
  if (U8 && (w & 7)) return NLT_ERR_UNSUPPORTED;                       // 8-byte pieces of a uint8 row
  if (!(alpha >= 0.f && alpha <= 1.f)) return NLT_ERR_UNSUPPORTED;     // LeakyReLU as max(v, alpha v)
  if (!nlt_aligned16(packed) || !nlt_aligned16(packed_l2) || !nlt_aligned16(fm1) || !nlt_aligned16(qtmp2) || !nlt_aligned16(otmp2))
    return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(in.base) || !nlt_aligned16(in.cvis) || !nlt_aligned16(in.lvis) || !nlt_aligned16(in.nn_rgb) ||
      !nlt_aligned16(in.nn_base))
    return NLT_ERR_UNSUPPORTED;                                        // row pieces are loaded 16 (8) bytes at a time
  if ((long long)n * k * h * w * 3 >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  const int ty = (h / 2 + SH - 1) / SH, tx = (w / 2 + SW - 1) / SW;
  const long tiles = (long)n * ty * tx;
  if (tiles >= (1l << 31)) return NLT_ERR_UNSUPPORTED;
  // one workgroup per CU (always a multiple of 8: one run of tiles per XCD); with fewer than 2048 strips its waves share them out
  const long per_xcd = (tiles + 7) / 8;
  long groups = per_xcd;                                               // workgroups per XCD: one per CU, fewer only below 32 strips per XCD
  if (groups > 32) groups = 32;
  const dim3 grid((unsigned)(8 * groups));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (products == 6)
    hipLaunchKernelGGL((front5_kernel<U8, 6>), grid, dim3(512), 0, s, in, k, h, w, ty, tx, (int)tiles, packed, add_base, alpha, fm1,
                       skip3, packed_l2, qtmp2, otmp2);
  else
    hipLaunchKernelGGL((front5_kernel<U8, 9>), grid, dim3(512), 0, s, in, k, h, w, ty, tx, (int)tiles, packed, add_base, alpha, fm1,
                       skip3, packed_l2, qtmp2, otmp2);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

}  // namespace

extern "C" int nlt_front5_forward(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                                  const float* nn_base, int n, int k, int h, int w, const float* packed,
                                  const float* packed_l2, int add_base, float alpha, float* fm1, float* skip3,
                                  float* qtmp2, float* otmp2, int products, void* stream) {
  Front5In in = {base, cvis, lvis, nn_rgb, nn_base, nullptr, nullptr};
  return front5_launch<false>(in, n, k, h, w, packed, packed_l2, add_base, alpha, fm1, skip3, qtmp2, otmp2, products, stream);
}

extern "C" int nlt_front5_forward_u8(const unsigned char* diffuse_store, const unsigned char* rgb_store,
                                     const unsigned char* cvis_store, const unsigned char* lvis_store,
                                     const int* ids, const int* nn_ids, int n, int k, int h, int w,
                                     const float* packed, const float* packed_l2, int add_base, float alpha,
                                     float* fm1, float* skip3, float* qtmp2, float* otmp2, int products, void* stream) {
  Front5In in = {diffuse_store, cvis_store, lvis_store, rgb_store, diffuse_store, ids, nn_ids};
  return front5_launch<true>(in, n, k, h, w, packed, packed_l2, add_base, alpha, fm1, skip3, qtmp2, otmp2, products, stream);
}
