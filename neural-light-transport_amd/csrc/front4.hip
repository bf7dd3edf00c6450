// Third-generation fused front kernel (inference): layers 0-1 of both paths, both observation means and level 2's
// stride-2 convs from the raw texel buffers (nlt/models/nlt.py:95-96,141-180) -- the arithmetic, weight blobs and
// results of front_kernel<true> (fused.hip), bit for bit -- organised so that NO workgroup barrier separates its stages:
//
//   one WAVE works on one 4 x 16 strip of level-1 (half-resolution) texels at a time, i.e. 8 x 32 raw texels plus the
//   halo, i.e. exactly ONE 16-texel MFMA column tile of level 2 (2 x 8).  Everything a strip needs is wave-private:
//   its raw rows (staged in LDS in their natural row layout by 16-byte loads, item by item), its haloed stage-1 tile
//   (5 x 17 -> 6 column tiles), its level-1 tile for the level-2 convs.  Eight such waves live on a CU (17.3 KB of LDS
//   and <= 256 registers each); they run independently, so one wave's staging / LDS traffic / stores run under another
//   wave's MFMAs instead of every workgroup of the chip marching through the same phases in lockstep (what the
//   barrier-separated stages of the first two generations did: their phases added up linearly -- profiles/README.md r02_a).
//
// Per observation (any k): stage 1 (folded L0 + L1 stride-2, 6 independent accumulators) -> stage 2 (L1 stride-1,
// 4 rows = 4 accumulators) -> stage 3 (level 2's stride-2 conv of that observation, 2 accumulators); only the running
// observation mean and the raw mean stay in registers.  The query path runs last.
//
// r04: the waves are PERSISTENT.  With one strip per one-wave workgroup (r02-r04_b) the k = 1 -> 4 sweep read 0.029 ms per
// observation and 0.144 ms for "everything else" -- the query path (1.5 observations' worth of MFMAs) plus, for every one of
// the 16 384 strips, a workgroup launch, the kernel-argument / weight / geometry prologue and an exposed HBM round trip for
// its first observation.  Now one workgroup = the eight waves of a CU, launched once; each wave walks its own sequence of
// strips and
//   * every weight and bias is fetched once per wave (the query path's level-2 fragments once per CU, into LDS) instead of
//     once per strip;
//   * the raw inputs of a strip are a sequence of staged items -- observation 0 .. k - 1, then the query inputs -- that
//     continues into the next strip: while item t is computed, item t + 1 is converted into LDS and item t + 2 is in flight
//     in registers, so no strip starts with a wait.
// The arithmetic of a strip was unchanged by r04's persistence (bit-identical to front_kernel<true> then).  r05: the biases became
// the initial values of the MFMA chains and the uint8 variant feeds raw bytes -- a re-association: <= 3.3e-7 max-abs from
// front_kernel<true>, float / uint8 forms <= 3e-7 rel-L2 apart (tests/test_gpu_front4.py: 5e-7 bars); float = train form bit for bit.
//
// U8 = true reads the resident uint8 capture store (nlt/datasets/nlt.py:131-136,173-181) and feeds the byte values themselves:
// see u8x4_unit.
#include "front_common.h"

namespace {

constexpr int SH = 4, SW = 16;             // level-1 strip of one wave
constexpr int AH = SH + 1, AW = SW + 1;    // haloed: 5 x 17 = 85 texels
constexpr int AT = AH * AW;
constexpr int NC = (AT + 15) / 16;         // 6 column tiles (96 slots)
constexpr int SLOTS = NC * 16;
constexpr int XH = 2 * AH;                 // raw rows: 10
constexpr int R3 = 104;                    // floats per staged 3-channel raw row (34 texels = 102, 26 float4)
constexpr int R1 = 40;                     // floats per staged 1-channel raw row (34 -> 5 x 8)
constexpr int W_RO = 0;                    // raw observation (nn_rgb - nn_base)
constexpr int W_RQ = W_RO + XH * R3;       // raw base
constexpr int W_RC = W_RQ + XH * R3;       // raw cvis
constexpr int W_RL = W_RC + XH * R1;       // raw lvis
constexpr int W_OT = W_RL + XH * R1;       // stage-1 tile [4 channel quads][96 slots][4]; also the level-1 tile of stage 3
constexpr int W_END = W_OT + 4 * SLOTS * 4;   // 4416 floats = 17664 B per wave
constexpr int NWAVES = 8;                  // waves of a workgroup = of a CU (two per SIMD)
constexpr int W_AQ3 = 0;                   // workgroup-shared: the query path's level-2 fragments [rt 2][c8 8][lane 64][4],
constexpr int W_AQ1 = W_AQ3 + 2 * 8 * 64 * 4;          // its stride-1 fragments [tap 4][lane 64][4]
constexpr int W_BIAS = W_AQ1 + 4 * 64 * 4;             // and every bias: [bq2 | bo2 | bq1 | bo1] (16 each), [bq3 | bo3] (32 each)
constexpr int W_WAVES = W_BIAS + 128;
constexpr int LDS_FLOATS = W_WAVES + NWAVES * W_END;   // 162 304 B of the CU's 163 840
constexpr int B_Q2 = 0, B_O2 = 16, B_Q1 = 32, B_O1 = 48, B_Q3 = 64, B_O3 = 96;

struct Front4In {
  const void *base, *cvis, *lvis, *nn_rgb, *nn_base;   // float buffers, or the uint8 stores (base = diffuse, nn_rgb = rgb store)
  const int *ids, *nn_ids;                              // U8 only: frame of each sample [n], of each observation [n,k] (-1: zeros)
};

// r05 (r04 review item 5): the uint8-store variant stages the raw BYTE values (v_cvt_f32_ubyteN: one instruction, exact) instead of
// `_load_data`'s float32(float64(u) / 255) (u8_unit: four instructions per byte, 192 per observation), and the 1 / 255 moves into
// the operands the raw values meet: the stage-1 A operands (the folded L0 + L1 stride-2 weights) and the head's skip rows are
// multiplied by fl(1 / 255) -- the first in registers, once per wave; the second at pack time (OFF_WSK8) -- and `+ base` becomes
// fma(byte, fl(1 / 255), .).  nn_rgb - nn_base is an exact integer in [-255, 255].  fl(W / 255) . u instead of W . fl(u / 255):
// the uint8 variant is no longer bit-identical to the float kernel on the assembled batch, it is <= 3e-7 rel-L2 from it
// (tests/test_gpu_front4.py; reported per output there).
__device__ __forceinline__ f32x4 u8x4_unit(unsigned v) {
  return (f32x4){(float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24)};
}

__device__ __forceinline__ void wave_sync() {       // orders this wave's LDS traffic for the compiler; no instruction
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// staged item in flight between its global loads and its LDS writes: float inputs = ten 16-byte pieces per lane (an
// observation: 5 of nn_rgb + 5 of nn_base; the query inputs: 5 of base + 2 of cvis + 2 of lvis), uint8 stores = six 8-byte ones
template <bool U8> struct Staged;
template <> struct Staged<false> { f32x4 v[10]; };
template <> struct Staged<true> { uint2 v[6]; };

// LeakyReLU for 0 <= alpha <= 1 (checked by the launcher): max(v, alpha * v), bit-identical to the select form.  r05: the
// products as <2 x float> multiplies (v_pk_mul_f32) and the maximum as v_med3_f32(v, alpha v, FLT_MAX) (= max for every finite value): 1.5 instead of 2 VALU
// instructions per value (fmaxf() on an MFMA result costs a third: the compiler quiets the operand first).
// NO inline-asm VALU in this kernel: the hazard recognizer pads MFMA -> VALU read / write distances with s_nop only for
// instructions it can see, and a v_max_f32 / v_pk_add_f32 written as asm landed inside an MFMA's write-back window in one
// build of the uint8 variant (obs texels wrong by 1e-1; which build depends on register allocation).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 lrelu4m(f32x4 v, f32x2 alpha2) {
  const f32x2 plo = (f32x2){v[0], v[1]} * alpha2, phi = (f32x2){v[2], v[3]} * alpha2;
  const float top = 3.4028234663852886e38f;   // FLT_MAX, not +inf: med3 with an infinity is folded back into the quieting max
  return (f32x4){__builtin_amdgcn_fmed3f(v[0], plo[0], top), __builtin_amdgcn_fmed3f(v[1], plo[1], top),
                 __builtin_amdgcn_fmed3f(v[2], phi[0], top), __builtin_amdgcn_fmed3f(v[3], phi[1], top)};
}

// a - b as two packed operations
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
  const f32x2 m1 = (f32x2){-1.f, -1.f};                 // fma(b, -1, a) = a - b, one rounding: v_pk_fma_f32
  const f32x2 lo = __builtin_elementwise_fma((f32x2){b[0], b[1]}, m1, (f32x2){a[0], a[1]});
  const f32x2 hi = __builtin_elementwise_fma((f32x2){b[2], b[3]}, m1, (f32x2){a[2], a[3]});
  return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}

// Two waves per SIMD (<= 256 registers).  A leaner variant (weights re-read per observation, query inputs fetched
// late, 10 KB of LDS, 168 registers, three waves per SIMD) measured SLOWER (0.27 vs 0.25 ms at k = 4, uint8): what
// limits the kernel is each wave's own MFMA duty cycle, not the number of waves (profiles/README.md r02_a; r04: fp32
// MFMAs and VALU instructions of a SIMD's waves do not overlap at all -- tools/micro/mfma_valu_overlap.hip).
// TRAIN = true additionally keeps what the backward pass reads (the maps nlt_front_forward_train keeps): the level-1
// stride-2 outputs of both paths (owned texels of the haloed stage-1 tile) and the per-observation level-1 maps.
struct Front4Keep { float *obs1, *qtmp1, *otmp1; };

template <bool U8, bool TRAIN>
__global__ __launch_bounds__(512, 1) void front4_kernel(
    Front4In in, int k, int h, int w, int tiles_y, int tiles_x, int ntiles, const float* __restrict__ blob, int add_base,
    float alpha, float* __restrict__ fm1, float* __restrict__ skip3, const float* __restrict__ blob3,
    float* __restrict__ qtmp2, float* __restrict__ otmp2, Front4Keep keep) {
  __shared__ __attribute__((aligned(16))) float lds_all[LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int kk = lane >> 4, j = lane & 15;
  const int h2 = h >> 1, w2 = w >> 1;
  const long hw = (long)h * w;
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x2 alpha2 = (f32x2){alpha, alpha};

  // ---- the query path's level-2 and stride-1 fragments and all biases: once per workgroup (128 registers of every wave otherwise)
#pragma unroll
  for (int u = tid; u < 2 * 8 * 64; u += 512)
    *reinterpret_cast<f32x4*>(lds_all + W_AQ3 + u * 4) = *reinterpret_cast<const f32x4*>(blob3 + OFF3_AQ + u * 4);
  if (tid < 256) *reinterpret_cast<f32x4*>(lds_all + W_AQ1 + tid * 4) = *reinterpret_cast<const f32x4*>(blob + OFF_AQ1 + tid * 4);
  else if (tid < 320) lds_all[W_BIAS + tid - 256] = blob[OFF_BQ2 + tid - 256];       // the four level-1 biases are contiguous in the blob
  else if (tid < 384) lds_all[W_BIAS + tid - 256] = blob3[OFF3_BQ + tid - 320];      // so are level 2's
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
  float* const lds = lds_all + W_WAVES + wv * W_END;

  // ---- staging geometry of the strip whose items are being LOADED (it runs ahead of the strip being computed): item =
  // pass * 64 + lane -> (raw row, piece of the row); LDS offset = item * piece floats
  constexpr int N3 = U8 ? 13 : 26, P3 = U8 ? 3 : 5, E3 = U8 ? 8 : 4;
  constexpr int N1 = U8 ? 5 : 9, P1 = U8 ? 1 : 2, E1 = U8 ? 8 : 4;
  // Loads are unconditional: wave-uniform frame pointer + 32-bit lane offset.  A piece that lies beyond the image's
  // bottom / right edge (or a lane without a piece) reads the frame's first bytes instead: whatever it delivers only
  // reaches stage-1 texels outside the image, whose outputs are forced to zero (`inside_m`), so no select is needed.
  unsigned g3[P3], g1[P1];                                               // BYTE offset inside a frame (32 bits: the load takes
                                                                         // it as the VGPR offset of an SGPR base -- no address VALU)
  int lf = 0;                                                            // sample (frame of the batch) of that strip
  // (`opaque`: the lane-only parts of these index computations are loop invariants; hoisted out of the strip loop they
  // would occupy ~40 registers of a kernel that has none to spare -- recomputing them per strip is ~100 VALU instructions)
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
      g3[p] = ok ? (unsigned)((gy * w + 2 * tx0) * 3 + E3 * i) * (U8 ? 1u : 4u) : 0u;
    }
#pragma unroll
    for (int p = 0; p < P1; ++p) {
      const int item = p * 64 + ln;
      const int r = item / N1, i = item - r * N1;
      const int gy = 2 * ty0 + r;
      const bool ok = item < XH * N1 && gy < h && 2 * tx0 + E1 * i < w;
      g1[p] = ok ? (unsigned)(gy * w + 2 * tx0 + E1 * i) * (U8 ? 1u : 4u) : 0u;
    }
  };
  Staged<U8> st;
  auto ld3 = [&](const void* arr, long frame, int at) {
#pragma unroll
    for (int p = 0; p < P3; ++p) {
      if constexpr (U8) st.v[at + p] = *reinterpret_cast<const uint2*>(static_cast<const unsigned char*>(arr) + frame * hw * 3 + g3[p]);
      else st.v[at + p] = *reinterpret_cast<const f32x4*>(static_cast<const unsigned char*>(arr) + frame * hw * 12 + g3[p]);
    }
  };
  auto ld1 = [&](const void* arr, long frame, int at) {
#pragma unroll
    for (int p = 0; p < P1; ++p) {
      if constexpr (U8) st.v[at + p] = *reinterpret_cast<const uint2*>(static_cast<const unsigned char*>(arr) + frame * hw + g1[p]);
      else st.v[at + p] = *reinterpret_cast<const f32x4*>(static_cast<const unsigned char*>(arr) + frame * hw * 4 + g1[p]);
    }
  };
  auto load_obs = [&](int i) {                                           // observation i of the strip being loaded
    long fr;
    if constexpr (U8) fr = in.nn_ids[lf * k + i]; else fr = (long)lf * k + i;
    if constexpr (U8) {                                                  // (float batches have every neighbour: the assembler wrote zeros)
      if (fr < 0) {                                                      // a missing neighbour (wave-uniform): zeros
#pragma unroll
        for (int p = 0; p < 2 * P3; ++p) st.v[p] = make_uint2(0u, 0u);
        return;
      }
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
  // registers -> LDS, natural row layout (a - b for an observation): 16-byte stores at consecutive addresses
  auto st3 = [&](float* dst, bool sub) {
#pragma unroll
    for (int p = 0; p < P3; ++p) {
      const int item = p * 64 + lane;
      // only the LAST pass can hold lanes without an item; spelled so that the full passes carry no lane-dependent branch
      // (the compiler does not derive lane < 64 here, and a branch per piece also makes its wait counts conservative)
      if ((p + 1) * 64 > XH * N3 && item >= XH * N3) continue;
      if constexpr (U8) {
        f32x4 lo = u8x4_unit(st.v[p].x), hi = u8x4_unit(st.v[p].y);
        if (sub) { lo -= u8x4_unit(st.v[P3 + p].x); hi -= u8x4_unit(st.v[P3 + p].y); }
        *reinterpret_cast<f32x4*>(dst + item * 8) = lo;
        *reinterpret_cast<f32x4*>(dst + item * 8 + 4) = hi;
      } else {
        *reinterpret_cast<f32x4*>(dst + item * 4) = sub ? sub4(st.v[p], st.v[P3 + p]) : st.v[p];
      }
    }
  };
  auto st1 = [&](float* dst, int at) {                                   // rows of R1 = 40 floats
    const int ln = opaque(lane);
#pragma unroll
    for (int p = 0; p < P1; ++p) {
      const int item = p * 64 + ln;
      if ((p + 1) * 64 > XH * N1 && item >= XH * N1) continue;
      const int r = item / N1, i = item - r * N1;
      if constexpr (U8) {
        *reinterpret_cast<f32x4*>(dst + r * R1 + 8 * i) = u8x4_unit(st.v[at + p].x);
        *reinterpret_cast<f32x4*>(dst + r * R1 + 8 * i + 4) = u8x4_unit(st.v[at + p].y);
      } else {
        *reinterpret_cast<f32x4*>(dst + r * R1 + 4 * i) = st.v[at + p];
      }
    }
  };
  auto store_obs = [&]() { st3(lds + W_RO, true); };
  auto store_query = [&]() { st3(lds + W_RQ, false); st1(lds + W_RC, P3); st1(lds + W_RL, P3 + P1); };

  // ---- weights: held in registers for every strip of the wave
  float ao2[3], aq2[8];
#pragma unroll
  for (int m = 0; m < 3; ++m) ao2[m] = blob[OFF_AO2 + m * 64 + lane];
#pragma unroll
  for (int m = 0; m < 8; ++m) aq2[m] = blob[OFF_AQ2 + m * 64 + lane];
  constexpr float INV255 = 1.0f / 255.0f;
  if constexpr (U8) {                                                    // stage 1 multiplies byte values (see u8x4_unit)
#pragma unroll
    for (int m = 0; m < 3; ++m) ao2[m] *= INV255;
#pragma unroll
    for (int m = 0; m < 8; ++m) aq2[m] *= INV255;
  }
  const float* const bl = lds_all + W_BIAS + 4 * kk;                     // biases: read where they are added
  auto bias4 = [&](int off) { return *reinterpret_cast<const f32x4*>(bl + off); };
  f32x4 ao1[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) ao1[t] = *reinterpret_cast<const f32x4*>(blob + OFF_AO1 + (t * 64 + lane) * 4);
  f32x4 ao3[2][4];                                                       // level 2, obs (2,2,16,32): [row tile][channel quad]
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) ao3[rt][c4] = *reinterpret_cast<const f32x4*>(blob3 + OFF3_AO + ((rt * 4 + c4) * 64 + lane) * 4);
  const float s0b = blob[OFF_BSK], s1b = blob[OFF_BSK + 1], s2b = blob[OFF_BSK + 2];
  const float inv_k = 1.f / (float)k;

  // ---- this lane's six stage-1 positions: haloed level-1 texel t = c * 16 + j, tap kk
  int rd3[NC];                                                           // float offset of the lane's raw texel in a 3-channel raw tile
  unsigned live_m = 0, own_m = 0;                                        // column tiles whose slot is a texel of the haloed tile / of the strip
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int t = c * 16 + j;
    const bool live = t < AT;
    const int hy = live ? t / AW : 0, hx = live ? t % AW : 0;
    rd3[c] = (2 * hy + (kk >> 1)) * R3 + (2 * hx + (kk & 1)) * 3;
    live_m |= (unsigned)live << c;
    own_m |= (unsigned)(live && hy < SH && hx < SW) << c;
  }
  float* const ot = lds + W_OT;

  // level-2 geometry (stage 3): lane = (tap kk, level-2 texel j = (Y, X) of the 2 x 8 tile)
  const int Y = j >> 3, X = j & 7;
  const int h4 = h2 >> 1, w4 = w2 >> 1;
  // level-1 tile in LDS: [channel quad][x parity][row 4][x / 2 8][4], parity plane 1 xor-swizzled by 8 slots so that the
  // 16 lanes a ds_read_b128 services (two taps x two rows) touch 16 different 16-byte slots
  const int l1_rd = ((kk & 1) * 32 + ((((2 * Y + (kk >> 1)) * 8) + X) ^ ((kk & 1) * 8))) * 4;
  auto l1_wr = [&](int row) { return (kk * 64 + (j & 1) * 32 + (((row * 8) + (j >> 1)) ^ ((j & 1) * 8))) * 4; };

  // stage 2: L1 stride-1 conv of the haloed tile in `ot` (TF 'same': taps (y + a, x + b)), four rows side by side
  auto stage2 = [&](const f32x4 (&a)[4], f32x4 bias, f32x4 (&out)[SH]) {
    const float* tilep = ot + kk * SLOTS * 4;
    f32x4 acc[SH] = {bias, bias, bias, bias};                            // r05: the bias is the chain's initial value (no add)                            // r05: the bias is the chain's initial value (no add)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 b[SH];
#pragma unroll
      for (int r = 0; r < SH; ++r) b[r] = *reinterpret_cast<const f32x4*>(tilep + ((r + (t >> 1)) * AW + j + (t & 1)) * 4);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int r = 0; r < SH; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s4], b[r][s4], acc[r], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < SH; ++r) out[r] = lrelu4m(acc[r], alpha2);
  };

  // ---- prologue: item 0 of the first strip into LDS, item 1 in flight
  load_geom(tile);
  load_obs(0);
  store_obs();
  if (k > 1) load_obs(1); else load_query();
  wave_sync();

  for (;;) {
    // ---- geometry of the strip being computed
    int tt = tile;
    const int tx0 = (tt % tiles_x) * SW; tt /= tiles_x;
    const int ty0 = (tt % tiles_y) * SH;
    const int f = tt / tiles_y;
    const int next = tile + stride;
    const bool has_next = next < t_hi;
    // a strip whose haloed tile lies inside the image needs no zero-padding masks (wave-uniform)
    const bool interior = ty0 + AH <= h2 && tx0 + AW <= w2;
    unsigned inside_m = live_m, owned_m = own_m;
    if (!interior) {
      inside_m = owned_m = 0;
      const int jo = opaque(j);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int t = c * 16 + jo;
        const bool live = (live_m >> c) & 1;
        const int hy = live ? t / AW : 0, hx = live ? t % AW : 0;
        const bool inside = live && ty0 + hy < h2 && tx0 + hx < w2;
        inside_m |= (unsigned)inside << c;
        owned_m |= (unsigned)(inside && hy < SH && hx < SW) << c;
      }
    }
    const int gy2 = (ty0 >> 1) + Y, gx2 = (tx0 >> 1) + X;
    const bool in2 = gy2 < h4 && gx2 < w4;
    const long tex2 = (long)gy2 * w4 + gx2;

    float xs[NC][3];
#pragma unroll
    for (int c = 0; c < NC; ++c) xs[c][0] = xs[c][1] = xs[c][2] = 0.f;
    f32x4 mean[SH] = {zero4, zero4, zero4, zero4};

    for (int i = 0; i < k; ++i) {
      // ---- stage 1: folded L0 + L1 stride-2 conv of observation i, three column tiles at a time (three independent
      // accumulators keep the matrix pipe issuing; six at once cost 21 more registers)
      f32x4 sv[NC];
#pragma unroll
      for (int c0 = 0; c0 < NC; c0 += 3) {
        const f32x4 b2 = bias4(B_O2);
        f32x4 acc[3] = {b2, b2, b2};
        float d[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* s = lds + W_RO + rd3[c0 + c];
          d[c][0] = s[0]; d[c][1] = s[1]; d[c][2] = s[2];
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ao2[m], d[c][m], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          xs[c0 + c][0] += d[c][0]; xs[c0 + c][1] += d[c][1]; xs[c0 + c][2] += d[c][2];
          sv[c0 + c] = lrelu4m(acc[c], alpha2);
        }
      }
      if (!interior) {                                                     // beyond the image: the stride-1 conv's zero padding.
        asm volatile("" ::: "memory");                                     // A BRANCH (wave-uniform, border strips only), not 24 selects in every strip
#pragma unroll
        for (int c = 0; c < NC; ++c)
          if (!((inside_m >> c) & 1)) sv[c] = zero4;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) *reinterpret_cast<f32x4*>(ot + (kk * SLOTS + c * 16 + j) * 4) = sv[c];
      if constexpr (TRAIN) {
        const int jt = opaque(j);
#pragma unroll
        for (int c = 0; c < NC; ++c)
          if ((owned_m >> c) & 1) {
            const int t = c * 16 + jt;
            float* o = keep.otmp1 + ((((long)f * k + i) * h2 + ty0 + t / AW) * w2 + tx0 + t % AW) * 16 + 4 * kk;
            *reinterpret_cast<f32x4*>(o) = sv[c];
          }
      }
      wave_sync();                                                         // every lane has its raw values: the raw tile is free
      // item i + 1 (requested an iteration ago) is converted and stored NEXT TO stage 2's MFMAs (same scheduling region,
      // no fence between them), item i + 2 is requested: the next observation, the query inputs, or the next strip's first
      if (i + 1 < k) store_obs(); else store_query();
      if (i + 2 < k) load_obs(i + 2);
      else if (i + 2 == k) load_query();
      else if (has_next) { load_geom(next); load_obs(0); }
      // ---- stage 2
      f32x4 o1[SH];
      stage2(ao1, bias4(B_O1), o1);
#pragma unroll
      for (int r = 0; r < SH; ++r) mean[r] += o1[r];
      if constexpr (TRAIN) {
#pragma unroll
        for (int r = 0; r < SH; ++r)
          if (ty0 + r < h2 && tx0 + j < w2)
            *reinterpret_cast<f32x4*>(keep.obs1 + ((((long)f * k + i) * h2 + ty0 + r) * w2 + tx0 + j) * 16 + 4 * kk) = o1[r];
      }
      wave_sync();                                                         // stage-2 reads of `ot` are done: it becomes the level-1 tile
      // ---- stage 3: level 2's stride-2 conv of this observation's level-1 strip
#pragma unroll
      for (int r = 0; r < SH; ++r) *reinterpret_cast<f32x4*>(ot + l1_wr(r)) = o1[r];
      wave_sync();
      {
        f32x4 a3[2] = {bias4(B_O3), bias4(B_O3 + 16)};                     // the biases: initial values of the two chains
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(ot + c4 * 256 + l1_rd);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a3[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ao3[0][c4][e], v[e], a3[0], 0, 0, 0);
            a3[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ao3[1][c4][e], v[e], a3[1], 0, 0, 0);
          }
        }
        if (in2) {
          float* o = otmp2 + (((long)f * k + i) * h4 * w4 + tex2) * 32 + 4 * kk;
          *reinterpret_cast<f32x4*>(o) = lrelu4m(a3[0], alpha2);
          *reinterpret_cast<f32x4*>(o + 16) = lrelu4m(a3[1], alpha2);
        }
      }
      wave_sync();                                                         // the level-1 tile is consumed: `ot` is free again
    }

    // ---- query path
    {
      float wsk[24];                                                       // wave-uniform: scalar loads
#pragma unroll
      for (int rr = 0; rr < 24; ++rr) wsk[rr] = blob[(U8 ? OFF_WSK8 : OFF_WSK) + rr];
      const int jq = opaque(j);
      // stage 1 (8 MFMAs per column tile): raw = (base r g b, cvis, lvis, mean raw observation r g b)
#pragma unroll
      for (int c0 = 0; c0 < NC; c0 += 3) {                                 // three column tiles at a time, as for the observations
        const f32x4 b2 = bias4(B_Q2);
        f32x4 acc[3] = {b2, b2, b2};
        float raw[3][8];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* s = lds + W_RQ + rd3[c0 + c];
          raw[c][0] = s[0]; raw[c][1] = s[1]; raw[c][2] = s[2];
          {
            const int t = (c0 + c) * 16 + jq;
            const bool live = (live_m >> (c0 + c)) & 1;
            const int hy = live ? t / AW : 0, hx = live ? t % AW : 0;
            const int o1c = (2 * hy + (kk >> 1)) * R1 + 2 * hx + (kk & 1);
            raw[c][3] = lds[W_RC + o1c]; raw[c][4] = lds[W_RL + o1c];
          }
          raw[c][5] = xs[c0 + c][0] * inv_k; raw[c][6] = xs[c0 + c][1] * inv_k; raw[c][7] = xs[c0 + c][2] * inv_k;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq2[m], raw[c][m], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          f32x4 v = lrelu4m(acc[c], alpha2);
          if (!interior) {
            asm volatile("" ::: "memory");
            if (!((inside_m >> (c0 + c)) & 1)) v = zero4;
          }
          *reinterpret_cast<f32x4*>(ot + (kk * SLOTS + (c0 + c) * 16 + j) * 4) = v;
          if constexpr (TRAIN) {
            if ((owned_m >> (c0 + c)) & 1) {
              const int t = (c0 + c) * 16 + jq;
              *reinterpret_cast<f32x4*>(keep.qtmp1 + (((long)f * h2 + ty0 + t / AW) * w2 + tx0 + t % AW) * 16 + 4 * kk) = v;
            }
          }
          if ((owned_m >> (c0 + c)) & 1) {                                 // the head's share of the L0 features (+ base)
            float s0 = s0b, s1 = s1b, s2 = s2b;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              s0 = fmaf(raw[c][rr], wsk[rr * 3], s0);
              s1 = fmaf(raw[c][rr], wsk[rr * 3 + 1], s1);
              s2 = fmaf(raw[c][rr], wsk[rr * 3 + 2], s2);
            }
            if (add_base) {
              if constexpr (U8) { s0 = fmaf(raw[c][0], INV255, s0); s1 = fmaf(raw[c][1], INV255, s1); s2 = fmaf(raw[c][2], INV255, s2); }
              else { s0 += raw[c][0]; s1 += raw[c][1]; s2 += raw[c][2]; }
            }
            const int t = (c0 + c) * 16 + jq;
            // (staging these rows through LDS for 16-byte stores was built and measured in r03: no change -- the stores are not
            // what the query path waits for)
            float* sk = skip3 + ((long)f * hw + (long)(2 * (ty0 + t / AW) + (kk >> 1)) * w + 2 * (tx0 + t % AW) + (kk & 1)) * 3;
            sk[0] = s0; sk[1] = s1; sk[2] = s2;
          }
        }
      }
      wave_sync();                                                         // the raw observation tile has long been free
      if (has_next) {                                                      // next strip: its observation 0 to LDS, its item 1 requested
        store_obs();
        if (k > 1) load_obs(1); else load_query();
      }
      f32x4 qv[SH];
      {
        f32x4 aq1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) aq1[t] = *reinterpret_cast<const f32x4*>(lds_all + W_AQ1 + (t * 64 + lane) * 4);
        stage2(aq1, bias4(B_Q1), qv);
      }
#pragma unroll
      for (int r = 0; r < SH; ++r) mean[r] *= inv_k;
      {
        const long hw2 = (long)h2 * w2;
        const int gx = tx0 + j;
#pragma unroll
        for (int r = 0; r < SH; ++r)
          if (ty0 + r < h2 && gx < w2) {
            float* o = fm1 + ((long)f * hw2 + (long)(ty0 + r) * w2 + gx) * 32 + 4 * kk;
            *reinterpret_cast<f32x4*>(o) = qv[r];
            *reinterpret_cast<f32x4*>(o + 16) = mean[r];
          }
      }
      wave_sync();
      // stage 3, query (2,2,32,32): slab 0 = q1 (c8 0..3), slab 1 = mean o1 (c8 4..7), accumulated in this order; fragments
      // from the workgroup's LDS copy
      f32x4 a3[2] = {bias4(B_Q3), bias4(B_Q3 + 16)};
#pragma unroll
      for (int slab = 0; slab < 2; ++slab) {
#pragma unroll
        for (int r = 0; r < SH; ++r) *reinterpret_cast<f32x4*>(ot + l1_wr(r)) = slab ? mean[r] : qv[r];
        wave_sync();
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(ot + c4 * 256 + l1_rd);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(lds_all + W_AQ3 + ((0 * 8 + slab * 4 + c4) * 64 + lane) * 4);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(lds_all + W_AQ3 + ((1 * 8 + slab * 4 + c4) * 64 + lane) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a3[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[e], v[e], a3[0], 0, 0, 0);
            a3[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[e], v[e], a3[1], 0, 0, 0);
          }
        }
        wave_sync();
      }
      if (in2) {
        float* o = qtmp2 + ((long)f * h4 * w4 + tex2) * 32 + 4 * kk;
        *reinterpret_cast<f32x4*>(o) = lrelu4m(a3[0], alpha2);
        *reinterpret_cast<f32x4*>(o + 16) = lrelu4m(a3[1], alpha2);
      }
    }
    if (!has_next) break;
    tile = next;
  }
}

template <bool U8, bool TRAIN = false>
int front4_launch(const Front4In& in, int n, int k, int h, int w, const float* packed, const float* packed_l2, int add_base,
                  float alpha, float* fm1, float* skip3, float* qtmp2, float* otmp2, int wps, void* stream,
                  Front4Keep keep = Front4Keep{nullptr, nullptr, nullptr}) {
  if (!in.base || !in.cvis || !in.lvis || !in.nn_rgb || !in.nn_base || !packed || !packed_l2 || !fm1 || !skip3 || !qtmp2 || !otmp2)
    return NLT_ERR_BAD_ARG;
  if (U8 && (!in.ids || !in.nn_ids)) return NLT_ERR_BAD_ARG;
  if (n <= 0 || k <= 0 || h <= 0 || w <= 0) return NLT_ERR_BAD_ARG;
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
  (void)wps;                                                           // one register allocation (2 waves per SIMD); kept in the ABI
  if (TRAIN && (!keep.obs1 || !keep.qtmp1 || !keep.otmp1 || !nlt_aligned16(keep.obs1) || !nlt_aligned16(keep.qtmp1) ||
                !nlt_aligned16(keep.otmp1)))
    return NLT_ERR_BAD_ARG;
  // one workgroup per CU (always a multiple of 8: one run of tiles per XCD); with fewer than 2048 strips its waves share them out
  const long per_xcd = (tiles + 7) / 8;
  long groups = per_xcd;                                               // workgroups per XCD: one per CU, fewer only below 32 strips per XCD
  if (groups > 32) groups = 32;
  hipLaunchKernelGGL((front4_kernel<U8, TRAIN>), dim3((unsigned)(8 * groups)), dim3(64 * NWAVES), 0, static_cast<hipStream_t>(stream),
                     in, k, h, w, ty, tx, (int)tiles, packed, add_base, alpha, fm1, skip3, packed_l2, qtmp2, otmp2, keep);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

}  // namespace

extern "C" int nlt_front4_forward(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                                  const float* nn_base, int n, int k, int h, int w, const float* packed,
                                  const float* packed_l2, int add_base, float alpha, float* fm1, float* skip3,
                                  float* qtmp2, float* otmp2, int waves_per_simd, void* stream) {
  Front4In in = {base, cvis, lvis, nn_rgb, nn_base, nullptr, nullptr};
  return front4_launch<false>(in, n, k, h, w, packed, packed_l2, add_base, alpha, fm1, skip3, qtmp2, otmp2, waves_per_simd, stream);
}

extern "C" int nlt_front4_forward_u8(const unsigned char* diffuse_store, const unsigned char* rgb_store,
                                     const unsigned char* cvis_store, const unsigned char* lvis_store,
                                     const int* ids, const int* nn_ids, int n, int k, int h, int w,
                                     const float* packed, const float* packed_l2, int add_base, float alpha,
                                     float* fm1, float* skip3, float* qtmp2, float* otmp2, int waves_per_simd,
                                     void* stream) {
  Front4In in = {diffuse_store, cvis_store, lvis_store, rgb_store, diffuse_store, ids, nn_ids};
  return front4_launch<true>(in, n, k, h, w, packed, packed_l2, add_base, alpha, fm1, skip3, qtmp2, otmp2, waves_per_simd, stream);
}

extern "C" int nlt_front4_forward_train(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                                        const float* nn_base, int n, int k, int h, int w, const float* packed,
                                        const float* packed_l2, int add_base, float alpha, float* fm1, float* skip3,
                                        float* qtmp2, float* otmp2, float* obs1, float* qtmp1, float* otmp1, void* stream) {
  Front4In in = {base, cvis, lvis, nn_rgb, nn_base, nullptr, nullptr};
  return front4_launch<false, true>(in, n, k, h, w, packed, packed_l2, add_base, alpha, fm1, skip3, qtmp2, otmp2, 0, stream,
                                    Front4Keep{obs1, qtmp1, otmp1});
}
