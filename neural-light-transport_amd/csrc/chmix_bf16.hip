// bf16 1x1 channel-mixing conv on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulate): the literal
// dense GEMM of the path (BASELINE config 5: 2048^2 UV, bf16, HBM-roofline stress; SURVEY.md 8d "x64-ch" point).
//   out[t][o] = act(sum_c x[t][c] * W[c][o] + b[o]),   x / out bf16 NHWC, W bf16 (packed), b fp32.
// 256 B per texel at 64 -> 64 channels against ~8 KFLOP: HBM-bound by two orders of magnitude, so the kernel is
// built around memory: a wave owns 16 texels x all output channels, reads each texel's channels with 16-byte loads
// (lane group g = 8 consecutive channels of a 32-channel slab = one MFMA K step), keeps every weight fragment in
// registers for all the texel groups it walks (grid-stride), and sends the result through a 2 KB LDS transpose so
// that the stores are 16 bytes per lane and contiguous per texel.
// K slots: both operands give slot e of lane group g the SAME channel (32 * slab + 8 * g + e), so the hardware's
// internal k numbering of the 8-element fragments does not matter.
#include "nlt_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned short u16;

__device__ __forceinline__ u16 f2bf(float f) {                        // round to nearest even (NaN kept quiet)
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

// packed weights: [ct = cout/16][slab = cin/32][lane 64][8]  =  W[c = 32*slab + 8*(lane>>4) + e][o = 16*ct + (lane&15)]
__global__ void chmix_pack_kernel(const float* __restrict__ w, int cin, int cout, long total, u16* __restrict__ wp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7, lane = (idx >> 3) & 63;
  long r = idx >> 9;
  const int slabs = cin >> 5;
  const int slab = r % slabs, ct = r / slabs;
  const int c = 32 * slab + 8 * (lane >> 4) + e, o = 16 * ct + (lane & 15);
  wp[idx] = f2bf(w[(long)c * cout + o]);
}

template <int SLABS, int CTS>
__global__ __launch_bounds__(256) void chmix_bf16_kernel(const u16* __restrict__ x, long texels, const u16* __restrict__ wp,
                                                         const float* __restrict__ bias, int act, float alpha,
                                                         u16* __restrict__ out) {
  constexpr int CIN = 32 * SLABS, COUT = 16 * CTS;
  __shared__ __attribute__((aligned(16))) u16 stage[4][16 * COUT];   // per-wave output tile [texel][cout]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  bf16x8 a[CTS][SLABS];
#pragma unroll
  for (int ct = 0; ct < CTS; ++ct)
#pragma unroll
    for (int s = 0; s < SLABS; ++s) a[ct][s] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)(ct * SLABS + s) * 64 + lane) * 8);
  f32x4 bv[CTS];
#pragma unroll
  for (int ct = 0; ct < CTS; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(bias + 16 * ct + 4 * g);

  const long groups = (texels + 15) >> 4;
  const long stride = (long)gridDim.x * 4;
  for (long grp = (long)blockIdx.x * 4 + wave; grp < groups; grp += stride) {
    const long t = grp * 16 + j;
    const long tc = t < texels ? t : texels - 1;                      // clamped address, result discarded
    bf16x8 b[SLABS];
#pragma unroll
    for (int s = 0; s < SLABS; ++s) b[s] = *reinterpret_cast<const bf16x8*>(x + tc * CIN + 32 * s + 8 * g);
    u16* st = stage[wave];
#pragma unroll
    for (int ct = 0; ct < CTS; ++ct) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < SLABS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ct][s], b[s], acc, 0, 0, 0);
      acc += bv[ct];
      if (act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] > 0.f ? acc[e] : alpha * acc[e];
      }
      // lane holds outputs 16*ct + 4*g .. +3 of texel j
      ushort4 o = make_ushort4(f2bf(acc[0]), f2bf(acc[1]), f2bf(acc[2]), f2bf(acc[3]));
      *reinterpret_cast<ushort4*>(st + j * COUT + 16 * ct + 4 * g) = o;
    }
    // LDS transpose read: the wave's 16 x COUT tile is 32 * COUT contiguous bytes in global memory
    constexpr int VECS = 16 * COUT / 8;                               // 16-byte vectors in the tile
#pragma unroll
    for (int v = lane; v < VECS; v += 64) {
      const long tt = grp * 16 + (v * 8) / COUT;
      if (tt < texels) *reinterpret_cast<uint4*>(out + grp * 16 * COUT + v * 8) = *reinterpret_cast<const uint4*>(st + v * 8);
    }
  }
}

template <int SLABS>
int launch_cts(int cts, const u16* x, long texels, const u16* wp, const float* bias, int act, float alpha, u16* out, hipStream_t s) {
  const long groups = (texels + 15) >> 4;
  long blocks = (groups + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;                              // grid-stride over the rest: weights stay in registers
#define NLT_CM(C) if (cts == C) { hipLaunchKernelGGL((chmix_bf16_kernel<SLABS, C>), dim3((unsigned)blocks), dim3(256), 0, s, x, texels, wp, bias, act, alpha, out); NLT_CHECK_LAUNCH(); return NLT_OK; }
  NLT_CM(2) NLT_CM(4) NLT_CM(8)
#undef NLT_CM
  return NLT_ERR_UNSUPPORTED;
}

bool shape_ok(int cin, int cout) {
  return (cin == 32 || cin == 64 || cin == 128) && (cout == 32 || cout == 64 || cout == 128);
}

}  // namespace

extern "C" long nlt_chmix_bf16_packed_elems(int cin, int cout) { return shape_ok(cin, cout) ? (long)cin * cout : -1; }

extern "C" int nlt_chmix_bf16_pack(const float* w_keras, int cin, int cout, unsigned short* packed, void* stream) {
  if (!w_keras || !packed) return NLT_ERR_BAD_ARG;
  if (!shape_ok(cin, cout)) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(packed)) return NLT_ERR_BAD_ARG;
  const long total = (long)cin * cout;
  hipLaunchKernelGGL(chmix_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w_keras, cin, cout, total, packed);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_chmix_bf16_forward(const unsigned short* x, long texels, int cin, const unsigned short* packed,
                                      const float* bias, int cout, int act, float alpha, unsigned short* out, void* stream) {
  if (!x || !packed || !bias || !out || texels <= 0) return NLT_ERR_BAD_ARG;
  if (!shape_ok(cin, cout)) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(x) || !nlt_aligned16(packed) || !nlt_aligned16(bias) || !nlt_aligned16(out)) return NLT_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (cin >> 5) {
    case 1: return launch_cts<1>(cout >> 4, x, texels, packed, bias, act, alpha, out, s);
    case 2: return launch_cts<2>(cout >> 4, x, texels, packed, bias, act, alpha, out, s);
    case 4: return launch_cts<4>(cout >> 4, x, texels, packed, bias, act, alpha, out, s);
  }
  return NLT_ERR_UNSUPPORTED;
}
