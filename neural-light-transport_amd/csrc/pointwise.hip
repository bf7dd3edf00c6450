// Per-texel (1x1) stages of the renderer: fused L0 stem, observation mean, output head,
// elementwise product.  All HBM-bound: 16-byte accesses, 4 lanes per texel where the texel
// vector is wide enough so that a wave's stores form 64-byte contiguous runs.
#include "nlt_common.h"

namespace {

// ---------------------------------------------------------------------------------------
// Stem: nlt/models/nlt.py:95-96 + layer 0 of both nets (convnet.py:44) + first obs mean.
// thread = (texel, quad of 4 output channels)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_kernel(
    const float* __restrict__ base, const float* __restrict__ cvis, const float* __restrict__ lvis,
    const float* __restrict__ nn_rgb, const float* __restrict__ nn_base, const float* __restrict__ obs_w,
    int n, int k, int hw, int c,
    const float* __restrict__ wq, const float* __restrict__ bq, const float* __restrict__ wo,
    const float* __restrict__ bo, float* __restrict__ fm0, float* __restrict__ obs0, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int quads = c >> 2;
  const int q = idx % quads;
  const long tex = idx / quads;          // over n*hw
  const int f = tex / hw;
  const long pix = tex - (long)f * hw;
  const int co = 4 * q;

  // query path: [base(3) | cvis | lvis] x wq(5,c)
  float in[5];
  in[0] = base[tex * 3 + 0]; in[1] = base[tex * 3 + 1]; in[2] = base[tex * 3 + 2];
  in[3] = cvis[tex]; in[4] = lvis[tex];
  f32x4 a = *reinterpret_cast<const f32x4*>(bq + co);
#pragma unroll
  for (int j = 0; j < 5; ++j) a += in[j] * *reinterpret_cast<const f32x4*>(wq + j * c + co);
  *reinterpret_cast<f32x4*>(fm0 + tex * 2 * c + co) = a;

  // observation path: (nn_rgb - nn_base) x wo(3,c) per observation; mean over k
  const f32x4 bov = *reinterpret_cast<const f32x4*>(bo + co);
  f32x4 w0 = *reinterpret_cast<const f32x4*>(wo + co);
  f32x4 w1 = *reinterpret_cast<const f32x4*>(wo + c + co);
  f32x4 w2 = *reinterpret_cast<const f32x4*>(wo + 2 * c + co);
  f32x4 mean = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < k; ++i) {
    const long ot = ((long)f * k + i) * hw + pix;
    const float d0 = nn_rgb[ot * 3 + 0] - nn_base[ot * 3 + 0];
    const float d1 = nn_rgb[ot * 3 + 1] - nn_base[ot * 3 + 1];
    const float d2 = nn_rgb[ot * 3 + 2] - nn_base[ot * 3 + 2];
    f32x4 o = bov + d0 * w0 + d1 * w1 + d2 * w2;
    *reinterpret_cast<f32x4*>(obs0 + ot * c + co) = o;
    mean += obs_w ? obs_w[f * k + i] * o : o;
  }
  mean *= 1.f / (float)k;
  *reinterpret_cast<f32x4*>(fm0 + tex * 2 * c + c + co) = mean;
}

__global__ __launch_bounds__(256) void obs_mean_kernel(const float* __restrict__ obs,
                                                       const float* __restrict__ obs_w, int n, int k,
                                                       int hw, int c, float* __restrict__ out, int ldo,
                                                       long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int quads = c >> 2;
  const int q = idx % quads;
  const long tex = idx / quads;
  const int f = tex / hw;
  const long pix = tex - (long)f * hw;
  f32x4 mean = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < k; ++i) {
    const f32x4 o = *reinterpret_cast<const f32x4*>(obs + (((long)f * k + i) * hw + pix) * c + 4 * q);
    mean += obs_w ? obs_w[f * k + i] * o : o;
  }
  mean *= 1.f / (float)k;
  *reinterpret_cast<f32x4*>(out + tex * ldo + 4 * q) = mean;
}

// ---------------------------------------------------------------------------------------
// Head: convnet.py:85 over [dec | skip] -> 3, + base (nlt.py:101-102), corner zero (:110).
// 8 lanes per texel: lane q takes channel quads q, q + 8, ... of the virtual concat, so a wave's loads of the
// 32-channel skip are 1 KB contiguous (one thread per texel made every 16-byte load touch 64 different cache
// lines); the three partial dot products are combined with 3 xor-shuffles and lane 0 finishes the texel.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ dec, int ldd, int cd,
                                                   const float* __restrict__ skip, int lds, int cs,
                                                   const float* __restrict__ wk, const float* __restrict__ bias,
                                                   const float* __restrict__ base, int hw, long total,
                                                   float* __restrict__ pred) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long tex = tid >> 3;
  const int q = tid & 7;
  const bool live = tex < total;
  const long tc = live ? tex : total - 1;                    // clamped: every lane of the shuffle group stays active
  const int qd = cd >> 2, quads = (cd + cs) >> 2;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int qq = q; qq < quads; qq += 8) {
    const f32x4 v = qq < qd ? *reinterpret_cast<const f32x4*>(dec + tc * ldd + 4 * qq)
                            : *reinterpret_cast<const f32x4*>(skip + tc * lds + 4 * (qq - qd));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* wr = wk + (4 * qq + j) * 3;
      a0 = fmaf(v[j], wr[0], a0); a1 = fmaf(v[j], wr[1], a1); a2 = fmaf(v[j], wr[2], a2);
    }
  }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off);
  }
  if (!live || q) return;
  a0 += bias[0]; a1 += bias[1]; a2 += bias[2];
  if (base) { a0 += base[tex * 3 + 0]; a1 += base[tex * 3 + 1]; a2 += base[tex * 3 + 2]; }
  if (tex % hw == 0) { a0 = 0.f; a1 = 0.f; a2 = 0.f; }     // texel (0,0) of every frame
  pred[tex * 3 + 0] = a0; pred[tex * 3 + 1] = a1; pred[tex * 3 + 2] = a2;
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  long count, float* __restrict__ out) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long n4 = count >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(a)[i] * reinterpret_cast<const f32x4*>(b)[i];
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) out[i] = a[i] * b[i];
}

inline unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

extern "C" int nlt_stem_forward(const float* base, const float* cvis, const float* lvis,
                                const float* nn_rgb, const float* nn_base, const float* obs_weights,
                                int n, int k, int h, int w, int c,
                                const float* wq, const float* bq, const float* wo, const float* bo,
                                float* fm0, float* obs0, void* stream) {
  if (!base || !cvis || !lvis || !nn_rgb || !nn_base || !wq || !bq || !wo || !bo || !fm0 || !obs0) return NLT_ERR_BAD_ARG;
  if (n <= 0 || k <= 0 || h <= 0 || w <= 0 || c <= 0) return NLT_ERR_BAD_ARG;
  if (c & 3) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(wq) || !nlt_aligned16(bq) || !nlt_aligned16(wo) || !nlt_aligned16(bo) ||
      !nlt_aligned16(fm0) || !nlt_aligned16(obs0)) return NLT_ERR_BAD_ARG;
  if ((long long)n * k * h * w * c >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  const long total = (long)n * h * w * (c >> 2);
  hipLaunchKernelGGL(stem_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     base, cvis, lvis, nn_rgb, nn_base, obs_weights, n, k, h * w, c, wq, bq, wo, bo, fm0, obs0, total);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_obs_mean_forward(const float* obs, const float* obs_weights, int n, int k, int hw, int c,
                                    float* out, int ldo, void* stream) {
  if (!obs || !out || n <= 0 || k <= 0 || hw <= 0 || c <= 0 || ldo < c) return NLT_ERR_BAD_ARG;
  if ((c & 3) || (ldo & 3)) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(obs) || !nlt_aligned16(out)) return NLT_ERR_BAD_ARG;
  const long total = (long)n * hw * (c >> 2);
  hipLaunchKernelGGL(obs_mean_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     obs, obs_weights, n, k, hw, c, out, ldo, total);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_head_forward(const float* dec, int ldd, int cd, const float* skip, int lds, int cs,
                                const float* w_keras, const float* bias, const float* base,
                                int n, int h, int w, float* pred, void* stream) {
  if (!dec || !w_keras || !bias || !pred || n <= 0 || h <= 0 || w <= 0 || cd <= 0 || cs < 0) return NLT_ERR_BAD_ARG;
  if (cs > 0 && !skip) return NLT_ERR_BAD_ARG;
  if (ldd < cd || (cs > 0 && lds < cs)) return NLT_ERR_BAD_ARG;
  if ((cd & 3) || (cs & 3) || (ldd & 3) || (cs > 0 && (lds & 3))) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(dec) || (cs > 0 && !nlt_aligned16(skip))) return NLT_ERR_BAD_ARG;
  const long total = (long)n * h * w;
  hipLaunchKernelGGL(head_kernel, dim3(blocks_for(total * 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dec, ldd, cd, skip, lds, cs, w_keras, bias, base, h * w, total, pred);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_mul_forward(const float* a, const float* b, long count, float* out, void* stream) {
  if (!a || !b || !out || count <= 0) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(a) || !nlt_aligned16(b) || !nlt_aligned16(out)) return NLT_ERR_BAD_ARG;
  long blocks = (count / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(mul_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, count, out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}
