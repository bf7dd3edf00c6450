// norm = layer / batch of nlt/networks/elements.py:51-56, the two trainable normalisations the config key `norm` offers
// (sits between each conv and its activation, nlt/networks/convnet.py:50-59,67-76).  Executed layer by layer
// (nlt_amd/generic.py).  Both are, per texel, y[c] = (x[c] - m) * r * gamma[c] + beta[c]:
//   kind 0, LayerNormalization(epsilon=0.001, center, scale), axis = channels: m, r = this texel's mean over its c
//           channels and rsqrt(biased variance + eps);
//   kind 1, BatchNormalization(momentum=0.99, epsilon=0.001) AS THE REFERENCE'S LOOP RUNS IT: nothing in nlt/ ever passes
//           `training=True` (networks/seq.py:36-41 calls `layer(x)`; models call `self.net[..](x)`), so Keras resolves the
//           layer to inference mode: m = moving_mean[c], r = rsqrt(moving_variance[c] + eps), and the moving statistics stay
//           at their initial (0, 1) because only training-mode calls update them.  gamma / beta still train.
// One WAVE per texel (lanes stride over the channels: coalesced 256-byte rows, shuffle reductions); the per-channel
// sums of the backward pass (dgamma, dbeta) are kept in registers across a wave's texels, added across the 4 waves of a
// workgroup through LDS in wave order, written as one partial row per workgroup and summed in workgroup order by a
// second launch: deterministic, no atomics.  c <= 1024 (16 channels per lane).
#include "nlt_common.h"

namespace {

constexpr int MAXJ = 16;            // channels per lane
constexpr int NORM_BLOCKS = 1024;   // upper bound of the backward grid (= rows of the partial-sum workspace)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// (m, r) of one texel for the lane's channel j (kind 1: per channel; kind 0: the same for all channels)
template <int KIND>
__device__ __forceinline__ void texel_stats(const float* __restrict__ p, int c, int lane, float eps, float& m, float& r) {
  if (KIND == 0) {
    float s = 0.f;
    for (int ch = lane; ch < c; ch += 64) s += p[ch];
    m = wave_sum(s) / (float)c;
    float v = 0.f;
    for (int ch = lane; ch < c; ch += 64) { const float d = p[ch] - m; v = fmaf(d, d, v); }
    r = 1.f / sqrtf(wave_sum(v) / (float)c + eps);
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const float* __restrict__ x, long texels, int c,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                       float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const long nw = (long)gridDim.x * 4;
  for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < texels; t += nw) {
    const float* p = x + t * c;
    float m = 0.f, r = 1.f;
    texel_stats<KIND>(p, c, lane, eps, m, r);
    for (int ch = lane; ch < c; ch += 64) {
      if (KIND == 1) { m = mean[ch]; r = 1.f / sqrtf(var[ch] + eps); }
      const float xh = (p[ch] - m) * r;
      y[t * c + ch] = fmaf(xh, gamma[ch], beta[ch]);
    }
  }
}

// dx, and this workgroup's partial sums of dgamma[c] = sum_texels g * xhat, dbeta[c] = sum_texels g
//   kind 0: g' = g * gamma; dx = r * (g' - mean_c(g') - xhat * mean_c(g' * xhat));   kind 1: dx = g * gamma * r
template <int KIND>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, long texels, int c,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ var, float eps, float* __restrict__ dx,
                                                       float* __restrict__ partial) {
  __shared__ float red[3][2 * 64 * MAXJ];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float dg[MAXJ], db[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) { dg[j] = 0.f; db[j] = 0.f; }
  const long nw = (long)gridDim.x * 4;
  for (long t = (long)blockIdx.x * 4 + wv; t < texels; t += nw) {
    const float* p = x + t * c;
    const float* q = g + t * c;
    float m = 0.f, r = 1.f, m1 = 0.f, m2 = 0.f;
    texel_stats<KIND>(p, c, lane, eps, m, r);
    if (KIND == 0) {
      float s1 = 0.f, s2 = 0.f;
      for (int ch = lane; ch < c; ch += 64) {
        const float gp = q[ch] * gamma[ch];
        s1 += gp;
        s2 = fmaf(gp, (p[ch] - m) * r, s2);
      }
      m1 = wave_sum(s1) / (float)c;
      m2 = wave_sum(s2) / (float)c;
    }
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int ch = lane + 64 * j;
      if (ch < c) {
        if (KIND == 1) { m = mean[ch]; r = 1.f / sqrtf(var[ch] + eps); }
        const float xh = (p[ch] - m) * r, gv = q[ch], gp = gv * gamma[ch];
        dx[t * c + ch] = KIND == 0 ? r * (gp - m1 - xh * m2) : gp * r;
        dg[j] = fmaf(gv, xh, dg[j]);
        db[j] += gv;
      }
    }
  }
  // waves 1..3 hand their sums to wave 0, which adds them in wave order
  if (wv > 0) {
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) { red[wv - 1][j * 64 + lane] = dg[j]; red[wv - 1][(MAXJ + j) * 64 + lane] = db[j]; }
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int ch = lane + 64 * j;
      if (ch < c) {
        float a = dg[j], b = db[j];
        for (int o = 0; o < 3; ++o) { a += red[o][j * 64 + lane]; b += red[o][(MAXJ + j) * 64 + lane]; }
        partial[(long)blockIdx.x * 2 * c + ch] = a;
        partial[(long)blockIdx.x * 2 * c + c + ch] = b;
      }
    }
  }
}

// dgamma[ch] += sum over workgroups (in order) of their partial sums; same for dbeta
__global__ __launch_bounds__(256) void norm_reduce_kernel(const float* __restrict__ partial, int blocks, int c,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * c) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += partial[(long)b * 2 * c + i];
  if (i < c) dgamma[i] += s; else dbeta[i - c] += s;
}

inline int norm_blocks(long texels) {
  const long b = (texels + 3) / 4;
  return (int)(b < NORM_BLOCKS ? b : NORM_BLOCKS);
}

}  // namespace

extern "C" long nlt_norm_workspace_floats(long texels, int c) {
  if (texels <= 0 || c <= 0 || c > 64 * MAXJ) return -1;
  return (long)norm_blocks(texels) * 2 * c;
}

extern "C" int nlt_norm_forward(int kind, const float* x, long texels, int c, const float* gamma, const float* beta,
                                const float* mean, const float* var, float eps, float* y, void* stream) {
  if (!x || !y || !gamma || !beta || texels <= 0 || c <= 0 || (kind != 0 && kind != 1)) return NLT_ERR_BAD_ARG;
  if (kind == 1 && (!mean || !var)) return NLT_ERR_BAD_ARG;
  if (c > 64 * MAXJ) return NLT_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = (int)((texels + 3) / 4 < 65536 ? (texels + 3) / 4 : 65536);
  if (kind == 0) hipLaunchKernelGGL(norm_fwd_kernel<0>, dim3(blocks), dim3(256), 0, s, x, texels, c, gamma, beta, mean, var, eps, y);
  else hipLaunchKernelGGL(norm_fwd_kernel<1>, dim3(blocks), dim3(256), 0, s, x, texels, c, gamma, beta, mean, var, eps, y);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_norm_backward(int kind, const float* g, const float* x, long texels, int c, const float* gamma,
                                 const float* mean, const float* var, float eps, float* dx, float* dgamma, float* dbeta,
                                 float* workspace, void* stream) {
  if (!g || !x || !gamma || !dx || !dgamma || !dbeta || !workspace || texels <= 0 || c <= 0 || (kind != 0 && kind != 1))
    return NLT_ERR_BAD_ARG;
  if (kind == 1 && (!mean || !var)) return NLT_ERR_BAD_ARG;
  if (c > 64 * MAXJ) return NLT_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = norm_blocks(texels);
  if (kind == 0) hipLaunchKernelGGL(norm_bwd_kernel<0>, dim3(blocks), dim3(256), 0, s, g, x, texels, c, gamma, mean, var, eps, dx, workspace);
  else hipLaunchKernelGGL(norm_bwd_kernel<1>, dim3(blocks), dim3(256), 0, s, g, x, texels, c, gamma, mean, var, eps, dx, workspace);
  NLT_CHECK_LAUNCH();
  hipLaunchKernelGGL(norm_reduce_kernel, dim3((2 * c + 255) / 256), dim3(256), 0, s, workspace, blocks, c, dgamma, dbeta);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}
