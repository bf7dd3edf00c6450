// Pointwise / pooling layers of the config branches the released .ini files do not take (nlt/networks/elements.py:
// act = elu (:74-75), norm = pixel (:103-121), pool = max / avg (:81-94)); `upconv` (:42-48) is the bilinear resize of
// warp.hip followed by a k2s1 conv.  Executed layer by layer (nlt_amd/generic.py): these branches are about coverage,
// not speed -- one thread per texel (norm, pool) or per element (activations), NHWC fp32, any channel count.
#include "nlt_common.h"

namespace {

static inline unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }

// kind 0: LeakyReLU(alpha) / ReLU (alpha = 0); kind 1: ELU(alpha): x > 0 ? x : alpha (exp(x) - 1)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, long n, int kind, float alpha, float* __restrict__ y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  y[i] = v > 0.f ? v : (kind == 0 ? alpha * v : alpha * (expf(v) - 1.f));
}

// dx = g * f'(x), written from the OUTPUT y (elu: f'(x) = y + alpha for x <= 0; y <= 0 <=> x <= 0 when alpha > 0)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, long n, int kind,
                                                      float alpha, float* __restrict__ dx) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = y[i];
  dx[i] = g[i] * (v > 0.f ? 1.f : (kind == 0 ? alpha : v + alpha));
}

// pixel norm: y = x * rsqrt(mean_c(x^2) + eps)   (elements.py:103-121, eps = 1e-8)
__global__ __launch_bounds__(256) void pixelnorm_fwd_kernel(const float* __restrict__ x, long texels, int c, float eps,
                                                            float* __restrict__ y) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= texels) return;
  const float* p = x + t * c;
  float s = 0.f;
  for (int i = 0; i < c; ++i) s = fmaf(p[i], p[i], s);
  const float r = 1.f / sqrtf(s / (float)c + eps);
  for (int i = 0; i < c; ++i) y[t * c + i] = p[i] * r;
}

// dx = r g - x r^3 <g, x> / c   with r = (mean x^2 + eps)^(-1/2)
__global__ __launch_bounds__(256) void pixelnorm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, long texels,
                                                            int c, float eps, float* __restrict__ dx) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= texels) return;
  const float* p = x + t * c;
  const float* q = g + t * c;
  float s = 0.f, d = 0.f;
  for (int i = 0; i < c; ++i) { s = fmaf(p[i], p[i], s); d = fmaf(q[i], p[i], d); }
  const float r = 1.f / sqrtf(s / (float)c + eps);
  const float k = r * r * r * d / (float)c;
  for (int i = 0; i < c; ++i) dx[t * c + i] = r * q[i] - p[i] * k;
}

// 2 x 2 / stride 2 pooling, TF 'same' on even sizes (no padding).  kind 0: max, 1: average.
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ x, int n, int h, int w, int c, int kind,
                                                       long total, float* __restrict__ y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ch = i % c;
  const long t = i / c;
  const int ow = w >> 1, oh = h >> 1;
  const int ox = t % ow, oy = (t / ow) % oh, f = t / ((long)ow * oh);
  const float* p = x + (((long)f * h + 2 * oy) * w + 2 * ox) * c + ch;
  const float a = p[0], b = p[c], d = p[(long)w * c], e = p[(long)w * c + c];
  y[i] = kind == 0 ? fmaxf(fmaxf(a, b), fmaxf(d, e)) : (a + b + d + e) * 0.25f;
}

// max: the gradient goes to the FIRST maximal tap in row-major window order (TF MaxPoolGrad); average: a quarter each
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, int n, int h, int w,
                                                       int c, int kind, long total, float* __restrict__ dx) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ch = i % c;
  const long t = i / c;
  const int ow = w >> 1, oh = h >> 1;
  const int ox = t % ow, oy = (t / ow) % oh, f = t / ((long)ow * oh);
  const long base = (((long)f * h + 2 * oy) * w + 2 * ox) * c + ch;
  const long off[4] = {0, c, (long)w * c, (long)w * c + c};
  const float gi = g[i];
  if (kind == 1) {
    for (int k = 0; k < 4; ++k) dx[base + off[k]] = gi * 0.25f;
    return;
  }
  int best = 0;
  float m = x[base];
  for (int k = 1; k < 4; ++k) { const float v = x[base + off[k]]; if (v > m) { m = v; best = k; } }
  for (int k = 0; k < 4; ++k) dx[base + off[k]] = k == best ? gi : 0.f;
}

__global__ __launch_bounds__(256) void sub_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] - b[i];
}

// pred = y (+ base), texel (0,0) of every frame zeroed (nlt/models/nlt.py:99-110)
__global__ __launch_bounds__(256) void finish_pred_kernel(const float* __restrict__ y, const float* __restrict__ base, long per_frame,
                                                          long total, float* __restrict__ pred) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float v = y[i];
  if (base) v += base[i];
  if (i % per_frame < 3) v = 0.f;
  pred[i] = v;
}

}  // namespace

extern "C" int nlt_sub_forward(const float* a, const float* b, long count, float* out, void* stream) {
  if (!a || !b || !out || count <= 0) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(sub_kernel, dim3(blocks_for(count)), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, count, out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_finish_pred(const float* y, const float* base, int n, int h, int w, float* pred, void* stream) {
  if (!y || !pred || n <= 0 || h <= 0 || w <= 0) return NLT_ERR_BAD_ARG;
  const long per = (long)h * w * 3, total = per * n;
  hipLaunchKernelGGL(finish_pred_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), y, base, per, total, pred);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_act_forward(const float* x, long count, int kind, float alpha, float* y, void* stream) {
  if (!x || !y || count <= 0 || (kind != 0 && kind != 1)) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks_for(count)), dim3(256), 0, static_cast<hipStream_t>(stream), x, count, kind, alpha, y);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_act_backward(const float* g, const float* y, long count, int kind, float alpha, float* dx, void* stream) {
  if (!g || !y || !dx || count <= 0 || (kind != 0 && kind != 1)) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(count)), dim3(256), 0, static_cast<hipStream_t>(stream), g, y, count, kind, alpha, dx);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_pixelnorm_forward(const float* x, long texels, int c, float eps, float* y, void* stream) {
  if (!x || !y || texels <= 0 || c <= 0) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(pixelnorm_fwd_kernel, dim3(blocks_for(texels)), dim3(256), 0, static_cast<hipStream_t>(stream), x, texels, c, eps, y);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_pixelnorm_backward(const float* g, const float* x, long texels, int c, float eps, float* dx, void* stream) {
  if (!g || !x || !dx || texels <= 0 || c <= 0) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(pixelnorm_bwd_kernel, dim3(blocks_for(texels)), dim3(256), 0, static_cast<hipStream_t>(stream), g, x, texels, c, eps, dx);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_pool2x2_forward(const float* x, int n, int h, int w, int c, int kind, float* y, void* stream) {
  if (!x || !y || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (kind != 0 && kind != 1)) return NLT_ERR_BAD_ARG;
  if ((h | w) & 1) return NLT_ERR_UNSUPPORTED;                         // TF 'same' would pad odd sizes
  const long total = (long)n * (h / 2) * (w / 2) * c;
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, h, w, c, kind, total, y);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_pool2x2_backward(const float* g, const float* x, int n, int h, int w, int c, int kind, float* dx, void* stream) {
  if (!g || !x || !dx || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (kind != 0 && kind != 1)) return NLT_ERR_BAD_ARG;
  if ((h | w) & 1) return NLT_ERR_UNSUPPORTED;
  const long total = (long)n * (h / 2) * (w / 2) * c;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), g, x, n, h, w, c, kind, total, dx);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}
