// Backward / optimizer kernels of the train step that are not conv GEMMs:
// LeakyReLU backward, observation-mean backward, L0 stem backward, head backward, resampler
// scatter-add (warp backward), bilinear-resize backward, L2 loss, fused Keras Adam-AMSGrad.
#include "nlt_common.h"

namespace {

inline unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }

// out[tex][ch] = g[tex][ch] * (y[tex][ch] > 0 ? 1 : alpha); c % 4 == 0; in place allowed
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* g, int ldg, const float* __restrict__ y, int ldy,
                                                        int c, long total, float alpha, float* out, int ldo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int quads = c >> 2;
  const int q = idx % quads;
  const long tex = idx / quads;
  f32x4 gv = *reinterpret_cast<const f32x4*>(g + tex * ldg + 4 * q);
  const f32x4 yv = *reinterpret_cast<const f32x4*>(y + tex * ldy + 4 * q);
#pragma unroll
  for (int j = 0; j < 4; ++j) gv[j] *= (yv[j] > 0.f) ? 1.f : alpha;
  *reinterpret_cast<f32x4*>(out + tex * ldo + 4 * q) = gv;
}

// dpre_obs[f,i,pix,:] = (partial[f,i,pix,:] + dmean[f,pix,:] * w_i / k) * lrelu'(obs_y[f,i,pix,:])
__global__ __launch_bounds__(256) void obs_mean_bwd_kernel(const float* __restrict__ dmean, int ldm,
                                                           const float* __restrict__ obs_y,
                                                           const float* __restrict__ obs_w, const float* partial,
                                                           int k, int hw, int c, float alpha, long total, float* out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int quads = c >> 2;
  const int q = idx % quads;
  const long tex = idx / quads;          // over n*hw
  const int f = tex / hw;
  const long pix = tex - (long)f * hw;
  const f32x4 dm = *reinterpret_cast<const f32x4*>(dmean + tex * ldm + 4 * q) * (1.f / (float)k);
  for (int i = 0; i < k; ++i) {
    const long o = (((long)f * k + i) * hw + pix) * c + 4 * q;
    f32x4 g = obs_w ? obs_w[f * k + i] * dm : dm;
    if (partial) g += *reinterpret_cast<const f32x4*>(partial + o);
    if (obs_y) {
      const f32x4 yv = *reinterpret_cast<const f32x4*>(obs_y + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] *= (yv[j] > 0.f) ? 1.f : alpha;
    }
    *reinterpret_cast<f32x4*>(out + o) = g;
  }
}

// One launch per level for both halves of dfm[l] = [query | observation mean] (the two launches above, fused):
//   query half, in place:  dfm[tex][0:c]  *= lrelu'(fm_y[tex][0:c])                      -> gradient w.r.t. q.s1's pre-activation
//   observation half:      dpre_obs[f,i]   = (partial[f,i] + dfm[tex][c:2c] w_i / k) * lrelu'(obs_y[f,i])
__global__ __launch_bounds__(256) void level_split_bwd_kernel(float* dfm, const float* __restrict__ fm_y, int ld,
                                                              const float* __restrict__ obs_y, const float* __restrict__ obs_w,
                                                              const float* partial, int k, int hw, int c, float alpha_q,
                                                              float alpha_o, long total, float* out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int quads = c >> 2;
  const int q = idx % quads;
  const long tex = idx / quads;          // over n*hw
  const int f = tex / hw;
  const long pix = tex - (long)f * hw;
  float* gq = dfm + tex * ld + 4 * q;
  f32x4 gv = *reinterpret_cast<const f32x4*>(gq);
  const f32x4 dm = *reinterpret_cast<const f32x4*>(gq + c) * (1.f / (float)k);
  const f32x4 yq = *reinterpret_cast<const f32x4*>(fm_y + tex * ld + 4 * q);
#pragma unroll
  for (int j = 0; j < 4; ++j) gv[j] *= (yq[j] > 0.f) ? 1.f : alpha_q;
  *reinterpret_cast<f32x4*>(gq) = gv;
  for (int i = 0; i < k; ++i) {
    const long o = (((long)f * k + i) * hw + pix) * c + 4 * q;
    f32x4 g = obs_w ? obs_w[f * k + i] * dm : dm;
    if (partial) g += *reinterpret_cast<const f32x4*>(partial + o);
    const f32x4 yv = *reinterpret_cast<const f32x4*>(obs_y + o);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] *= (yv[j] > 0.f) ? 1.f : alpha_o;
    *reinterpret_cast<f32x4*>(out + o) = g;
  }
}

// Stem backward: weight/bias gradients of the two L0 1x1 convs.  thread = (texel lane, quad);
// register accumulation over a grid-stride of texels, LDS atomics per block, global atomics.
__global__ __launch_bounds__(256) void stem_bwd_kernel(
    const float* __restrict__ base, const float* __restrict__ cvis, const float* __restrict__ lvis,
    const float* __restrict__ nn_rgb, const float* __restrict__ nn_base, const float* __restrict__ obs_w,
    int k, int hw, int c, long texels, const float* __restrict__ dfm0, const float* __restrict__ dobs0,
    float* dwq, float* dbq, float* dwo, float* dbo) {
  extern __shared__ __attribute__((aligned(16))) float part[];   // [10 rows][c]: wq 5, bq 1, wo 3, bo 1
  const int quads = c >> 2;
  const int tpb = blockDim.x / quads;
  const int tl = threadIdx.x / quads, q = threadIdx.x % quads;
  for (int i = threadIdx.x; i < 10 * c; i += blockDim.x) part[i] = 0.f;
  __syncthreads();
  if (tl < tpb) {
    const int co = 4 * q;
    f32x4 aq[6], ao[4];
#pragma unroll
    for (int j = 0; j < 6; ++j) aq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) ao[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (long tex = (long)blockIdx.x * tpb + tl; tex < texels; tex += (long)gridDim.x * tpb) {
      const int f = tex / hw;
      const long pix = tex - (long)f * hw;
      const f32x4 gq = *reinterpret_cast<const f32x4*>(dfm0 + tex * 2 * c + co);
      const f32x4 gm = *reinterpret_cast<const f32x4*>(dfm0 + tex * 2 * c + c + co) * (1.f / (float)k);
      aq[0] += base[tex * 3 + 0] * gq; aq[1] += base[tex * 3 + 1] * gq; aq[2] += base[tex * 3 + 2] * gq;
      aq[3] += cvis[tex] * gq; aq[4] += lvis[tex] * gq; aq[5] += gq;
      for (int i = 0; i < k; ++i) {
        const long ot = ((long)f * k + i) * hw + pix;
        f32x4 g = obs_w ? obs_w[f * k + i] * gm : gm;
        if (dobs0) g += *reinterpret_cast<const f32x4*>(dobs0 + ot * c + co);
        ao[0] += (nn_rgb[ot * 3 + 0] - nn_base[ot * 3 + 0]) * g;
        ao[1] += (nn_rgb[ot * 3 + 1] - nn_base[ot * 3 + 1]) * g;
        ao[2] += (nn_rgb[ot * 3 + 2] - nn_base[ot * 3 + 2]) * g;
        ao[3] += g;
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&part[j * c + co + e], aq[j][e]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&part[(6 + j) * c + co + e], ao[j][e]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 10 * c; i += blockDim.x) {
    const int row = i / c, col = i - row * c;
    float* dst = row < 5 ? dwq + row * c + col : row == 5 ? dbq + col : row < 9 ? dwo + (row - 6) * c + col : dbo + col;
    atomicAdd(dst, part[i]);
  }
}

// Head backward.  thread = (texel lane, quad of input channels over [dec | skip]).
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dec, int ldd, int cd,
                                                       const float* __restrict__ skip, int lds, int cs,
                                                       const float* __restrict__ wk, const float* __restrict__ dpred,
                                                       int hw, long texels, float* __restrict__ d_dec, int ldgd,
                                                       float* __restrict__ d_skip, int ldgs, float* dw, float* db) {
  extern __shared__ __attribute__((aligned(16))) float part[];   // [(cd+cs)*3 + 3]
  const int cin = cd + cs;
  const int quads = cin >> 2;
  const int tpb = blockDim.x / quads;
  const int tl = threadIdx.x / quads, q = threadIdx.x % quads;
  for (int i = threadIdx.x; i < cin * 3 + 3; i += blockDim.x) part[i] = 0.f;
  __syncthreads();
  if (tl < tpb) {
    const int c0 = 4 * q;
    const bool from_dec = c0 < cd;
    float wr[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int o = 0; o < 3; ++o) wr[j][o] = wk[(c0 + j) * 3 + o];
    float aw[4][3] = {{0.f}};
    float ab[3] = {0.f, 0.f, 0.f};
    for (long tex = (long)blockIdx.x * tpb + tl; tex < texels; tex += (long)gridDim.x * tpb) {
      float g[3] = {dpred[tex * 3 + 0], dpred[tex * 3 + 1], dpred[tex * 3 + 2]};
      if (tex % hw == 0) { g[0] = 0.f; g[1] = 0.f; g[2] = 0.f; }       // d(set_left_top_corner)
      const float* xp = from_dec ? dec + tex * ldd + c0 : skip + tex * lds + (c0 - cd);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xp);
      f32x4 dx;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dx[j] = g[0] * wr[j][0] + g[1] * wr[j][1] + g[2] * wr[j][2];
#pragma unroll
        for (int o = 0; o < 3; ++o) aw[j][o] += xv[j] * g[o];
      }
      float* dp = from_dec ? d_dec + tex * ldgd + c0 : d_skip + tex * ldgs + (c0 - cd);
      *reinterpret_cast<f32x4*>(dp) = dx;
      if (q == 0) { ab[0] += g[0]; ab[1] += g[1]; ab[2] += g[2]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int o = 0; o < 3; ++o) atomicAdd(&part[(c0 + j) * 3 + o], aw[j][o]);
    if (q == 0)
#pragma unroll
      for (int o = 0; o < 3; ++o) atomicAdd(&part[cin * 3 + o], ab[o]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cin * 3; i += blockDim.x) atomicAdd(dw + i, part[i]);
  if (threadIdx.x < 3) atomicAdd(db + threadIdx.x, part[cin * 3 + threadIdx.x]);
}

// Resampler backward w.r.t. data: 4-corner scatter-add with the forward weights.  Texel (0,0)
// is skipped: every background pixel maps there and its gradient is discarded by the corner
// mask anyway (nlt/models/nlt.py:110), so the one hot atomic address never exists.
//
// Device-scope float atomics are resolved behind the XCDs' L2s, one read-modify-write per touched line and instruction,
// so the cost is the number of (instruction, line) pairs.  Lane = (camera pixel, one of the 6 contiguous floats of the
// texel pair [fx, cx] x rgb): a wave covers 10 consecutive pixels, and ONE atomic instruction per texel row carries both
// corners and all three channels of all of them (neighbouring pixels of a chart: one or two lines), instead of one
// channel of one corner of 64 pixels.
constexpr int WARPB_PX = 10;
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ dcam, const float* __restrict__ warp,
                                                       int uvh, int uvw, int hcwc, long total, float* dpred) {
  const int lane = threadIdx.x & 63;
  const int j = lane / 6, sub = lane - 6 * j;
  const int right = sub >= 3 ? 1 : 0, ch = sub - 3 * right;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long p = wave * WARPB_PX + j;
  if (j >= WARPB_PX || p >= total) return;
  const int f = p / hcwc;
  const float x = warp[p * 2 + 0] * (float)uvw;
  const float y = warp[p * 2 + 1] * (float)uvh;
  if (!(x > -1.f && y > -1.f && x < (float)uvw && y < (float)uvh)) return;
  const int fx = (int)floorf(x), fy = (int)floorf(y);
  const int cx = fx + 1, cy = fy + 1;
  const float dx = (float)cx - x, dy = (float)cy - y;
  const float wx = right ? 1.f - dx : dx;
  const float g = dcam[p * 3 + ch];
  const int xi = fx + right;
  if (xi < 0 || xi > uvw - 1) return;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int yi = r ? cy : fy;
    // the forward's weights, formed the same way: (fx,fy) dx*dy, (cx,fy) (1-dx)*dy, (fx,cy) dx*(1-dy), (cx,cy) (1-dx)*(1-dy)
    const float wt = wx * (r ? 1.f - dy : dy);
    if (yi < 0 || yi > uvh - 1 || (xi == 0 && yi == 0) || wt == 0.f) continue;
    atomicAdd(dpred + (((long)f * uvh + yi) * uvw + xi) * 3 + ch, wt * g);
  }
}

__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dout, int h, int w, int c, int oh,
                                                         int ow, long total, float* dx) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const int ox = p % ow;
  const int oy = (p / ow) % oh;
  const int f = p / ((long)ow * oh);
  const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
  const float srcy = ((float)oy + 0.5f) * sy - 0.5f, srcx = ((float)ox + 0.5f) * sx - 0.5f;
  const float fly = floorf(srcy), flx = floorf(srcx);
  const int ylo = max((int)fly, 0), yhi = min((int)ceilf(srcy), h - 1);
  const int xlo = max((int)flx, 0), xhi = min((int)ceilf(srcx), w - 1);
  const float ly = srcy - fly, lx = srcx - flx;
  // out = (1-ly)*((1-lx)*tl + lx*tr) + ly*((1-lx)*bl + lx*br)
  for (int ch = 0; ch < c; ++ch) {
    const float g = dout[p * c + ch];
    atomicAdd(dx + (((long)f * h + ylo) * w + xlo) * c + ch, (1.f - ly) * (1.f - lx) * g);
    atomicAdd(dx + (((long)f * h + ylo) * w + xhi) * c + ch, (1.f - ly) * lx * g);
    atomicAdd(dx + (((long)f * h + yhi) * w + xlo) * c + ch, ly * (1.f - lx) * g);
    atomicAdd(dx + (((long)f * h + yhi) * w + xhi) * c + ch, ly * lx * g);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// loss[f] = mean_{h,w,c} (gt - pred)^2   (blockIdx.y = frame); loss zeroed by the launcher
__global__ __launch_bounds__(256) void l2_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                     long per, float* loss) {
  __shared__ float ws[4];
  const int f = blockIdx.y;
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
    const float d = gt[f * per + i] - pred[f * per + i];
    s += d * d;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss + f, (ws[0] + ws[1] + ws[2] + ws[3]) / (float)per);
}

// Keras `sample_weight` on MeanSquaredError(reduction='none') (nlt/losses.py:42-43): the [N,H,W] per-texel loss (mean over the
// c channels) times the weight map wt [N,H,W], then the mean over H,W: loss[f] = sum_px wt[px] * (sum_c d^2 / c) / hw
__global__ __launch_bounds__(256) void l2w_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                      const float* __restrict__ wt, long hw, int c, float* loss) {
  __shared__ float ws[4];
  const int f = blockIdx.y;
  float s = 0.f;
  for (long px = (long)blockIdx.x * blockDim.x + threadIdx.x; px < hw; px += (long)gridDim.x * blockDim.x) {
    const long e = (f * hw + px) * c;
    float m = 0.f;
    for (int ch = 0; ch < c; ++ch) {
      const float d = gt[e + ch] - pred[e + ch];
      m += d * d;
    }
    s += wt[f * hw + px] * (m / (float)c);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss + f, (ws[0] + ws[1] + ws[2] + ws[3]) / (float)hw);
}

// dpred = gloss[f] * wt[px] * 2 (pred - gt) / (c hw)
__global__ __launch_bounds__(256) void l2w_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                      const float* __restrict__ wt, const float* __restrict__ gloss,
                                                      long hw, int c, long total, float* __restrict__ dpred) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long px = i / c;
  dpred[i] = gloss[px / hw] * wt[px] * 2.f * (pred[i] - gt[i]) / ((float)c * (float)hw);
}

// dpred = gloss[f] * 2 (pred - gt) / per
__global__ __launch_bounds__(256) void l2_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                     const float* __restrict__ gloss, long per, long total,
                                                     float* __restrict__ dpred) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  dpred[i] = gloss[i / per] * 2.f * (pred[i] - gt[i]) / (float)per;
}

// The l2 train step's loss glue in one pass (gt = rgb * fg; per-example mean square; sum over examples / global batch; its
// gradient): replaces mul + l2 forward + sum + divide + their autograd mirror + l2 backward = 12 launches between the
// resampler and the first backward kernel.  blockIdx.y = frame; loss (one float, zeroed by the launcher) += sum_f mean_f / gbs.
__global__ __launch_bounds__(256) void l2_step_kernel(const float* __restrict__ pred, const float* __restrict__ rgb,
                                                      const float* __restrict__ fg, long per, float inv_gbs,
                                                      float* __restrict__ gt, float* __restrict__ dpred, float* loss) {
  __shared__ float ws[4];
  const int f = blockIdx.y;
  float s = 0.f;
  const float two_gbs = inv_gbs * 2.f, fper = (float)per;           // dpred = gloss[f] * 2 (pred - gt) / per with gloss = 1 / gbs (l2_bwd_kernel's form)
  const uintptr_t al = reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(rgb) | reinterpret_cast<uintptr_t>(fg) |
                       reinterpret_cast<uintptr_t>(gt) | reinterpret_cast<uintptr_t>(dpred);
  if ((per & 3) == 0 && (al & 15) == 0) {
    // 16-byte accesses, two quads per thread and pass in flight (the 4-byte form ran 64 dependent passes per thread: 38 us
    // for 63 MB); same per-element arithmetic, the block's partial sums meet in the same order
    const long q4 = per >> 2, stride = (long)gridDim.x * blockDim.x;
    const f32x4* p4 = reinterpret_cast<const f32x4*>(pred + f * per);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(rgb + f * per);
    const f32x4* f4 = reinterpret_cast<const f32x4*>(fg + f * per);
    f32x4* g4 = reinterpret_cast<f32x4*>(gt + f * per);
    f32x4* d4 = reinterpret_cast<f32x4*>(dpred + f * per);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < q4; i += 2 * stride) {
      const long j = i + stride;
      const bool two = j < q4;
      const f32x4 pa = p4[i], ra = r4[i], fa = f4[i];
      const f32x4 pb = two ? p4[j] : pa, rb = two ? r4[j] : ra, fb = two ? f4[j] : fa;
      const f32x4 ga = ra * fa, da = pa - ga;
      g4[i] = ga; d4[i] = (two_gbs * da) / fper;
      s += (da[0] * da[0] + da[1] * da[1]) + (da[2] * da[2] + da[3] * da[3]);
      if (two) {
        const f32x4 gb = rb * fb, db = pb - gb;
        g4[j] = gb; d4[j] = (two_gbs * db) / fper;
        s += (db[0] * db[0] + db[1] * db[1]) + (db[2] * db[2] + db[3] * db[3]);
      }
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
      const long e = f * per + i;
      const float g = rgb[e] * fg[e];
      const float d = pred[e] - g;
      gt[e] = g;
      dpred[e] = two_gbs * d / fper;
      s += d * d;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (ws[0] + ws[1] + ws[2] + ws[3]) / (float)per * inv_gbs);
}

// x[f, :] *= s[f]
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                         long per, long total, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  out[i] = x[i] * s[i / per];
}

// Keras Adam(amsgrad=True), TF 2.2 OptimizerV2 form (epsilon OUTSIDE the bias-corrected lr):
//   m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; vhat = max(vhat, v); p -= lr_t * m / (sqrt(vhat) + eps)
__global__ __launch_bounds__(256) void adam_amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           float* __restrict__ vhat, long count, float lr_t,
                                                           float b1, float b2, float eps) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  const float vh = fmaxf(vhat[i], vi);
  m[i] = mi; v[i] = vi; vhat[i] = vh;
  p[i] -= lr_t * mi / (sqrtf(vh) + eps);
}

// Keras `clipnorm` = tf.clip_by_norm per variable: t * clip / max(||t||_2, clip), evaluated in that order (multiply, then
// divide -- also when nothing is clipped).  One workgroup per slot of the flat gradient bucket; the sum of squares is
// reduced in a fixed order (lane-strided partial sums, then a tree), so the result is deterministic.
__global__ __launch_bounds__(256) void clip_slots_kernel(float* __restrict__ g, const long* __restrict__ slots, float clip) {
  __shared__ float part[256];
  const long off = slots[2 * blockIdx.x], cnt = slots[2 * blockIdx.x + 1];
  float s = 0.f;
  for (long i = threadIdx.x; i < cnt; i += 256) { const float v = g[off + i]; s = fmaf(v, v, s); }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  const float l2 = part[0];
  const float norm = l2 > 0.f ? sqrtf(l2) : l2;
  const float den = fmaxf(norm, clip);
  for (long i = threadIdx.x; i < cnt; i += 256) g[off + i] = (g[off + i] * clip) / den;
}

}  // namespace

extern "C" int nlt_clip_by_norm_slots(float* grad, const long* slots, int n_slots, float clipnorm, void* stream) {
  if (!grad || !slots || n_slots <= 0 || !(clipnorm > 0.f)) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(clip_slots_kernel, dim3(n_slots), dim3(256), 0, static_cast<hipStream_t>(stream), grad, slots, clipnorm);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_lrelu_backward(const float* g, int ldg, const float* y, int ldy, int c, long texels, float alpha,
                                  float* out, int ldo, void* stream) {
  if (!g || !y || !out || c <= 0 || texels <= 0 || ldg < c || ldy < c || ldo < c) return NLT_ERR_BAD_ARG;
  if ((c & 3) || (ldg & 3) || (ldy & 3) || (ldo & 3)) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(g) || !nlt_aligned16(y) || !nlt_aligned16(out)) return NLT_ERR_BAD_ARG;
  const long total = texels * (c >> 2);
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     g, ldg, y, ldy, c, total, alpha, out, ldo);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_obs_mean_backward(const float* dmean, int ldm, const float* obs_y, const float* obs_weights,
                                     const float* dobs_partial, int n, int k, int hw, int c, float alpha,
                                     float* dpre_obs, void* stream) {
  if (!dmean || !dpre_obs || n <= 0 || k <= 0 || hw <= 0 || c <= 0 || ldm < c) return NLT_ERR_BAD_ARG;
  if ((c & 3) || (ldm & 3)) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(dmean) || !nlt_aligned16(dpre_obs) || (obs_y && !nlt_aligned16(obs_y)) ||
      (dobs_partial && !nlt_aligned16(dobs_partial))) return NLT_ERR_BAD_ARG;
  const long total = (long)n * hw * (c >> 2);
  hipLaunchKernelGGL(obs_mean_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dmean, ldm, obs_y, obs_weights, dobs_partial, k, hw, c, alpha, total, dpre_obs);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_level_split_backward(float* dfm, const float* fm_y, int ld, const float* obs_y, const float* obs_weights,
                                        const float* dobs_partial, int n, int k, int hw, int c, float alpha_q, float alpha_o,
                                        float* dpre_obs, void* stream) {
  if (!dfm || !fm_y || !obs_y || !dpre_obs || n <= 0 || k <= 0 || hw <= 0 || c <= 0 || ld < 2 * c) return NLT_ERR_BAD_ARG;
  if ((c & 3) || (ld & 3)) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(dfm) || !nlt_aligned16(fm_y) || !nlt_aligned16(dpre_obs) || !nlt_aligned16(obs_y) ||
      (dobs_partial && !nlt_aligned16(dobs_partial))) return NLT_ERR_BAD_ARG;
  const long total = (long)n * hw * (c >> 2);
  hipLaunchKernelGGL(level_split_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dfm, fm_y, ld, obs_y, obs_weights, dobs_partial, k, hw, c, alpha_q, alpha_o, total, dpre_obs);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_stem_backward(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                                 const float* nn_base, const float* obs_weights, int n, int k, int h, int w, int c,
                                 const float* dfm0, const float* dobs0_partial,
                                 float* dwq, float* dbq, float* dwo, float* dbo, void* stream) {
  if (!base || !cvis || !lvis || !nn_rgb || !nn_base || !dfm0 || !dwq || !dbq || !dwo || !dbo) return NLT_ERR_BAD_ARG;
  if (n <= 0 || k <= 0 || h <= 0 || w <= 0 || c <= 0) return NLT_ERR_BAD_ARG;
  if ((c & 3) || c > 64) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(dfm0) || (dobs0_partial && !nlt_aligned16(dobs0_partial))) return NLT_ERR_BAD_ARG;
  const long texels = (long)n * h * w;
  const int tpb = 256 / (c >> 2);
  long blocks = (texels + tpb - 1) / tpb;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(stem_bwd_kernel, dim3((unsigned)blocks), dim3(256), (size_t)10 * c * sizeof(float),
                     static_cast<hipStream_t>(stream), base, cvis, lvis, nn_rgb, nn_base, obs_weights, k, h * w, c,
                     texels, dfm0, dobs0_partial, dwq, dbq, dwo, dbo);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_head_backward(const float* dec, int ldd, int cd, const float* skip, int lds, int cs,
                                 const float* w_keras, const float* dpred, int n, int h, int w,
                                 float* d_dec, int ldgd, float* d_skip, int ldgs, float* dw, float* db, void* stream) {
  if (!dec || !w_keras || !dpred || !d_dec || !dw || !db || n <= 0 || h <= 0 || w <= 0 || cd <= 0 || cs < 0) return NLT_ERR_BAD_ARG;
  if (cs > 0 && (!skip || !d_skip)) return NLT_ERR_BAD_ARG;
  if ((cd & 3) || (cs & 3) || (ldd & 3) || (ldgd & 3) || (cs > 0 && ((lds & 3) || (ldgs & 3)))) return NLT_ERR_UNSUPPORTED;
  if (cd + cs > 256) return NLT_ERR_UNSUPPORTED;
  const long texels = (long)n * h * w;
  const int tpb = 256 / ((cd + cs) >> 2);
  long blocks = (texels + tpb - 1) / tpb;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)blocks), dim3(256), (size_t)((cd + cs) * 3 + 3) * sizeof(float),
                     static_cast<hipStream_t>(stream), dec, ldd, cd, skip, lds, cs, w_keras, dpred, h * w, texels,
                     d_dec, ldgd, d_skip, ldgs, dw, db);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_warp_backward(const float* dpred_cam, const float* warp, int n, int uvh, int uvw, int hc, int wc,
                                 float* dpred, void* stream) {
  if (!dpred_cam || !warp || !dpred || n <= 0 || uvh <= 0 || uvw <= 0 || hc <= 0 || wc <= 0) return NLT_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(dpred, 0, (size_t)n * uvh * uvw * 3 * sizeof(float), s) != hipSuccess) return NLT_ERR_LAUNCH;
  const long total = (long)n * hc * wc;
  hipLaunchKernelGGL(warp_bwd_kernel, dim3((unsigned)((total + 4 * WARPB_PX - 1) / (4 * WARPB_PX))), dim3(256), 0, s, dpred_cam, warp, uvh, uvw, hc * wc,
                     total, dpred);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_resize_bilinear_backward(const float* dout, int n, int h, int w, int c, int oh, int ow, float* dx,
                                            void* stream) {
  if (!dout || !dx || n <= 0 || h <= 0 || w <= 0 || c <= 0 || oh <= 0 || ow <= 0) return NLT_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(dx, 0, (size_t)n * h * w * c * sizeof(float), s) != hipSuccess) return NLT_ERR_LAUNCH;
  const long total = (long)n * oh * ow;
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, s, dout, h, w, c, oh, ow, total, dx);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_l2_loss_forward(const float* pred, const float* gt, int n, long per_example, float* loss,
                                   void* stream) {
  if (!pred || !gt || !loss || n <= 0 || per_example <= 0) return NLT_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(loss, 0, (size_t)n * sizeof(float), s) != hipSuccess) return NLT_ERR_LAUNCH;
  long bx = (per_example + 255) / 256;
  if (bx > 48) bx = 48;          // one float atomic per workgroup and frame: device-scope atomics on ONE address serialise behind the L2s
                                 // (512 per frame cost 25 us of a 31 us launch); 48 x 256 threads per frame still stream at HBM speed
  hipLaunchKernelGGL(l2_fwd_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, s, pred, gt, per_example, loss);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_l2_loss_backward(const float* pred, const float* gt, const float* gloss, int n, long per_example,
                                    float* dpred, void* stream) {
  if (!pred || !gt || !gloss || !dpred || n <= 0 || per_example <= 0) return NLT_ERR_BAD_ARG;
  const long total = (long)n * per_example;
  hipLaunchKernelGGL(l2_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), pred, gt,
                     gloss, per_example, total, dpred);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_l2_loss_weighted_forward(const float* pred, const float* gt, const float* weights, int n, long hw, int c,
                                            float* loss, void* stream) {
  if (!pred || !gt || !weights || !loss || n <= 0 || hw <= 0 || c <= 0) return NLT_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(loss, 0, (size_t)n * sizeof(float), s) != hipSuccess) return NLT_ERR_LAUNCH;
  long bx = (hw + 255) / 256;
  if (bx > 48) bx = 48;
  hipLaunchKernelGGL(l2w_fwd_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, s, pred, gt, weights, hw, c, loss);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_l2_loss_weighted_backward(const float* pred, const float* gt, const float* weights, const float* gloss, int n,
                                             long hw, int c, float* dpred, void* stream) {
  if (!pred || !gt || !weights || !gloss || !dpred || n <= 0 || hw <= 0 || c <= 0) return NLT_ERR_BAD_ARG;
  const long total = (long)n * hw * c;
  hipLaunchKernelGGL(l2w_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), pred, gt, weights,
                     gloss, hw, c, total, dpred);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_l2_train_loss(const float* pred, const float* rgb, const float* fg, int n, long per_example, float inv_global_bs,
                                 float* gt, float* dpred, float* loss, void* stream) {
  if (!pred || !rgb || !fg || !gt || !dpred || !loss || n <= 0 || per_example <= 0) return NLT_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(loss, 0, sizeof(float), s) != hipSuccess) return NLT_ERR_LAUNCH;
  long bx = (per_example + 255) / 256;
  if (bx > 48) bx = 48;
  hipLaunchKernelGGL(l2_step_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, s, pred, rgb, fg, per_example, inv_global_bs,
                     gt, dpred, loss);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_scale_rows(const float* x, const float* scale, int n, long per_row, float* out, void* stream) {
  if (!x || !scale || !out || n <= 0 || per_row <= 0) return NLT_ERR_BAD_ARG;
  const long total = (long)n * per_row;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     scale, per_row, total, out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_adam_amsgrad_step(float* param, const float* grad, float* m, float* v, float* vhat, long count,
                                     float lr_t, float beta1, float beta2, float eps, void* stream) {
  if (!param || !grad || !m || !v || !vhat || count <= 0) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(adam_amsgrad_kernel, dim3(blocks_for(count)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     param, grad, m, v, vhat, count, lr_t, beta1, beta2, eps);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}
