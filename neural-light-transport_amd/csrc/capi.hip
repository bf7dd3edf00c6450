// extern "C" dispatch for the conv family + library metadata.
#include "nlt_common.h"

extern "C" const char* nlt_version(void) { return "nlt_hip 0.1 (gfx950)"; }

extern "C" const char* nlt_status_string(int status) {
  switch (status) {
    case NLT_OK: return "ok";
    case NLT_ERR_BAD_ARG: return "bad argument (null/size/alignment)";
    case NLT_ERR_UNSUPPORTED: return "unsupported shape or algorithm";
    case NLT_ERR_LAUNCH: return "HIP launch failed";
  }
  return "unknown status";
}

extern "C" int nlt_conv_forward(int mode, int algo, int tile_hint,
                                const float* src0, int ld0, int c0,
                                const float* src1, int ld1, int c1,
                                int n, int h, int w,
                                const float* w_keras, const float* w_packed, const float* bias,
                                int cout, float* out, int ldo,
                                int act, float alpha,
                                const float* mask_src, int ldm, int accumulate,
                                void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  ConvP p;
  if (algo == NLT_ALGO_AUTO) {
    algo = NLT_ALGO_DIRECT;
    if (w_packed) {
      const int st = nlt_fill_conv_params(p, mode, src0, ld0, c0, src1, ld1, c1, n, h, w, w_packed, bias, cout,
                                          out, ldo, act, alpha, mask_src, ldm, accumulate);
      if (st != NLT_OK) return st;
      if (nlt_conv_mfma_supported(mode, p)) algo = NLT_ALGO_MFMA;
    }
  }
  const float* wgt = (algo == NLT_ALGO_MFMA) ? w_packed : w_keras;
  if (algo != NLT_ALGO_MFMA && algo != NLT_ALGO_DIRECT) return NLT_ERR_BAD_ARG;
  const int st = nlt_fill_conv_params(p, mode, src0, ld0, c0, src1, ld1, c1, n, h, w, wgt, bias, cout, out, ldo,
                                      act, alpha, mask_src, ldm, accumulate);
  if (st != NLT_OK) return st;
  if (algo == NLT_ALGO_MFMA) return nlt_conv_mfma_launch(mode, p, tile_hint, s);
  return nlt_conv_direct_launch(mode, p, s);
}

extern "C" long nlt_conv_splitk_workspace_floats(int mode, int n, int h, int w, int cout, int ksplit) {
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0 || ksplit == 0) return -1;
  long rows = (long)n * h * w;
  if (mode == NLT_CONV_K2S2) rows /= 4;
  const long ncols = mode == NLT_DECONV_K2S2 ? 4l * cout : cout;
  if (ksplit < 0) return NLT_SPLITK_COUNTERS + (ksplit < -1 ? -ksplit * rows * ((ncols + 15) / 16 * 16) : 0);   // two launches: a slab per slice
  const long groups = (ksplit + 3) / 4;                            // a workgroup adds 4 (or 16) slices in LDS; only groups meet in memory
  return NLT_SPLITK_COUNTERS + (groups > 1 ? groups * rows * ((ncols + 15) / 16 * 16) : 0);
}

extern "C" int nlt_conv_forward_splitk(int mode, int tile_hint, int ksplit, float* workspace,
                                       const float* src0, int ld0, int c0, const float* src1, int ld1, int c1,
                                       int n, int h, int w, const float* w_packed, const float* bias,
                                       int cout, float* out, int ldo, int act, float alpha,
                                       const float* mask_src, int ldm, int accumulate, void* stream) {
  if (ksplit == 0 || ((ksplit > 1 || ksplit < -1) && (!workspace || !nlt_aligned16(workspace)))) return NLT_ERR_BAD_ARG;
  ConvP p;
  const int st = nlt_fill_conv_params(p, mode, src0, ld0, c0, src1, ld1, c1, n, h, w, w_packed, bias, cout, out, ldo,
                                      act, alpha, mask_src, ldm, accumulate);
  if (st != NLT_OK) return st;
  return nlt_conv_mfma_launch(mode, p, tile_hint, static_cast<hipStream_t>(stream), ksplit, workspace);
}

extern "C" int nlt_conv_forward_map(int mode, int tile_hint, int ksplit, float* workspace,
                                    const float* src0, int ld0, int c0, const float* src1, int ld1, int c1,
                                    int n, int h, int w, const float* w_packed, const float* bias,
                                    int cout, float* out, int ldo, int act, float alpha,
                                    const float* bias_map, int map_frames, void* stream) {
  if (ksplit == 0 || ((ksplit > 1 || ksplit < -1) && (!workspace || !nlt_aligned16(workspace)))) return NLT_ERR_BAD_ARG;
  if (!bias_map || !nlt_aligned16(bias_map) || (map_frames != 1 && map_frames != n)) return NLT_ERR_BAD_ARG;
  ConvP p;
  const int st = nlt_fill_conv_params(p, mode, src0, ld0, c0, src1, ld1, c1, n, h, w, w_packed, bias, cout, out, ldo,
                                      act, alpha, nullptr, 0, 0);
  if (st != NLT_OK) return st;
  p.bmap = bias_map;
  p.bmap_mod = map_frames == 1 ? p.oh * p.ow : 0;
  return nlt_conv_mfma_launch(mode, p, tile_hint, static_cast<hipStream_t>(stream), ksplit, workspace);
}

extern "C" int nlt_conv_backward_data(int adj_mode, int tile_hint, int ksplit, float* workspace,
                                      const float* dpre, int ldp, int cpre, int n, int h, int w,
                                      const float* w_packed, const float* zero_bias, int cout, float* out, int ldo,
                                      const float* mask_src, int ldm, float mask_alpha, int accumulate,
                                      int split_c, const float* split_y, float* split_d, float split_alpha, int split_partial,
                                      void* stream) {
  if (ksplit == 0 || ((ksplit > 1 || ksplit < -1) && (!workspace || !nlt_aligned16(workspace)))) return NLT_ERR_BAD_ARG;
  ConvP p;
  const int st = nlt_fill_conv_params(p, adj_mode, dpre, ldp, cpre, nullptr, 0, 0, n, h, w, w_packed, zero_bias, cout, out, ldo,
                                      0, mask_alpha, mask_src, ldm, accumulate);
  if (st != NLT_OK) return st;
  if (split_c) {
    if (split_c < 0 || (split_c & 3) || split_c >= cout || !split_y || !split_d) return NLT_ERR_BAD_ARG;
    if (!nlt_aligned16(split_y) || !nlt_aligned16(split_d)) return NLT_ERR_BAD_ARG;
    p.split_c = split_c; p.split_y = split_y; p.split_d = split_d; p.split_alpha = split_alpha; p.split_partial = split_partial;
  }
  return nlt_conv_mfma_launch(adj_mode, p, tile_hint, static_cast<hipStream_t>(stream), ksplit, workspace);
}
