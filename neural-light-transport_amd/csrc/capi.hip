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
