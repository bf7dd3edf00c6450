// Native replay of a recorded launch tape (nlt_amd/_capi.py: tape_begin / tape_end / replay).
//
// A plan (engine.RenderPlan) issues the same ~40 (forward) / ~150 (train step) C calls with the same arguments every step.  The
// Python-side tape already skips re-deriving the arguments; replaying it from Python still costs a ctypes call (argument
// conversion of ~20 values) per launch, ~7 us each -- 1.3-1.7 ms per train step, MORE than the GPU needs at the released training
// shape (512^2 UV, bs 4: the step was host-bound).  Here the recorded calls are an array of (entry point, integer-class arguments,
// float arguments) and ONE C call walks it.
//
// The generic call relies on the x86-64 System V calling convention (the only host this library is built for): integer-class
// arguments (ints, longs, pointers) are assigned in declaration order to rdi, rsi, rdx, rcx, r8, r9 and then to the stack,
// float arguments in declaration order to xmm0..7, the two sequences independently of each other; surplus integer arguments
// are ignored by the callee.  So an entry point with i <= 32 integer-class and f <= 4 float parameters (no doubles) can be
// called as  int (*)(long x 32, float x f).  Entries that do not fit are replayed from Python.
#include "nlt_common.h"

extern "C" int nlt_event_record(void* event, void* stream) {
  return hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)) == hipSuccess ? NLT_OK : NLT_ERR_LAUNCH;
}

extern "C" int nlt_stream_wait_event(void* stream, void* event) {
  return hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0) == hipSuccess ? NLT_OK : NLT_ERR_LAUNCH;
}

// Events for GPU-to-GPU ordering between two streams of ONE device.  hipEventRecord of an ordinary event also performs a
// system-scope release fence (so that the host, or another device, sees coherent memory once the event reads as recorded); the
// stream that records waits for it -- measured r04: 7-12 us between the recording launch's end and the next launch's start on
// the observation chain, six times per forward pass.  hipEventDisableSystemFence leaves the fences to the kernels' own packets
// (agent scope: what a consumer on the same device needs).
extern "C" int nlt_event_create(int no_system_fence, void** event) {
  if (!event) return NLT_ERR_BAD_ARG;
  hipEvent_t e = nullptr;
  const unsigned flags = hipEventDisableTiming | (no_system_fence ? hipEventDisableSystemFence : 0u);
  if (hipEventCreateWithFlags(&e, flags) != hipSuccess) return NLT_ERR_LAUNCH;
  *event = e;
  return NLT_OK;
}

extern "C" int nlt_event_destroy(void* event) {
  return hipEventDestroy(static_cast<hipEvent_t>(event)) == hipSuccess ? NLT_OK : NLT_ERR_LAUNCH;
}

#define L8 long, long, long, long, long, long, long, long
#define A8(b) a[b], a[b + 1], a[b + 2], a[b + 3], a[b + 4], a[b + 5], a[b + 6], a[b + 7]
#define A32 A8(0), A8(8), A8(16), A8(24)

extern "C" int nlt_tape_play(const nlt_tape_call* calls, int n, int* failed_index) {
  if (!calls || n < 0) return NLT_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i) {
    const nlt_tape_call& c = calls[i];
    const long* a = c.iargs;
    const float* f = c.fargs;
    int rc;
    switch (c.n_float) {
      case 0: rc = reinterpret_cast<int (*)(L8, L8, L8, L8)>(c.fn)(A32); break;
      case 1: rc = reinterpret_cast<int (*)(L8, L8, L8, L8, float)>(c.fn)(A32, f[0]); break;
      case 2: rc = reinterpret_cast<int (*)(L8, L8, L8, L8, float, float)>(c.fn)(A32, f[0], f[1]); break;
      case 3: rc = reinterpret_cast<int (*)(L8, L8, L8, L8, float, float, float)>(c.fn)(A32, f[0], f[1], f[2]); break;
      case 4: rc = reinterpret_cast<int (*)(L8, L8, L8, L8, float, float, float, float)>(c.fn)(A32, f[0], f[1], f[2], f[3]); break;
      default: rc = NLT_ERR_BAD_ARG;
    }
    if (rc != NLT_OK) {
      if (failed_index) *failed_index = i;
      return rc;
    }
  }
  return NLT_OK;
}
