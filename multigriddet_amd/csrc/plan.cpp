// Launch plans: a step's sequence of C-ABI calls recorded once and replayed from ONE C entry point.
//
// The reference drives a training step through Keras (`model.fit`, multigriddet/trainers/trainer.py:572-581): one Python call
// per batch, the graph executor does the rest.  Here a step is ~600 kernel launches on two streams; enqueueing them from Python
// costs 6 ms of interpreter time per 12.5-ms step under the GIL - time the loader's threads (data/generators.py) cannot use.
// Every pointer of a step is fixed once the arenas exist (engine.Network.arena), so the sequence is static: Python records it
// (multigriddet_amd/_lib.py, Recorder) - entry point name, arguments, stream slot, and the cross-stream waits - and
// mgd_plan_run() replays it without the interpreter (ctypes releases the GIL for the call).  This is not a hipGraph: the two
// streams stay two hardware queues (ROCm 7.2 serialises captured streams), the plan only moves the host side to C.
//
// An argument is one 64-bit word + a kind: integer / pointer value, float, double, offset into the plan's blob (a descriptor
// the call takes by pointer, copied at record time), stream slot, or external parameter slot (a pointer that changes from run
// to run: the batch's images and boxes).  Trampolines are generated from the entry points' own prototypes, so a call is
// replayed with exactly the C types it was declared with.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/mgd_hip.h"

int mgd_set_error(int code, const char* fmt, ...);

namespace {

enum : uint8_t { K_INT = 0, K_F32 = 1, K_F64 = 2, K_BLOB = 3, K_STREAM = 4, K_PARAM = 5 };

struct Slot {
  int64_t w;
  uint8_t kind;
};

struct Ctx {
  const unsigned char* blob;
  void* const* streams;
  void* const* params;
};

template <typename T>
inline T arg(const Slot& s, const Ctx& c) {
  if constexpr (std::is_pointer_v<T>) {
    if (s.kind == K_BLOB) return (T)(c.blob + s.w);
    if (s.kind == K_STREAM) return (T)c.streams[s.w];
    if (s.kind == K_PARAM) return (T)c.params[s.w];
    return (T)(uintptr_t)s.w;
  } else if constexpr (std::is_floating_point_v<T>) {
    if (s.kind == K_F32) { float f; memcpy(&f, &s.w, 4); return (T)f; }
    if (s.kind == K_F64) { double d; memcpy(&d, &s.w, 8); return (T)d; }
    return (T)s.w;
  } else {
    return (T)s.w;
  }
}

typedef int (*Tramp)(const Slot*, const Ctx&);

template <auto F>
struct tramp;
template <typename... A, int (*F)(A...)>
struct tramp<F> {
  static constexpr int nargs = (int)sizeof...(A);
  template <size_t... I>
  static int go(const Slot* s, const Ctx& c, std::index_sequence<I...>) { return F(arg<A>(s[I], c)...); }
  static int call(const Slot* s, const Ctx& c) { return go(s, c, std::index_sequence_for<A...>{}); }
};

struct Entry {
  const char* name;
  Tramp call;
  int nargs;
};
#define E(fn) {#fn, &tramp<&fn>::call, tramp<&fn>::nargs}
// every int-returning, stream-taking entry point of include/mgd_hip.h that a training / inference step may issue
const Entry kTable[] = {
    E(mgd_conv_gather_gemm), E(mgd_conv_gather_gemm_classes), E(mgd_conv_dgrad_s2_patch), E(mgd_conv_wgrad), E(mgd_stem_fwd), E(mgd_stem_fwd_act), E(mgd_stem_wgrad),
    E(mgd_stem_wgrad_bn), E(mgd_pack_weights), E(mgd_pack_weights_batch), E(mgd_stem_im2col), E(mgd_bn_finalize), E(mgd_bn_act_fwd),
    E(mgd_bn_act_fwd_fused), E(mgd_bn_act_bwd_reduce), E(mgd_bn_act_bwd_apply), E(mgd_upsample_concat_fwd), E(mgd_upsample_concat_bwd),
    E(mgd_bias_grad), E(mgd_f32_to_bf16), E(mgd_bf16_to_f32), E(mgd_adam_step), E(mgd_adam_step_dev), E(mgd_sgd_step),
    E(mgd_build_targets), E(mgd_loss_fwd_bwd), E(mgd_decode), E(mgd_nms), E(mgd_wbf), E(mgd_mosaic), E(mgd_gridmask), E(mgd_mixup),
    E(mgd_letterbox_u8), E(mgd_memset_async),
};
#undef E

struct Op {
  int fn;            // index into kTable; -1: cross-stream wait (a = waiting slot, b = signalling slot, ev = event index)
  int first, n;      // slots
  int a, b, ev;
};

}  // namespace

struct mgd_plan {
  std::vector<Op> ops;
  std::vector<Slot> slots;
  std::vector<unsigned char> blob;
  std::vector<hipEvent_t> events;
  int max_stream = -1, max_param = -1;
};

extern "C" int mgd_plan_create(mgd_plan** out) {
  if (!out) return mgd_set_error(MGD_EINVAL, "plan_create: null pointer");
  *out = new mgd_plan();
  return MGD_OK;
}

extern "C" int mgd_plan_destroy(mgd_plan* p) {
  if (!p) return MGD_OK;
  for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
  delete p;
  return MGD_OK;
}

extern "C" int mgd_plan_size(const mgd_plan* p) { return p ? (int)p->ops.size() : 0; }

extern "C" int mgd_plan_add_call(mgd_plan* p, const char* name, const int64_t* words, const uint8_t* kinds, int nargs, const void* blob,
                                 int64_t blob_bytes) {
  if (!p || !name || nargs < 0 || (nargs && (!words || !kinds)) || blob_bytes < 0 || (blob_bytes && !blob))
    return mgd_set_error(MGD_EINVAL, "plan_add_call: arguments");
  int fn = -1;
  for (size_t i = 0; i < sizeof(kTable) / sizeof(kTable[0]); ++i)
    if (!strcmp(kTable[i].name, name)) { fn = (int)i; break; }
  if (fn < 0) return mgd_set_error(MGD_EINVAL, "plan_add_call: %s cannot be replayed from a plan", name);
  if (kTable[fn].nargs != nargs) return mgd_set_error(MGD_EINVAL, "plan_add_call: %s takes %d arguments, got %d", name, kTable[fn].nargs, nargs);
  const int64_t base = (int64_t)((p->blob.size() + 15) / 16 * 16);
  p->blob.resize((size_t)(base + blob_bytes));
  if (blob_bytes) memcpy(p->blob.data() + base, blob, (size_t)blob_bytes);
  Op op{fn, (int)p->slots.size(), nargs, 0, 0, 0};
  for (int i = 0; i < nargs; ++i) {
    Slot s{words[i], kinds[i]};
    if (s.kind > K_PARAM) return mgd_set_error(MGD_EINVAL, "plan_add_call: %s: argument kind %d", name, (int)s.kind);
    if (s.kind == K_BLOB) {
      if (s.w < 0 || s.w >= blob_bytes) return mgd_set_error(MGD_EINVAL, "plan_add_call: %s: blob offset out of range", name);
      s.w += base;
    }
    if (s.kind == K_STREAM) { if (s.w < 0 || s.w > 15) return mgd_set_error(MGD_EINVAL, "plan_add_call: stream slot"); if (s.w > p->max_stream) p->max_stream = (int)s.w; }
    if (s.kind == K_PARAM) { if (s.w < 0 || s.w > 63) return mgd_set_error(MGD_EINVAL, "plan_add_call: parameter slot"); if (s.w > p->max_param) p->max_param = (int)s.w; }
    p->slots.push_back(s);
  }
  p->ops.push_back(op);
  return MGD_OK;
}

// streams[waiting] waits for everything enqueued so far on streams[signalling] (an event record + a stream wait at replay)
extern "C" int mgd_plan_add_wait(mgd_plan* p, int waiting, int signalling) {
  if (!p || waiting < 0 || waiting > 15 || signalling < 0 || signalling > 15 || waiting == signalling)
    return mgd_set_error(MGD_EINVAL, "plan_add_wait: stream slots");
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return mgd_set_error(MGD_ELAUNCH, "plan_add_wait: event");
  p->events.push_back(e);
  p->ops.push_back(Op{-1, 0, 0, waiting, signalling, (int)p->events.size() - 1});
  if (waiting > p->max_stream) p->max_stream = waiting;
  if (signalling > p->max_stream) p->max_stream = signalling;
  return MGD_OK;
}

extern "C" int mgd_plan_run(const mgd_plan* p, void* const* streams, int nstreams, void* const* params, int nparams) {
  if (!p || (nstreams && !streams) || (nparams && !params)) return mgd_set_error(MGD_EINVAL, "plan_run: null pointer");
  if (p->max_stream >= nstreams) return mgd_set_error(MGD_EINVAL, "plan_run: the plan uses stream slot %d, %d given", p->max_stream, nstreams);
  if (p->max_param >= nparams) return mgd_set_error(MGD_EINVAL, "plan_run: the plan uses parameter slot %d, %d given", p->max_param, nparams);
  const Ctx c{p->blob.data(), streams, params};
  const Slot* slots = p->slots.data();
  for (const Op& op : p->ops) {
    if (op.fn < 0) {
      if (hipEventRecord(p->events[op.ev], (hipStream_t)streams[op.b]) != hipSuccess ||
          hipStreamWaitEvent((hipStream_t)streams[op.a], p->events[op.ev], 0) != hipSuccess)
        return mgd_set_error(MGD_ELAUNCH, "plan_run: cross-stream wait failed");
      continue;
    }
    const int rc = kTable[op.fn].call(slots + op.first, c);
    if (rc != MGD_OK) return rc;                               // (mgd_last_error holds the callee's message)
  }
  return MGD_OK;
}

extern "C" int mgd_memset_async(void* p, int value, int64_t bytes, void* stream) {
  if (!p || bytes < 0) return mgd_set_error(MGD_EINVAL, "memset_async: arguments");
  if (bytes && hipMemsetAsync(p, value, (size_t)bytes, (hipStream_t)stream) != hipSuccess) return mgd_set_error(MGD_ELAUNCH, "memset_async failed");
  return MGD_OK;
}
