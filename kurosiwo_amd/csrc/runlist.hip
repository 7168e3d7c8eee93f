// Launch-list executor: one C-ABI call issues a whole segment of a plan's prepared launches on the streams of the train step.
//
// A plan (kurosiwo_amd/*_plan.py) is a flat list of prepared entry-point calls plus ordering entries; until round 5 the host walked it in
// Python -- one ctypes call, one torch stream switch and up to three torch event calls per launch: 9 ms of host time for the ~290
// launches of an SNUNet step, 29 ms for ChangeFormer's ~730 (bench `host_issue_ms_per_step`), i.e. the step was about to become
// host-bound.  Here the list is compiled ONCE into an array of ksmi_op (function pointer, thunk of its
// signature, 64-bit argument slots, lane / side-stream tags) and ksmi_run_list replays a segment of it: the same launches on the same
// streams with the same event record / wait pairs as snunet_plan.LaunchList.run (the Python walk stays as the timed / hooked path and
// as the cross-check: tests/test_gpu_graph.py holds the two to the same bits).
//
// Streams (snunet_plan.StepStreams): main = lane 0, lane1 = second compute lane, side / side2 = weight-gradient streams.
//   CALL, plain:     launch on the stream of its lane.
//   CALL, side:      the side stream waits for an event recorded on the launch's lane stream ("fork"), the launch goes to the side
//                    stream, and a tagged launch records an event there that a later WAIT_SIDE(tag) consumes.
//   ORDER(a, b):     lane b waits for everything lane a was handed so far.
//   WAIT_SIDE(tag):  the issuing lane waits for that tagged side-stream launch (tag < 0: for the whole side stream, if it is busy).
//   JOIN:            main waits for every other stream that was handed work (end of the step).
// Events come from a fixed ring created once (hipEventDisableTiming): hipStreamWaitEvent captures the record it finds when it is
// enqueued, so a ring slot may be re-recorded as soon as the wait call has returned.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unordered_map>
#include <vector>
#include <hip/hip_runtime.h>
#include "../../include/ksmi.h"
#include "errors.h"
#include "thunks_gen.h"

namespace {

constexpr int kRing = 64;

struct Runner {
  hipStream_t main = nullptr, lane1 = nullptr, side = nullptr, side2 = nullptr;
  bool lanes = false, use_side = false;
  bool dirty = false, side_busy = false;
  int cur = 0;                                   // lane of the last launch of the list being walked
  hipEvent_t ring[kRing];
  int next = 0;
  bool have_ring = false;
  std::unordered_map<int, hipEvent_t> tagged;    // side tag -> event recorded behind that launch
  std::vector<hipEvent_t> tag_pool;              // events of consumed tags, for reuse

  hipStream_t stream(int lane) const { return (lanes && lane == 1 && lane1) ? lane1 : main; }
  int ensure_ring() {
    if (have_ring) return 0;
    for (int i = 0; i < kRing; ++i)
      if (hipEventCreateWithFlags(&ring[i], hipEventDisableTiming) != hipSuccess) return ksmi_fail(KSMI_E_UNSUPPORTED, "run_list: hipEventCreate");
    have_ring = true;
    return 0;
  }
  // dst waits for everything src was handed so far
  int order(hipStream_t src, hipStream_t dst) {
    if (src == dst) return 0;
    if (int rc = ensure_ring()) return rc;              // (created at the first cross-stream edge: single-stream lists never need events)
    hipEvent_t ev = ring[next];
    next = (next + 1) % kRing;
    if (hipEventRecord(ev, src) != hipSuccess || hipStreamWaitEvent(dst, ev, 0) != hipSuccess) return ksmi_fail(KSMI_E_UNSUPPORTED, "run_list: event record / wait");
    return 0;
  }
};

}  // namespace

extern "C" {

int ksmi_thunk_id(const char* sig) {
  if (!sig) return -1;
  for (int i = 0; i < ksmi_n_thunks; ++i)
    if (!strcmp(sig, ksmi_thunk_sigs[i])) return i;
  return -1;
}

void* ksmi_runner_create(void) { return new Runner(); }

int ksmi_runner_destroy(void* r_) {
  Runner* r = (Runner*)r_;
  if (!r) return 0;
  if (r->have_ring) for (int i = 0; i < kRing; ++i) (void)hipEventDestroy(r->ring[i]);
  for (auto& kv : r->tagged) (void)hipEventDestroy(kv.second);
  for (hipEvent_t e : r->tag_pool) (void)hipEventDestroy(e);
  delete r;
  return 0;
}

int ksmi_runner_set_streams(void* r_, void* main, void* lane1, void* side, void* side2) {
  Runner* r = (Runner*)r_;
  if (!r) return ksmi_fail(KSMI_E_ARG, "runner_set_streams: null runner");
  r->main = (hipStream_t)main; r->lane1 = (hipStream_t)lane1; r->side = (hipStream_t)side; r->side2 = (hipStream_t)side2;
  r->lanes = lane1 != nullptr;
  r->use_side = side != nullptr;
  return 0;
}

int ksmi_run_list(void* r_, const ksmi_op* ops, int first, int last, const uint8_t* skip, int32_t* failed_at) {
  Runner* r = (Runner*)r_;
  if (!r || !ops || first < 0 || last < first) return ksmi_fail(KSMI_E_ARG, "run_list: bad arguments");
  if (failed_at) *failed_at = -1;
  int rc = 0;
  if (first == 0) r->cur = 0;
  for (int i = first; i < last; ++i) {
    const ksmi_op& op = ops[i];
    switch (op.kind) {
      case KSMI_OP_CALL: {
        if (skip && skip[i]) break;
        if (op.sig < 0 || op.sig >= ksmi_n_thunks || !op.fn) { if (failed_at) *failed_at = i; return ksmi_fail(KSMI_E_ARG, "run_list: bad thunk / function"); }
        const int lane = r->lanes ? op.lane : 0;
        hipStream_t S = r->stream(lane);
        r->cur = lane;
        if (op.side && r->use_side) {
          hipStream_t SS = (op.side == 2 && r->side2) ? r->side2 : r->side;
          if ((rc = r->order(S, SS))) { if (failed_at) *failed_at = i; return rc; }
          r->dirty = r->side_busy = true;
          rc = ksmi_thunks[op.sig](op.fn, op.args, SS);
          if (rc == 0 && op.tag >= 0) {
            hipEvent_t ev;
            auto it = r->tagged.find(op.tag);
            if (it != r->tagged.end()) ev = it->second;
            else if (!r->tag_pool.empty()) { ev = r->tag_pool.back(); r->tag_pool.pop_back(); r->tagged[op.tag] = ev; }
            else { if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { if (failed_at) *failed_at = i; return ksmi_fail(KSMI_E_UNSUPPORTED, "run_list: hipEventCreate"); } r->tagged[op.tag] = ev; }
            if (hipEventRecord(ev, SS) != hipSuccess) { if (failed_at) *failed_at = i; return ksmi_fail(KSMI_E_UNSUPPORTED, "run_list: event record"); }
          }
        } else {
          rc = ksmi_thunks[op.sig](op.fn, op.args, S);
        }
        if (rc) { if (failed_at) *failed_at = i; return rc; }
        break;
      }
      case KSMI_OP_ORDER:
        if (r->lanes) {
          if ((rc = r->order(r->stream(op.a), r->stream(op.b)))) { if (failed_at) *failed_at = i; return rc; }
          r->dirty = true;
        }
        break;
      case KSMI_OP_WAIT_SIDE:
        if (r->use_side) {
          // the waiting stream: the lane the list was on when the entry was appended (op.lane) AND the lane of the last launch
          hipStream_t w0 = r->stream(r->lanes ? op.lane : 0), w1 = r->stream(r->cur);
          if (op.tag < 0) {
            if (r->side_busy) {
              for (hipStream_t SS : {r->side, r->side2}) {
                if (!SS) continue;
                if ((rc = r->order(SS, w0)) || (w1 != w0 && (rc = r->order(SS, w1)))) { if (failed_at) *failed_at = i; return rc; }
              }
              r->side_busy = false;
            }
          } else {
            auto it = r->tagged.find(op.tag);
            if (it != r->tagged.end()) {
              if (hipStreamWaitEvent(w0, it->second, 0) != hipSuccess || (w1 != w0 && hipStreamWaitEvent(w1, it->second, 0) != hipSuccess)) {
                if (failed_at) *failed_at = i;
                return ksmi_fail(KSMI_E_UNSUPPORTED, "run_list: wait on a tagged side event");
              }
              r->tag_pool.push_back(it->second);
              r->tagged.erase(it);
            }
          }
        }
        break;
      default:
        if (failed_at) *failed_at = i;
        return ksmi_fail(KSMI_E_ARG, "run_list: unknown op kind");
    }
  }
  return 0;
}

int ksmi_runner_join(void* r_) {
  Runner* r = (Runner*)r_;
  if (!r) return ksmi_fail(KSMI_E_ARG, "runner_join: null runner");
  int rc = 0;
  if (r->dirty) {
    for (hipStream_t s : {r->lane1, r->side, r->side2})
      if (s && s != r->main && (rc = r->order(s, r->main))) return rc;
  }
  r->dirty = r->side_busy = false;
  for (auto& kv : r->tagged) r->tag_pool.push_back(kv.second);
  r->tagged.clear();
  r->cur = 0;
  return 0;
}

}  // extern "C"
