"""Static launch plan of one SNUNet-ECAM forward/backward at a fixed (B, H, W, dtype, mode).

The plan owns every NHWC activation / gradient buffer and a flat list of prepared C-ABI
calls; running it is a loop of ctypes calls on the current HIP stream (no allocation, no
host sync), so a whole train step can be captured into one HIP graph.

Reference computation: /root/reference/models/snunet.py:118-153 (forward graph),
:11-29 (conv_block_nested), :32-46 (up), :49-62 + :146-151 (ECAM head).
"""
import ctypes as C
import os

import torch

from . import _lib
from .runtime import DT, Act, SrcSpec, conv_grid_m, conv_npad, conv_stats_rows, make_conv, make_pack, make_wgrad, packed_weight_numel, stream_ptr
from .snunet import BN_EPS, BN_MOMENTUM


_TAG_IDS = {}


def _tag_id(tag):
    """side-stream tags of the plans (arbitrary hashables) as the small integers ksmi_op carries"""
    return _TAG_IDS.setdefault(tag, len(_TAG_IDS))


def _sig_codes(argtypes):
    """signature string of tools/gen_thunks.py: one letter per argument (p pointer, i int, u unsigned, l int64, z size_t, f float, d double)"""
    out = []
    for t in argtypes:
        if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and (issubclass(t, C._Pointer) or issubclass(t, C.Array))):
            out.append("p")
        else:
            out.append({C.c_int: "i", C.c_int32: "i", C.c_uint32: "u", C.c_int64: "l", C.c_size_t: "z", C.c_float: "f", C.c_double: "d"}[t])
    return "".join(out)


def _slot(code, v, struct):
    """one prepared argument as the 64-bit slot the call thunks read (include/ksmi.h ksmi_op)"""
    if hasattr(v, "value") and not hasattr(v, "_obj"):        # a ctypes scalar (c_void_p, c_int, ...)
        v = v.value
    if code == "p":
        if v is None:
            return 0
        if isinstance(v, int):
            return v
        if hasattr(v, "_obj"):                                # ctypes.byref(x)
            return C.addressof(v._obj)
        if isinstance(v, (C.Array, C.Structure)):
            return C.addressof(v)
        if isinstance(v, C._Pointer):
            return C.cast(v, C.c_void_p).value or 0
        raise _lib.KsmiError(f"launch list: cannot take the address of a {type(v).__name__} argument")
    if code == "f":
        return struct.unpack("<I", struct.pack("<f", float(v)))[0]
    if code == "d":
        return struct.unpack("<Q", struct.pack("<d", float(v)))[0]
    return int(v) & 0xFFFFFFFFFFFFFFFF


_PLAIN_RUNNERS = {}


def _plain_runner(st):
    """the executor state of single-stream runs (model(x) outside a train step): everything on the caller's current stream"""
    dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
    lib = _lib.load()
    r = _PLAIN_RUNNERS.get(dev)
    if r is None:
        r = _PLAIN_RUNNERS[dev] = C.c_void_p(lib.ksmi_runner_create())
    lib.ksmi_runner_set_streams(r, st, None, None, None)
    return r


class LaunchList:
    """(name, argfn, meta) triples; argfn() is evaluated once, after all scratch buffers exist.
    meta = {"kind": kernel class, "bytes": algorithmic HBM bytes, "flops": 2*MAC} for the roofline, plus the scheduling tags
    "lane" (compute lane the launch belongs to: 0 = the caller's stream, 1 = the second lane) and "side" (weight gradient: may
    run on the side stream).  ("@wait", (a, b)) entries order lane b behind everything lane a was handed so far."""

    def __init__(self):
        self.pending, self.calls = [], []
        self.cur_lane = 0
        self.cur_stage = None      # measurement tag of the launches appended from here on (bench.py roofline.stages)

    def add(self, name, argfn, meta=None):
        meta = dict(meta) if meta else {"kind": name[5:], "bytes": 0, "flops": 0}
        meta["lane"] = self.cur_lane
        meta.setdefault("stage", self.cur_stage)
        self.pending.append((name, argfn, meta))

    def add_wait(self, src, dst):
        self.pending.append(("@wait", lambda: (src, dst), {"kind": "wait", "bytes": 0, "flops": 0, "lane": dst}))

    def add_allreduce(self, tensor_fn, meta=None):
        """SyncBN (optional): SUM `tensor_fn()` (a small fp32 statistics tensor) over the ranks, in place, on the stream of
        the issuing lane, between the launch that wrote it and the launch that reads it"""
        m = {"kind": "syncbn_allreduce", "bytes": 0, "flops": 0, "lane": self.cur_lane}
        m.setdefault("stage", self.cur_stage)
        self.pending.append(("@allreduce", lambda: (tensor_fn(),), m | (meta or {})))

    def add_wait_side(self, tag=None):
        """the issuing lane waits for the side-stream launch that carries meta["side_tag"] == tag (None: for everything handed to the
        side stream so far): placed before a launch that overwrites an operand of that weight gradient (plans that recycle buffers)"""
        self.pending.append(("@wait_side", lambda: (tag,), {"kind": "wait", "bytes": 0, "flops": 0, "lane": self.cur_lane}))

    def resolve(self, lib):
        self.calls = [(None if name.startswith("@") else getattr(lib, name), tuple(argfn()), name, meta) for name, argfn, meta in self.pending]
        self._compiled = None          # (the compiled form holds the argument values of the previous resolution)

    # ---- compiled form (round 6): the list as an array of ksmi_op walked by ONE C-ABI call per segment (csrc/runlist.hip) instead of one
    # ctypes call + stream switch + up to three torch event calls per launch in Python (host_issue_ms_per_step: 9 ms of a 14 ms SNUNet
    # step, 29 of 34 ms for ChangeFormer).  The Python walk below stays for timed runs (a kernel timer brackets single launches), for hooks
    # without an index list, for SyncBN's collectives, and as the cross-check (KSMI_RUN_LIST=0; tests/test_gpu_graph.py).
    fast = os.environ.get("KSMI_RUN_LIST", "1") != "0"
    _compiled = None

    def _compile(self):
        import struct
        lib = _lib.load()
        n = len(self.calls)
        ops = (_lib.Op * max(n, 1))()
        slots, where, skips, names, ok = [], [], [], [], True
        for i, (fn, args, name, meta) in enumerate(self.calls):
            op = ops[i]
            op.tag, op.sig = -1, -1
            op.lane = int(meta.get("lane", 0))
            names.append(name)
            if fn is None:
                if name == "@wait":
                    op.kind, op.a, op.b = _lib.OP_ORDER, int(args[0]), int(args[1])
                elif name == "@wait_side":
                    op.kind, op.tag = _lib.OP_WAIT_SIDE, (-1 if args[0] is None else _tag_id(args[0]))
                else:
                    ok = False                   # ("@allreduce": SyncBN's collectives are issued by torch.distributed)
                continue
            if name not in _lib.SIGNATURES:       # (a stubbed library in the host-only tests: the Python walk)
                ok = False
                continue
            restype, argtypes = _lib.SIGNATURES[name]
            codes = _sig_codes(argtypes[:-1])
            sig = lib.ksmi_thunk_id(codes.encode())
            if sig < 0 or len(codes) != len(args):
                raise _lib.KsmiError(f"launch list: no call thunk for {name} ({codes!r}, {len(args)} arguments): re-run tools/gen_thunks.py")
            op.kind, op.sig, op.nargs = _lib.OP_CALL, sig, len(args)
            op.fn = C.cast(fn, C.c_void_p).value
            op.side = (2 if meta.get("side_ix", 0) else 1) if meta.get("side") else 0
            if op.side and meta.get("side_tag") is not None:
                op.tag = _tag_id(meta["side_tag"])
            where.append((i, len(slots)))
            slots += [_slot(c, v, struct) for c, v in zip(codes, args)]
            if meta.get("skip_if") is not None:
                skips.append((i, meta["skip_if"]))
        arr = (C.c_uint64 * max(len(slots), 1))(*slots)
        base = C.addressof(arr)
        for i, off in where:
            ops[i].args = base + 8 * off
        self._compiled = {"ops": ops, "slots": arr, "n": n, "skips": skips, "names": names, "ok": ok, "failed": C.c_int32(-1),
                          "skipbuf": (C.c_uint8 * max(n, 1))() if skips else None}
        return self._compiled

    def _run_fast(self, hook, hook_at, streams):
        cp = self._compiled
        lib = _lib.load()
        if streams is not None:
            streams.begin()
            runner = streams.runner()
        else:
            runner = _plain_runner(stream_ptr())
        skip = None
        if cp["skips"]:
            skip = cp["skipbuf"]
            for i, fn in cp["skips"]:
                skip[i] = 1 if fn() else 0
        n = cp["n"]
        cuts = sorted({i + 1 for i in hook_at if 0 <= i < n} | {n}) if hook is not None else [n]
        a = 0
        for b in cuts:
            rc = lib.ksmi_run_list(runner, cp["ops"], a, b, skip, C.byref(cp["failed"]))
            if rc != 0:
                at = cp["failed"].value
                _lib.check(rc, cp["names"][at] if 0 <= at < n else "ksmi_run_list")
            if hook is not None and (b - 1) in hook_at:
                hook(b - 1)
            a = b

    def run(self, timer=None, hook=None, streams=None, hook_at=None):
        """hook_at: the list indices at which `hook` has work to do (dp.BucketedAllReduce.hook_indices); with it (or without a hook) and
        without a timer the compiled list runs (see above).
        streams = StepStreams or None.  None: every launch on the current stream, in list order (always a valid order; the
        "@wait" entries are no-ops).  With streams: launches of lane 1 go to the second compute stream, launches tagged "side" (the
        weight gradients: nothing on the critical path of the backward pass reads them) to the side stream behind an event recorded
        on the issuing lane's stream at that point of the list, so that independent work fills the machine next to the
        bandwidth-bound BatchNorm / elementwise launches of the critical path; the caller joins (StepStreams.join) before
        anything outside the lists reads the results."""
        if (timer is None or not getattr(timer, "active", True)) and self.fast and self.calls and (hook is None or hook_at is not None):
            cp = self._compiled or self._compile()
            if cp["ok"]:
                return self._run_fast(hook, hook_at or (), streams)
        if streams is not None:
            streams.begin()
        st, cur = stream_ptr(), 0
        try:
            for idx, (fn, args, name, meta) in enumerate(self.calls):
                if fn is None and name == "@allreduce":
                    lane = meta["lane"] if streams is not None and streams.lanes else 0
                    if lane != cur:
                        torch.cuda.set_stream(streams.stream(lane))
                        st, cur = stream_ptr(), lane
                    from . import distributed as D
                    D.all_reduce_sum_(args[0])                       # (stream-ordered on RCCL; gloo stages through the host)
                    if hook is not None:
                        hook(idx)
                    continue
                if fn is None:
                    if name == "@wait_side":
                        if streams is not None and streams.use_side:
                            streams.wait_side(args[0])
                    elif streams is not None and streams.lanes:
                        streams.order(*args)
                    if hook is not None:
                        hook(idx)
                    continue
                if meta.get("skip_if") is not None and meta["skip_if"]():      # (plan_base: the bf16 mirror the optimiser just wrote)
                    continue
                lane = meta["lane"] if streams is not None and streams.lanes else 0
                if lane != cur:
                    torch.cuda.set_stream(streams.stream(lane))
                    st, cur = stream_ptr(), lane
                timed = timer is not None and timer.wants(meta["kind"])
                if timed:
                    timer.begin(meta["kind"], meta)
                if streams is not None and streams.use_side and not timed and meta.get("side"):     # (a timed launch is bracketed by events on its lane's stream)
                    six = meta.get("side_ix", 0)
                    rc = fn(*args, streams.fork_side(six) if six else streams.fork_side())
                    if meta.get("side_tag") is not None:
                        streams.mark_side(meta["side_tag"], six) if six else streams.mark_side(meta["side_tag"])
                else:
                    rc = fn(*args, st)
                if timed:
                    timer.end()
                if rc != 0:
                    _lib.check(rc, name)
                if hook is not None:
                    hook(idx)
        finally:
            if cur != 0:
                torch.cuda.set_stream(streams.main)


class StepStreams:
    """The HIP streams of one train step: main = the caller's current stream (lane 0), lane1 = a second compute lane for the
    deeper decoder blocks (SNUNetPlan: they depend on the level-0 blocks only through the Up1_j edges), side = the weight gradients.
    Cross-stream ordering is plain event record / wait pairs, so a step that uses them still captures into one HIP graph."""

    def __init__(self, device, lanes=True, side=True):
        self.side = torch.cuda.Stream(device=device)       # (stream priorities measured no better)
        # KSMI_SIDE2=1 (experiment): a second side stream; the SNUNet plan alternates its weight gradients between the two by parameter
        # (meta["side_ix"]: launches that accumulate into one gradient stay on one stream), so the slab reducer of one weight gradient
        # runs beside the main kernel of the next
        self.side2 = torch.cuda.Stream(device=device) if (side and os.environ.get("KSMI_SIDE2", "0") == "1") else None
        self.side2_ptr = C.c_void_p(self.side2.cuda_stream) if self.side2 is not None else None
        self.lane1 = torch.cuda.Stream(device=device) if lanes else None
        self.lanes, self.use_side = bool(lanes), bool(side)
        self.side_ptr = C.c_void_p(self.side.cuda_stream)
        self.main = None
        self.dirty = False
        self.side_busy = False
        self.events = {}           # side_tag -> event recorded behind that launch on the side stream (LaunchList.add_wait_side)

    _runner = None

    def begin(self):
        if self.main is None:
            self.main = torch.cuda.current_stream()
            if self._runner is not None:
                self._bind_runner()

    def _bind_runner(self):
        _lib.load().ksmi_runner_set_streams(self._runner, C.c_void_p(self.main.cuda_stream),
                                            C.c_void_p(self.lane1.cuda_stream) if self.lanes else None,
                                            self.side_ptr if self.use_side else None, self.side2_ptr if self.use_side else None)

    def runner(self):
        """executor state of the compiled launch lists (csrc/runlist.hip) bound to this step's streams; call after begin()"""
        if self._runner is None:
            self._runner = C.c_void_p(_lib.load().ksmi_runner_create())
            self._bind_runner()
        return self._runner

    def __del__(self):
        try:
            if self._runner is not None:
                _lib.load().ksmi_runner_destroy(self._runner)
        except Exception:
            pass

    def stream(self, lane):
        return self.main if lane == 0 else self.lane1

    def _event(self, stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        return ev

    def order(self, src, dst):
        self.stream(dst).wait_event(self._event(self.stream(src)))
        self.dirty = True

    def fork_side(self, ix=0):
        two = ix and self.side2 is not None
        (self.side2 if two else self.side).wait_event(self._event(torch.cuda.current_stream()))
        self.dirty = self.side_busy = True
        return self.side2_ptr if two else self.side_ptr

    def mark_side(self, tag, ix=0):
        self.events[tag] = self._event(self.side2 if (ix and self.side2 is not None) else self.side)

    def wait_side(self, tag):
        if tag is None:
            if self.side_busy:                 # (nothing handed to the side stream since the last such wait: nothing to wait for)
                torch.cuda.current_stream().wait_stream(self.side)
                if self.side2 is not None:
                    torch.cuda.current_stream().wait_stream(self.side2)
                self.side_busy = False
        elif tag in self.events:
            torch.cuda.current_stream().wait_event(self.events.pop(tag))

    def all_streams(self):
        """every stream a launch of the step may have run on"""
        return [s for s in (self.main, self.lane1 if self.lanes else None, self.side if self.use_side else None,
                            self.side2 if self.use_side else None) if s is not None]

    def join(self):
        """the current stream waits for every other stream of the step"""
        if self.dirty:
            cur = torch.cuda.current_stream()
            for s in (self.main, self.lane1, self.side, self.side2):
                if s is not None and s.cuda_stream != cur.cuda_stream:
                    cur.wait_stream(s)
        if self.main is not None and torch.cuda.current_stream().cuda_stream == self.main.cuda_stream:
            self.dirty = False

    def end(self):
        self.join()
        if self._runner is not None:                 # (the compiled lists keep their own dirty / tag state: main joins the other streams)
            _lib.check(_lib.load().ksmi_runner_join(self._runner), "ksmi_runner_join")
        self.main = None
        self.events.clear()
        self.side_busy = False


class _Saved:
    """saved statistics of one BatchNorm call: rows = mean, rstd, scale, shift"""

    def __init__(self, Cch, device):
        self.t = torch.zeros((4, Cch), dtype=torch.float32, device=device)
        self.mean, self.rstd, self.scale, self.shift = (self.t[i].data_ptr() for i in range(4))
        self.scale_t, self.shift_t = self.t[2], self.t[3]


class SNUNetPlan:
    side_wgrad = True          # weight gradients on the train step's side stream (see LaunchList.run; plan_base.PlanBase.side_wgrad)
    two_lanes = True           # the decoder launches carry lane tags and hand-over entries (StepStreams)
    bn_fused = os.environ.get("KSMI_BN_FUSED", "1") != "0"     # statistics finish inside the consuming pass (csrc/bnfused.hip)
    im2col_late = os.environ.get("KSMI_IM2COL_LATE", "1") != "0"   # first-layer im2col in the backward list, next to its reader
    up_gemm = os.environ.get("KSMI_UP_GEMM", "1") != "0"           # ConvTranspose2d(k2, s2) with C >= 128 as token GEMMs (ksmi_up_*)
    up_wgrad64 = os.environ.get("KSMI_UP_WGRAD64", "1") != "0"     # ... and the level-0 Up weight gradients (C = 64)
    up_wgrad128 = os.environ.get("KSMI_UP_WGRAD128", "1") != "0"   # ... and the level-1 ones (C = 128; after the reducer was parallelised)

    def __init__(self, model, B, H, W, dtype, training, with_backward, tail=0, sync_bn=False):
        self.m, self.B, self.H, self.W, self.dtype = model, B, H, W, dtype
        # SyncBN (optional, SURVEY.md §8(e) "second-order items"): BatchNorm statistics of the GLOBAL batch -- the statistics rows of every
        # BatchNorm call (forward: sum, sum of squares; backward: sum g, sum g * xhat) are summed over the ranks before they are
        # finished, with the global pixel count.  Data parallelism with it equals one process on the whole batch
        # (tests/test_gpu_dp.py::test_syncbn_two_ranks_equal_single_process); the default (off) is the reference's per-rank BatchNorm
        # (models/snunet.py:16,18 under DataParallel-free training).  Uses the unfused statistics passes (one tiny collective per call).
        from . import distributed as _D
        self.sync_bn = bool(sync_bn) and training
        self.bn_world = _D.world_size() if self.sync_bn else 1
        if self.sync_bn:
            self.bn_fused = False
        self.training, self.with_backward = training, with_backward
        self.dev = model.flat_params.device
        self.dt = DT[dtype]
        self.lib = _lib.load()
        self.packs, self.fwd, self.bwd = LaunchList(), LaunchList(), LaunchList()
        self.keep = []
        self._pinit = set()
        self._pack_descs = []      # every weight-pack descriptor of the plan -> ONE batched launch per step
        self._up_packs = []        # (ConvTranspose weight, its [4C][C] bf16 image, C) of the `up` layers on the token-GEMM path
        self.param_ready = {}      # parameter key -> index of the last backward launch writing its gradient
        self._need, self._bufs, self._later = {}, {}, []
        self._rowsums = []         # deferred row reductions (bias gradients): (RowsumDesc, parameter key) -> ONE launch ending the backward
        n, c = model.base_channel, model.in_channels
        self.n = n
        f = [n, 2 * n, 4 * n, 8 * n, 16 * n]
        # `tail` trailing input channels (the DEM) are shared by both dates and live in their own buffer: the first conv reads
        # channels [0, c - tail) from xA / xB and the rest from xtail (the trainer's torch.cat((image, dem), 1) as an address choice)
        self.tail = tail
        self.xA = torch.empty((B, c - tail, H, W), dtype=torch.float32, device=self.dev)
        self.xB = torch.empty_like(self.xA)
        self.xtail = torch.empty((B, tail, H, W), dtype=torch.float32, device=self.dev) if tail else None
        self.logits = torch.empty((B, 3, H, W), dtype=torch.float32, device=self.dev)
        self.dlogits = torch.empty_like(self.logits) if with_backward else None
        self.bwd_builders = []

        def A(name, lvl, ch):
            return Act(name, B, H >> lvl, W >> lvl, ch, dtype, self.dev)

        # ---- forward graph (snunet.py:118-153) -----------------------------------------------
        # Siamese encoder, level by level: date A on lane 0, date B on lane 1, B one block behind A.  Both dates share every parameter:
        # BatchNorm running statistics are updated A first, then B (the reference's order, snunet.py:121-130), and in the backward list
        # (built in reverse) the date-B block writes the shared BatchNorm gradients ("=") before the date-A block accumulates ("+=");
        # the hand-over entries between the two blocks of a level keep exactly that order on two streams.
        L = self._lane
        L(0); x0_0A, p0A = self._block("conv0_0", "A", [self.xA], A("x0_0A", 0, f[0]), first=True, pool="p0A")
        self._handover(0, 1, back=(1, 0))
        L(1); x0_0B, p0B = self._block("conv0_0", "B", [self.xB], A("x0_0B", 0, f[0]), first=True, pool="p0B")
        L(0); x1_0A, p1A = self._block("conv1_0", "A", [self._pool_at(x0_0A, p0A, "p0A")], A("x1_0A", 1, f[1]), pool="p1A")
        self._handover(0, 1, back=(1, 0))
        L(1); x1_0B, p1B = self._block("conv1_0", "B", [self._pool_at(x0_0B, p0B, "p0B")], A("x1_0B", 1, f[1]), pool="p1B")
        L(0); x2_0A, p2A = self._block("conv2_0", "A", [self._pool_at(x1_0A, p1A, "p1A")], A("x2_0A", 2, f[2]), pool="p2A")
        self._handover(0, 1, back=(1, 0))
        L(1); x2_0B, p2B = self._block("conv2_0", "B", [self._pool_at(x1_0B, p1B, "p1B")], A("x2_0B", 2, f[2]), pool="p2B")
        L(0); x3_0A = self._block("conv3_0", "A", [self._pool_at(x2_0A, p2A, "p2A")], A("x3_0A", 3, f[3]))
        self._handover(0, 1, back=(1, 0))
        L(1); x3_0B, p3B = self._block("conv3_0", "B", [self._pool_at(x2_0B, p2B, "p2B")], A("x3_0B", 3, f[3]), pool="p3B")
        x4_0B = self._block("conv4_0", "B", [self._pool_at(x3_0B, p3B, "p3B")], A("x4_0B", 4, f[4]))
        self._handover(1, 0, back=(0, 1))                  # encoder | decoder: both lanes have everything of the other side

        # Decoder on two lanes (LaunchList.run / StepStreams): lane 0 = the level-0 column blocks + the head, lane 1 = every deeper block and the Ups
        # leaving it (Up1_0, fed by the encoder, stays with lane 0).  Lane 1 needs the encoder only; lane 0 needs lane 1 through
        # x1_j -> Up1_j (forward) and lane 1 needs lane 0 through d Up1_j -> d x1_j (backward): three ordered hand-overs each way.
        # The level-0 blocks carry the large BatchNorm / elementwise passes (224^2 maps), the deeper blocks are convolution-bound.
        self._handover(0, 1, back=(1, 0))
        L(0); x0_1 = self._block("conv0_1", "", [x0_0A, x0_0B, self._up("Up1_0", x1_0B)], A("x0_1", 0, f[0]))
        L(1); x1_1 = self._block("conv1_1", "", [x1_0A, x1_0B, self._up("Up2_0", x2_0B)], A("x1_1", 1, f[1]))
        u1_1 = self._up("Up1_1", x1_1)                      # (the Up1_j ride on lane 1 with their source block: lane 0 is the longer one)
        self._handover(1, 0, back=(0, 1))
        L(0); x0_2 = self._block("conv0_2", "", [x0_0A, x0_0B, x0_1, u1_1], A("x0_2", 0, f[0]))
        L(1); x2_1 = self._block("conv2_1", "", [x2_0A, x2_0B, self._up("Up3_0", x3_0B)], A("x2_1", 2, f[2]))
        x1_2 = self._block("conv1_2", "", [x1_0A, x1_0B, x1_1, self._up("Up2_1", x2_1)], A("x1_2", 1, f[1]))
        u1_2 = self._up("Up1_2", x1_2)
        self._handover(1, 0, back=(0, 1))
        L(0); x0_3 = self._block("conv0_3", "", [x0_0A, x0_0B, x0_1, x0_2, u1_2], A("x0_3", 0, f[0]))
        L(1); x3_1 = self._block("conv3_1", "", [x3_0A, x3_0B, self._up("Up4_0", x4_0B)], A("x3_1", 3, f[3]))
        x2_2 = self._block("conv2_2", "", [x2_0A, x2_0B, x2_1, self._up("Up3_1", x3_1)], A("x2_2", 2, f[2]))
        x1_3 = self._block("conv1_3", "", [x1_0A, x1_0B, x1_1, x1_2, self._up("Up2_2", x2_2)], A("x1_3", 1, f[1]))
        u1_3 = self._up("Up1_3", x1_3)
        self._handover(1, 0, back=(0, 1))
        L(0); x0_4 = self._block("conv0_4", "", [x0_0A, x0_0B, x0_1, x0_2, x0_3, u1_3], A("x0_4", 0, f[0]))
        self._head([x0_1, x0_2, x0_3, x0_4])
        self.acts = {a.name: a for a in (x0_0A, x0_0B, x1_0A, x1_0B, x3_0B, x4_0B, x0_1, x1_1, x0_2, x0_3, x0_4, x1_3, x3_1)}

        if with_backward:
            for lane, build in reversed(self.bwd_builders):
                self._lane(lane)
                build()
            self._lane(0)
            if self._rowsums:
                import ctypes
                nr = len(self._rowsums)
                arr_r = (_lib.RowsumDesc * nr)(*[r for r, _ in self._rowsums])
                raw_r = bytes(ctypes.string_at(ctypes.addressof(arr_r), ctypes.sizeof(arr_r)))
                rtable = torch.frombuffer(bytearray(raw_r), dtype=torch.uint8).to(self.dev)
                self.keep.append(rtable)
                self.bwd.add("ksmi_reduce_rows_batched", lambda: (rtable.data_ptr(), nr))
                self._mark(*[k for _, k in self._rowsums])
        for i in range(0, len(self._up_packs), 16):           # KSMI_UP_PACK_MAX tensors per launch
            grp = self._up_packs[i:i + 16]
            import ctypes
            wt_a = (ctypes.c_void_p * len(grp))(*[w.data_ptr() for w, _, _ in grp])
            wb_a = (ctypes.c_void_p * len(grp))(*[b.data_ptr() for _, b, _ in grp])
            c_a = (ctypes.c_int * len(grp))(*[c for _, _, c in grp])
            self.keep += [wt_a, wb_a, c_a]
            self.packs.add("ksmi_up_pack_weights_batched", lambda wt_a=wt_a, wb_a=wb_a, c_a=c_a, k=len(grp): (wt_a, wb_a, c_a, k),
                           {"kind": "up_pack_weight", "bytes": sum(4 * c * c * 6 for _, _, c in grp), "flops": 0})
        # all weight packs of the step as one launch over a device-resident descriptor table
        if self._pack_descs:
            import ctypes
            n = len(self._pack_descs)
            arr = (_lib.PackDesc * n)(*self._pack_descs)
            raw = bytes(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr)))
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
            self.keep.append(table)
            self.packs.add("ksmi_pack_weights_batched", lambda: (table.data_ptr(), n, self.dt))
        # scratch allocation, descriptor patching, argument resolution
        for name, nbytes in self._need.items():
            self._bufs[name] = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.dev)
        for fn in self._later:
            fn()
        for ll in (self.packs, self.fwd, self.bwd):
            ll.resolve(self.lib)

    # ---------------------------------------------------------------- helpers
    def _lane(self, lane):
        """launches appended from here on belong to compute lane `lane`"""
        self.fwd.cur_lane = self.bwd.cur_lane = lane

    def _stage(self, H):
        """measurement tag: the resolution level of the maps a module works on (L0 = full resolution ... L4 = 1/16)"""
        lvl = {self.H >> k: k for k in range(5)}.get(H)
        self.fwd.cur_stage = self.bwd.cur_stage = None if lvl is None else f"L{lvl}"

    def _sname(self, name):
        """scratch buffers that live from one launch to the next of the same block are per lane"""
        return name if self.fwd.cur_lane == 0 else f"{name}@{self.fwd.cur_lane}"

    def _handover(self, src, dst, back):
        """forward list: lane dst continues behind everything lane src was handed so far; backward list (built in reverse): lane
        back[1] behind lane back[0] at the mirrored position"""
        self.fwd.add_wait(src, dst)
        a, b = back
        self.bwd_builders.append((0, lambda: self.bwd.add_wait(a, b)))

    def need(self, name, nbytes):
        self._need[name] = max(self._need.get(name, 0), int(nbytes))

    def scr(self, name):
        return self._bufs[name].data_ptr()

    def patch(self, desc, field, name):
        self._later.append(lambda: setattr(desc, field, self.scr(name)))

    def _acc_param(self, key):
        acc = 1 if key in self._pinit else 0
        self._pinit.add(key)
        return acc

    def _packed(self, key, table, taps, N, n_mod, sK, sN, sD, sT, flip):
        Npad = conv_npad(N)                          # (= ksmi_conv_desc.Npad of the layer: 32 columns below 16 channels, runtime.conv_npad)
        out = torch.empty(packed_weight_numel(table, taps, Npad, self.dtype), dtype=self.dtype, device=self.dev)
        d = make_pack(self.m._p(key), out, table, taps, N, Npad, n_mod, sK, sN, sD, sT, flip)
        self.keep += [d, out]
        self._pack_descs.append(d)
        return out

    def _rows(self, npix):
        return max(1, min(512, npix // 256))

    def _mark(self, *keys):
        """the launch just appended to self.bwd is (so far) the last writer of these gradients"""
        for k in keys:
            self.param_ready[k] = len(self.bwd.pending) - 1

    def _defer_rowsum(self, key, partial, rows, K, k, Cstride, Cc):
        """grad[key][c] (+)= sum_rows partial[(r*K + k)*Cstride + c] in the batched launch that ends the backward; partial = None: zeros"""
        r = _lib.RowsumDesc()
        r.partial = partial.data_ptr() if partial is not None else None
        r.dst = self.m._g(key).data_ptr()
        r.rows, r.K, r.k, r.Cstride, r.C = (rows if partial is not None else 0), K, k, Cstride, Cc
        r.accumulate, r.head, r.next = 0, 1, -1
        if self._acc_param(key):                      # a second call of the same module (siamese branches): chain behind the first entry
            prev = [q for q, kk in self._rowsums if kk == key][-1]
            prev.next, r.head = len(self._rowsums), 0
        if partial is not None:
            self.keep.append(partial)
        self._rowsums.append((r, key))

    def _es(self):
        return 2 if self.dtype == torch.bfloat16 else 4

    def _conv(self, ll, d, tag="fwd"):
        self.keep.append(d)
        d.dir = 1 if "dgrad" in tag else 0            # (profiling tag: kernel names carry the direction, ksmi.h)
        taps, es = d.KH * d.KW, self._es()
        ktot = sum(d.src[i].c_len for i in range(d.nsrc))
        pin, pout = d.B * d.Hin * d.Win, d.B * d.Hout * d.Wout
        elems = pin * ktot + sum(pout * d.dst[i].n_len * (2 if d.dst[i].accumulate else 1) for i in range(d.ndst))
        if d.mask_src:
            elems += pout * d.N
        nt = 2 if d.Npad >= 32 else 1
        meta = {"kind": f"igemm_{tag}<{d.KH}x{d.KW}s{d.stride},BN{16 * nt}>", "bytes": elems * es + taps * ktot * d.N * es,
                "flops": 2 * pout * d.N * ktot * taps}
        meta["tag"] = f"K={ktot} N={d.N} {d.Hout}x{d.Wout} nsrc={d.nsrc}"
        ll.add("ksmi_conv_forward", lambda: (C.byref(d), self.dt), meta)

    def _side_ix(self, key):
        """side stream of a parameter's weight gradient (KSMI_SIDE2=1: alternating by first appearance; a parameter keeps its stream, so
        the two launches of a shared encoder weight stay ordered)"""
        if os.environ.get("KSMI_SIDE2", "0") != "1":
            return 0
        if not hasattr(self, "_side_of"):
            self._side_of = {}
        if key not in self._side_of:
            self._side_of[key] = len(self._side_of) & 1
        return self._side_of[key]

    def _wgrad(self, d, ws, *keys):
        self.keep.append(d)
        # the partial-slab scratch is per compute lane: with the side stream off (or a timed launch) the weight gradients of the two
        # lanes run concurrently on their lanes' streams (and per side stream: two weight gradients in flight own two slabs)
        six = self._side_ix(keys[0]) if keys else 0
        sW = self._sname("wgrad" + ("_s2" if six else ""))
        self.need(sW, ws)
        self.patch(d, "partial", sW)
        taps, es = d.KH * d.KW, self._es()
        ktot = sum(d.src[i].c_len for i in range(d.nsrc))
        pin, pout = d.B * d.Hin * d.Win, d.B * d.Hout * d.Wout
        meta = {"kind": f"igemm_wgrad<{d.KH}x{d.KW}s{d.stride}>", "bytes": (pin * ktot + pout * d.N) * es + taps * ktot * d.N * 4,
                "flops": 2 * pout * d.N * ktot * taps}
        meta["tag"] = f"{keys[0] if keys else '?'} K={ktot} N={d.N} {d.Hout}x{d.Wout}"
        meta["side"] = True                  # off the critical path: eligible for the side stream (LaunchList.run)
        meta["side_ix"] = six
        self.bwd.add("ksmi_conv_wgrad", lambda: (C.byref(d), self.dt), meta)
        self._mark(*keys)

    # ---------------------------------------------------------------- "virtual sum" input gradient
    def _emit_dgrad(self, act, bias_key=None, gate=None):
        """d act = sum over the 3x3 convs that read `act` of convT(di_j, W_j[:, slice_j]) as ONE implicit GEMM
        whose K axis walks the consumers' `di` tensors (the dual of the forward's virtual concat): every
        gradient tensor is written once instead of read-modify-written per consumer, and K grows from
        Cout_j to sum_j Cout_j.  Consumers registered themselves in their own build_bwd (earlier in the
        backward order)."""
        cons = getattr(act, "consumers", [])
        if not cons:
            return False
        B, H, W, Cc = act.B, act.H, act.W, act.C
        srcs = [SrcSpec(r, cj) for (r, cj, _, _, _) in cons]
        acc = act.take_acc_flag()
        d, table = make_conv(srcs, [(act.grad(), Cc, 0, 0, Cc, acc)], act.grad(), None, None, B, H, W, H, W, 3, 3, 1, 1, Cc, self.dtype)
        Npad = conv_npad(Cc)
        wpk = torch.empty(packed_weight_numel(table, 9, Npad, self.dtype), dtype=self.dtype, device=self.dev)
        kc = 32 if self.dtype == torch.bfloat16 else 16
        slab = 9 * Npad * kc
        ch0 = 0
        for (r, cj, wkey, coff, ktot) in cons:
            nch = -(-cj // kc)
            tj = [(0, c0, c0, min(kc, cj - c0)) for c0 in range(0, cj, kc)]
            # element (chunk, tap', col = c_local, kk = n_local) = W_j[n][coff + c_local][flip(tap')]
            wview = self.m._p(wkey)[coff * 9:]
            out = wpk[ch0 * slab:(ch0 + nch) * slab]
            pd = make_pack(wview, out, tj, 9, Cc, Npad, Cc, ktot * 9, 9, 0, 1, 1)
            self.keep += [pd, wview, out]
            self._pack_descs.append(pd)
            ch0 += nch
        d.wpk = wpk.data_ptr()
        self.keep.append(wpk)
        gated = None
        if gate is not None:
            g_out, g_z, g_sv = gate
            d.gate_src, d.xhat_src, d.g_mean, d.g_rstd = g_out.data_ptr(), g_z.data_ptr(), g_sv.mean, g_sv.rstd
            if self.lib.ksmi_conv_gate_supported(C.byref(d), self.dt):
                rows_g = conv_stats_rows(d, self.dtype)
                st = torch.empty(rows_g * 2 * Npad, dtype=torch.float32, device=self.dev)
                d.stats = st.data_ptr()
                self.keep.append(st)
                gated = (st, rows_g, Npad)
            else:
                d.gate_src = d.xhat_src = d.g_mean = d.g_rstd = None
        if bias_key is not None:
            # `act` is the output of a ConvTranspose2d whose bias gradient is sum_pixels d act: the statistics epilogue of this launch
            # emits the per-tile sums (was: one more pass over the gradient tensor, ksmi_channel_sum)
            rows_g = conv_stats_rows(d, self.dtype)
            st = torch.empty(rows_g * 2 * Npad, dtype=torch.float32, device=self.dev)
            d.stats = st.data_ptr()
            self._defer_rowsum(bias_key, st, rows_g, 2, 0, Npad, Cc)
        self._conv(self.bwd, d, "dgrad")
        return gated if gated is not None else True

    # ---------------------------------------------------------------- nn.MaxPool2d(2,2)  (snunet.py:73)
    def _pool(self, x, name):
        y = Act(name, x.B, x.H // 2, x.W // 2, x.C, self.dtype, self.dev)
        self._stage(x.H)
        self.fwd.add("ksmi_maxpool2x2_forward", lambda: (x.t.data_ptr(), y.t.data_ptr(), x.B, x.H, x.W, x.C, self.dt))
        self._pool_bwd(x, y)
        return y

    def _pool_at(self, x, y, name):
        """the pooled copy of an encoder block output at this point of the lists: y from the fused block tail (only its backward is
        registered here, at the place the stand-alone pool has in the launch order), or None -> the stand-alone launch"""
        if y is None:
            return self._pool(x, name)
        self._pool_bwd(x, y)
        return y

    def _pool_bwd(self, x, y):
        """backward of y = maxpool2x2(x) (y written by ksmi_maxpool2x2_forward or by the fused block tail)"""
        def build_bwd():
            self._stage(x.H)
            self._emit_dgrad(y)
            acc = x.take_acc_flag()
            gy, gx = y.grad(), x.grad()
            self.bwd.add("ksmi_maxpool2x2_backward", lambda: (x.t.data_ptr(), gy.data_ptr(), gx.data_ptr(), acc,
                                                              x.B, x.H, x.W, x.C, self.dt))
        self.bwd_builders.append((self.fwd.cur_lane, build_bwd))

    def _up_conv_forward(self, x, y, wkey, bkey, B, H, W, Cc):
        d, table = make_conv([SrcSpec(x.t, Cc)], [(y.t, Cc, 0, 0, 4 * Cc, 0)], x.t, self.m._p(bkey), None,
                             B, H, W, H, W, 1, 1, 1, 0, 4 * Cc, self.dtype, ps_cout=Cc)
        # Wt[c][n][dy][dx]: GEMM column j = d*C + n, k = c
        wpk = self._packed(wkey, table, 1, 4 * Cc, Cc, Cc * 4, 4, 1, 0, 0)
        d.wpk = wpk.data_ptr()
        self._conv(self.fwd, d)

    # ---------------------------------------------------------------- up = ConvTranspose2d(k2,s2)  (snunet.py:32-46)
    def _up(self, name, x):
        Cc, B, H, W = x.C, x.B, x.H, x.W
        y = Act(name, B, 2 * H, 2 * W, Cc, self.dtype, self.dev)
        self._stage(2 * H)                                   # (an Up is booked on the level it writes)
        wkey, bkey = f"{name}.up.weight", f"{name}.up.bias"
        # levels with C >= 128 (Up2_j, Up3_j, Up4_0): the transposed convolution and both of its gradients as token GEMMs over the
        # "depth rows" of the output (csrc/gemm2.hip, ksmi_up_*): 400-600 TFLOP/s kernels instead of the 140-240 TFLOP/s the k2 s2
        # shapes reach on the first-generation convolution kernels
        up_gemm = (self.up_gemm and self.dtype == torch.bfloat16 and bool(self.lib.ksmi_up_gemm_supported(B, H, W, Cc, self.dt)))
        # measured per operation (profiles/r04_up_gemm.txt, one stream): the input gradient (K = 4C) wins from C = 128 on (76 -> 49 us at
        # 56^2), forward and weight gradient from C = 256 on (77 -> 57 us, 79 -> ~40 us at 28^2); at C = 128 the forward GEMM has two K
        # steps per tile and loses to the persistent 1x1 kernel (33-45 -> 80 us), the weight gradient ties (73 -> 70 us)
        up_fwd_gemm = up_gemm and Cc >= 256
        up_wgrad_gemm = up_fwd_gemm or (self.up_gemm and ((self.up_wgrad64 and Cc == 64) or (self.up_wgrad128 and Cc == 128))
                                        and self.dtype == torch.bfloat16 and bool(self.lib.ksmi_up_wgrad_supported(B, H, W, Cc, self.dt)))
        es = self._es()
        if up_gemm:
            wb = torch.empty(4 * Cc * Cc, dtype=torch.bfloat16, device=self.dev)
            self.keep.append(wb)
            self._up_packs.append((self.m._p(wkey), wb, Cc))        # one batched launch for all of them (end of __init__)
        if up_fwd_gemm:
            self.fwd.add("ksmi_up_forward", lambda: (x.t.data_ptr(), wb.data_ptr(), self.m._p(bkey).data_ptr(), y.t.data_ptr(), B, H, W, Cc),
                         {"kind": "up_gemm_fwd", "bytes": B * H * W * Cc * 5 * es, "flops": 2 * B * H * W * Cc * 4 * Cc, "tag": f"K={Cc} N={4 * Cc} {H}x{W}"})
        else:
            self._up_conv_forward(x, y, wkey, bkey, B, H, W, Cc)

        def build_bwd():
            self._stage(2 * H)
            fused_bias = self._emit_dgrad(y, bias_key=bkey)
            gy = y.grad()
            s2 = [SrcSpec(gy, Cc)]
            acc = x.take_acc_flag()
            if up_gemm:
                gx = x.grad()
                self.bwd.add("ksmi_up_dgrad", lambda: (gy.data_ptr(), wb.data_ptr(), gx.data_ptr(), acc, B, H, W, Cc),
                             {"kind": "up_gemm_dgrad", "bytes": B * H * W * Cc * (5 + acc) * es, "flops": 2 * B * H * W * Cc * 4 * Cc,
                              "tag": f"K={4 * Cc} N={Cc} {H}x{W}"})
            else:
                # input gradient = 2x2 stride-2 conv over dUp: K = n, N = c
                d2, t2 = make_conv(s2, [(x.grad(), Cc, 0, 0, Cc, acc)], gy, None, None,
                                   B, 2 * H, 2 * W, H, W, 2, 2, 2, 0, Cc, self.dtype)
                w2 = self._packed(wkey, t2, 4, Cc, Cc, 4, Cc * 4, 0, 1, 0)
                d2.wpk = w2.data_ptr()
                self._conv(self.bwd, d2, "dgrad")
            if up_wgrad_gemm:
                six = self._side_ix(wkey)
                sW = self._sname("wgrad" + ("_s2" if six else ""))
                self.need(sW, self.lib.ksmi_up_wgrad_workspace(B, H, W, Cc))
                a_w = self._acc_param(wkey)
                self.bwd.add("ksmi_up_wgrad", lambda: (x.t.data_ptr(), gy.data_ptr(), self.scr(sW), self.m._g(wkey).data_ptr(), a_w, B, H, W, Cc),
                             {"kind": "up_gemm_wgrad", "bytes": B * H * W * Cc * 5 * es + 16 * Cc * Cc, "flops": 2 * B * H * W * Cc * 4 * Cc,
                              "tag": f"{wkey} K={Cc} N={4 * Cc} {H}x{W}", "side": True, "side_ix": six})
                self._mark(wkey)
            else:
                # weight gradient: G[tap d][k = n][col = c] -> grad[c*(4C) + n*4 + d]
                dw, ws = make_wgrad(s2, x.t, Cc, 0, Cc, self.m._g(wkey), 4, Cc * 4, 1, self._acc_param(wkey),
                                    B, 2 * H, 2 * W, H, W, 2, 2, 2, 0, self.dtype)
                self._wgrad(dw, ws, wkey)
            if not fused_bias:                                      # (no 3x3 consumer: separate pass over the gradient)
                npix = B * 4 * H * W
                rows = self._rows(npix)
                pb = torch.empty(rows * Cc, dtype=torch.float32, device=self.dev)
                self.bwd.add("ksmi_channel_sum", lambda: (gy.data_ptr(), pb.data_ptr(), rows, npix, Cc, self.dt))
                self._defer_rowsum(bkey, pb, rows, 1, 0, Cc, Cc)
        self.bwd_builders.append((self.fwd.cur_lane, build_bwd))
        return y

    # ---------------------------------------------------------------- conv_block_nested  (snunet.py:11-29)
    def _block(self, name, branch, sources, out, first=False, pool=None):
        """pool: name of the max-pooled copy of the block output (encoder blocks, snunet.py:121-130).  With the fused BatchNorm glue
        (csrc/bnfused.hip) the pooled tensor comes out of the launch that writes the block output; returns (out, pooled) then."""
        m, B, H, W, Cc = self.m, out.B, out.H, out.W, out.C
        npix = B * H * W
        dtype, dt, training = self.dtype, self.dt, self.training
        self._stage(H)
        i_act = Act(f"{name}{branch}.i", B, H, W, Cc, dtype, self.dev)
        z_act = Act(f"{name}{branch}.z", B, H, W, Cc, dtype, self.dev)
        sv1, sv2 = _Saved(Cc, self.dev), _Saved(Cc, self.dev)
        self.keep += [i_act, z_act, sv1, sv2]
        P = lambda s: m._p(f"{name}.{s}").data_ptr()
        G = lambda s: m._g(f"{name}.{s}").data_ptr()
        Bf = lambda s: m._b(f"{name}.{s}").data_ptr()
        Npad = conv_npad(Cc)
        sS, sR = self._sname("stats"), self._sname("red")             # per-lane scratch (this block's lane, forward and backward)
        stats = (lambda: self.scr(sS)) if training else (lambda: None)

        # ---- conv1 ----------------------------------------------------------------------
        if first:
            # forward: direct fp32 conv on the raw NCHW image (exact fp32 operands even in bf16 mode: quantising
            # the input image to bf16 costs ~2 points of gradient cosine downstream).  For the weight gradient
            # the image is also laid out as im2col [B,H,W,Kpad] (k = c*9+t) so dW runs on the MFMA wgrad kernel.
            x_img = sources[0]
            chead, cin = x_img.shape[1], m.in_channels
            x_tail = self.xtail.data_ptr() if self.tail else None
            kc = 32 if dtype == torch.bfloat16 else 16
            Kpad = -(-(cin * 9) // kc) * kc
            rows1, cpad1, Ktot = self.lib.ksmi_conv_first_stats_rows(B, H, W), Cc, cin
            self.need(sS, rows1 * 2 * Cc * 4)
            # raw tiles (model.set_input_pipeline): clamp / NaN / Normalize happen in the image load of both kernels
            raw = m._raw_ptrs(self.dev)
            self.fwd.add("ksmi_conv_first_forward_raw", lambda: (x_img.data_ptr(), x_tail, chead, P("conv1.weight"), P("conv1.bias"),
                                                                 i_act.t.data_ptr(), stats(), B, cin, H, W, Cc, *raw, dt))
            if self.with_backward:
                # the im2col image of the raw tile for the first-layer weight gradient is laid out in the BACKWARD list, right in front
                # of its only reader (the input buffers of the plan stay untouched until the next batch is set): the 103 MB it writes
                # are still in the 256 MB memory-side cache when the weight gradient reads them, and the forward pass does not carry them
                col = torch.empty((B, H, W, Kpad), dtype=dtype, device=self.dev)
                self.keep.append(col)
                im2col_args = lambda: (x_img.data_ptr(), x_tail, chead, col.data_ptr(), B, cin, H, W, Kpad, *raw, dt)
                if not self.im2col_late:
                    self.fwd.add("ksmi_im2col3x3_raw", im2col_args)
                src1 = [SrcSpec(col, Kpad)]
        else:
            srcs = [SrcSpec(a.t, a.C) for a in sources]
            Ktot = sum(a.C for a in sources)
            d1, t1 = make_conv(srcs, [(i_act.t, Cc, 0, 0, Cc, 0)], i_act.t, m._p(f"{name}.conv1.bias"), None,
                               B, H, W, H, W, 3, 3, 1, 1, Cc, dtype)
            w1 = self._packed(f"{name}.conv1.weight", t1, 9, Cc, Cc, 9, Ktot * 9, 0, 1, 0)
            d1.wpk = w1.data_ptr()
            rows1, cpad1 = conv_stats_rows(d1, dtype), Npad
            if training:
                self.need(sS, rows1 * 2 * Npad * 4)
                self.patch(d1, "stats", sS)
            self._conv(self.fwd, d1)

        gcount = float(npix * self.bn_world)                         # pixels the statistics describe (SyncBN: of all ranks)

        def bn_fin(bn, sv, rows, cpad):
            nbt = m._c(f"{name}.{bn}.num_batches_tracked").data_ptr()
            if self.sync_bn:                                          # rows of all ranks add up element-wise; the finish sums the rows
                self.fwd.add_allreduce(lambda rows=rows, cpad=cpad: self._bufs[sS][:rows * 2 * cpad * 4].view(torch.float32))
            self.fwd.add("ksmi_bn_finalize", lambda: (stats(), rows, cpad, Cc, gcount, P(f"{bn}.weight"), P(f"{bn}.bias"),
                                                      Bf(f"{bn}.running_mean"), Bf(f"{bn}.running_var"), nbt,
                                                      BN_MOMENTUM, BN_EPS, 1 if training else 0,
                                                      sv.mean, sv.rstd, sv.scale, sv.shift))
        bn_fin("bn1", sv1, rows1, cpad1)

        # ---- conv2 (BN1-apply + ReLU fused into the operand load) ---------------------------
        src2 = [SrcSpec(i_act.t, Cc, scale=sv1.scale_t, shift=sv1.shift_t, relu=1)]
        d2, t2 = make_conv(src2, [(z_act.t, Cc, 0, 0, Cc, 0)], z_act.t, m._p(f"{name}.conv2.bias"), None,
                           B, H, W, H, W, 3, 3, 1, 1, Cc, dtype)
        w2 = self._packed(f"{name}.conv2.weight", t2, 9, Cc, Cc, 9, Cc * 9, 0, 1, 0)
        d2.wpk = w2.data_ptr()
        rows2 = conv_stats_rows(d2, dtype)
        if training:
            self.need(sS, rows2 * 2 * Npad * 4)
            self.patch(d2, "stats", sS)
        self._conv(self.fwd, d2)
        fused = training and self.bn_fused and bool(self.lib.ksmi_bn_fused_supported(Cc, Npad, dt))
        pooled = None
        if fused:
            # statistics finish + BN2-apply + residual + ReLU (+ the 2x2 max-pool of the encoder blocks) in ONE launch
            if pool is not None:
                pooled = Act(pool, B, H // 2, W // 2, Cc, dtype, self.dev)
            nbt2 = m._c(f"{name}.bn2.num_batches_tracked").data_ptr()
            pp = pooled.t.data_ptr() if pooled is not None else None
            self.fwd.add("ksmi_bn_fin_add_relu", lambda: (stats(), rows2, Npad, Cc, float(npix), P("bn2.weight"), P("bn2.bias"),
                                                          Bf("bn2.running_mean"), Bf("bn2.running_var"), nbt2, BN_MOMENTUM, BN_EPS,
                                                          sv2.mean, sv2.rstd, sv2.scale, sv2.shift, z_act.t.data_ptr(), i_act.t.data_ptr(),
                                                          out.t.data_ptr(), pp, B, H, W, dt),
                         {"kind": "bn_fin_add_relu", "bytes": npix * Cc * self._es() * (3.25 if pooled is not None else 3), "flops": 0})
        else:
            bn_fin("bn2", sv2, rows2, Npad)
            self.fwd.add("ksmi_bn_add_relu", lambda: (z_act.t.data_ptr(), i_act.t.data_ptr(), sv2.scale, sv2.shift,
                                                      out.t.data_ptr(), npix, Cc, dt))

        # ---- backward ----------------------------------------------------------------------------
        def build_bwd():
            self._stage(H)
            # (the virtual-sum input gradient below is the last writer of d out: the other producers -- pool / Up input gradients, the
            # ECAM head -- belong to modules later in the forward order, i.e. earlier in this list)
            gated = self._emit_dgrad(out, gate=(out.t, z_act.t, sv2)) if os.environ.get("KSMI_NO_GATE") is None else self._emit_dgrad(out)
            gated = gated if isinstance(gated, tuple) else None
            rows = self._rows(npix)
            self.need(sR, rows * 2 * Cc * 4)
            sums1 = torch.zeros((2, Cc), dtype=torch.float32, device=self.dev)
            sums2 = torch.zeros((2, Cc), dtype=torch.float32, device=self.dev)
            dz = torch.empty_like(z_act.t)     # grad wrt conv2 output
            r = torch.empty_like(i_act.t)      # relu-masked dgrad of conv2, then di (grad wrt conv1 output)
            self.keep += [sums1, sums2, dz, r]
            gout = out.grad().data_ptr()
            s1p, s2p = sums1.data_ptr(), sums2.data_ptr()
            a_bn2 = self._acc_param(f"{name}.bn2")
            es = self._es()
            if gated is not None:
                # d out arrives gated (x (out > 0)) with the BatchNorm2-backward sums in the statistics rows of its producer
                gst, grows, gpad = gated
                if fused and self.lib.ksmi_bn_fused_supported(Cc, gpad, dt):
                    self.bwd.add("ksmi_bn_bwd_fin_apply_gated", lambda: (gst.data_ptr(), grows, gpad, s2p, G("bn2.weight"), G("bn2.bias"), a_bn2,
                                                                         gout, z_act.t.data_ptr(), sv2.mean, sv2.rstd, P("bn2.weight"),
                                                                         dz.data_ptr(), float(npix), npix, Cc, dt),
                                 {"kind": "bn_bwd_fin_apply_gated", "bytes": npix * Cc * es * 3, "flops": 0})
                    self._mark(f"{name}.bn2.weight", f"{name}.bn2.bias")
                else:
                    self.bwd.add("ksmi_reduce_rows", lambda: (gst.data_ptr(), grows, 2, gpad, Cc, s2p, G("bn2.weight"), G("bn2.bias"), a_bn2))
                    self._mark(f"{name}.bn2.weight", f"{name}.bn2.bias")
                    if self.sync_bn:          # (the parameter gradients above stay LOCAL sums: the gradient all-reduce adds them; d x needs the global ones)
                        self.bwd.add_allreduce(lambda: sums2)
                    self.bwd.add("ksmi_bn_bwd_apply_gated", lambda: (gout, z_act.t.data_ptr(), sv2.mean, sv2.rstd, P("bn2.weight"), s2p,
                                                                     dz.data_ptr(), gcount, npix, Cc, dt))
            else:
                self.bwd.add("ksmi_bnrelu_bwd_reduce", lambda: (gout, out.t.data_ptr(), z_act.t.data_ptr(), sv2.mean, sv2.rstd,
                                                                self.scr(sR), rows, npix, Cc, dt))
                if fused:
                    self.bwd.add("ksmi_bnrelu_bwd_fin_apply", lambda: (self.scr(sR), rows, Cc, s2p, G("bn2.weight"), G("bn2.bias"), a_bn2,
                                                                       gout, out.t.data_ptr(), z_act.t.data_ptr(), sv2.mean, sv2.rstd,
                                                                       P("bn2.weight"), dz.data_ptr(), float(npix), npix, Cc, dt),
                                 {"kind": "bnrelu_bwd_fin_apply", "bytes": npix * Cc * es * 5, "flops": 0})
                    self._mark(f"{name}.bn2.weight", f"{name}.bn2.bias")
                else:
                    self.bwd.add("ksmi_reduce_rows", lambda: (self.scr(sR), rows, 2, Cc, Cc, s2p, G("bn2.weight"), G("bn2.bias"), a_bn2))
                    self._mark(f"{name}.bn2.weight", f"{name}.bn2.bias")
                    if self.sync_bn:
                        self.bwd.add_allreduce(lambda: sums2)
                    self.bwd.add("ksmi_bnrelu_bwd_apply", lambda: (gout, out.t.data_ptr(), z_act.t.data_ptr(), sv2.mean, sv2.rstd,
                                                                   P("bn2.weight"), s2p, dz.data_ptr(), gcount, npix, Cc, dt))
            # conv2.bias feeds a train-mode BatchNorm: its gradient sum(dz) is analytically 0 (the reference holds
            # ~1e-6 of rounding noise there); write exact zeros instead of two reduction launches.
            if training:
                self._defer_rowsum(f"{name}.conv2.bias", None, 0, 1, 0, Cc, Cc)
            else:                                                   # eval-mode BN: d bias = sum(dz)
                pz = torch.empty(rows * Cc, dtype=torch.float32, device=self.dev)
                self.bwd.add("ksmi_channel_sum", lambda: (dz.data_ptr(), pz.data_ptr(), rows, npix, Cc, dt))
                self._defer_rowsum(f"{name}.conv2.bias", pz, rows, 1, 0, Cc, Cc)
            # dgrad of conv2 with fused ReLU mask + BN1-backward statistics in the epilogue
            dg2, tg2 = make_conv([SrcSpec(dz, Cc)], [(r, Cc, 0, 0, Cc, 0)], dz, None, None, B, H, W, H, W, 3, 3, 1, 1, Cc, dtype,
                                 mask=(i_act.t, sv1.t[0], sv1.t[1], sv1.t[2], sv1.t[3]))
            wg2 = self._packed(f"{name}.conv2.weight", tg2, 9, Cc, Cc, Cc * 9, 9, 0, 1, 1)
            dg2.wpk = wg2.data_ptr()
            rows_g = conv_stats_rows(dg2, dtype)
            self.need(sS, rows_g * 2 * Npad * 4)
            self.patch(dg2, "stats", sS)
            self._conv(self.bwd, dg2, "dgrad")
            a_bn1 = self._acc_param(f"{name}.bn1")
            if not fused:
                self.bwd.add("ksmi_reduce_rows", lambda: (self.scr(sS), rows_g, 2, Npad, Cc, s1p, G("bn1.weight"), G("bn1.bias"), a_bn1))
                self._mark(f"{name}.bn1.weight", f"{name}.bn1.bias")
                if self.sync_bn:
                    self.bwd.add_allreduce(lambda: sums1)
            # weight gradient of conv2: X = relu(bn1(i)) recomputed on load, dY = dz
            dw2, ws2 = make_wgrad(src2, dz, Cc, 0, Cc, m._g(f"{name}.conv2.weight"), 9, Cc * 9, 1,
                                  self._acc_param(f"{name}.conv2.weight"), B, H, W, H, W, 3, 3, 1, 1, dtype)
            self._wgrad(dw2, ws2, f"{name}.conv2.weight")
            # di = g + BN1 backward (over r, in place); conv1 bias gradient
            pb1 = torch.empty(rows * Cc, dtype=torch.float32, device=self.dev)      # per-block partial sums of di (conv1 bias gradient)
            if fused:
                # the BN1-backward sums are finished from the statistics rows of conv2's input gradient inside the apply pass.  (The
                # weight gradient of conv2 above runs on the side stream and does not touch `stats`; nothing else of this lane writes
                # the rows between the two launches.)
                rows_b = min(rows, self.lib.ksmi_bn_fused_max_rows())
                self.bwd.add("ksmi_bn_bwd_fin_apply_add", lambda: (self.scr(sS), rows_g, Npad, s1p, G("bn1.weight"), G("bn1.bias"), a_bn1,
                                                                   r.data_ptr(), gout, i_act.t.data_ptr(), sv1.mean, sv1.rstd, P("bn1.weight"),
                                                                   pb1.data_ptr(), rows_b, float(npix), npix, Cc, dt),
                             {"kind": "bn_bwd_fin_apply_add", "bytes": npix * Cc * self._es() * 4, "flops": 0})
                self._mark(f"{name}.bn1.weight", f"{name}.bn1.bias")
                self._defer_rowsum(f"{name}.conv1.bias", pb1, rows_b, 1, 0, Cc, Cc)
            else:
                self.bwd.add("ksmi_bn_bwd_apply_add", lambda: (r.data_ptr(), gout, i_act.t.data_ptr(), sv1.mean, sv1.rstd,
                                                               P("bn1.weight"), s1p, pb1.data_ptr(), rows, gcount, npix, Cc, dt))
                self._defer_rowsum(f"{name}.conv1.bias", pb1, rows, 1, 0, Cc, Cc)
            a_w1 = self._acc_param(f"{name}.conv1.weight")
            if first:
                # dW[n][c*9+t] = sum_px im2col[px][c*9+t] * di[px][n]  (1x1 weight-gradient GEMM over the saved im2col)
                if self.im2col_late:
                    self.bwd.add("ksmi_im2col3x3_raw", im2col_args, {"kind": "im2col3x3_raw", "bytes": B * H * W * (cin * 4 + Kpad * self._es()), "flops": 0})
                dw1, ws1 = make_wgrad(src1, r, Cc, 0, Cc, m._g(f"{name}.conv1.weight"), 1, cin * 9, 0, a_w1,
                                      B, H, W, H, W, 1, 1, 1, 0, dtype)
                for ci in range(dw1.nchunks):
                    dw1.k_len[ci] = max(0, min(dw1.k_len[ci], cin * 9 - dw1.k_off[ci]))
                self._wgrad(dw1, ws1, f"{name}.conv1.weight")
            else:
                srcs = [SrcSpec(a.t, a.C) for a in sources]
                # input gradients are NOT launched here: every source registers this block's `di` as one of
                # its consumers and its producer gathers all of them in ONE "virtual sum" dgrad (_emit_dgrad)
                nb = 0
                for a in sources:
                    a.consumers.append((r, Cc, f"{name}.conv1.weight", nb, Ktot))
                    nb += a.C
                dw1, ws1 = make_wgrad(srcs, r, Cc, 0, Cc, m._g(f"{name}.conv1.weight"), 9, Ktot * 9, 1, a_w1,
                                      B, H, W, H, W, 3, 3, 1, 1, dtype)
                self._wgrad(dw1, ws1, f"{name}.conv1.weight")
        self.bwd_builders.append((self.fwd.cur_lane, build_bwd))
        if pool is not None:
            return out, pooled           # pooled = None: the caller's _pool_at() launches the stand-alone max-pool
        return out

    # ---------------------------------------------------------------- ECAM head  (snunet.py:49-62,146-151)
    def _head(self, xs):
        m, B, n = self.m, self.B, self.n
        HW = self.H * self.W
        dev, dt = self.dev, self.dt
        self.fwd.cur_stage = self.bwd.cur_stage = "head"
        f32 = dict(dtype=torch.float32, device=dev)
        avg, mx = torch.zeros((B, 5 * n), **f32), torch.zeros((B, 5 * n), **f32)
        argmax = torch.zeros((B, 5 * n), dtype=torch.int32, device=dev)
        ca, ca1 = torch.zeros((B, 4 * n), **f32), torch.zeros((B, n), **f32)
        hidden = torch.zeros((B, 2, 4 * n // 16 + n // 4), **f32)
        xarr = (C.c_void_p * 4)(*[a.t.data_ptr() for a in xs])
        ws_pool = torch.empty(self.lib.ksmi_ecam_pool_workspace(B, HW, n), dtype=torch.uint8, device=dev)
        self.keep += [avg, mx, argmax, ca, ca1, hidden, xarr, ws_pool]
        self.head = dict(avg=avg, mx=mx, argmax=argmax, ca=ca, ca1=ca1)
        P = lambda key: m._p(key).data_ptr()
        G = lambda key: m._g(key).data_ptr()
        T0 = B * HW * n * self._es()             # one [B,H,W,n] map
        self.fwd.add("ksmi_ecam_pool", lambda: (xarr, avg.data_ptr(), mx.data_ptr(), argmax.data_ptr(), ws_pool.data_ptr(), B, HW, n, dt),
                     {"kind": "ecam_pool", "bytes": 4 * T0, "flops": 0})
        self.fwd.add("ksmi_ecam_mlp", lambda: (avg.data_ptr(), mx.data_ptr(), P("ca.fc1.weight"), P("ca.fc2.weight"),
                                               P("ca1.fc1.weight"), P("ca1.fc2.weight"), ca.data_ptr(), ca1.data_ptr(),
                                               hidden.data_ptr(), B, n))
        self.fwd.add("ksmi_ecam_final_forward", lambda: (xarr, ca.data_ptr(), ca1.data_ptr(), P("conv_final.weight"),
                                                         P("conv_final.bias"), self.logits.data_ptr(), B, HW, n, 3, dt),
                     {"kind": "ecam_final_forward", "bytes": 4 * T0 + B * HW * 3 * 4, "flops": 2 * B * HW * 4 * n * 3})

        def build_bwd():
            self.fwd.cur_stage = self.bwd.cur_stage = "head"
            dca, dca1 = torch.zeros((B, 4 * n), **f32), torch.zeros((B, n), **f32)
            davg, dmax = torch.zeros((B, 5 * n), **f32), torch.zeros((B, 5 * n), **f32)
            ws_b = torch.empty(self.lib.ksmi_ecam_bwd_workspace(B, HW, n, 3), dtype=torch.uint8, device=dev)
            ws_m = torch.empty(self.lib.ksmi_ecam_mlp_bwd_workspace(B, n), dtype=torch.uint8, device=dev)
            garr = (C.c_void_p * 4)(*[a.grad().data_ptr() for a in xs])
            for a in xs:
                a.take_acc_flag()
            for key in ("conv_final.weight", "conv_final.bias", "ca.fc1.weight", "ca.fc2.weight", "ca1.fc1.weight", "ca1.fc2.weight"):
                self._acc_param(key)
            self.keep += [dca, dca1, davg, dmax, ws_b, ws_m, garr]
            dl = self.dlogits.data_ptr()
            self.bwd.add("ksmi_ecam_final_backward_reduce", lambda: (
                xarr, dl, ca.data_ptr(), ca1.data_ptr(), P("conv_final.weight"), dca.data_ptr(), dca1.data_ptr(),
                G("conv_final.weight"), G("conv_final.bias"), ws_b.data_ptr(), B, HW, n, 3, dt),
                {"kind": "ecam_final_backward_reduce", "bytes": 4 * T0 + B * HW * 3 * 4, "flops": 2 * B * HW * 4 * n * 3})
            self._mark("conv_final.weight", "conv_final.bias")
            self.bwd.add("ksmi_ecam_mlp_backward", lambda: (
                avg.data_ptr(), mx.data_ptr(), hidden.data_ptr(), ca.data_ptr(), ca1.data_ptr(), dca.data_ptr(), dca1.data_ptr(),
                P("ca.fc1.weight"), P("ca.fc2.weight"), P("ca1.fc1.weight"), P("ca1.fc2.weight"), davg.data_ptr(), dmax.data_ptr(),
                G("ca.fc1.weight"), G("ca.fc2.weight"), G("ca1.fc1.weight"), G("ca1.fc2.weight"), ws_m.data_ptr(), B, n))
            self._mark("ca.fc1.weight", "ca.fc2.weight", "ca1.fc1.weight", "ca1.fc2.weight")
            self.bwd.add("ksmi_ecam_final_backward_dx", lambda: (
                garr, dl, ca.data_ptr(), P("conv_final.weight"), davg.data_ptr(), dmax.data_ptr(), argmax.data_ptr(), B, HW, n, 3, dt),
                {"kind": "ecam_final_backward_dx", "bytes": 4 * T0 + B * HW * 3 * 4, "flops": 2 * B * HW * 4 * n * 3})
        self.bwd_builders.append((self.fwd.cur_lane, build_bwd))

    # ---------------------------------------------------------------- execution
    def run_forward(self, xA, xB, tail=None):
        if xA.data_ptr() != self.xA.data_ptr():
            self.xA.copy_(xA)
        if xB.data_ptr() != self.xB.data_ptr():
            self.xB.copy_(xB)
        if (tail is None) != (self.xtail is None):
            raise _lib.KsmiError("plan and call disagree about the shared tail channels (DEM)")
        if tail is not None and tail.data_ptr() != self.xtail.data_ptr():
            self.xtail.copy_(tail)
        self.packs.run()
        self.fwd.run()
        return self.logits

    def run_backward(self, dlogits=None):
        if not self.with_backward:
            raise _lib.KsmiError("plan was built without backward")
        if dlogits is not None and dlogits.data_ptr() != self.dlogits.data_ptr():
            self.dlogits.copy_(dlogits)
        self.bwd.run()
