#!/usr/bin/env python3
"""Headline benchmark: SAR tiles/sec of one SNUNet-ECAM change-detection TRAIN STEP
(forward + CE+Dice loss + backward + Adam [+ gradient all-reduce]) on 224x224 tiles.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): SNUNet-ECAM, 2 dates x 2-ch GRD (VV,VH) 224x224, per-GPU
batch 32, bf16 activations / fp32 parameters+accumulation, synthetic tiles, random-init weights.
Weak scaling: every rank processes its own 32 tiles per step; value = total tiles / max-over-ranks time.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel class,
HIP-event timed inside the timed region) and `cpu_baseline` (CPU oracle port on host cores).
"""
import argparse
import json
import os
import sys
import time

if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("KSMI_DP_FORCE"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "7")     # five streams under data parallelism: kurosiwo_amd/distributed.py set_hw_queues (before the HIP runtime starts)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0    # ... and the copy rate it measures on the part (6.29 TB/s, 79 %): roofline.frac_of_measured_hbm
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16


IN_STEP_STEPS = 4    # multi-stream steps AFTER the timed region whose dominant-class launches carry HIP event pairs (roofline.in_step);
                     # the timed region itself carries one event per step boundary only (ms_per_step_p50)


class KernelTimer:
    """HIP events around selected launches on the stream they are launched on (torch's current
    stream == the stream handed to the C-ABI)."""

    def __init__(self, kinds=None, names=False):
        self.kinds = kinds          # None = every launch
        self.rec = []               # (kind, meta, ev0, ev1)
        self._cur = None
        self.active = True          # False during the timed region: no launch of it is bracketed (round 5; the pairs cost ~1.7 % of a step)
        self.steps_on = 0
        self.names = names          # also ask the library which kernel instantiation(s) each launch started (ksmi_last_kernels)
        self.kernels = []           # parallel to rec: tuple of kernel names ('' for elementwise launches)
        if names:
            import ctypes
            from kurosiwo_amd import _lib
            self._lib, self._buf = _lib.load(), ctypes.create_string_buffer(2048)

    def wants(self, kind):
        return self.active and (self.kinds is None or kind in self.kinds)

    def begin(self, kind, meta=None):
        if self.names:
            self._lib.ksmi_last_kernels(self._buf, 2048)          # (forget what earlier, untimed launches started)
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self._cur = (kind, meta or {"bytes": 0, "flops": 0}, e0)

    def end(self):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.rec.append(self._cur + (e1,))
        if self.names:
            n = self._lib.ksmi_last_kernels(self._buf, 2048)
            self.kernels.append(tuple(self._buf.value.decode().split(";")) if n else ())
        self._cur = None

    def summary(self):
        out = {}
        for kind, meta, e0, e1 in self.rec:
            d = out.setdefault(kind, {"ms": 0.0, "n": 0, "bytes": 0, "flops": 0})
            d["ms"] += e0.elapsed_time(e1)
            d["n"] += 1
            d["bytes"] += meta.get("bytes", 0)
            d["flops"] += meta.get("flops", 0)
        return out


# bench kernel class -> regular expressions over the rocprofv3 kernel names that make up one "launch" of it.  The table they are
# looked up in (profiles/r03_<model>_traffic.json) is written by profiles/summarize.py from the --pmc FETCH_SIZE / WRITE_SIZE passes of
# THIS command (tools/profile.sh) and is keyed by the full kernel name; the persistent kernels carry the direction of a launch in their
# name (ksmi_conv_desc.dir: igemm4_kernel<WM, NF, AFF, EPI, DBG, DIR, ROT>, igemm3_kernel<.., MASK, DIR>; AFF = forward with the fused BN
# operand, EPI 1 / 2 = the two input-gradient epilogues), so forward and input-gradient launches are separate rows.
IG4 = r"igemm4_kernel<\d, \d, "
TRAFFIC_KERNELS = {
    "igemm_fwd<3x3s1,BN32>": (IG4 + r"true, 0, false, 0, (true|false)>", IG4 + r"false, 0, false, 0, (true|false)>", r"igemm3_kernel<3, 3, \d, \d, true, 1, false, 0>",
                              r"igemm3_kernel<3, 3, \d, \d, false, 1, false, 0>"),
    "igemm_dgrad<3x3s1,BN32>": (IG4 + r"false, [12], false, 0, (true|false)>", IG4 + r"false, 0, false, 1, (true|false)>", r"igemm3_kernel<3, 3, \d, \d, false, 1, false, 1>",
                                r"igemm2_fwd_kernel<bf16, 2, 3, 3, false, false, 1, true, 1>"),      # (the lean tile kernel: K = 32 mask epilogue)
    "igemm_wgrad<3x3s1>": (r"wgrad3_kernel<", r"wgrad3_reduce_kernel", r"igemm_wgrad_kernel<bf16, 2, 3, 3", r"wgrad_reduce_kernel"),
    # ChangeFormer / BIT-CD: plain 3x3 convolutions of the decoder (forward: ReLU / residual epilogues included) and their input gradients
    "igemm_conv3x3<3x3s1>": (IG4 + r"(true|false), 0, false, 0, (true|false)>", r"igemm2_fwd_kernel<bf16, 2, 3, 3, (true|false), true"),
    "igemm_conv3x3_dgrad<3x3s1>": (IG4 + r"false, [12], false, 0, (true|false)>", IG4 + r"false, 0, false, 1, (true|false)>"),
    "igemm_wgrad<1x1s1>": (r"gemm2_tn_kernel", r"tn_reduce_kernel", r"igemm_wgrad_kernel<bf16, [24], 1, 1", r"wgrad_reduce_kernel"),
    "gemm_nt": (r"gemm2_kernel<\d+, false>", r"gemm_nt_kernel"),
    "gemm_nn": (r"gemm2_kernel<\d+, true>", r"gemm_nn_kernel"),
}
TRAFFIC_FILE = {"snunet": "snunet", "changeformer": "changeformer", "floodvit": "floodvit", "unet": "unet", "mae": "mae"}
TRAFFIC_ROUNDS = ("r06", "r05", "r04")     # newest committed table first


def measured_traffic(kind, model="snunet"):
    """HBM bytes per launch of the dominant kernel class by the PMC counters (None when no table / no mapping): call-weighted mean of
    2 x FETCH_SIZE + WRITE_SIZE over the kernels of the class (x2: the guide's gfx950 FETCH_SIZE correction), plus the rows it used."""
    import re
    path = next((q for q in (os.path.join(ROOT, "profiles", f"{r}_{TRAFFIC_FILE.get(model, model)}_traffic.json") for r in TRAFFIC_ROUNDS)
                 if os.path.exists(q)), None)
    if kind not in TRAFFIC_KERNELS or path is None:
        return None, []
    tab = json.load(open(path))["kernels"]
    pats = [re.compile(p) for p in TRAFFIC_KERNELS[kind]]
    tot, main_calls, rows = 0.0, 0, []
    for name, row in tab.items():
        if any(p.match(name) for p in pats) and row.get("fetch_kb_raw") is not None and row.get("write_kb_raw") is not None:
            b = (2.0 * row["fetch_kb_raw"] + row["write_kb_raw"]) * 1024.0
            tot += row["calls"] * b
            rows.append({"kernel": name, "calls": row["calls"], "bytes_per_launch": round(b), "avg_us": round(row["avg_us"], 1),
                         "mfma_busy_pct": None if row.get("mfma_busy_pct") is None else round(row["mfma_busy_pct"], 1)})
            if "reduce" not in name:
                main_calls += row["calls"]
    return (round(tot / main_calls) if main_calls else None), rows


def measure_hbm_peaks(dev, gib=1):
    """Device-memory rates of THIS part, measured here ("measure ... on the box and quote both"): ksmi_hbm_probe
    passes over 1 GiB operands (4 x the 256 MB memory-side cache), HIP events on the launching stream, best of 3 after a warm-up.
    GB/s count every byte moved (copy = 2 x n, triad = 3 x n)."""
    from kurosiwo_amd import _lib
    from kurosiwo_amd.runtime import stream_ptr
    lib = _lib.load()
    n = gib << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1)
    b, c = torch.empty_like(a), torch.empty_like(a)
    b.fill_(2)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    out = {}
    for name, kind, moved in (("read_dma", 0, n), ("read_vec", 1, n), ("copy", 2, 2 * n), ("triad", 3, 3 * n), ("fill", 4, n)):
        best, how = None, None
        variants = [kind] if kind == 0 else [kind | nt << 3 | g << 4 for nt in ((0, 1) if kind > 1 else (0,)) for g in range(4)]
        if kind > 1:
            variants += [kind | 64 | g << 4 for g in range(4)]
        for mode in variants:                      # (non-temporal or not, 8192 ... 1024 workgroups: the best variant is the part's rate)
            for it in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.ksmi_hbm_probe(mode, (c if kind == 4 else a).data_ptr(), b.data_ptr(), c.data_ptr(), n, sink.data_ptr(), stream_ptr()), "hbm_probe")
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1)
                if it and (best is None or ms < best):
                    best, how = ms, mode
        out[f"{name}_GBs"] = round(moved / best / 1e6, 1)
        if kind:
            out[f"{name}_variant"] = (f"contiguous chunks, {(8192 >> ((how >> 4) & 3)) * 16} workgroups" if how & 64 else
                                      f"{'nt ' if (how >> 3) & 1 else ''}{8192 >> ((how >> 4) & 3)} workgroups, grid-stride")
    # the same three operations by torch's own elementwise kernels (a second opinion on the part, not a kernel of this library)
    for name, fn, moved in (("copy", lambda: b.copy_(a), 2 * n), ("fill", lambda: c.fill_(3), n),
                            ("triad", lambda: torch.add(a.view(torch.float32), b.view(torch.float32), alpha=0.5, out=c.view(torch.float32)), 3 * n)):
        best = None
        for it in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1)
            if it and (best is None or ms < best):
                best = ms
        out[f"{name}_torch_GBs"] = round(moved / best / 1e6, 1)
    out["how"] = f"ksmi_hbm_probe over {gib} GiB operands, HIP events, best of 2 per variant (csrc/probe.hip)"
    del a, b, c
    # "... and a hipBLASLt bf16 GEMM on the box": the vendor library's dense bf16 rate through torch.matmul (hipBLASLt
    # behind it).  A second opinion on the PART next to the 2.5 PFLOP/s spec every mfma_frac is quoted against -- not a kernel of this
    # library, never part of `value`.
    try:
        best_tf, best_n = 0.0, 0
        for nn in (4096, 8192):
            ga = (torch.randn(nn, nn, device=dev) * 0.1).to(torch.bfloat16)
            gb = (torch.randn(nn, nn, device=dev) * 0.1).to(torch.bfloat16)
            gc = torch.empty(nn, nn, dtype=torch.bfloat16, device=dev)
            for it in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    torch.matmul(ga, gb, out=gc)
                e1.record(); e1.synchronize()
                tf = 4 * 2.0 * nn ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12
                if it and tf > best_tf:
                    best_tf, best_n = tf, nn
            del ga, gb, gc
        out["gemm_bf16_torch_TFs"] = round(best_tf, 1)
        out["gemm_bf16_how"] = f"torch.matmul (hipBLASLt) bf16 {best_n}^3, fp32 accumulate, best of 3 x 4 back-to-back calls, random operands"
    except Exception as e:                     # (a part / build without the library: the HBM rates stand alone)
        out["gemm_bf16_torch_TFs"] = None
        out["gemm_bf16_how"] = f"not measured: {type(e).__name__}"
    torch.cuda.empty_cache()
    return out


PROFILE_ROUNDS = ("r06", "r05", "r04", "r03")


def profile_table(model, solo=True):
    """per-kernel rows of the committed rocprofv3 passes (profiles/<round>_<model>[_solo]_traffic.json, written by profiles/summarize.py
    from tools/profile.sh): calls, average duration, FETCH_SIZE / WRITE_SIZE, MFMA-busy; {} when no table is committed"""
    for rnd in PROFILE_ROUNDS:
        for tag in (("_solo", "") if solo else ("",)):
            path = os.path.join(ROOT, "profiles", f"{rnd}_{TRAFFIC_FILE.get(model, model)}{tag}_traffic.json")
            if os.path.exists(path):
                return json.load(open(path))["kernels"], os.path.relpath(path, ROOT)
    return {}, None


def _row_bytes(row):
    if row.get("fetch_kb_raw") is None or row.get("write_kb_raw") is None:
        return None
    return (2.0 * row["fetch_kb_raw"] + row["write_kb_raw"]) * 1024.0        # (x2: the guide's gfx950 FETCH_SIZE correction)


def build_roofline(args, step, solo_timer, solo_steps, d, dominant, timer_steps, ms_step, step_bytes, step_flops, peaks, hbm_meas, two_streams):
    """The `roofline` object of the bench line.

    kernel / achieved / frac: ONE kernel instantiation -- the convolution / GEMM kernel with the largest total duration in the
    single-stream pass -- by its rocprofv3 name; achieved = sum of the algorithmic bytes (or flops) of its launches / sum of their HIP-event
    durations on one stream (launch metadata: the plans' meta["bytes"] / meta["flops"]); profile_avg_launch_ms = the same
    kernel's average in the committed rocprofv3 table.  stages[]: every launch of the step grouped by the plan's stage tag (SNUNet:
    resolution level).  in_step: the dominant kernel CLASS inside the timed (multi-stream) region, the figure earlier rounds reported."""
    avg_ms = d["ms"] / d["n"]
    cls_gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
    cls_tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
    in_step = {"kernel_class": dominant, "avg_launch_ms": round(avg_ms, 4), "launches_timed": d["n"],
               "algorithmic_GBs": round(cls_gbs, 1), "hbm_frac": round(cls_gbs / HBM_PEAK_GBS, 4), "tflops": round(cls_tf, 1),
               "mfma_frac": round(cls_tf / MFMA_BF16_PEAK_TF, 4), "share_of_step": round((d["ms"] / timer_steps) / ms_step, 3)}
    if two_streams:
        in_step["note"] = ("launches of the timed region share the machine with the other streams of the step (config.hip_streams): "
                           "durations are per launch, not per machine-second")
    # the step against the roofs.  TWO byte counts, labelled: step_conv_* = the convolution / linear launches alone (in + out activation
    # elements of every convolution, SURVEY.md §8(d)'s "algorithmic bytes": the figure the north-star's ">= 60 % of the HBM roofline on
    # the conv stages" is graded on); step_allpass_* = every launch of the step incl. the BatchNorm / pooling / elementwise / loss /
    # optimiser passes' own bytes (a larger, builder-defined count: reported, never the headline).
    calls = step.plan.fwd.calls + step.plan.bwd.calls
    conv_bytes = sum(c[3]["bytes"] for c in calls if c[3].get("kind", "").startswith(("igemm", "gemm", "up_gemm", "attention", "sr_attention", "token_cross", "bmm")))
    gemm_peak = (peaks or {}).get("gemm_bf16_torch_TFs")
    common = {"measured_peaks": peaks, "measured_hbm_peak_GBs": hbm_meas, "in_step": in_step,
              "step_conv_algorithmic_GB": round(conv_bytes / 1e9, 3),
              "step_conv_hbm_frac": round(conv_bytes / ms_step / 1e6 / HBM_PEAK_GBS, 4),
              "step_conv_frac_of_measured_hbm": round(conv_bytes / ms_step / 1e6 / hbm_meas, 4),
              "step_allpass_GB": round(step_bytes / 1e9, 3),
              "step_allpass_hbm_frac": round(step_bytes / ms_step / 1e6 / HBM_PEAK_GBS, 4),
              "step_allpass_frac_of_measured_hbm": round(step_bytes / ms_step / 1e6 / hbm_meas, 4),
              "step_GFLOP": round(step_flops / 1e9, 1),
              "step_mfma_frac": round(step_flops / ms_step / 1e9 / MFMA_BF16_PEAK_TF, 4),
              "step_frac_of_measured_gemm": None if not gemm_peak else round(step_flops / ms_step / 1e9 / gemm_peak, 4),
              "step_note": "step_conv_* counts the convolution / linear / attention launches only (SURVEY.md §8(d) algorithmic bytes, the graded "
                           "figure); step_allpass_* adds the BatchNorm / pooling / elementwise / loss / optimiser passes (rounds 1-4 printed it as step_hbm_frac)"}
    if solo_timer is None or not solo_timer.rec:
        hbm = not (d["flops"] / MFMA_BF16_PEAK_TF / 1e12 > d["bytes"] / HBM_PEAK_GBS / 1e9 and args.precision == "bf16")
        traffic, rows = measured_traffic(dominant, args.model)
        return ({"kernel": dominant, "bound": "hbm", "achieved": round(cls_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(cls_gbs / HBM_PEAK_GBS, 4)}
                if hbm else
                {"kernel": dominant, "bound": "mfma", "achieved": round(cls_tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(cls_tf / MFMA_BF16_PEAK_TF, 4)}
                ) | {"traffic": traffic, "traffic_rows": rows, "note": "no single-stream pass in this run: kernel = the dominant kernel class of the timed region"} | common
    table, table_path = profile_table(args.model)
    by_kernel, stages = {}, {}
    for (kind, meta, e0, e1), names in zip(solo_timer.rec, solo_timer.kernels):
        ms = e0.elapsed_time(e1)
        main = next((n for n in names if "reduce" not in n), None)
        row = table.get(main) if main else None
        if main:
            k = by_kernel.setdefault(main, {"ms": 0.0, "n": 0, "bytes": 0, "flops": 0})
            k["ms"] += ms; k["n"] += 1; k["bytes"] += meta.get("bytes", 0); k["flops"] += meta.get("flops", 0)
        st = stages.setdefault(meta.get("stage") or "other", {"ms": 0.0, "n": 0, "conv_bytes": 0, "bytes": 0, "flops": 0, "counted": 0.0, "counted_n": 0,
                                                               "busy_ms": 0.0, "busy_w": 0.0, "named_ms": 0.0})
        st["ms"] += ms; st["n"] += 1; st["bytes"] += meta.get("bytes", 0); st["flops"] += meta.get("flops", 0)
        if main:
            st["conv_bytes"] += meta.get("bytes", 0)
            st["named_ms"] += ms
            if row is not None:
                rb = sum(filter(None, (_row_bytes(table[n]) for n in names if n in table)))
                if rb:
                    st["counted"] += rb; st["counted_n"] += 1
                if row.get("mfma_busy_pct") is not None:
                    st["busy_ms"] += ms; st["busy_w"] += ms * row["mfma_busy_pct"]
    if os.environ.get("BENCH_LAUNCH_MAP"):         # per-launch table of the LAST single-stream step (profiles/summarize.py aligns traces with it)
        per = len(solo_timer.rec) // max(solo_steps, 1)
        rows_ = [{"i": i, "kind": kind, "stage": meta.get("stage"), "tag": meta.get("tag", ""), "lane": meta.get("lane", 0), "side": bool(meta.get("side")),
                  "kernels": list(names), "bytes": meta.get("bytes", 0),
                  "flops": meta.get("flops", 0), "ms": round(e0.elapsed_time(e1), 5)}
                 for i, ((kind, meta, e0, e1), names) in enumerate(zip(solo_timer.rec[-per:], solo_timer.kernels[-per:]))]
        json.dump(rows_, open(os.environ["BENCH_LAUNCH_MAP"], "w"))
    # ---- the dominant kernel FAMILY (round 6): template instantiations merged -- `wgrad3_kernel<4, 1, 4, ...>` and
    # `wgrad3_kernel<4, 1, 2, ...>` are one kernel with different tile parameters; split ~50 ways the largest single row covered 7 % of the
    # step and described nothing.  `frac` = algorithmic bytes (flops) of the family's launches / their HIP-event durations on one stream
    # (live, this run); `profile` = the same arithmetic with the committed rocprofv3 durations of the same kernels (so the figure
    # reproduces from profiles/: HIP-event brackets are ~10 % longer than the kernel's own duration on short launches).
    fam_of = lambda n: n.split("<")[0]
    by_family = {}
    for n, v in by_kernel.items():
        f = by_family.setdefault(fam_of(n), {"ms": 0.0, "n": 0, "bytes": 0, "flops": 0, "members": []})
        for q in ("ms", "n", "bytes", "flops"):
            f[q] += v[q]
        f["members"].append(n)
    solo_ms = sum(v["ms"] for v in stages.values())
    name = max(by_family, key=lambda n: by_family[n]["ms"])
    k = by_family[name]
    gbs, tf = k["bytes"] / (k["ms"] * 1e-3) / 1e9, k["flops"] / (k["ms"] * 1e-3) / 1e12
    hbm = not (k["flops"] / MFMA_BF16_PEAK_TF / 1e12 > k["bytes"] / HBM_PEAK_GBS / 1e9 and args.precision == "bf16")
    # the family's rows of the committed rocprofv3 table: call-weighted duration, traffic and MFMA-busy
    prow = None
    rows_f = [(n, table[n]) for n in table if fam_of(n) == name]
    if rows_f:
        calls = sum(r["calls"] for _, r in rows_f)
        tot_us = sum(r["calls"] * r["avg_us"] for _, r in rows_f)
        tb = [(_row_bytes(r), r["calls"]) for _, r in rows_f if _row_bytes(r) is not None]
        busy = [(r["mfma_busy_pct"], r["calls"] * r["avg_us"]) for _, r in rows_f if r.get("mfma_busy_pct") is not None]
        # this run's algorithmic work per launch at the table's average duration (the plan, hence bytes / flops per launch, is the same)
        p_ms = tot_us / calls / 1e3
        p_gbs, p_tf = k["bytes"] / k["n"] / (p_ms * 1e-3) / 1e9, k["flops"] / k["n"] / (p_ms * 1e-3) / 1e12
        prow = {"table": table_path, "instantiations": len(rows_f), "calls": calls, "avg_launch_ms": round(p_ms, 4),
                "traffic_bytes_per_launch": None if not tb else round(sum(b * c for b, c in tb) / sum(c for _, c in tb)),
                "mfma_busy_pct": None if not busy else round(sum(b * w for b, w in busy) / sum(w for _, w in busy), 1),
                "achieved": round(p_gbs if hbm else p_tf, 1), "frac": round((p_gbs / HBM_PEAK_GBS) if hbm else (p_tf / MFMA_BF16_PEAK_TF), 4),
                "note": "this run's algorithmic bytes / flops per launch at the committed table's call-weighted average duration of the family"}
    roof = ({"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
             "frac_of_measured_hbm": round(gbs / hbm_meas, 4)} if hbm else
            {"kernel": name, "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TF, 4),
             "frac_of_measured_gemm": None if not gemm_peak else round(tf / gemm_peak, 4)})
    roof |= {"traffic": None if prow is None else prow["traffic_bytes_per_launch"], "traffic_unit": "bytes/launch (call-weighted over the family's instantiations)",
             "kernel_is": "a kernel FAMILY: every template instantiation of this __global__ function (instances[])",
             "algorithmic_bytes_per_launch": round(k["bytes"] / k["n"]), "launches_timed": k["n"], "avg_launch_ms": round(k["ms"] / k["n"], 4),
             "algorithmic_GBs": round(gbs, 1), "tflops": round(tf, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "mfma_frac": round(tf / MFMA_BF16_PEAK_TF, 4),
             "share_of_solo_step": round(k["ms"] / solo_ms, 3),
             "how": f"single stream, {solo_steps} steps after the timed region, HIP events around every launch on the launching stream; kernel names from ksmi_last_kernels",
             "profile": prow,
             "instances": [{"kernel": n, "launches_per_step": by_kernel[n]["n"] // solo_steps, "avg_launch_ms": round(by_kernel[n]["ms"] / by_kernel[n]["n"], 4),
                            "ms_per_step": round(by_kernel[n]["ms"] / solo_steps, 3), "algorithmic_GBs": round(by_kernel[n]["bytes"] / by_kernel[n]["ms"] / 1e6, 1),
                            "tflops": round(by_kernel[n]["flops"] / by_kernel[n]["ms"] / 1e9, 1)}
                           for n in sorted(k["members"], key=lambda n: -by_kernel[n]["ms"])],
             "families": [{"kernel": n, "instantiations": len(v["members"]), "launches_per_step": v["n"] // solo_steps, "ms_per_step": round(v["ms"] / solo_steps, 3),
                           "share_of_solo_step": round(v["ms"] / solo_ms, 3), "algorithmic_GBs": round(v["bytes"] / v["ms"] / 1e6, 1),
                           "tflops": round(v["flops"] / v["ms"] / 1e9, 1), "hbm_frac": round(v["bytes"] / v["ms"] / 1e6 / HBM_PEAK_GBS, 4),
                           "mfma_frac": round(v["flops"] / v["ms"] / 1e9 / MFMA_BF16_PEAK_TF, 4)}
                          for n, v in sorted(by_family.items(), key=lambda kv: -kv[1]["ms"])[:8]]}
    # exact per-stage counters of one single-stream step (profiles/stage_traffic.py: per-dispatch PMC rows aligned with this plan's launches)
    stage_pmc, stage_pmc_path = {}, None
    for rnd in PROFILE_ROUNDS:
        sp = os.path.join(ROOT, "profiles", f"{rnd}_{TRAFFIC_FILE.get(args.model, args.model)}_stage_traffic.json")
        if os.path.exists(sp):
            stage_pmc, stage_pmc_path = json.load(open(sp))["stages"], os.path.relpath(sp, ROOT)
            break
    order = sorted(stages, key=lambda t: (not t.startswith("L"), t))
    roof["stages"] = [{"stage": t, "launches_per_step": stages[t]["n"] // solo_steps, "ms_per_step_solo": round(stages[t]["ms"] / solo_steps, 3),
                       "conv_algorithmic_GB": round(stages[t]["conv_bytes"] / solo_steps / 1e9, 3), "all_passes_GB": round(stages[t]["bytes"] / solo_steps / 1e9, 3),
                       "counted_conv_GB": None if not stages[t]["counted_n"] else round(stages[t]["counted"] / solo_steps / 1e9, 3),
                       "GBs": round(stages[t]["bytes"] / max(stages[t]["ms"], 1e-9) / 1e6, 1),
                       "hbm_frac": round(stages[t]["bytes"] / max(stages[t]["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
                       "frac_of_measured_hbm": round(stages[t]["bytes"] / max(stages[t]["ms"], 1e-9) / 1e6 / hbm_meas, 4),
                       "GFLOP": round(stages[t]["flops"] / solo_steps / 1e9, 1), "tflops": round(stages[t]["flops"] / max(stages[t]["ms"], 1e-9) / 1e9, 1),
                       "mfma_frac": round(stages[t]["flops"] / max(stages[t]["ms"], 1e-9) / 1e9 / MFMA_BF16_PEAK_TF, 4),
                       "mfma_busy_pct": None if not stages[t]["busy_ms"] else round(stages[t]["busy_w"] / stages[t]["busy_ms"], 1)}
                      | ({"counted_GB": stage_pmc[t]["counted_GB"], "counted_over_all_passes": round(stage_pmc[t]["counted_GB"] / (stages[t]["bytes"] / solo_steps / 1e9), 3) if stages[t]["bytes"] else None,
                          "mfma_busy_pct": stage_pmc[t]["mfma_busy_pct"]} if t in stage_pmc else {})
                      for t in order]
    if stage_pmc_path:
        roof["stages_counters"] = stage_pmc_path
    roof["stages_note"] = ("all_passes_GB = algorithmic bytes of EVERY launch of the stage (convolutions + BatchNorm / pooling / elementwise passes), "
                           "conv_algorithmic_GB = the convolution / GEMM launches alone (SURVEY.md §8(d) counts these); counted_conv_GB and mfma_busy_pct "
                           "use the per-kernel averages of the committed PMC table (a kernel instantiation that serves several stages contributes its "
                           "average to each), null without a table; with `stages_counters`, counted_GB (every pass of the stage) and mfma_busy_pct are "
                           "the per-dispatch counters of one single-stream step aligned with the launches (profiles/stage_traffic.py)")
    if "attention_block" in stages:            # north_star: ">= 40 % of the bf16 MFMA peak on the ChangeFormer attention block"
        a = stages["attention_block"]
        roof["attention_block"] = {"what": "Block.forward first half, forward + backward: norm1 -> q / sr-conv / norm / kv linears -> softmax(q k^T) v -> proj -> "
                                           "dropout / DropPath + residual (models/changeformer.py:148-208, 239-243), all 13 blocks, both dates",
                                   "launches_per_step": a["n"] // solo_steps, "GFLOP_per_step": round(a["flops"] / solo_steps / 1e9, 1),
                                   "ms_per_step_solo": round(a["ms"] / solo_steps, 3), "tflops": round(a["flops"] / a["ms"] / 1e9, 1),
                                   "mfma_frac": round(a["flops"] / a["ms"] / 1e9 / MFMA_BF16_PEAK_TF, 4), "target": 0.40}
    return roof | common


def cpu_baseline(budget_s=12.0):
    """The CPU oracle (port of the reference's train step, pinned to it by tests/golden) on the
    host cores: SNUNet-ECAM c=2 bc=32, bs=4, fp32, ce+dice, Adam; 1 warm-up + timed steps."""
    from oracle import snunet_ref as R
    from oracle.seeded import seeded_fill_
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    B = 4
    batch = make_batch(B, seed=424242)
    (xA, xB), mask = cd_inputs(batch)
    sd = seeded_fill_(R.new_state_dict(2, 3, 32))
    opt = R.AdamRef(sd, lr=1e-3)
    R.train_step(sd, opt, xA, xB, mask)          # warm-up
    # thread sweep (round 6): torch's default = every hardware thread of the host (128 on the MI355X boxes), which oversubscribes the
    # oracle's small convolutions -- round 5 reported 0.63 tiles/s on 128 threads where 8 threads of the build container give 1.5.  One
    # step per candidate, then the timed steps on the best; `cores` = the threads actually used.
    default_threads = torch.get_num_threads()
    sweep = {}
    for nt in sorted({t for t in (8, 16, 32, 64, default_threads) if t <= default_threads}):
        torch.set_num_threads(nt)
        t0 = time.time()
        R.train_step(sd, opt, xA, xB, mask)
        sweep[nt] = round(B / (time.time() - t0), 3)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t0, n = time.time(), 0
    while True:
        R.train_step(sd, opt, xA, xB, mask)
        n += 1
        if time.time() - t0 > budget_s or n >= 8:
            break
    dt = time.time() - t0
    torch.set_num_threads(default_threads)
    return {"value": round(B * n / dt, 3), "unit": "tiles/s", "cores": best, "kind": "port",
            "sample": f"{n} timed train steps of SNUNet-ECAM(2,3,32) bs={B} 224x224 fp32 ce+dice Adam (oracle/snunet_ref.py), {dt:.1f}s, on the best "
                      f"thread count of a one-step sweep",
            "thread_sweep_tiles_per_s": {str(k): v for k, v in sweep.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="snunet", choices=["snunet", "floodvit", "changeformer", "unet", "mae", "siam-conc", "siam-diff", "bit-cd"],
                    help="snunet = BASELINE.json configs[1] (the headline); changeformer = configs[3] (bs 32); "
                         "floodvit = configs[4] per-GPU shard (bs 16)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling); default 32 snunet/changeformer, 16 floodvit")
    ap.add_argument("--base-channel", type=int, default=32)
    ap.add_argument("--channels", type=int, default=2, help="bands per date: 2 = GRD (VV, VH), 4 = SLC (BASELINE.json configs[3] as written)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--net-g", default="base_resnet18", help="--model bit-cd: net_G of define_G (models/bit_cd.py:686-707)")
    ap.add_argument("--time-all", action="store_true", help="HIP-event time every kernel class (diagnostic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solo", action="store_true",
                    help="skip the single-stream steps after the timed region (roofline.solo); the rocprof passes use it so that a trace "
                         "holds the launches of ONE mode")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as ONE captured HIP graph in the timed region (single GPU); the per-kernel roofline figures "
                         "then come from eager steps run after it")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Libraries write banners to the C-level stdout (RCCL prints its version block when a
    # communicator is created / destroyed): send file descriptor 1 to stderr for the life of the process and keep a private copy
    # of the real stdout for the result line.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # KSMI_DIST_BACKEND=gloo: the same ranks with the collectives on the CPU backend (several ranks may then share one GPU): the
    # comparison run of tests/test_gpu_dp.py::test_two_gpus_* -- never the measured configuration
    backend = os.environ.get("KSMI_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1 or os.environ.get("KSMI_DP_FORCE"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cdev = torch.device("cpu") if backend == "gloo" else dev          # where the small bookkeeping collectives live

    from kurosiwo_amd.synthetic import cd_inputs, make_batch, seg_inputs

    B, H, W = args.batch or (16 if args.model == "floodvit" else 32), 224, 224
    torch.manual_seed(999)                      # same init on every rank (reference default seed, main.py:36)
    batch = make_batch(B, H, W, seed=999 + rank, channels=args.channels if args.model == "changeformer" else 2)
    if args.model == "snunet":
        from kurosiwo_amd.snunet import SNUNet_ECAM
        from kurosiwo_amd.trainer import CDTrainStep
        model = SNUNet_ECAM(2, 3, base_channel=args.base_channel, precision=args.precision).to(dev).train()
        step = CDTrainStep(model, B, H, W, loss_function="ce+dice", lr=1e-3, bucket_mb=8.0)
        (xA, xB), mask = cd_inputs(batch, ("pre_event_1", "post_event"))
        step.set_batch(xA.to(dev), xB.to(dev), mask.to(dev))          # inputs resident in HBM before the timed region
        workload = ("BASELINE.json configs[1]: SNUNet-ECAM CD, 2 dates x 2-ch GRD 224x224, "
                    f"per-GPU batch {B}, ce+dice loss, Adam lr 1e-3, fwd+loss+bwd+optimizer")
        metric = "SAR tiles/sec (224x224, SNUNet-ECAM change-detection train step)"
    elif args.model == "changeformer":
        from kurosiwo_amd.changeformer import ChangeFormerV6
        from kurosiwo_amd.optim import FusedSGD
        from kurosiwo_amd.trainer import CDTrainStep
        model = ChangeFormerV6(input_nc=args.channels, output_nc=3, decoder_softmax=True, embed_dim=256, precision=args.precision).to(dev).train()
        opt = FusedSGD(model.parameters(), lr=6e-4, momentum=0.99, weight_decay=1e-5)      # configs/method/changeformer/changeformer.json
        step = CDTrainStep(model, B, H, W, loss_function="ce+dice", optimizer=opt, bucket_mb=16.0)
        (xA, xB), mask = cd_inputs(batch, ("pre_event_1", "post_event"))
        step.set_batch(xA.to(dev), xB.to(dev), mask.to(dev))
        workload = (f"BASELINE.json configs[3]: ChangeFormerV6 CD (embed 256), 2 dates x {args.channels}-ch {'SLC' if args.channels == 4 else 'GRD'} 224x224, "
                    f"per-GPU batch {B}, ce+dice on the sigmoid map, SGD(0.99, wd 1e-5), fwd+loss+bwd+optimizer")
        metric = "SAR tiles/sec (224x224, ChangeFormerV6 change-detection train step)"
    elif args.model in ("siam-conc", "siam-diff", "bit-cd"):
        # the other siamese change-detection baselines of the reference with their shipped method configs
        from kurosiwo_amd.trainer import CDTrainStep
        if args.model == "bit-cd":
            from kurosiwo_amd.bitcd import define_G
            from kurosiwo_amd.optim import FusedSGD
            model = define_G({"net_G": args.net_g}, args.channels, precision=args.precision).to(dev).train()
            opt = FusedSGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=5e-4)       # configs/method/bit-cd/bit_cd.json
            desc = (f"BIT-CD (net_G {args.net_g}: " + ("siamese ResNet-18 + difference head" if args.net_g == "base_resnet18" else
                    "ResNet-18 to layer3 + semantic tokenizer + token encoder / decoder") + "), SGD(0.9, wd 5e-4)")
        else:
            from kurosiwo_amd.fcsiam import SiamUnet_conc, SiamUnet_diff
            from kurosiwo_amd.optim import FusedAdam
            model = (SiamUnet_conc if args.model == "siam-conc" else SiamUnet_diff)(args.channels, 3, precision=args.precision).to(dev).train()
            opt = FusedAdam(model.parameters(), lr=1e-5)                                         # configs/method/siam-conc/siam_conc.json
            desc = f"FC-{args.model} (Dropout2d 0.2 on), Adam 1e-5"
        step = CDTrainStep(model, B, H, W, loss_function="ce+dice", optimizer=opt, bucket_mb=8.0)
        (xA, xB), mask = cd_inputs(batch, ("pre_event_1", "post_event"))
        step.set_batch(xA.to(dev), xB.to(dev), mask.to(dev))
        workload = f"SURVEY.md §8(f) N2: {desc}, 2 dates x {args.channels}-ch 224x224, per-GPU batch {B}, ce+dice, fwd+loss+bwd+optimizer"
        metric = f"SAR tiles/sec (224x224, {args.model} change-detection train step)"
    elif args.model == "unet":
        from kurosiwo_amd.trainer import SegTrainStep
        from kurosiwo_amd.unet import Unet
        model = Unet("resnet18", encoder_weights=None, in_channels=2, classes=3, precision=args.precision).to(dev).train()
        step = SegTrainStep(model, B, loss_function="cross_entropy", lr=1e-3, bucket_mb=8.0, image_size=(H, W))
        x, mask = seg_inputs(batch, ("post_event",))
        step.set_batch(x.to(dev), mask.to(dev))
        workload = ("BASELINE.json configs[0] model on the GPU: Unet(resnet18) segmentation, 2-ch GRD 224x224, "
                    f"per-GPU batch {B}, CE, Adam, fwd+loss+bwd+optimizer")
        metric = "SAR tiles/sec (224x224, Unet-resnet18 segmentation train step)"
    elif args.model == "mae":
        from kurosiwo_amd.config import load_json5
        from kurosiwo_amd.mae import build_mae
        from kurosiwo_amd.trainer import MAETrainStep
        mc = load_json5(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs/method/mae/mae.json"))
        model = build_mae(mc, precision=args.precision, channels=2).to(dev).train()
        step = MAETrainStep(model, B, lr=mc["learning_rate"], bucket_mb=32.0)
        x, _ = seg_inputs(batch, ("post_event",))
        step.set_batch(x.to(dev))                                     # a fresh random permutation is drawn every step (mae.py:73)
        workload = ("SURVEY.md §8(f) N3: MAE pre-training (configs/method/mae/mae.json: ViT d1024 L24 h16 encoder on 25 % of 196 patches, "
                    f"decoder d512 L8 h16), 2-ch GRD 224x224, per-GPU batch {B}, MSE on the masked patches, Adam, fwd+loss+bwd+optimizer")
        metric = "SAR tiles/sec (224x224, MAE pre-training step)"
    else:
        from kurosiwo_amd.floodvit import FinetunerSegmentation, ViT
        from kurosiwo_amd.trainer import SegTrainStep
        enc = ViT(image_size=224, patch_size=16, num_classes=1000, dim=1024, depth=24, heads=16, mlp_dim=2048, channels=6)
        model = FinetunerSegmentation(enc, {"decoder": True, "num_classes": 3}, precision=args.precision).to(dev).train()
        step = SegTrainStep(model, B, loss_function="cross_entropy", lr=1e-4, bucket_mb=32.0)
        x, mask = seg_inputs(batch)                                   # [post, pre1, pre2] = 3 dates x 2 ch
        step.set_batch(x.to(dev), mask.to(dev))
        workload = ("BASELINE.json configs[4] per-GPU shard: FloodViT (ViT d1024 L24 h16 mlp2048, 6 ch) + Decoder head, "
                    f"224x224, per-GPU batch {B}, weighted CE, Adam, fwd+loss+bwd+optimizer")
        metric = "SAR tiles/sec (224x224, FloodViT segmentation train step)"

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (one of the warm-up steps also picks the dominant kernel class with every launch timed; not the very first
    # step when there are two or more: first launches carry one-off code-object loads) -------------
    if args.warmup >= 2:
        step.run()
    calib = KernelTimer(None)
    step.timer = calib
    step.run()
    torch.cuda.synchronize()
    cs = calib.summary()
    dominant = max((k for k in cs if k.startswith("igemm") or k.startswith("gemm")), key=lambda k: cs[k]["ms"])
    step.timer = None
    for _ in range(max(args.warmup - 2, 0)):
        step.run()
    # ---- timed region ----------------------------------------------------------------------------
    timer = KernelTimer(None if args.time_all else {dominant})
    graph = args.graph and world == 1 and hasattr(step, "capture_graph")
    if graph:
        step.capture_graph()
    else:
        step.timer = timer
    timer.active = bool(args.time_all)          # (--time-all is a diagnostic: its value is not the benchmark's)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    sync()
    t0 = time.perf_counter()
    issue_marks = [t0]
    for i in range(args.steps):
        marks[i].record()                       # one event per step boundary on the caller's stream (every step joins its streams there)
        step.run()
        issue_marks.append(time.perf_counter())
    marks[args.steps].record()
    t_issue = time.perf_counter() - t0            # host time to ENQUEUE the K steps (no wait inside): >= dt would mean a host-bound step
    # ... but once the runtime's queues are full the host is paced by the GPU, so the K-step average approaches the step time whatever the
    # host costs: the first steps after the sync (empty queues) are the host's own issue time
    issue_first = sorted(issue_marks[i + 1] - issue_marks[i] for i in range(min(3, args.steps)))[0]
    sync()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    p50_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    # roofline.in_step: the dominant kernel class inside the multi-stream step, bracketed in a few steps BEHIND the timed region
    timer.active = True
    timer_steps = args.steps if args.time_all else min(args.steps, IN_STEP_STEPS)
    if graph:           # HIP events cannot bracket the nodes of a replayed graph: eager steps
        step._graph = None
    if not args.time_all:
        step.timer = timer
        for _ in range(timer_steps):
            step.run()
        sync()
    # ---- the kernels alone on the machine ("solo"): with more than one stream (trainer.py overlap_wgrad / overlap_lanes) the launches
    # timed above shared the GPU with the other streams, so their durations describe the overlapped step, not the kernel.  A few more
    # steps on ONE stream with EVERY launch bracketed by HIP events (on the stream it is launched on) give per-kernel durations that
    # a rocprofv3 --kernel-trace of a single-stream run reproduces; the library reports which kernel instantiation each launch
    # started (ksmi_last_kernels), so every record carries the name it has in the rocprofv3 tables under profiles/.
    n_streams = (1 + int(bool(getattr(step, "overlap_wgrad", False) and getattr(step.plan, "side_wgrad", False)))
                 + int(bool(getattr(step, "overlap_lanes", False) and getattr(step.plan, "two_lanes", False))))
    two_streams = n_streams > 1
    solo_timer, solo_steps = None, 0
    if not args.no_solo and hasattr(step, "overlap_wgrad"):
        solo_timer = KernelTimer(None, names=True)
        keep = (step.overlap_wgrad, step.overlap_lanes, getattr(step, "_graph", None))
        step.timer, step.overlap_wgrad, step.overlap_lanes, step._graph = solo_timer, False, False, None
        solo_steps = min(args.steps, 3)
        for _ in range(solo_steps):
            step.run()
        sync()
        step.timer, step.overlap_wgrad, step.overlap_lanes, step._graph = (timer if not graph else None), keep[0], keep[1], keep[2]
    dp_check = None
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
        # data-parallel self-check: every rank started from the same parameters and applied the same averaged gradients, so the
        # parameter arenas must agree bit for bit (an integer checksum of the fp32 words) although every rank saw different tiles
        fp = step.model.flat_params
        mine = torch.stack([fp.view(torch.int32).to(torch.int64).sum(), torch.isfinite(fp).all().to(torch.int64),
                            step.loss_out.detach().double().mul(1e9).round().to(torch.int64)[0]]).to(cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        dp_check = {"ranks": world, "backend": backend, "params_equal": all(int(e[0]) == int(every[0][0]) for e in every),
                    "params_finite": all(int(e[1]) == 1 for e in every), "param_checksum": int(every[0][0]),
                    "rank_losses": [int(e[2]) / 1e9 for e in every],
                    "grad_wire": step.reducer.grad_dtype, "dp_mode": step.reducer.mode}
    loss = step.loss_out.cpu().tolist()

    if rank == 0:
        ts = timer.summary()
        d = ts[dominant]
        step_flops = sum(c[3]["flops"] for c in step.plan.fwd.calls + step.plan.bwd.calls)
        step_bytes = sum(c[3]["bytes"] for c in step.plan.fwd.calls + step.plan.bwd.calls)
        ms_step = dt / args.steps * 1e3
        peaks = measure_hbm_peaks(dev) if world == 1 and not args.no_solo else None      # (--no-solo: the rocprofv3 passes trace the step only)
        hbm_meas = max(peaks["copy_GBs"], peaks["triad_GBs"], peaks["copy_torch_GBs"], peaks["triad_torch_GBs"]) if peaks else HBM_MEASURED_GBS
        res = {
            "metric": metric,
            "value": round(B * world * args.steps / dt, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "ms_per_step_p50": round(p50_ms, 3), "ms_per_step_min_max": [round(step_ms[0], 3), round(step_ms[-1], 3)],
            "host_issue_ms_per_step": round(t_issue / args.steps * 1e3, 3),
            "host_issue_ms_first_steps": round(issue_first * 1e3, 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": workload + ("+RCCL all-reduce" if world > 1 else "") + (" [HIP graph replay]" if graph else ""),
                       "global_batch": B * world, "base_channel": args.base_channel, "parallelism": f"dp{world}",
                       "hip_streams": n_streams,
                       "timed_step": "forward + loss + backward + optimizer on a batch resident in HBM; the metric update of the reference's "
                                     "iteration (argmax -> confusion matrix, SURVEY.md §8 M1 / T1) is OUTSIDE the timed step, as §8(d) permits; "
                                     "ms_per_step = wall clock / steps (the value), ms_per_step_p50 = median of per-step HIP-event intervals",
                       "loss_last": [round(x, 5) for x in loss]} | ({"dp_check": dp_check} if dp_check else {}),
        }
        res["roofline"] = build_roofline(args, step, solo_timer, solo_steps, d, dominant, timer_steps, ms_step, step_bytes, step_flops,
                                         peaks, hbm_meas, two_streams)
        if args.time_all:
            tot = sum(v["ms"] for v in ts.values())
            res["kernels"] = {k: {"ms_per_step": round(v["ms"] / args.steps, 3), "launches": v["n"] // args.steps,
                                  "GBs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1),
                                  "TFs": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)}
                              for k, v in sorted(ts.items(), key=lambda kv: -kv[1]["ms"])}
            res["kernels_total_ms_per_step"] = round(tot / args.steps, 3)
        if os.environ.get("BENCH_DETAIL"):
            det = {}
            for kind, meta, e0, e1 in timer.rec:
                if os.environ["BENCH_DETAIL"] in kind:
                    dd = det.setdefault((kind, meta.get("tag", "")), [0.0, 0, meta])
                    dd[0] += e0.elapsed_time(e1); dd[1] += 1
            for (kind, tag), (ms, n, meta) in sorted(det.items(), key=lambda kv: -kv[1][0]):
                print(f"DETAIL {kind:28s} {tag:48s} {ms / timer_steps:7.3f} ms/step x{n // timer_steps} "
                      f"{meta['flops'] * n / ms / 1e9:7.1f} TF/s {meta['bytes'] * n / ms / 1e6:7.1f} GB/s", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline and args.model == "snunet":
            res["cpu_baseline"] = cpu_baseline()
        result_line = json.dumps(res)
    else:
        result_line = None
    if dist.is_initialized():
        dist.destroy_process_group()
    if result_line is not None:
        real_stdout.write(result_line + "\n")
        real_stdout.flush()


if __name__ == "__main__":
    main()
