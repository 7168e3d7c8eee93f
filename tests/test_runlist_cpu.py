"""Host side of the launch-list executor (csrc/runlist.hip, include/ksmi.h ksmi_run_list): the generated call thunks are in step with
the binding table, every entry point a plan may list has one, and the argument slots carry the values the thunks read.  The executor
itself is exercised end to end by the GPU tests (tests/test_gpu_graph.py: compiled list == Python walk, bit for bit); here a one-entry
list calls a host-only entry point through it (ksmi_tiles_fill_nodata takes no stream... so the CALL path is covered on the GPU only)."""
import ctypes as C
import os
import struct
import subprocess
import sys

from kurosiwo_amd import _lib, snunet_plan as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_thunks_are_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_thunks.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_listable_entry_point_has_a_thunk():
    lib = _lib.load()
    n = 0
    for name, (res, args) in _lib.SIGNATURES.items():
        if res is C.c_int and args and args[-1] is C.c_void_p and not name.startswith(("ksmi_runner_", "ksmi_run_list", "ksmi_thunk_")):
            codes = sp._sig_codes(args[:-1])
            assert lib.ksmi_thunk_id(codes.encode()) >= 0, (name, codes)
            n += 1
    assert n > 100
    assert lib.ksmi_thunk_id(b"no-such-signature") == -1


def test_argument_slots():
    d = _lib.ConvDesc()
    arr = (C.c_void_p * 4)(1, 2, 3, 4)
    assert sp._slot("p", None, struct) == 0
    assert sp._slot("p", 0x7F00DEADBEEF, struct) == 0x7F00DEADBEEF
    assert sp._slot("p", C.c_void_p(4096), struct) == 4096
    assert sp._slot("p", C.byref(d), struct) == C.addressof(d)
    assert sp._slot("p", arr, struct) == C.addressof(arr)
    assert sp._slot("i", -1, struct) == 0xFFFFFFFFFFFFFFFF              # (the thunk narrows through int64 -> int)
    assert sp._slot("l", -(1 << 40), struct) == (1 << 64) - (1 << 40)
    assert sp._slot("f", 1.5, struct) == 0x3FC00000
    assert sp._slot("d", -2.0, struct) == 0xC000000000000000
    assert sp._sig_codes([C.c_void_p, C.POINTER(_lib.ConvDesc), C.c_void_p * 4, C.c_int, C.c_int64, C.c_size_t, C.c_uint32, C.c_float, C.c_double]) == "pppilzufd"


def test_runner_lifecycle_and_argument_checks_without_a_gpu():
    lib = _lib.load()
    r = C.c_void_p(lib.ksmi_runner_create())
    assert r.value
    assert lib.ksmi_runner_set_streams(r, None, None, None, None) == 0
    ops = (_lib.Op * 1)()
    failed = C.c_int32(7)
    assert lib.ksmi_run_list(r, ops, 0, 0, None, C.byref(failed)) == 0 and failed.value == -1        # an empty segment touches nothing
    assert lib.ksmi_run_list(None, ops, 0, 0, None, None) != 0
    assert lib.ksmi_runner_destroy(r) == 0


def test_compile_resolves_calls_waits_and_tags():
    """LaunchList._compile on a list of real entry points (never run here): op kinds, lanes, side flags, tag ids, argument addresses"""
    ll = sp.LaunchList()
    d = _lib.ConvDesc()
    ll.add("ksmi_conv_forward", lambda: (C.byref(d), 1), {"kind": "conv", "bytes": 0, "flops": 0})
    ll.cur_lane = 1
    ll.add("ksmi_conv_wgrad", lambda: (C.byref(d), 1), {"kind": "wgrad", "bytes": 0, "flops": 0, "side": True, "side_tag": ("w", 3)})
    ll.add_wait(1, 0)
    ll.add_wait_side(("w", 3))
    ll.add_wait_side(None)
    ll.add("ksmi_gelu_forward", lambda: (4096, 8192, 1 << 33, 1), {"kind": "gelu", "bytes": 0, "flops": 0, "skip_if": lambda: True})
    ll.resolve(_lib.load())
    cp = ll._compile()
    ops = cp["ops"]
    assert cp["ok"] and cp["n"] == 6
    assert [ops[i].kind for i in range(6)] == [_lib.OP_CALL, _lib.OP_CALL, _lib.OP_ORDER, _lib.OP_WAIT_SIDE, _lib.OP_WAIT_SIDE, _lib.OP_CALL]
    assert (ops[0].lane, ops[0].side, ops[0].tag) == (0, 0, -1) and (ops[1].lane, ops[1].side) == (1, 1)
    assert ops[1].tag == ops[3].tag >= 0 and ops[4].tag == -1 and (ops[2].a, ops[2].b) == (1, 0)
    a0 = C.cast(ops[0].args, C.POINTER(C.c_uint64))
    assert a0[0] == C.addressof(d) and a0[1] == 1
    a5 = C.cast(ops[5].args, C.POINTER(C.c_uint64))
    assert [a5[i] for i in range(4)] == [4096, 8192, 1 << 33, 1] and ops[5].nargs == 4
    assert [i for i, _ in cp["skips"]] == [5]
