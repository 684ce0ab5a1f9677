"""Runtime contexts (csrc/runtime.hip, include/rte_hip_ext.h): two host threads on two contexts -- own stream, arena, plan
caches -- run the LW and the SW chain CONCURRENTLY and reproduce the single-context results bit for bit; the error channel
(sticky mode) records a failing HIP call instead of aborting and makes the context a no-op until cleared.
The reference intends concurrent calls on distinct buffers (examples/all-sky/rrtmgp_allsky.F90:331)."""
import ctypes
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rte_rrtmgp_amd import frontend, hiplib, synth  # noqa: E402

pytestmark = pytest.mark.gpu
NLAY = 60


def _ctx_api(hip):
    create, setc, destroy = hip.raw("rte_hip_ctx_create"), hip.raw("rte_hip_ctx_set_current"), hip.raw("rte_hip_ctx_destroy")
    create.restype = ctypes.c_void_p
    create.argtypes = [ctypes.c_int, ctypes.c_void_p]
    setc.restype = ctypes.c_void_p
    setc.argtypes = [ctypes.c_void_p]
    destroy.argtypes = [ctypes.c_void_p]
    return create, setc, destroy


def _chain(hip, xp, kind, kd, atm, ncol):
    A = xp.asarray
    go = frontend.GasOptics(hip, kd, xp)
    if kind == "lw":
        b = go.gas_optics_lw(ncol, NLAY, A(atm.play), A(atm.plev), A(atm.tlay), A(atm.tsfc), A(atm.col_gas), A(atm.tlev), atm.top_at_1)
        r = frontend.rte_lw(hip, xp, ncol, NLAY, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"], xp.full((ncol, kd.ngpt), 0.98), b["sfc_src"])
        keys = ("flux_up", "flux_dn")
    else:
        b = go.gas_optics_sw(ncol, NLAY, A(atm.play), A(atm.plev), A(atm.tlay), A(atm.col_gas), A(atm.col_dry))
        r = frontend.rte_sw(hip, xp, ncol, NLAY, kd.ngpt, atm.top_at_1, b["tau"], b["ssa"], b["g"], xp.full((ncol, NLAY), 0.86), b["toa_src"],
                            xp.full((ncol, kd.ngpt), 0.06), xp.full((ncol, kd.ngpt), 0.06))
        keys = ("flux_up", "flux_dn", "flux_dir")
    hiplib.ext_call(hip, "rte_hip_sync", [])
    out = {k: np.array(xp.to_numpy(r[k])) for k in keys}
    out["tau"] = np.array(xp.to_numpy(b["tau"])[::7])
    return out


def test_two_threads_on_two_contexts_reproduce_the_serial_results():
    import torch

    hip = hiplib.load()
    create, setc, destroy = _ctx_api(hip)
    xp = frontend.TorchArrays("cuda:0")
    ncol = 4096  # production kernels
    cases = {}
    for kind in ("lw", "sw"):
        kd = synth.make_kdist(kind)
        cases[kind] = (kd, synth.make_atmosphere(ncol, NLAY, seed=5 if kind == "lw" else 6, kdist=kd))
    serial = {kind: _chain(hip, xp, kind, *cases[kind], ncol) for kind in cases}
    torch.cuda.synchronize()
    results, errors = {}, []

    def work(kind, reps=3):
        try:
            ctx = create(-1, None)  # own non-blocking stream
            assert ctx
            setc(ctx)
            torch.cuda.set_device(0)
            try:
                for _ in range(reps):
                    results[kind] = _chain(hip, xp, kind, *cases[kind], ncol)
            finally:
                setc(None)
                assert destroy(ctx) == 0
        except Exception as e:  # noqa: BLE001
            errors.append((kind, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in cases]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for kind in cases:
        for k, v in serial[kind].items():
            assert np.array_equal(results[kind][k], v), (kind, k, float(np.max(np.abs(results[kind][k] - v))))
    torch.cuda.synchronize()


def test_sticky_error_mode_records_instead_of_aborting():
    hip = hiplib.load()
    create, setc, destroy = _ctx_api(hip)
    ctx = create(-1, None)
    setc(ctx)
    try:
        hiplib.ext_call(hip, "rte_hip_error_mode", ["i"], 1)
        last = hip.raw("rte_hip_last_error")
        buf = ctypes.create_string_buffer(600)
        assert last(buf, 600) == 0
        # an 8 TB staging request (host arrays of an absurd declared size): hipMalloc fails before anything is read
        small = np.zeros(16)
        xp = frontend.NumpyArrays()
        hip.rte_sum_broadband(1 << 20, 1 << 10, 1 << 10, small, small)
        code = last(buf, 600)
        assert code != 0 and b"rte_sum_broadband" in buf.value, (code, buf.value)
        # the context is a no-op until the error is cleared ...
        out = np.full((4, 3), 7.0, order="F")
        hip.zero_array_2D(4, 3, out)
        assert np.all(out == 7.0)
        # ... and works again afterwards
        hiplib.ext_call(hip, "rte_hip_clear_error", [])
        assert last(buf, 600) == 0
        hip.zero_array_2D(4, 3, out)
        assert np.all(out == 0.0)
        del xp
    finally:
        hiplib.ext_call(hip, "rte_hip_error_mode", ["i"], 0)
        setc(None)
        destroy(ctx)
