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
            torch.cuda.set_device(0)
            # one stream per thread for BOTH the library's kernels and torch's own (fills, copies made by the array
            # container): the frontend mirror assumes that the two are ordered, as bench.py arranges with rte_hip_set_stream
            st = torch.cuda.Stream()
            ctx = create(-1, ctypes.c_void_p(st.cuda_stream))
            assert ctx
            setc(ctx)
            try:
                with torch.cuda.stream(st):
                    for _ in range(reps):
                        results[kind] = _chain(hip, xp, kind, *cases[kind], ncol)
                    st.synchronize()
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


def test_host_mirror_mode_with_numpy_arrays_and_table_copies():
    """Host-mirror mode driven from Python with pageable numpy arrays (the arrays persist, unlike Fortran automatics): the LW
    chain gives the staged mode's arrays bit for bit, intermediates are readable after rte_hip_writeback, and the cached device
    copy of a host table follows a table that is replaced in place (fingerprint of the contents)."""
    hip = hiplib.load()
    create, setc, destroy = _ctx_api(hip)
    xp = frontend.NumpyArrays()
    kd = synth.make_kdist("lw", ngpt=64, nbnd=4)
    ncol = 600
    atm = synth.make_atmosphere(ncol, NLAY, seed=9, kdist=kd)
    stat = hip.raw("rte_hip_mirror_stat")
    stat.restype = ctypes.c_longlong
    wb = hip.raw("rte_hip_writeback")
    wb.argtypes = [ctypes.c_void_p]

    def chain(go):
        b = go.gas_optics_lw(ncol, NLAY, atm.play, atm.plev, atm.tlay, atm.tsfc, atm.col_gas, atm.tlev, atm.top_at_1)
        r = frontend.rte_lw(hip, xp, ncol, NLAY, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"], xp.full((ncol, kd.ngpt), 0.98), b["sfc_src"])
        return b, r

    ctx = create(-1, None)
    setc(ctx)
    try:
        go = frontend.GasOptics(hip, kd, xp)
        b0, r0 = chain(go)                       # staged
        up0, tau0 = r0["flux_up"].copy(), b0["tau"].copy()
        hiplib.ext_call(hip, "rte_hip_host_mirror", ["i"], 1)
        stat(ctypes.c_int(-1))
        b1, r1 = chain(go)
        assert np.array_equal(r1["flux_up"], up0)
        assert stat(ctypes.c_int(0)) > 10 and stat(ctypes.c_int(3)) < 8 * ncol * (NLAY + 1) * 4  # hits; only the fluxes came back
        assert not np.array_equal(b1["tau"], tau0)           # the host copy is unspecified (canaries) ...
        assert wb(b1["tau"].ctypes.data) == 1                 # ... until it is written back
        assert np.array_equal(b1["tau"], tau0)
        assert wb(b1["tau"].ctypes.data) == 0
        # the k-distribution replaced in place: the cached device copy must not be used
        kmajor = go.t["kmajor"]
        kmajor *= 2.0
        hiplib.ext_call(hip, "rte_hip_host_mirror", ["i"], 0)
        b2, r2 = chain(go)
        up2 = r2["flux_up"].copy()
        hiplib.ext_call(hip, "rte_hip_host_mirror", ["i"], 1)
        b3, r3 = chain(go)
        assert np.array_equal(r3["flux_up"], up2) and not np.array_equal(up2, up0)
        kmajor *= 0.5
    finally:
        hiplib.ext_call(hip, "rte_hip_host_mirror", ["i"], 0)
        setc(None)
        destroy(ctx)


@pytest.mark.gpu
def test_host_mirror_mode_serves_unchanged_inputs_from_the_device_and_sees_changed_ones():
    """Host-mirror mode keeps a device copy and a host-side shadow of inputs the host produced: the same bytes coming again
    (the same range, or another array with the same contents) are not uploaded; ONE changed element in the same range is."""
    hip = hiplib.load()
    create, setc, destroy = _ctx_api(hip)
    stat = hip.raw("rte_hip_mirror_stat")
    stat.restype = ctypes.c_longlong
    ncol, nlev, ngpt = 512, 61, 64  # 16 MB of spectral fluxes
    rng = np.random.default_rng(7)
    x = np.asfortranarray(rng.random((ncol, nlev, ngpt)))
    out = np.zeros((ncol, nlev), order="F")
    ctx = create(-1, None)
    setc(ctx)
    os.environ["RTE_HIP_INPUT_CACHE"] = "1"  # (read at the context's first staged input; by default on from the second staging context on)
    try:
        hiplib.ext_call(hip, "rte_hip_host_mirror", ["i"], 1)
        stat(ctypes.c_int(-1))
        hip.rte_sum_broadband(ncol, nlev, ngpt, x, out)
        want = x.sum(axis=2)
        assert np.allclose(out, want, rtol=1e-13) and stat(ctypes.c_int(10)) == 0
        up0 = stat(ctypes.c_int(2))
        out[:] = 0
        hip.rte_sum_broadband(ncol, nlev, ngpt, x, out)              # same range, same bytes
        assert np.allclose(out, want, rtol=1e-13)
        assert stat(ctypes.c_int(10)) == 1 and stat(ctypes.c_int(2)) == up0 and stat(ctypes.c_int(11)) == x.nbytes
        y = x.copy(order="F")                                        # another range, same bytes
        hip.rte_sum_broadband(ncol, nlev, ngpt, y, out)
        assert np.allclose(out, want, rtol=1e-13) and stat(ctypes.c_int(10)) == 2
        x[ncol // 2, nlev // 2, ngpt // 2] += 1.0                    # one element changed in place
        hip.rte_sum_broadband(ncol, nlev, ngpt, x, out)
        assert stat(ctypes.c_int(10)) == 2 and stat(ctypes.c_int(2)) == up0 + x.nbytes
        assert abs(out[ncol // 2, nlev // 2] - want[ncol // 2, nlev // 2] - 1.0) < 1e-12
        assert np.allclose(out, x.sum(axis=2), rtol=1e-13)
    finally:
        os.environ.pop("RTE_HIP_INPUT_CACHE", None)
        hiplib.ext_call(hip, "rte_hip_host_mirror", ["i"], 0)
        setc(None)
        destroy(ctx)


@pytest.mark.gpu
def test_input_cache_does_not_refill_an_entry_already_handed_out_in_the_same_call():
    """Fortran temporaries swap addresses between calls.  Call 1 passes (play = P at address A, tlay = T at address
    B).  Call 2 passes play = a NEW array at address C whose bytes equal T (served from the entry of B by content) and tlay = NEW
    bytes at address B: refilling B's entry for tlay would overwrite the device copy play is about to be read from.  Checked with
    rte_hip_tlev_interp (two same-sized inputs) against the C oracle."""
    from oracle import oracle as O

    hip = hiplib.load()
    create, setc, destroy = _ctx_api(hip)
    ncol, nlay = 4096, 40  # 1.3 MB per array: above the cache's 64 KB floor
    rng = np.random.default_rng(11)
    F = np.asfortranarray
    plev = F(np.sort(rng.uniform(100.0, 101000.0, (ncol, nlay + 1)), axis=1)[:, ::-1])
    a_buf = F(0.5 * (plev[:, 1:] + plev[:, :-1]))          # address A: pressures
    b_buf = F(rng.uniform(180.0, 320.0, (ncol, nlay)))       # address B: temperatures T
    oracle = O.load_c()

    def tlev_of(lib, play, tlay):
        out = np.zeros((ncol, nlay + 1), order="F")
        assert hiplib.ext_call(lib, "rte_hip_tlev_interp", "iiaaaa", ncol, nlay, play, plev, tlay, out) == 0
        return out

    ctx = create(-1, None)
    setc(ctx)
    os.environ["RTE_HIP_INPUT_CACHE"] = "1"
    try:
        hiplib.ext_call(hip, "rte_hip_host_mirror", ["i"], 1)
        first = tlev_of(hip, a_buf, b_buf)
        assert np.array_equal(first, tlev_of(oracle, a_buf, b_buf))
        c_buf = b_buf.copy(order="F")                        # address C: the bytes of T, handed in as "play"
        b_buf[:] = F(rng.uniform(200.0, 300.0, (ncol, nlay)))  # address B: new contents, handed in as "tlay"
        want = tlev_of(oracle, c_buf, b_buf)
        got = tlev_of(hip, c_buf, b_buf)
        assert np.array_equal(got, want), float(np.max(np.abs(got - want)))
        assert np.array_equal(tlev_of(hip, c_buf, b_buf), want)  # and again, from whatever the cache holds now
    finally:
        os.environ.pop("RTE_HIP_INPUT_CACHE", None)
        hiplib.ext_call(hip, "rte_hip_host_mirror", ["i"], 0)
        setc(None)
        destroy(ctx)
