import sys, time, torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
for name in ("rte_hip_defer_zero", "rte_hip_share_geometry"): hiplib.ext_call(lib, name, ["i"], 1)
aux = int(sys.argv[1])
hiplib.ext_call(lib, "rte_hip_aux_stream", ["i"], aux)
NLAY = 60; kd = synth.make_kdist("lw"); go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
for B in (1024, 4096):
    atm = synth.make_atmosphere(B, NLAY, seed=42, kdist=kd)
    play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
    emis = xp.full((B, kd.ngpt), 0.98); bufs, rb = {}, {}
    def chain():
        go.gas_optics_lw(B, NLAY, play, plev, tlay, tsfc, col_gas, tlev, atm.top_at_1, buffers=bufs)
        frontend.rte_lw(lib, xp, B, NLAY, kd.ngpt, atm.top_at_1, bufs["tau"], bufs["lay_src"], bufs["lev_src"], emis, bufs["sfc_src"], buffers=rb)
    for _ in range(3): chain()
    torch.cuda.synchronize(); n = 50; t0 = time.perf_counter()
    for _ in range(n): chain()
    torch.cuda.synchronize(); tq = (time.perf_counter() - t0) / n
    g = hiplib.CallGraph(lib, chain)
    for _ in range(3): g.launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.launch()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n
    print("aux", aux, "ncol", B, "queued %.3f graph %.3f ms" % (tq * 1e3, tg * 1e3))
