"""GPU-box helper: the LW chain at small column counts (what a host model's block loop calls): wall time per chain with the launches
queued back to back, the same with a synchronisation per chain (latency), and the host time the calls themselves take."""
import sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
for name in ("rte_hip_defer_zero", "rte_hip_share_geometry"):
    hiplib.ext_call(lib, name, ["i"], 1)
NLAY = 60
kd = synth.make_kdist("lw"); go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
for B in [int(x) for x in sys.argv[1:]] or [512, 1024, 2048, 4096, 8192, 16384]:
    atm = synth.make_atmosphere(B, NLAY, seed=42, kdist=kd)
    play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
    emis = xp.full((B, kd.ngpt), 0.98); bufs, rb = {}, {}
    def chain():
        go.gas_optics_lw(B, NLAY, play, plev, tlay, tsfc, col_gas, tlev, atm.top_at_1, buffers=bufs)
        frontend.rte_lw(lib, xp, B, NLAY, kd.ngpt, atm.top_at_1, bufs["tau"], bufs["lay_src"], bufs["lev_src"], emis, bufs["sfc_src"], buffers=rb)
    for _ in range(3): chain()
    torch.cuda.synchronize(); n = 50
    t0 = time.perf_counter()
    for _ in range(n): chain()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); t_q = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): chain(); torch.cuda.synchronize()
    t_s = (time.perf_counter() - t0) / n
    print("ncol %6d: queued %.3f ms/chain (%.2f M col/s), host issue %.3f ms, with a sync per chain %.3f ms; ideal at 18 ms per 1e5: %.3f ms"
          % (B, t_q * 1e3, B / t_q / 1e6, t_host * 1e3, t_s * 1e3, 18.0 * B / 1e5), flush=True)
