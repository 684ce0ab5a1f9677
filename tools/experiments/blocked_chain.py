"""GPU-box helper: the LW chain (gas optics + lw_solver_noscat, the bench's opt-ins) over 1e5 columns processed in BLOCKS of columns
through the same entry points, the blocks reusing one set of intermediate arrays -- does a block whose tau / source arrays fit the
256 MB Infinity Cache run faster per column than the monolithic call?  usage: python tools/blocked_chain.py [block sizes...]"""
import sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
for name in ("rte_hip_defer_zero", "rte_hip_share_geometry"):
    hiplib.ext_call(lib, name, ["i"], 1)
NCOL, NLAY = 100000, 60
kd = synth.make_kdist("lw"); go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
sizes = [int(x) for x in sys.argv[1:]] or [100000, 25000, 12800, 6400, 3200, 2048, 1024]
for B in sizes:
    atm = synth.make_atmosphere(B, NLAY, seed=42, kdist=kd)
    play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
    emis = xp.full((B, kd.ngpt), 0.98)
    bufs, rb = {}, {}
    nblk = (NCOL + B - 1) // B
    def chain():
        go.gas_optics_lw(B, NLAY, play, plev, tlay, tsfc, col_gas, tlev, atm.top_at_1, buffers=bufs)
        frontend.rte_lw(lib, xp, B, NLAY, kd.ngpt, atm.top_at_1, bufs["tau"], bufs["lay_src"], bufs["lev_src"], emis, bufs["sfc_src"], buffers=rb)
    for _ in range(2): chain()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        for _ in range(nblk): chain()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    # host time alone (launches are asynchronous): the same loop without waiting at the end is bounded by it
    print("block %6d x %3d blocks: %.2f ms per %d columns = %.2f M col/s" % (B, nblk, dt * 1e3, B * nblk, B * nblk / dt / 1e6), flush=True)
    del bufs, rb; torch.cuda.empty_cache()
