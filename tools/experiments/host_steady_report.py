import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import stream_io
for nrep in (2, 14):
    r = stream_io.measure_frontend_driver("lw", 98304, 4096, ("mirror",), nrep=nrep, threads=8, env_extra={"REF_DRIVER_TIMING": "1"})["mirror"]
    print("nrep", nrep, "passes", " | ".join(p.split(",")[0] for p in r["passes"]))
    for rep in r["reports"][:2]: print("   ", rep[:420])
    for ln in r["detail"][:8]: print("   ", ln)
