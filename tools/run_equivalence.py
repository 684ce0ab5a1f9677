"""GPU-box helper: oracle/_ref/bin/ref_equivalence_driver (the invariances of the reference's tests/check_equivalence.F90 on
synthetic streams, through the reference's unchanged frontend) on the HIP library; prints every check's worst deviation.
usage: run_equivalence.py [lw|sw] [ncol] [nlay] [top_at_1: 0|1] [mirror: 0|1]"""
import sys, tempfile, pathlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import stream_io, test_extern_frontend as T
kind = sys.argv[1] if len(sys.argv) > 1 else "lw"
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 96
nlay = int(sys.argv[3]) if len(sys.argv) > 3 else 24
top = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
mirror = sys.argv[5] if len(sys.argv) > 5 else "0"
with tempfile.TemporaryDirectory() as d:
    raw, kd, atm, kf, af = T._frontend_case(pathlib.Path(d), kind, ncol, nlay, ncol, top, False, True)
    rc, checks, log = stream_io.run_equivalence_driver("ref_equivalence_driver", kf, af, T.GASES, env={"RTE_HIP_HOST_MIRROR": mirror})
    print(f"{kind} ncol {ncol} nlay {nlay} top_at_1 {top} mirror {mirror}: rc {rc}")
    for k, (w, lim, ok) in checks.items():
        print(f"   {k:44s} {w:10.2f} spacings (limit {lim:4.0f}) {'ok' if ok else 'FAIL'}")
    if rc != 0:
        print(log[-1500:])
