"""GPU-box helper: build experiment variants of the HIP library (extra -D flags) and time the LW gas-optics kernels.
usage: python tools/variants.py tag1:-DFLAG1,-DFLAG2 tag2: ...   (built here or on the box; run on the box)"""
import os, subprocess, sys
sys.path.insert(0, ".")
from rte_rrtmgp_amd import hiplib
mode = sys.argv[1]
for spec in sys.argv[2:]:
    tag, _, flags = spec.partition(":")
    name = f"librte_rrtmgp_hip_x{tag}.so"
    if mode == "build":
        hiplib.LIB_NAMES["dp"] = name
        hiplib.build("dp", force=True, extra=[f for f in flags.split(",") if f])
    else:
        code = (f"import sys; sys.path.insert(0,'.'); from rte_rrtmgp_amd import hiplib; hiplib.LIB_NAMES['dp']='{name}'; "
                "import os; sys.argv=['x']+os.environ.get('VARIANT_ARGS','100000').split(); exec(open(os.environ.get('VARIANT_SCRIPT','tools/time_gas_optics.py')).read())")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        print(tag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
