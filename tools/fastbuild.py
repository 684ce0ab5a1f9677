"""Build the HIP library from per-file objects compiled in parallel and cached by (source, flags) content hash -- for
experiment builds: a variant that changes one translation unit recompiles only that one.
usage: python tools/fastbuild.py [tag[:file.hip=-DFLAG1,-DFLAG2[;file2.hip=...]]] ...   (tag "" = the product library)
The product build of record stays rte-rrtmgp_amd/hiplib.py (one hipcc command over all sources)."""
import hashlib, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rte_rrtmgp_amd  # noqa
from rte_rrtmgp_amd import hiplib

CACHE = os.environ.get("RTE_OBJ_CACHE", "/tmp/rte_obj_cache")
os.makedirs(CACHE, exist_ok=True)
BASE = [f for f in hiplib.HIPCC_FLAGS if f != "-shared"]
INC = ["-I", os.path.join(hiplib.ROOT, "include"), "-I", hiplib.CSRC]


def headers_digest():
    h = hashlib.sha1()
    for p in sorted(os.listdir(hiplib.CSRC)) + ["../../include/rte_rrtmgp_kernels.h", "../../include/rte_hip_ext.h"]:
        if p.endswith(".h"):
            h.update(open(os.path.join(hiplib.CSRC, p), "rb").read())
    return h.hexdigest()


HD = headers_digest()


def obj_for(src, extra):
    h = hashlib.sha1((HD + " ".join(BASE + extra)).encode() + open(src, "rb").read()).hexdigest()[:20]
    out = os.path.join(CACHE, os.path.basename(src) + "." + h + ".o")
    if not os.path.exists(out):
        subprocess.check_call(["hipcc"] + BASE + extra + INC + ["-c", src, "-o", out + ".tmp"])
        os.replace(out + ".tmp", out)
    return out


def build(tag, per_file):
    name = "librte_rrtmgp_hip.so" if not tag else f"librte_rrtmgp_hip_x{tag}.so"
    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(lambda s: obj_for(s, per_file.get(os.path.basename(s), []) + per_file.get("*", [])), hiplib.sources()))
    out = os.path.join(hiplib.PKG_DIR, name)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    return out


if __name__ == "__main__":
    for spec in (sys.argv[1:] or [""]):
        tag, _, rest = spec.partition(":")
        per = {}
        for part in filter(None, rest.split(";")):
            f, _, flags = part.partition("=")
            per[f] = [x for x in flags.split(",") if x]
        print(build(tag, per))
