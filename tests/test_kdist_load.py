"""Load-time reductions of a k-distribution: rte-rrtmgp_amd/kdist_load.py against the
REFERENCE's own ``ty_gas_optics_rrtmgp%load`` (compiled with flang by oracle/build_load_check.sh into
oracle/_ref/bin/ref_load_driver).  The driver loads a raw table, calls ``gas_optics`` once, and a recorder standing in
for the kernels (oracle/abi_recorder.c) writes every table argument that crosses the kernel C ABI -- i.e. the arrays
after the reference's reductions, exactly as the kernels would see them.  Cases: all file gases available, and a host
model that lacks some minor-absorber gases (intervals dropped, kminor compacted, indices remapped).
GPU: the HIP LW chain driven from a ``.npz`` produced by these reductions, against the oracle."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rte_rrtmgp_amd import frontend, kdist_io, kdist_load, synth  # noqa: E402
from stream_io import write_kdist_stream as _write_stream  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "bin", "ref_load_driver")
ALL = list(kdist_load.FILE_GASES)
SUBSET = ["h2o", "co2", "o3", "n2o", "co", "ch4", "o2"]  # no n2, ccl4, cfc11: their minor intervals must go


def _read_record(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            tag = f.read(32)
            if len(tag) < 32:
                break
            kind, rank = struct.unpack("<ii", f.read(8))
            dims = struct.unpack("<" + "i" * rank, f.read(4 * rank))
            n = int(np.prod(dims)) if rank else 1
            a = np.frombuffer(f.read(8 * n), dtype="<f8") if kind == 1 else np.frombuffer(f.read(4 * n), dtype="<i4")
            a = a.reshape(dims, order="F")
            out[tag.decode().strip()] = a.astype(bool) if kind == 2 else a
    return out


@pytest.mark.parametrize("kind", ["lw", "sw"])
@pytest.mark.parametrize("gases", [ALL, SUBSET], ids=["all-gases", "subset"])
def test_reductions_match_the_reference_load(kind, gases, tmp_path):
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/bin/ref_load_driver absent (needs /root/reference + flang: oracle/build_load_check.sh)")
    raw = kdist_load.synth_raw(kind)
    fin, fout, frec = (str(tmp_path / n) for n in ("raw.bin", "toa.bin", "abi.bin"))
    _write_stream(fin, raw, kind == "lw")
    r = subprocess.run(f"ulimit -s unlimited 2>/dev/null; exec '{DRIVER}' '{fin}' '{fout}' '{','.join(gases)}'", shell=True,
                       capture_output=True, text=True, env=dict(os.environ, RTE_ABI_RECORD=frec), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ref = _read_record(frec)
    kd = kdist_load.init_from_raw(raw, gases)
    kdist_io.validate(kd)
    ngas, nflav, neta, npres, ntemp = (int(x) for x in ref["dims_interp"])
    assert (kd.ngas, kd.nflav, kd.neta, kd.npres, kd.ntemp) == (ngas, nflav, neta, npres, ntemp)
    assert int(ref["idx_h2o"][0]) == kd.idx_h2o
    names = ["flavor", "press_ref_log", "temp_ref", "vmr_ref", "gpoint_flavor", "band_lims_gpt", "kmajor", "kminor_lower",
             "kminor_upper", "minor_limits_gpt_lower", "minor_limits_gpt_upper", "minor_scales_with_density_lower",
             "minor_scales_with_density_upper", "scale_by_complement_lower", "scale_by_complement_upper", "idx_minor_lower",
             "idx_minor_upper", "idx_minor_scaling_lower", "idx_minor_scaling_upper", "kminor_start_lower", "kminor_start_upper"]
    names += ["gpoint_bands", "planck_frac", "totplnk"] if kind == "lw" else ["krayl"]
    for n in names:
        assert ref[n].shape == kd.arrays[n].shape, (n, ref[n].shape, kd.arrays[n].shape)
        assert np.array_equal(ref[n], kd.arrays[n]), n
    sc = ref["scalars"]
    for i, k in enumerate(("press_ref_log_delta", "temp_ref_min", "temp_ref_delta", "press_ref_trop_log")):
        assert sc[i] == kd.scalars[k], k
    if kind == "lw":
        assert ref["totplnk_delta"][0] == kd.scalars["totplnk_delta"]
    else:
        toa = np.fromfile(fout, dtype="<f8")
        assert np.array_equal(toa, kd.arrays["solar_source"])
    if gases is SUBSET:  # the reduction really removed something
        assert kd.arrays["idx_minor_lower"].size < len(raw["minor_gases_lower"])


def test_missing_key_species_is_an_error():
    raw = kdist_load.synth_raw("lw")
    with pytest.raises(ValueError, match="required gases"):
        kdist_load.init_from_raw(raw, ["co2", "o3"])  # h2o is a key species


def test_npz_round_trip_of_a_reduced_table(tmp_path):
    kd = kdist_load.init_from_raw(kdist_load.synth_raw("sw"), SUBSET)
    gas_names = kd.scalars.pop("gas_names")
    assert gas_names == SUBSET
    path = str(tmp_path / "k.npz")
    kdist_io.save_kdist(path, kd)
    back = kdist_io.load_kdist(path)
    for k, v in kd.arrays.items():
        assert np.array_equal(back.arrays[k], v), k


@pytest.mark.gpu
def test_hip_chain_from_a_loaded_table(tmp_path):
    """raw table -> load-time reductions (subset of gases) -> .npz -> load -> HIP LW chain, against the C oracle."""
    from oracle import oracle as O
    from rte_rrtmgp_amd import hiplib

    kd = kdist_load.init_from_raw(kdist_load.synth_raw("lw", ngpt=128, nbnd=8, nminor_lower=19, nminor_upper=13), SUBSET)
    kd.scalars.pop("gas_names")
    path = str(tmp_path / "lw.npz")
    kdist_io.save_kdist(path, kd)
    kd = kdist_io.load_kdist(path)
    ncol, nlay = 900, 30
    atm = synth.make_atmosphere(ncol, nlay, seed=5, kdist=kd, ngas=kd.ngas)
    res = {}
    for name, lib, xp in (("oracle", O.load_c(), frontend.NumpyArrays()), ("hip", hiplib.load(), frontend.TorchArrays("cuda:0"))):
        A = xp.asarray
        go = frontend.GasOptics(lib, kd, xp)
        b = go.gas_optics_lw(ncol, nlay, A(atm.play), A(atm.plev), A(atm.tlay), A(atm.tsfc), A(atm.col_gas), A(atm.tlev), atm.top_at_1)
        r = frontend.rte_lw(lib, xp, ncol, nlay, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                            xp.full((ncol, kd.ngpt), 0.98), b["sfc_src"])
        xp.sync()
        res[name] = {k: np.array(xp.to_numpy(v)) for k, v in dict(tau=b["tau"], lay_src=b["lay_src"], up=r["flux_up"], dn=r["flux_dn"]).items()}
    for k in ("tau", "lay_src"):
        assert np.max(np.abs(res["hip"][k] - res["oracle"][k])) / np.max(np.abs(res["oracle"][k])) <= 1e-12, k
    for k in ("up", "dn"):
        assert np.max(np.abs(res["hip"][k] - res["oracle"][k])) / np.max(np.abs(res["oracle"][k])) <= 1e-10, k
