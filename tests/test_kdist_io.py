"""Flat k-distribution file format (kdist_io): round trip, validation, and that a table read back drives
the oracle to the same results as the in-memory one."""
import numpy as np
import pytest

from rte_rrtmgp_amd import frontend, kdist_io, synth


@pytest.mark.parametrize("kind", ["lw", "sw"])
def test_round_trip_is_exact(tmp_path, kind):
    kd = synth.make_kdist(kind, ngpt=32, nbnd=2)
    path = str(tmp_path / f"{kind}.npz")
    kdist_io.save_kdist(path, kd)
    back = kdist_io.load_kdist(path)
    assert (back.kind, back.ngpt, back.nbnd, back.nflav, back.ngas) == (kd.kind, kd.ngpt, kd.nbnd, kd.nflav, kd.ngas)
    assert set(back.arrays) == set(kd.arrays) and back.scalars == pytest.approx(kd.scalars)
    for k, v in kd.arrays.items():
        assert back.arrays[k].dtype == v.dtype and back.arrays[k].flags.f_contiguous, k
        assert np.array_equal(back.arrays[k], v), k


def test_loaded_table_gives_identical_optical_depths(tmp_path):
    from oracle import oracle as O

    kd = synth.make_kdist("lw", ngpt=32, nbnd=2)
    path = str(tmp_path / "lw.npz")
    kdist_io.save_kdist(path, kd)
    kd2 = kdist_io.load_kdist(path)
    atm = synth.make_atmosphere(6, 9, seed=4, kdist=kd)
    xp, lib, outs = frontend.NumpyArrays(), O.load_c(), []
    for table in (kd, kd2):
        b = frontend.GasOptics(lib, table, xp).gas_optics_lw(6, 9, atm.play, atm.plev, atm.tlay, atm.tsfc, atm.col_gas,
                                                            atm.tlev, atm.top_at_1)
        outs.append({k: np.array(b[k]) for k in ("tau", "lay_src", "lev_src")})
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_validation_catches_a_broken_table(tmp_path):
    kd = synth.make_kdist("sw", ngpt=32, nbnd=2)
    bad = dict(kd.arrays)
    bad["band_lims_gpt"] = synth.F(np.array([[1, 18], [16, 32]], dtype=np.int32))  # gap between the bands
    broken = synth.KDist(kind=kd.kind, ngas=kd.ngas, nflav=kd.nflav, neta=kd.neta, npres=kd.npres, ntemp=kd.ntemp,
                         nbnd=kd.nbnd, ngpt=kd.ngpt, arrays=bad, scalars=kd.scalars)
    path = str(tmp_path / "bad.npz")
    kdist_io.save_kdist(path, broken)
    with pytest.raises(ValueError, match="band_lims_gpt"):
        kdist_io.load_kdist(path)
    with pytest.raises(ValueError, match="not a rte-rrtmgp-kdist"):
        np.savez(str(tmp_path / "x.npz"), __meta__=np.frombuffer(b'{"format": "other"}', dtype=np.uint8))
        kdist_io.load_kdist(str(tmp_path / "x.npz"))
