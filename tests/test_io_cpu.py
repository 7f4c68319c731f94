"""CPU tests of the on-disk formats either side of the path (fluidnet_amd/io.py)."""
import numpy as np
import pytest

import scenes
from fluidnet_amd import io


@pytest.mark.parametrize("dims", [(1, 12, 10), (6, 7, 8)])
def test_manta_bin_round_trip(tmp_path, dims):
    sc = scenes.make_scene(dims, seed=3, empty_cells=True)
    fn = str(tmp_path / "000000.bin")
    io.saveMantaFile(fn, sc["p"], sc["U"], sc["flags"], sc["density"])
    p, U, flags, density, is3d = io.loadMantaFile(fn)
    assert is3d == sc["is3d"]
    for a, b in ((p, sc["p"]), (U, sc["U"]), (flags, sc["flags"]), (density, sc["density"])):
        assert a.dtype == np.float32 and np.array_equal(a, b)
    raw = open(fn, "rb").read()
    assert np.frombuffer(raw, "<i4", 5).tolist() == [0, dims[2], dims[1], dims[0], int(dims[0] > 1)]
    nfloat = np.prod(dims) * ((3 if dims[0] > 1 else 2) + 3)
    assert len(raw) == 20 + 4 * nfloat                      # header + Ux,Uy,[Uz],p,flags(int32),density


def _write_binvox(path, vol_d1d2d3):
    flat = vol_d1d2d3.astype(np.uint8).reshape(-1)
    runs, i = [], 0
    while i < flat.size:
        j = i
        while j < flat.size and flat[j] == flat[i] and j - i < 255:
            j += 1
        runs.append((int(flat[i]), j - i))
        i = j
    d = vol_d1d2d3.shape
    with open(path, "wb") as f:
        f.write(("#binvox 1\ndim %d %d %d\ntranslate -0.5 0.25 1\nscale 2.5\ndata\n" % d).encode())
        f.write(bytes(b for r in runs for b in r))


def test_binvox_reader(tmp_path):
    rng = np.random.RandomState(0)
    vol = np.zeros((8, 8, 8), np.uint8)
    vol[2:6, 1:5, 3:7] = rng.rand(4, 4, 4) > 0.3     # ends with a run of zeros, like real models
    fn = str(tmp_path / "m.binvox")
    _write_binvox(fn, vol)
    plain = io.loadVoxelData(fn, reference_quirks=False)
    assert plain["dims"] == [8, 8, 8] and plain["scale"] == 2.5 and plain["translation"] == [-0.5, 0.25, 1.0]
    assert np.array_equal(plain["data"], vol.transpose(0, 2, 1).astype(np.float32))
    # reference behaviour (obstacles_import_binvox.lua:77-104): runs are written over count+1 cells and the
    # pair that reaches end-of-file is skipped, so a model ending in a run of zeros gains exactly one voxel:
    # the first cell of that final run keeps the previous run's value.
    quirky = io.loadVoxelData(fn)
    diff = np.argwhere(quirky["data"] != plain["data"])
    assert len(diff) == 1 and quirky["data"][tuple(diff[0])] == 1.0
    flat_q = quirky["data"].transpose(0, 2, 1).reshape(-1)
    last_one = np.nonzero(vol.reshape(-1))[0][-1]
    assert np.nonzero(flat_q)[0][-1] == last_one + 1
    assert quirky["data"].dtype == np.float32


def test_voxel_utils():
    v = np.zeros((6, 6, 6), np.float32)
    v[1:3, 2:5, 3:4] = 1
    bb = io.calculateBoundingBox(v)
    assert bb == dict(min=[2, 3, 4], max=[3, 5, 4])
    padded = io.padVoxelsToDims(16, 12, 10, v, 0, 1, 0)
    assert padded.shape == (10, 12, 16) and padded.sum() == v.sum()
    pk, pb, pl = max((10 - 2) // 2, 1), max(int(np.floor((12 - 3) / 2 + 1)), 1), max(int(np.floor((16 - 1) / 2)), 1)
    assert padded[pk:pk + 2, pb:pb + 3, pl:pl + 1].sum() == v.sum()
    c = np.arange(27, dtype=np.float32).reshape(3, 3, 3)
    assert np.array_equal(io.flipDiagonal(c, 0), c.transpose(0, 2, 1))
    assert np.array_equal(io.flipDiagonal(c, 1), c.transpose(2, 1, 0))
    assert np.array_equal(io.flipDiagonal(c, 2), c.transpose(1, 0, 2))
    flags = scenes.empty_domain(1, 10, 12, 16, True)
    io.voxelsToFlags(flags, padded)
    assert (flags[0, 0, 1:-1, 1:-1, 1:-1] == 2).sum() == padded[1:-1, 1:-1, 1:-1].sum()
    assert (flags[0, 0, 0] == 2).all() and set(np.unique(flags)) == {1.0, 2.0}


def test_vbox_writer(tmp_path):
    fn = str(tmp_path / "density.vbox")
    rng = np.random.RandomState(1)
    frames = rng.rand(3, 4, 5, 6).astype(np.float32)            # [F, Z, Y, X]
    with io.VboxWriter(fn, 6, 5, 4, 3) as w:
        for fr in frames:
            w.write(fr.reshape(1, 1, 4, 5, 6))
    raw = open(fn, "rb").read()
    assert np.frombuffer(raw, "<i4", 4).tolist() == [6, 5, 4, 3] and len(raw) == 16 + 4 * frames.size
    first = np.frombuffer(raw, "<f4", 4 * 5 * 6, 16).reshape(6, 5, 4)   # x slowest
    assert np.array_equal(first, frames[0].transpose(2, 1, 0))
    back, n = io.readVbox(fn)
    assert n == 3 and np.array_equal(back, frames)
