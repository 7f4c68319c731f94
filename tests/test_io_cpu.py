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


# ---- byte-level fixtures written by hand from the reference's readers / writers (no io.py writer involved) ----------
def test_manta_bin_bytes_as_the_reference_reads_them(tmp_path):
    """lib/load_manta_file.lua:15-61 reads: int32 {transpose, nx, ny, nz, is3D}, then float32[numel] Ux, Uy, (Uz if 3-D),
    p, int32[numel] flags, float32[numel] density, every field x-fastest and resized to (1, 1, nz, ny, nx)."""
    import struct
    for (nx, ny, nz, is3d) in [(3, 2, 1, 0), (2, 3, 2, 1)]:
        n = nx * ny * nz
        blob = struct.pack("<5i", 7, nx, ny, nz, is3d)                       # transpose = 7: "Legacy. Never used."
        blob += struct.pack("<%df" % n, *[100 + q for q in range(n)])         # Ux
        blob += struct.pack("<%df" % n, *[200 + q for q in range(n)])         # Uy
        if is3d:
            blob += struct.pack("<%df" % n, *[300 + q for q in range(n)])     # Uz
        blob += struct.pack("<%df" % n, *[0.5 * q for q in range(n)])         # p
        blob += struct.pack("<%di" % n, *[1 + (q % 3 == 0) for q in range(n)])  # flags as int32: 2 (obstacle) / 1 (fluid)
        blob += struct.pack("<%df" % n, *[-1.0 - q for q in range(n)])        # density
        fn = str(tmp_path / ("f%d.bin" % is3d))
        open(fn, "wb").write(blob)
        p, U, flags, density, got3d = io.loadMantaFile(fn)
        assert got3d == bool(is3d) and U.shape == (1, 3 if is3d else 2, nz, ny, nx) and p.shape == (1, 1, nz, ny, nx)
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    q = (k * ny + j) * nx + i
                    assert U[0, 0, k, j, i] == 100 + q and U[0, 1, k, j, i] == 200 + q
                    assert (not is3d) or U[0, 2, k, j, i] == 300 + q
                    assert p[0, 0, k, j, i] == 0.5 * q and density[0, 0, k, j, i] == -1.0 - q
                    assert flags[0, 0, k, j, i] == (2.0 if q % 3 == 0 else 1.0)
        assert all(a.dtype == np.float32 for a in (p, U, flags, density))


def _lua_binvox_reader(raw):
    """obstacles_import_binvox.lua:40-119 transliterated line by line (1-based indices kept), as an independent witness
    for io.loadVoxelData's vectorised reader."""
    lines, pos = [], 0
    for _ in range(5):
        e = raw.index(b"\n", pos)
        lines.append(raw[pos:e].decode())
        pos = e + 1
    dims = [int(v) for v in lines[1].split()[1:4]]
    end_position = len(raw)
    voxel_count = dims[0] * dims[1] * dims[2]
    data1d = [0] * (voxel_count + 2)          # the C buffer behind pData1D; [voxel_count] absorbs the count+1 overrun
    index, end_index = 1, 1
    while end_index < voxel_count and pos < end_position:
        value, count = raw[pos], raw[pos + 1]
        pos += 2
        if pos < end_position:
            end_index = index + count
            assert end_index <= voxel_count
            for i in range(index, end_index + 1):
                data1d[i - 1] = value
            index = end_index
    vol = np.array(data1d[:voxel_count], np.float32).reshape(dims)
    return np.ascontiguousarray(vol.transpose(0, 2, 1))       # view(d1, d2, d3):permute(1, 3, 2)


def test_binvox_bytes_as_the_reference_reads_them(tmp_path):
    """A hand-written 2x2x3 binvox file. Runs (value, count): (0,2) (1,3) (0,1) (1,4) (0,2) over 12 cells. Plain binvox:
    0 0 1 1 1 0 1 1 1 1 0 0. The reference writes count+1 cells per run (the extra cell is overwritten by the next run) and
    never applies the pair that ends the file, so it yields 0 0 1 1 1 0 1 1 1 1 1 0: the first cell of the final zero run
    keeps the previous run's 1."""
    raw = b"#binvox 1\ndim 2 2 3\ntranslate 0 0 0\nscale 1\ndata\n" + bytes([0, 2, 1, 3, 0, 1, 1, 4, 0, 2])
    fn = str(tmp_path / "tiny.binvox")
    open(fn, "wb").write(raw)
    want_flat = np.array([0, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0], np.float32)
    want = np.ascontiguousarray(want_flat.reshape(2, 2, 3).transpose(0, 2, 1))
    got = io.loadVoxelData(fn)
    assert got["dims"] == [2, 2, 3] and got["data"].shape == (2, 3, 2)
    assert np.array_equal(got["data"], want)
    assert np.array_equal(_lua_binvox_reader(raw), want)
    plain = io.loadVoxelData(fn, reference_quirks=False)["data"].transpose(0, 2, 1).reshape(-1)
    assert plain.tolist() == [0, 0, 1, 1, 1, 0, 1, 1, 1, 1, 0, 0]
    # and on a larger random model the two readers agree cell for cell
    rng = np.random.RandomState(4)
    vol = (rng.rand(6, 5, 7) > 0.55).astype(np.uint8)
    vol[-1, -1, -3:] = 0
    fn2 = str(tmp_path / "r.binvox")
    _write_binvox(fn2, vol)
    assert np.array_equal(io.loadVoxelData(fn2)["data"], _lua_binvox_reader(open(fn2, "rb").read()))


def test_vbox_bytes_as_the_reference_writes_them(tmp_path):
    """fluid_net_3d_sim.lua:164-169, 286-290: int32 {res, res, res, numFrames}, then per frame
    density:mean(2):squeeze():permute(3, 2, 1):float():contiguous() of the [Z, Y, X] grid, i.e. x slowest, z fastest."""
    import struct
    Z, Y, X = 2, 3, 4
    g = np.fromfunction(lambda k, j, i: 100 * k + 10 * j + i, (Z, Y, X)).astype(np.float32)
    fn = str(tmp_path / "d.vbox")
    with io.VboxWriter(fn, X, Y, Z, 2) as w:
        w.write(g.reshape(1, 1, Z, Y, X))
        w.write(2 * g)
    raw = open(fn, "rb").read()
    assert struct.unpack("<4i", raw[:16]) == (X, Y, Z, 2) and len(raw) == 16 + 2 * 4 * Z * Y * X
    frame = struct.unpack("<%df" % (Z * Y * X), raw[16:16 + 4 * Z * Y * X])
    want = [100 * k + 10 * j + i for i in range(X) for j in range(Y) for k in range(Z)]
    assert list(frame) == want
    frames, nf = io.readVbox(fn)
    assert nf == 2 and np.array_equal(frames[0], g) and np.array_equal(frames[1], 2 * g)
