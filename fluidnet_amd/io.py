"""On-disk formats either side of the simulate() path -- host-side plumbing, numpy only.

  loadMantaFile / saveMantaFile   Manta `.bin` frames      torch/lib/load_manta_file.lua:15-61
  loadVoxelData                   binvox RLE obstacles     torch/lib/obstacles_import_binvox.lua:29-119
  calculateBoundingBox, padVoxelsToDims, flipDiagonal      torch/lib/voxel_utils.lua:20-50, 176-203, 225-277
  voxelsToFlags                   occupancy -> flags, interior cells only   torch/fluid_net_3d_sim.lua:118-129
  VboxWriter                      `.vbox` density / geometry dump for the Blender scripts
                                                           torch/fluid_net_3d_sim.lua:164-190, 266-291
Arrays are numpy, [B, C, Z, Y, X] float32 like everything else on the path.
"""
import struct

import numpy as np


# ---- Manta .bin ------------------------------------------------------------------------------------
def loadMantaFile(fn):
    """-> p[1,1,Z,Y,X], U[1,C,Z,Y,X], flags[1,1,Z,Y,X] (float, read from int32), density, is3D."""
    with open(fn, "rb") as f:
        _transpose, nx, ny, nz, is3d = struct.unpack("<5i", f.read(20))   # transpose: legacy, never used
        is3d = is3d == 1
        n = nx * ny * nz

        def rd(dt):
            return np.frombuffer(f.read(4 * n), dt, n)
        ux, uy = rd("<f4"), rd("<f4")
        uz = rd("<f4") if is3d else None
        p = rd("<f4")
        flags = rd("<i4").astype(np.float32)
        density = rd("<f4")
    sh = (1, 1, nz, ny, nx)
    comps = [ux, uy] + ([uz] if is3d else [])
    U = np.concatenate([c.reshape(sh) for c in comps], axis=1).astype(np.float32)
    return (p.reshape(sh).astype(np.float32), np.ascontiguousarray(U), flags.reshape(sh),
            density.reshape(sh).astype(np.float32), is3d)


def saveMantaFile(fn, p, U, flags, density):
    """Inverse of loadMantaFile (the layout manta's scenes/_trainingData.py writes); B must be 1."""
    _, C, nz, ny, nx = U.shape
    is3d = C == 3
    with open(fn, "wb") as f:
        f.write(struct.pack("<5i", 0, nx, ny, nz, 1 if is3d else 0))
        for c in range(C):
            f.write(np.ascontiguousarray(U[0, c], "<f4").tobytes())
        f.write(np.ascontiguousarray(p[0, 0], "<f4").tobytes())
        f.write(np.ascontiguousarray(flags[0, 0]).astype("<i4").tobytes())
        f.write(np.ascontiguousarray(density[0, 0], "<f4").tobytes())


# ---- binvox ------------------------------------------------------------------------------------------
def loadVoxelData(filename, reference_quirks=True):
    """binvox reader. Header: '#binvox 1' / 'dim a b c' / 'translate ..' / 'scale ..' / 'data', then
    (value, count) byte pairs. Returns {'dims', 'translation', 'scale', 'data'} with data float32 of
    shape (dims[0], dims[2], dims[1]) -- the reference's view(d1,d2,d3):permute(1,3,2).

    reference_quirks=True reproduces obstacles_import_binvox.lua:77-104 literally: each run is written
    over count+1 cells (the extra one is overwritten by the next run) and the pair that ends at the end of
    the file is skipped ('file:position() < endPosition'), so the cells of the last run keep the value 0
    except its first cell, which keeps the previous run's value. False = the plain binvox semantics."""
    with open(filename, "rb") as f:
        raw = f.read()
    lines, pos = [], 0
    for _ in range(5):
        e = raw.index(b"\n", pos)
        lines.append(raw[pos:e].decode("latin-1"))
        pos = e + 1
    dims = [int(v) for v in lines[1].split()[1:4]]
    translation = [float(v) for v in lines[2].split()[1:4]]
    scale = float(lines[3].split()[1])
    body = np.frombuffer(raw, np.uint8, offset=pos)
    n = dims[0] * dims[1] * dims[2]
    data = np.zeros(n + 1, np.uint8)
    npairs = len(body) // 2
    index = 0   # 0-based start of the current run
    for k in range(npairs):
        value, count = int(body[2 * k]), int(body[2 * k + 1])
        if index + 1 >= n and reference_quirks:
            break       # while (endIndex < voxelCount)
        if reference_quirks and k == npairs - 1:
            break       # the pair that reaches the end of the file is read but not applied
        end = index + count
        if end > n:
            raise ValueError("binvox run overruns the grid")
        data[index:end + (1 if reference_quirks else 0)] = value
        index = end
    vox = data[:n].reshape(dims[0], dims[1], dims[2]).transpose(0, 2, 1)
    return dict(dims=dims, translation=translation, scale=scale, data=np.ascontiguousarray(vox, np.float32))


def saveVoxelData(filename, voxels, translation=(0.0, 0.0, 0.0), scale=1.0):
    """binvox writer (the format of the voxelizer's output the reference loads): `voxels` = a {0,1} grid in the layout
    loadVoxelData returns, shape (d0, d2, d1). Plain run-length pairs (value, count <= 255). The reference's reader
    drops the pair that ends the file (see loadVoxelData), so a grid that is to survive it must end in two empty cells
    (file order: the last two x of the last row) -- true of every model that does not touch the last corner of its
    box; asserted here."""
    v = np.ascontiguousarray(np.asarray(voxels).transpose(0, 2, 1) != 0).astype(np.uint8)
    d0, d1, d2 = v.shape
    flat = v.reshape(-1)
    assert flat[-1] == 0 and flat[-2] == 0, "the last two cells (file order) must be empty: the reference's reader skips the final run"
    change = np.flatnonzero(np.diff(flat)) + 1
    starts = np.concatenate(([0], change))
    lengths = np.diff(np.concatenate((starts, [flat.size])))
    runs = [(int(flat[st]), int(ln)) for st, ln in zip(starts, lengths)]
    # the reference's reader writes each run over count+1 cells and never applies the pair that ends the file: end with
    # a run of its own holding the last empty cell, so that the spill of the run before it lands on an empty cell too
    val, ln = runs[-1]
    runs[-1:] = [(0, ln - 1), (0, 1)]
    out = bytearray()
    for val, ln in runs:
        while ln > 0:
            c = min(ln, 255)
            out += bytes((val, c))
            ln -= c
    with open(filename, "wb") as f:
        f.write(("#binvox 1\ndim %d %d %d\ntranslate %g %g %g\nscale %g\ndata\n"
                 % (d0, d1, d2, translation[0], translation[1], translation[2], scale)).encode("latin-1"))
        f.write(bytes(out))


def calculateBoundingBox(voxels):
    """voxel_utils.lua:20-50: 1-based inclusive first/last non-zero index along each of the 3 dims."""
    assert voxels.ndim == 3 and voxels.sum() > 0
    mn, mx = [], []
    for d in range(3):
        nz = np.nonzero(voxels.sum(axis=tuple(a for a in range(3) if a != d)))[0]
        mn.append(int(nz[0]) + 1)
        mx.append(int(nz[-1]) + 1)
    return dict(min=mn, max=mx)


def padVoxelsToDims(width, height, depth, voxels, offsetX=0, offsetY=0, offsetZ=0):
    """voxel_utils.lua:176-203: crop to the bounding box, paste centred (+offset) into [depth, height, width]."""
    assert voxels.ndim == 3
    assert voxels.shape[0] <= depth and voxels.shape[1] <= height and voxels.shape[2] <= width
    bb = calculateBoundingBox(voxels)
    v = voxels[bb["min"][0] - 1:bb["max"][0], bb["min"][1] - 1:bb["max"][1], bb["min"][2] - 1:bb["max"][2]]
    pl = max(int(np.floor((width - v.shape[2]) / 2 + offsetX)), 1)
    pb = max(int(np.floor((height - v.shape[1]) / 2 + offsetY)), 1)
    pk = max(int(np.floor((depth - v.shape[0]) / 2 + offsetZ)), 1)
    out = np.zeros((depth, height, width), voxels.dtype)
    out[pk:pk + v.shape[0], pb:pb + v.shape[1], pl:pl + v.shape[2]] = v
    assert out.sum() == v.sum(), "Lost some voxels."
    return out


def flipDiagonal(voxels, axis):
    """voxel_utils.lua:225-277: transpose the two dims other than `axis` (0-based), in place semantics
    of the reference returned as a new array."""
    assert voxels.ndim == 3 and 0 <= axis <= 2
    perm = {0: (0, 2, 1), 1: (2, 1, 0), 2: (1, 0, 2)}[axis]
    a, b = [d for d in range(3) if d != axis]
    assert voxels.shape[a] == voxels.shape[b]
    return np.ascontiguousarray(voxels.transpose(perm))


def voxelsToFlags(flags, occupancy):
    """fluid_net_3d_sim.lua:118-129: copy {0,1} occupancy into a flags grid as Obstacle(2)/Fluid(1), only
    inside the 1-cell border. flags: [B,1,Z,Y,X] (modified in place), occupancy: [Z,Y,X]."""
    occ = occupancy[1:-1, 1:-1, 1:-1]
    flags[:, 0, 1:-1, 1:-1, 1:-1] = occ * 2.0 + (1.0 - occ) * 1.0
    return flags


# ---- .vbox ---------------------------------------------------------------------------------------------
class VboxWriter:
    """`.vbox`: int32 {X, Y, Z, frames} then per frame a float32 volume stored x-slowest, i.e. the
    reference's squeeze():permute(3,2,1) of a [Z,Y,X] grid (fluid_net_3d_sim.lua:164-169, 286-290)."""

    def __init__(self, filename, xdim, ydim, zdim, num_frames):
        self.f = open(filename, "wb")
        self.f.write(struct.pack("<4i", xdim, ydim, zdim, num_frames))
        self.shape = (zdim, ydim, xdim)

    def write(self, grid):
        """grid: [..., Z, Y, X] scalar field (leading unit dims / channel mean handled by the caller)."""
        g = np.asarray(grid, np.float32).reshape(self.shape)
        self.f.write(np.ascontiguousarray(g.transpose(2, 1, 0), "<f4").tobytes())

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def readVbox(filename):
    """Reader for tests / tooling: -> frames [F, Z, Y, X]."""
    with open(filename, "rb") as f:
        x, y, z, frames = struct.unpack("<4i", f.read(16))
        data = np.frombuffer(f.read(), "<f4")
    nf = data.size // (x * y * z)
    return data[:nf * x * y * z].reshape(nf, x, y, z).transpose(0, 3, 2, 1).copy(), frames
