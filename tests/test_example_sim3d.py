"""examples/sim3d.py (the reference's fluid_net_3d_sim.lua driver on the HIP path): binvox -> flags -> simulate -> .vbox.
The .vbox the run leaves behind is read back and compared with the oracle stepping the same scene."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))

from fluidnet_amd import io  # noqa: E402


def test_binvox_writer_round_trips_through_the_reference_reader(tmp_path):
    """saveVoxelData -> loadVoxelData, both with the reference's run quirks (obstacles_import_binvox.lua:77-104) and
    with plain binvox semantics: a model whose box ends in empty cells comes back unchanged either way."""
    rng = np.random.RandomState(3)
    n = 16
    vox = (rng.rand(n, n, n) < 0.3).astype(np.float32)
    vox[-1, -3:, -1] = 0.0                       # file order is (d0, d1, d2) = data[d0, d2, d1]: the file ends in empty cells
    fn = str(tmp_path / "m.binvox")
    io.saveVoxelData(fn, vox, translation=(0.5, -1.0, 2.0), scale=3.0)
    for quirks in (True, False):
        got = io.loadVoxelData(fn, reference_quirks=quirks)
        assert got["dims"] == [n, n, n] and got["scale"] == 3.0 and got["translation"] == [0.5, -1.0, 2.0]
        assert np.array_equal(got["data"], vox), quirks
    long_run = np.zeros((8, 8, 8), np.float32)
    long_run[2:6] = 1.0                          # runs longer than 255 cells are split
    io.saveVoxelData(fn, long_run)
    assert np.array_equal(io.loadVoxelData(fn)["data"], long_run)


@pytest.mark.gpu
def test_sim3d_example_against_the_oracle(tmp_path):
    import torch
    import sim3d
    from fluidnet_amd import FluidNetModel
    from oracle import simulate_np as S
    from oracle.oracle import OracleTfluids
    res, frames, dec = 32, 6, 3
    out = str(tmp_path / "run")
    layers = S.default_3d_layers(seed=4)
    model = FluidNetModel(layers, True)
    batch, mconf, _ = sim3d.run(res, frames, out, "procedural", "convnet", decimation=dec, model=model, quiet=True)
    assert os.path.exists(os.path.join(out, "procedural_16.binvox"))
    dens, nframes = io.readVbox(os.path.join(out, "density_output.vbox"))
    assert nframes == frames and dens.shape == (frames // dec, res, res, res)
    geom, _ = io.readVbox(os.path.join(out, "geom_output.vbox"))
    geom_b, _ = io.readVbox(os.path.join(out, "geom_output_blender.vbox"))
    # the obstacle the run saw = the binvox model inside a shell of border cells
    flags0 = sim3d.build(res, os.path.join(out, "procedural_16.binvox"), torch.device("cuda:0"))["flags"].cpu().numpy()
    assert np.array_equal(geom[0], (flags0[0, 0] == 2).astype(np.float32))
    inner = (flags0[0, 0] == 2)[1:-1, 1:-1, 1:-1]
    assert inner.sum() > 100 and np.array_equal(geom_b[0][1:-1, 1:-1, 1:-1], inner.astype(np.float32))
    assert geom_b[0][0].sum() == 0 and geom_b[0][:, :, -1].sum() == 0
    # the same scene on the checker
    nb = dict(pDiv=np.zeros((1, 1, res, res, res), np.float32), UDiv=np.zeros((1, 3, res, res, res), np.float32),
              flags=flags0.copy(), density=np.zeros((1, 1, res, res, res), np.float32))
    S.create_plume_bcs(nb, [1.0], 1.0 * (res / 128.0), 0.15)
    ora = OracleTfluids()
    k = 0
    for i in range(1, frames + 1):
        S.simulate(ora, mconf, nb, layers)
        if i % dec == 0:
            want = nb["density"].mean(axis=1)[0]
            rel = float(np.linalg.norm(dens[k] - want) / max(np.linalg.norm(want), 1e-30))
            assert rel <= 1e-5, (i, rel)
            k += 1
    assert float(np.abs(dens[-1]).max()) > 0.1
    got_u = batch["UDiv"].cpu().numpy()
    assert float(np.linalg.norm(got_u - nb["UDiv"]) / np.linalg.norm(nb["UDiv"])) <= 1e-5
