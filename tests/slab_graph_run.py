"""Helper of tests/test_native_transport.py::test_rank_step_graph_equals_eager_step: own process (the library binds ONE
RCCL per process: tests/stub_rccl.cpp in STUB_RCCL_NULL mode, named by TFL_RCCL_LIBRARY). A middle rank of a 4-rank
layout steps through the library's native transport, once eagerly and once as a replayed HIP graph
(tfl_slab_graph_create): same kernels, same transport calls, so the two states must be equal bit for bit -- with absent
neighbours the numbers are not a simulation, but they are deterministic. Then the same for a slab without neighbours
(world 1), where the result is also the unsplit step's."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import torch
    import test_hip_simulate as T
    from fluidnet_amd import FluidNetModel, tfluids
    from fluidnet_amd.dist import RcclComm, SlabLayout, SlabSimulation
    from fluidnet_amd.simulate import simulate_native
    from oracle import simulate_np as S
    dev = torch.device("cuda:0")
    world, Y, X = 4, 24, 32
    Zt = 12 * world
    b = T._plume_batch((Zt, Y, X), 0.15, 0.6, obstacles_seed=11)
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0,
                 gravityScale=0.2, vorticityConfinementAmp=2.0, simMethod="convnet")
    layers = S.default_3d_layers(seed=2)
    ref = T._to_dev(b, dev)
    model = FluidNetModel(layers, True)
    for _ in range(3):
        simulate_native(None, mconf, ref, model)           # a developed state to cut the slabs from
    lib, ctx = tfluids._context(ref["flags"])
    assert lib.tfl_rccl_comm_origin(ctx).decode() == os.environ["TFL_RCCL_LIBRARY"]
    for overlap in (0, 1):
        for rank in (1, 0):
            out = {}
            for graph in (False, True):
                lay = SlabLayout(Zt, world, rank)
                loc = {k: (lay.extract(v) if torch.is_tensor(v) else v) for k, v in ref.items()}
                comm = RcclComm(ctx, RcclComm.unique_id(ctx), rank, world)
                assert comm.struct.capturable == 1
                sim = SlabSimulation(loc, mconf, FluidNetModel(layers, True), lay, comm, overlap=overlap, graph=graph)
                for n in range(5):
                    # (the fourth step of the recorded run goes through the eager call again -- SlabSimulation.step(eager=True), what
                    # bench.py's per-kernel profile pass does -- and the fifth back through the graph: the forms interleave)
                    sim.step(eager=graph and n == 3)
                sim.drain()
                torch.cuda.synchronize()
                assert (sim.graph is not None) == graph, sim.graph_error
                if graph:
                    assert sim.graph_nodes >= 12, sim.graph_nodes
                    assert sim.slab.in_flight & 0xF == 0
                out[graph] = {k: loc[k].clone() for k in ("pDiv", "UDiv", "density")}
                sim.close()
            for k in out[False]:
                assert torch.equal(out[False][k], out[True][k]), (overlap, rank, k)
    # a slab without neighbours: the recorded step is the whole step, and equals the unsplit one
    lay = SlabLayout(Zt, 1, 0)
    loc = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in ref.items()}
    sim = SlabSimulation(loc, mconf, FluidNetModel(layers, True), lay, None, graph=True)
    for _ in range(4):
        sim.step()
        simulate_native(None, mconf, ref, model)
    torch.cuda.synchronize()
    assert sim.graph is not None
    for k in ("pDiv", "UDiv", "density"):
        rel = float((loc[k] - ref[k]).norm() / ref[k].norm().clamp_min(1e-30))
        assert rel <= 1e-7, (k, rel)
    sim.close()
    print("slab graph ok")


if __name__ == "__main__":
    main()
