"""Helper of tests/test_native_transport.py: runs in its own process (the library binds ONE RCCL per process, here the
stub named by TFL_RCCL_LIBRARY) and steps `world` virtual z-slab ranks through the library's native transport."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main(world, overlap):
    import torch
    import test_hip_simulate as T
    from fluidnet_amd import FluidNetModel, _lib, tfluids
    from fluidnet_amd.dist import run_virtual_ranks
    from fluidnet_amd.simulate import simulate_native
    from oracle import simulate_np as S
    dev = torch.device("cuda:0")
    Zt, Y, X = 12 * world, 24, 32
    b = T._plume_batch((Zt, Y, X), 0.15, 0.6, obstacles_seed=11)
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0,
                 gravityScale=0.2, vorticityConfinementAmp=2.0, simMethod="convnet")
    layers = S.default_3d_layers(seed=2)
    ref = T._to_dev(b, dev)
    lib, ctx = tfluids._context(ref["flags"])
    assert lib.tfl_rccl_available(ctx) == 1, lib.tfl_last_error(ctx)
    origin = lib.tfl_rccl_comm_origin(ctx).decode()
    assert origin == os.environ["TFL_RCCL_LIBRARY"], origin
    model = FluidNetModel(layers, True)
    sims = T._slab_sims(ref, mconf, world, layers, overlap=overlap, transport="native")
    for _ in range(3):
        for _ in range(2):
            simulate_native(None, mconf, ref, model)
        run_virtual_ranks(sims, 2)
        T._assert_slabs_equal(sims, ref, 1e-7)
    for s in sims:
        lay = s.lay
        for k, below, above in (("UDiv", 2, 2), ("pDiv", 4, 3)):
            a = lay.c0 - (below if lay.has_lower else 0)
            e = lay.c1 + (above if lay.has_upper else 0)
            assert torch.equal(s.batch[k][:, :, a:e], ref[k][:, :, lay.lo + a:lay.lo + e]), (lay.rank, k)
        s.close()
    print("native transport ok: world %d overlap %d origin %s" % (world, overlap, origin))


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))
