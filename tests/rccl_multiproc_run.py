"""One rank of tests/test_rccl_multiproc.py: a real process on its own GPU, the library's native RCCL transport
(fluidnet_amd/csrc/comm_rccl.cpp: ncclCommInitRank through tfl_rccl_comm_create, ncclSend / ncclRecv groups, ncclAllReduce).
argv: rank world rendezvous_dir. Rank 0 writes the RCCL unique id of each communicator into the rendezvous directory; the
others wait for the file -- no torch.distributed anywhere: this is the path a LuaJIT or C host takes.
Every rank also steps the WHOLE grid on its own GPU (the single-GPU step, itself held to the reference by
test_hip_fullsize.py) and compares its owned planes with it."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def exchange_id(rdv, name, rank, make):
    path = os.path.join(rdv, name)
    if rank == 0:
        uid = make()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
        return uid
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 120:
            raise SystemExit("rank %d: no unique id from rank 0 after 120 s" % rank)
        time.sleep(0.05)
    return open(path, "rb").read()


def main(rank, world, rdv):
    import torch
    import test_hip_simulate as T
    from fluidnet_amd import FluidNetModel, tfluids
    from fluidnet_amd.dist import RcclComm, SlabLayout, SlabSimulation
    from fluidnet_amd.simulate import simulate_native
    from oracle import simulate_np as S
    # TFL_RCCL_ONE_GPU=1 (tests/test_rccl_multiproc.py on a one-GPU box): every rank on device 0, each with its own NCCL_HOSTID, so
    # that RCCL takes the ranks for processes of different hosts and moves the messages through its socket transport
    index = 0 if os.environ.get("TFL_RCCL_ONE_GPU") == "1" else rank
    torch.cuda.set_device(index)
    dev = torch.device("cuda", index)
    Zt, Y, X = 16 * world, 24, 32
    b = T._plume_batch((Zt, Y, X), 0.15, 0.6, obstacles_seed=11)
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0,
                 gravityScale=0.2, vorticityConfinementAmp=2.0, simMethod="convnet")
    layers = S.default_3d_layers(seed=2)
    for mode, overlap in (("inplace", 0), ("packed", 1)):
        # TFL_RCCL_PACKED is read when the communicator is made: staged messages (pack / unpack kernels) vs in-place chunks
        if mode == "packed":
            os.environ["TFL_RCCL_PACKED"] = "1"
        else:
            os.environ.pop("TFL_RCCL_PACKED", None)
        ref = T._to_dev(b, dev)
        lib, ctx = tfluids._context(ref["flags"])
        assert lib.tfl_rccl_available(ctx) == 1, lib.tfl_last_error(ctx)
        uid = exchange_id(rdv, "uid_" + mode, rank, lambda: RcclComm.unique_id(ctx))
        lay = SlabLayout(Zt, world, rank)
        loc = {k: (lay.extract(v) if torch.is_tensor(v) else v) for k, v in ref.items()}
        sim = SlabSimulation(loc, mconf, FluidNetModel(layers, True), lay, lambda c: RcclComm(c, uid, rank, world), overlap=overlap)
        model = FluidNetModel(layers, True)
        for n in range(3):
            simulate_native(None, mconf, ref, model)
            try:
                sim.step()
            except tfluids.TfluidsError as e:
                # the communicator is made inside the first step (ncclCommInitRank is collective): an RCCL that cannot
                # bootstrap in this environment (no usable network interface) is a reason to skip, not a failure of the path
                if n == 0 and mode == "inplace" and ("ncclCommInitRank" in str(e) or "rccl transport" in str(e)):
                    print("RCCL_INIT_FAILED: %s" % e)
                    sys.exit(77)
                raise
        sim.drain()
        torch.cuda.synchronize()
        for k in ("pDiv", "UDiv", "density"):
            got, want = lay.owned(sim.batch[k]), ref[k][:, :, lay.z0:lay.z1]
            rel = float((got - want).norm() / want.norm().clamp_min(1e-30))
            assert rel <= 1e-7, (mode, rank, k, rel)
        assert float(ref["UDiv"].abs().max()) > 0.05
        sim.close()
        print("rank %d/%d %s: owned planes equal the single-GPU step (origin %s)" % (rank, world, mode, lib.tfl_rccl_comm_origin(ctx).decode()))
    # ---- round 6: the rank-step RECORDED into a HIP graph with the real ncclSend / ncclRecv / ncclAllReduce inside
    # (tfl_slab_graph_create). Recording is collective in effect (every rank records at the same step index); a rank whose
    # recording fails keeps stepping eagerly -- the two forms post the same messages in the same order, so mixed ranks stay in
    # step. The result must equal the single-GPU step either way; how each rank stepped is printed.
    os.environ.pop("TFL_RCCL_PACKED", None)
    ref = T._to_dev(b, dev)
    lib, ctx = tfluids._context(ref["flags"])
    uid = exchange_id(rdv, "uid_graph", rank, lambda: RcclComm.unique_id(ctx))
    lay = SlabLayout(Zt, world, rank)
    loc = {k: (lay.extract(v) if torch.is_tensor(v) else v) for k, v in ref.items()}
    sim = SlabSimulation(loc, mconf, FluidNetModel(layers, True), lay, RcclComm(ctx, uid, rank, world), overlap=0, graph=None)
    sim.graph_mode = None          # "try, fall back to eager" (what TFL_SLAB_GRAPH=1 selects)
    model = FluidNetModel(layers, True)
    for n in range(6):
        simulate_native(None, mconf, ref, model)
        sim.step()
    sim.drain()
    torch.cuda.synchronize()
    for k in ("pDiv", "UDiv", "density"):
        got, want = lay.owned(sim.batch[k]), ref[k][:, :, lay.z0:lay.z1]
        rel = float((got - want).norm() / want.norm().clamp_min(1e-30))
        assert rel <= 1e-7, ("graph", rank, k, rel)
    print("rank %d/%d recorded step: %s" % (rank, world, ("HIP graph of %d nodes" % sim.graph_nodes) if sim.graph is not None else ("eager (%s)" % sim.graph_error)))
    sim.close()
    # ---- round 6: the exact reach mode over real RCCL -- a flow of 1.5 cells per step through the cuts: every rank is refused
    # before the step (the reach flags go through ncclAllReduce), widens its halos through ncclSend / ncclRecv
    # (tfl_slab_exchange) and ends equal to the single-GPU step
    b2 = T._plume_batch((Zt, Y, X), 0.15, 0.6)
    b2["UDiv"][:, 2, 4:Zt - 4, 4:Y - 4, 4:X - 4] = 15.0
    m2 = dict(mconf, gravityScale=0, vorticityConfinementAmp=1.0)
    ref = T._to_dev(b2, dev)
    uid = exchange_id(rdv, "uid_reach", rank, lambda: RcclComm.unique_id(ctx))
    loc = {k: (lay.extract(v) if torch.is_tensor(v) else v) for k, v in ref.items()}
    sim = SlabSimulation(loc, m2, FluidNetModel(layers, True), lay, RcclComm(ctx, uid, rank, world), overlap=0, check_reach="exact", graph=False)
    for n in range(3):
        simulate_native(None, m2, ref, model)
        sim.step()
    sim.drain()
    torch.cuda.synchronize()
    assert sim.relayouts == [2], sim.relayouts
    for k in ("pDiv", "UDiv", "density"):
        got, want = sim.lay.owned(sim.batch[k]), ref[k][:, :, sim.lay.z0:sim.lay.z1]
        rel = float((got - want).norm() / want.norm().clamp_min(1e-30))
        assert rel <= 1e-7, ("exact reach", rank, k, rel)
    sim.close()
    print("rccl multiproc ok rank %d" % rank)


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
