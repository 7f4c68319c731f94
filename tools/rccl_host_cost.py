"""One rank of tools/rccl_host_cost.sh: what a rank-step costs the HOST when the transport is the REAL RCCL (round 6: ranks share
this box's GPU under their own NCCL_HOSTID, tools/rccl_one_gpu.sh). The device side of such a run means nothing (socket
transport, N processes on one GPU); the host side does: the time the calling thread spends inside step() -- kernel launches
plus ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd / ncclAllReduce of the real library -- eager against the recorded
rank-step (ONE hipGraphLaunch with the RCCL operations inside). argv: rank world rendezvous_dir res."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(rank, world, rdv, res):
    import torch
    import bench
    from rccl_multiproc_run import exchange_id
    from fluidnet_amd import FluidNetModel, tfluids
    from fluidnet_amd.dist import RcclComm, SlabLayout, SlabSimulation
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lay = SlabLayout(res, world, rank)
    for mode in ("eager", "graph"):
        batch, mconf = bench.build_scene(res, res, lay, dev)
        lib, ctx = tfluids._context(batch["flags"])
        uid = exchange_id(rdv, "uid_" + mode, rank, lambda: RcclComm.unique_id(ctx))
        sim = SlabSimulation(batch, mconf, FluidNetModel.default_3d(seed=1), lay, RcclComm(ctx, uid, rank, world), overlap=0,
                             check_reach=False, graph=(mode == "graph"))
        for _ in range(6):
            sim.step()
        sim.drain()
        torch.cuda.synchronize()
        best = None
        for rep in range(5):
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                sim.step()
            t1 = time.perf_counter()
            sim.drain()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host = (t1 - t0) / n * 1e3
            best = host if best is None else min(best, host)
            wall = (t2 - t0) / n * 1e3
        how = ("HIP graph of %d nodes" % sim.graph_nodes) if sim.graph is not None else "eager"
        assert (sim.graph is not None) == (mode == "graph"), sim.graph_error
        print("res %d rank %d of %d (%d planes), real RCCL, %s: host %.4f ms per step() call (best of 5 x %d calls; wall %.2f ms per step: "
              "socket transport, shared GPU -- not a measurement)" % (res, rank, world, lay.nloc if hasattr(lay, "nloc") else lay.z1 - lay.z0, how, best, n, wall))
        assert bool(torch.isfinite(batch["UDiv"]).all())
        sim.close()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 128)
