#!/usr/bin/env python
"""bench.py -- simulate() throughput of the MI355X-native tfluids path.

Workload (BASELINE.json metric: "simulate() steps/s + Mcells/s, 3D 128^3 ConvNet projection, 1/2/4/8 MI355X"):
BASELINE config 4 -- the scene of torch/fluid_net_3d_sim.lua:62-87 at res 128: plume BCs
(createPlumeBCs(batch, {1}, plumeScale=1, rad=0.15)), buoyancyScale 2, vorticityConfinementAmp 3,
dt 0.1, maccormackOurs with strength 0.6, ConvNet projection (3-D `default` topology, seeded weights:
the reference ships no 3-D model), plus a procedural voxel obstacle standing in for the bunny binvox
(not in the reference tree). One "step" = one tfluids.simulate() call on the whole grid; fp32.

  python bench.py [--gpus N] [--steps K] [--warmup W]      (N>1: launched by torch.distributed.run)

N = 1: the whole 128^3 grid on one GPU through ONE C-ABI call per step (tfl_simulate_step).
N > 1: STRONG scaling of the same 128^3 grid -- the metric's 1/2/4/8 series: z-slabs of 128/N planes (+4 halo planes per
neighbour), one tfl_simulate_step_slab call per step and rank, halo messages through the library's own transport
(ncclSend / ncclRecv issued natively, fluidnet_amd/csrc/comm_rccl.cpp; torch.distributed only hands out the unique id
and provides the timing barrier). TFL_DIST_BACKEND=gloo swaps in a host-staged transport for control-flow checks on a
box with fewer GPUs than ranks; the line names the transport that actually ran.
After the timed region the run also measures BASELINE configs 1-3 (short, through HIP-graph replay where they are
launch-bound) and a 1 GiB device copy (the HBM rate this box actually delivers) -- reported under "configs" and
"hbm_measured_peak_GBps"; every roofline fraction is given against the 8 TB/s pin rate AND against that.
Every run also reports BASELINE config 5 (3-D 256^3, no obstacle, no confinement; cut into N z-slabs) under
"config5_256" -- the north-star's ">= 100 steps/s at 256^3 on 8 GPUs" figure -- measured after the timed region.
Rank 0 prints ONE JSON line. Inputs are resident in HBM before the timed region.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# cpu_baseline mixes two OpenMP runtimes (libgomp in oracle/_ref, torch's for conv3d); with spinning
# workers they starve each other on a many-core host. Passive waits give the CPU baseline its best time.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # same guide: fp32 matrix (= vector) peak
F16_MFMA_PEAK_TFLOPS = 2500.0   # same guide: bf16 / fp16 MFMA, dense
# conv_mfma16.hip (the default 3-D conv path): MFMA flops ISSUED per voxel. One v_mfma_f32_16x16x32_f16 = 16384 flop; a
# wave issues 72 of them per 128 output voxels in the first layer (9 (dz, dy) taps per output row of 16 voxels, one
# activation term): every fp32 product is four fp16 products (hi/lo of both operands) and one K group of four is idle.
# The 8->8 layers run K-PACKED by default (k_conv3_m16p: the 27 taps in 7 groups of <= 4 K slices): 56 per 64 voxels;
# TFL_M16_KPACK=0 brings back k_conv3_m16z (one (dz, dy) per MFMA, 72).
_KPACK = os.environ.get("TFL_M16_KPACK", "1")
_M16_MID = (56 if _KPACK != "0" else 72) * 16384 / 64.0
# the first layer K-packed (k_conv3_m16p_in, the default): 7 MFMAs per output row of 16 voxels = 28 per 64 voxels (round 5: the
# constant still described the 32 x 4 x 4 tile kernel's 72 per 128); the tail since round 5 also issues one
# v_mfma_f32_16x16x16_f16 (8192 flop) per output row for its 8 -> 8 (k = 1) layer
_M16_TAIL = _M16_MID + (4 * 8192 / 64.0 if os.environ.get("TFL_M16_TAIL_MFMA", "1") != "0" and _KPACK != "0" else 0.0)
M16_ISSUED_FLOP_PER_VOXEL = {"k_conv3_in": (28 * 16384 / 64.0 if _KPACK not in ("0", "2") else 72 * 16384 / 128.0),
                             "k_conv3_mid": _M16_MID, "k_conv3_tail": _M16_TAIL}

# Algorithmic HBM bytes per CELL per launch (fp32, 3-D; each distinct input read once, each output
# written once) -- SURVEY.md 8d restated per kernel of the fused implementation.
ALG_BYTES_PER_CELL = {
    "k_scalar_fwd": 24,      # s, U3, flags -> fwd                      (advectScalar pass A; + 8 B of clamp bounds)
    "k_scalar_bwd": 28,      # fwd, s, U3, flags -> dst                 (advectScalar pass B; + 8 B)  A+B = 52
    "k_stream_copy": 8,      # the yard-stick: float4 copy
    "k_vel_fwd": 28,         # U3, flags -> fwd3                        (advectVel pass A)
    "k_vel_bwd": 40,         # fwd3, U3, flags -> dst3                  (advectVel pass B)     A+B = 68
    "k_add_buoyancy": 32,    # U3, flags, rho -> U3
    "k_curl": 28,            # U3 -> curl3, |curl|                      (vorticity pass A)
    "k_confine": 44,         # curl3, |curl|, flags, U3 -> U3           (vorticity pass B)     A+B = 72
    "k_vort_fused": 28,      # U3, flags -> U3                          (curl + confinement in one launch, curl in LDS)
    "k_bcs_div_stats": 30 if os.environ.get("TFL_WALL_PLAN", "1") != "0" else 32,   # U3, wall codes (2 B; round 6: tfl_wall_plan) | flags (4 B) -> U3_bc, div (+ 2 scalars)
    "k_net_input": 24,       # pDiv, div, flags -> 3 input planes
    "k_project": 34 if os.environ.get("TFL_WALL_PLAN", "1") != "0" else 36,   # pPred, wall codes (2 B) | flags (4 B), U3 -> U3, p  (the plume's U pair is sparse: applied on its four rows, the dense
                             # UBC3 / mask3 tensors -- 24 B/cell more -- are not read; round 3 counted them: 60)
    "k_set_wall_bcs": 28, "k_divergence": 20, "k_velocity_update": 32, "k_jacobi": 16,
    # round 6, z-slab step only (advect_pair3.hip): the passes A / the passes B of advectScalar AND advectVel as one launch each
    "k_adv_fwd_pair": 24 + 28, "k_adv_bwd_pair": 28 + 40,
}


def build_scene(res_xy, z_total, layout, device):
    """Config-4 scene on a res_xy x res_xy x z_total grid; returns (batch, mconf) holding the planes
    [layout.lo, layout.hi) (the rank's z-slab plus halos; layout=None: the whole grid). Only the local
    planes are ever materialised. Obstacle = sphere + torus of ~res/2 extent (stand-in for the bunny)."""
    from fluidnet_amd import simulate as sim
    X = Y = res_xy
    lo, hi = (0, z_total) if layout is None else (layout.lo, layout.hi)
    Zl = hi - lo
    zz, yy, xx = torch.meshgrid(torch.arange(lo, hi), torch.arange(Y), torch.arange(X), indexing="ij")
    border = (xx == 0) | (xx == X - 1) | (yy == 0) | (yy == Y - 1) | (zz == 0) | (zz == z_total - 1)
    cx, cz = X / 2.0, z_total / 2.0
    sphere = (xx - cx) ** 2 + (yy - 0.50 * Y) ** 2 + (zz - cz) ** 2 <= (0.11 * X) ** 2
    rho = torch.sqrt((xx - cx) ** 2 + (zz - cz) ** 2) - 0.22 * X
    torus = rho ** 2 + (yy - 0.72 * Y) ** 2 <= (0.045 * X) ** 2
    flags = torch.where(border | sphere | torus, 2.0, 1.0).to(torch.float32).view(1, 1, Zl, Y, X).contiguous()
    full = dict(pDiv=torch.zeros(1, 1, Zl, Y, X), UDiv=torch.zeros(1, 3, Zl, Y, X), flags=flags,
                density=torch.zeros(1, 1, Zl, Y, X))
    scale = res_xy / 128.0
    sim.createPlumeBCs(full, [1.0], 1.0 * scale, 0.15, zOffset=lo, zTotal=z_total)
    batch = {k: (None if v is None else v.contiguous().to(device)) for k, v in full.items()}
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=2.0 * scale,
                 gravityScale=0, vorticityConfinementAmp=3.0, simMethod="convnet")
    return batch, mconf


class _TimedOps:
    """Wall time per tfluids operator of the CPU checker (per-op reference-CPU ms beside each number)."""

    def __init__(self, ops):
        self._ops, self.seconds, self.calls = ops, {}, {}

    def __getattr__(self, name):
        fn = getattr(self._ops, name)
        if not callable(fn):
            return fn

        def timed(*a, **kw):
            t0 = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                self.seconds[name] = self.seconds.get(name, 0.0) + time.perf_counter() - t0
                self.calls[name] = self.calls.get(name, 0) + 1
        return timed


def conv_witness(ops, model, dev):
    """How the `f32` label of the conv stack is earned: the projection of a developed 48^3 plume by the
    HIP path (default: fp16 hi/lo pairs on the matrix cores) and by PyTorch-CPU fp32 convolutions, each against the SAME
    graph with fp64 convolutions: ratio = our error / PyTorch-fp32's error (< 1: closer to the exact answer than an fp32
    fmaf chain is). Checker work: runs inside the cpu_baseline leg, outside every timed region."""
    from fluidnet_amd.simulate import simulate_native
    from oracle import simulate_np as S
    b, m = build_scene(48, 48, None, dev)
    for _ in range(12):
        simulate_native(None, m, b, model)
    p, U, f = (b[k].cpu().numpy().copy() for k in ("pDiv", "UDiv", "flags"))
    pg, Ug = model.forward([b["pDiv"], b["UDiv"], b["flags"]])
    p32, U32 = S.model_forward(ops, model.layers, p, U, f)
    p64, U64 = S.model_forward(ops, model.layers, p, U, f, conv_dtype="float64")

    def rel(a, ref):
        a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
        return float(np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-300))
    e_ours, e_torch = rel(pg.cpu().numpy(), p64), rel(p32, p64)
    return {"grid": "48^3 developed plume + obstacle (12 steps)", "rel_l2_p_ours_vs_fp64_conv": e_ours,
            "rel_l2_p_pytorch_fp32_vs_fp64_conv": e_torch, "ratio": e_ours / max(e_torch, 1e-300),
            "rel_l2_U_ours_vs_fp64_conv": rel(Ug.cpu().numpy(), U64), "rel_l2_U_pytorch_fp32_vs_fp64_conv": rel(U32, U64)}


def cpu_baseline(batch, mconf, layers, max_seconds=25.0, max_steps=6, model=None, dev=None):
    """The reference's own CPU tfluids code (oracle/_ref, -O3 build) -- or the C restatement if that
    library did not travel -- driving the same simulate() on the host cores, from the same state."""
    from oracle import simulate_np as S
    kind = "reference"
    try:
        from oracle import ref as refmod
        if refmod.available(fast=True):
            ops = refmod.RefTfluids(fast=True)
        elif refmod.available():
            ops = refmod.RefTfluids()
        else:
            raise ImportError
    except Exception:
        from oracle.oracle import OracleTfluids
        ops, kind = OracleTfluids(), "port"
    nb = {k: (v.cpu().numpy().copy() if torch.is_tensor(v) else v) for k, v in batch.items()}
    cells = nb["flags"].size
    tops = _TimedOps(ops)
    t0 = time.time()
    n = 0
    while n < max_steps and (n == 0 or time.time() - t0 < max_seconds):
        S.simulate(tops, mconf, nb, layers)
        n += 1
    dt = time.time() - t0
    per_op = {k: {"ms_per_step": v / n * 1e3, "calls_per_step": tops.calls[k] / n} for k, v in sorted(tops.seconds.items())}
    per_op["conv_stack_and_numpy_glue"] = {"ms_per_step": (dt - sum(tops.seconds.values())) / n * 1e3, "calls_per_step": 1,
                                           "what": "PyTorch-CPU conv3d x5 + the Lua-side arithmetic restated in numpy (setConstVals, scale, joins)"}
    out = {"value": cells * n / dt / 1e6, "unit": "Mcells/s", "cores": os.cpu_count(), "kind": kind,
           "steps_per_s": n / dt, "ms_per_step": dt / n * 1e3, "per_op_ms": per_op,
           "sample": "%d full simulate() steps of the same %s grid from the post-warm-up state "
                     "(tfluids ops: %s, OpenMP on all host cores; conv stack: PyTorch-CPU conv3d)"
                     % (n, "x".join(str(s) for s in nb["flags"].shape[2:]),
                        "reference CPU sources -O3" if kind == "reference" else "C restatement")}
    if model is not None and dev is not None:
        try:
            out["conv_witness"] = conv_witness(ops, model, dev)
        except Exception as e:      # noqa: BLE001 -- the baseline must not take the line down
            out["conv_witness"] = {"error": repr(e)}
    return out


# Planes a slab rank computes beyond its own, per kernel (below + above), from the z-windows of tfl_simulate_step_slab
# (fluidnet_amd/csrc/simulate.cpp); kernels not listed run on the owned planes only.
SLAB_EXTRA_PLANES = {"k_minmax3": (2, 2), "k_scalar_fwd": (1, 1), "k_vel_fwd": (1, 1), "k_adv_fwd_pair": (1, 1), "k_add_buoyancy": (3, 4),
                     "k_add_gravity": (3, 4), "k_curl": (2, 2), "k_confine": (0, 1), "k_conv3_mfma_in": (3, 2),
                     "k_conv3_mfma": (2, 1), "k_conv3_mfma_tail": (1, 0), "k_conv3_in": (3, 2), "k_conv3_mid": (2, 1),
                     "k_conv3_tail": (1, 0)}


def config5_scene(res, layout, device):
    """BASELINE config 5 / config 3's scene at `res`: plume only (no obstacle, no vorticity confinement)."""
    batch, mconf = build_scene(res, res, layout, device)
    lo, hi = (0, res) if layout is None else (layout.lo, layout.hi)
    zz, yy, xx = torch.meshgrid(torch.arange(lo, hi), torch.arange(res), torch.arange(res), indexing="ij")
    border = (xx == 0) | (xx == res - 1) | (yy == 0) | (yy == res - 1) | (zz == 0) | (zz == res - 1)
    batch["flags"] = torch.where(border, 2.0, 1.0).to(torch.float32).view(1, 1, hi - lo, res, res).contiguous().to(device)
    return batch, dict(mconf, vorticityConfinementAmp=0)


TRANSPORT = {"name": "none (single GPU)"}


def make_stepper(res, world, rank, dev, model, scene):
    """(batch, mconf, step, slab simulation or None) for a res^3 grid on `world` ranks."""
    from fluidnet_amd.simulate import simulate_native
    if world > 1:
        import torch.distributed as dist
        from fluidnet_amd import tfluids
        from fluidnet_amd.dist import DistComm, RcclComm, SlabLayout, SlabSimulation
        layout = SlabLayout(res, world, rank)
        batch, mconf = scene(res, layout, dev)
        comm = None
        if dist.get_backend() == "nccl" and not os.environ.get("TFL_BENCH_TORCH_TRANSPORT"):
            # the library's own transport: rank 0's RCCL unique id travels over the process group, the halo traffic does not.
            # Every rank must end up on the same transport: a failure anywhere sends all of them to torch.distributed.
            lib, ctx = tfluids._context(batch["UDiv"])
            ok = 1
            try:
                box = [RcclComm.unique_id(ctx) if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                comm = RcclComm(ctx, box[0], rank, world)
            except Exception as e:      # noqa: BLE001
                sys.stderr.write("rank %d: native RCCL transport unavailable (%s)\n" % (rank, e))
                ok = 0
            flag = torch.tensor([ok], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                TRANSPORT["name"] = "native RCCL send/recv (csrc/comm_rccl.cpp, %s)" % lib.tfl_rccl_comm_origin(ctx).decode()
            else:
                if comm is not None:
                    comm.close()
                comm = None
        want_graph = None if os.environ.get("TFL_SLAB_GRAPH", "0") == "1" else False     # the recorded rank-step: opt-in
        sim = None
        if comm is not None:
            # the native transport has never run between real GPUs (no multi-GPU box in any round): ONE trial step decides, on every
            # rank together, whether the run stays on it -- a failure anywhere sends all ranks to the torch.distributed transport
            ok = 1
            try:
                sim = SlabSimulation(batch, mconf, model, layout, comm, graph=want_graph)
                sim.step()
                sim.drain()
                torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001
                sys.stderr.write("rank %d: the native RCCL transport failed its trial step (%s)\n" % (rank, e))
                ok = 0
            flag = torch.tensor([ok], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                try:
                    if sim is not None:
                        sim.close()
                    else:
                        comm.close()
                except Exception:      # noqa: BLE001
                    pass
                sim, comm = None, None
                batch, mconf = scene(res, layout, dev)      # the trial step may have touched the state
        if comm is None:
            comm = DistComm(rank, world)
            TRANSPORT["name"] = ("torch.distributed nccl (RCCL) batch_isend_irecv" if dist.get_backend() == "nccl" else
                                 "torch.distributed %s, staged through the host (control-flow check, not a measurement)" % dist.get_backend())
        if sim is None:
            sim = SlabSimulation(batch, mconf, model, layout, comm, graph=False)
        return batch, mconf, sim.step, sim
    batch, mconf = scene(res, None, dev)

    def step():
        # the whole simulate() step through ONE C-ABI call (tfl_simulate_step, csrc/simulate.cpp)
        simulate_native(None, mconf, batch, model)
    return batch, mconf, step, None


def measured_hbm_GBps(dev, mib=1024, reps=10):
    """read + write rate of a 1 GiB device copy by the library's own float4 kernel (k_stream_copy, stencil.hip), timed
    per launch with the built-in HIP-event profiler: what this box's HBM delivers to a streaming kernel."""
    import ctypes
    from fluidnet_amd import tfluids
    n = mib * (1 << 20) // 4
    a = torch.empty(n, device=dev).normal_()
    b = torch.empty_like(a)
    lib, ctx = tfluids._context(a)

    def copy():
        rc = lib.tfl_stream_copy(ctx, ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(a.data_ptr()), n)
        assert rc == 0, lib.tfl_last_error(ctx)
    for _ in range(2):
        copy()
    with tfluids.profile(a) as prof:
        for _ in range(reps):
            copy()
    rec = prof.kernels["k_stream_copy"]
    ok = bool(torch.equal(a[:4096], b[:4096]) and torch.equal(a[-4096:], b[-4096:]))
    del a, b
    torch.cuda.empty_cache()
    assert ok, "k_stream_copy did not copy"
    return 2.0 * n * 4 / (rec["ms"] / rec["calls"] * 1e-3) / 1e9


def _plume_scene(dims, rad, uscale, dev):
    """flags = a border of obstacle cells, zero fields, plume BCs (simulate.lua:47-123) on a [Z][Y][X] grid (Z = 1: 2-D)"""
    from fluidnet_amd import simulate as sim
    Z, Y, X = dims
    C = 3 if Z > 1 else 2
    zz, yy, xx = torch.meshgrid(torch.arange(Z), torch.arange(Y), torch.arange(X), indexing="ij")
    border = (xx == 0) | (xx == X - 1) | (yy == 0) | (yy == Y - 1)
    if Z > 1:
        border |= (zz == 0) | (zz == Z - 1)
    flags = torch.where(border, 2.0, 1.0).to(torch.float32).view(1, 1, Z, Y, X).contiguous()
    full = dict(pDiv=torch.zeros(1, 1, Z, Y, X), UDiv=torch.zeros(1, C, Z, Y, X), flags=flags, density=torch.zeros(1, 1, Z, Y, X))
    sim.createPlumeBCs(full, [1.0], uscale, rad)
    return {k: (None if v is None else v.contiguous().to(dev)) for k, v in full.items()}


def other_configs(dev):
    """BASELINE configs 1-3 on this GPU, steps/s (after the timed region). Small grids are launch-bound:
    they run as HIP-graph replays (GraphedSimulate), the way a host would drive them; config 3 eagerly through
    tfl_simulate_step. Config 2 uses the reference's shipped 2-D model (data/models/myModel2D, parsed into
    tests/golden/myModel2D_weights.npz: a data fixture, no checker code is imported)."""
    from fluidnet_amd import FluidNetModel
    from fluidnet_amd.simulate import GraphedSimulate, simulate_native
    out = {}

    def run(key, what, dims, mconf, model, rad, usc, graph, steps):
        b = _plume_scene(dims, rad, usc, dev)
        if graph:
            g = GraphedSimulate(None, mconf, b, model, native=True)    # the one-call native step, captured
            stepfn = g.step
        else:
            stepfn = lambda: simulate_native(None, mconf, b, model)   # noqa: E731
        for _ in range(20):
            stepfn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            stepfn()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        assert bool(torch.isfinite(b["UDiv"]).all())
        out[key] = {"workload": what, "steps_per_s": 1.0 / el, "ms_per_step": el * 1e3,
                    "mcells_per_s": dims[0] * dims[1] * dims[2] / el / 1e6, "steps": steps,
                    "driver": "HIP-graph replay of tfl_simulate_step (GraphedSimulate native=True)" if graph else "tfl_simulate_step"}

    m2 = dict(dt=4 / 60, advectionMethod="maccormackOurs", maccormackStrength=0.75, buoyancyScale=1.0, gravityScale=0,
              vorticityConfinementAmp=0)
    run("config1", "2-D 64x64 smoke plume, Jacobi pressure (20 iterations)", (1, 64, 64), dict(m2, simMethod="jacobi", maxIter=20),
        None, 0.05, 10.0, True, 300)
    wpath = os.path.join(ROOT, "tests", "golden", "myModel2D_weights.npz")
    if os.path.exists(wpath):
        z = np.load(wpath)
        model2, wsrc = FluidNetModel([(z["w%d" % i], z["b%d" % i]) for i in range(5)], False), "shipped myModel2D weights"
        run("config2", "2-D 128x128 ConvNet pressure projection (2-D default topology, %s)" % wsrc, (1, 128, 128),
            dict(m2, simMethod="convnet"), model2, 0.05, 10.0, True, 300)
    else:
        out["config2"] = {"skipped": "tests/golden/myModel2D_weights.npz (the reference's shipped 2-D model) did not travel"}
    m3 = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, gravityScale=0, simMethod="convnet",
              buoyancyScale=1.0, vorticityConfinementAmp=0)
    run("config3", "3-D 64^3 plume, MacCormack advection + ConvNet projection", (64, 64, 64), m3,
        FluidNetModel.default_3d(seed=1), 0.15, 0.5, os.environ.get("TFL_BENCH_CONFIG3_GRAPH", "1") != "0", 200)
    return out


def error_line(n_gpus, why):
    """the ONE JSON line of a run that could not measure: same keys a reader of the metric looks at, value null"""
    return {"metric": "simulate_mcells_per_s", "value": None, "unit": "Mcells/s", "n_gpus": n_gpus, "error": why,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}


def SHARE_GPU():
    """TFL_RANKS_SHARE_GPU=1: more nccl ranks than GPUs (a control-flow check of the WHOLE --gpus N path against the real RCCL on
    a one-GPU box, never a measurement). RCCL refuses two ranks of one host on one device ("Duplicate GPU detected": host hash +
    bus id), so each rank calls itself a host of its own (NCCL_HOSTID) and RCCL moves the messages through its socket transport
    over the loopback interface: same ncclSend / ncclRecv / ncclAllReduce calls, same stream semantics, another wire."""
    return os.environ.get("TFL_RANKS_SHARE_GPU", "0") == "1"


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute this script under torch.distributed.run with N ranks on this
    node (rendezvous on 127.0.0.1, a free port), pass its output through and return its exit code. nccl (= RCCL, the measured
    configuration) needs one GPU per rank: with fewer visible GPUs the run is refused with rc 2 and a JSON error line --
    unless TFL_DIST_BACKEND=gloo asks for the host-staged control-flow check, where ranks share GPUs."""
    import socket
    import subprocess
    backend = os.environ.get("TFL_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have == 0 or (backend == "nccl" and have < n and not SHARE_GPU()):
        print(json.dumps(error_line(n, "--gpus %d over %s needs %d visible GPUs, this node shows %d%s" % (
            n, backend, n, have, "" if have == 0 else " (control-flow checks with ranks sharing GPUs: TFL_RANKS_SHARE_GPU=1 keeps "
            "RCCL -- its socket transport between the ranks --, TFL_DIST_BACKEND=gloo stages the messages through the host)"))))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool's host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    last = ""
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        if line.strip():
            last = line.strip()
    rc = proc.wait()
    measured = False
    try:
        measured = "metric" in json.loads(last)
    except Exception:      # noqa: BLE001
        pass
    if rc != 0 and not measured:
        print(json.dumps(error_line(n, "torch.distributed.run ended with exit code %d before rank 0 printed its line (stderr above)" % rc)))
    return rc if rc != 0 else (0 if measured else 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--preroll", type=int, default=16, help="untimed steps that develop the plume before warm-up")
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps each; ms_per_step = their median")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true", help="skip the extra 256^3 (BASELINE config 5) measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 1-3")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: become the launcher (one rank per GPU, this node only)
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        if rank == 0:
            print(json.dumps(error_line(args.gpus, "--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))))
        raise SystemExit(2)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # TFL_DIST_BACKEND=gloo (+ ranks sharing GPUs) exists only to validate the multi-rank control flow on a box
    # with fewer GPUs than ranks; the measured configuration is one rank per GPU over nccl (= RCCL).
    backend = os.environ.get("TFL_DIST_BACKEND", "nccl")
    shared = backend == "nccl" and world > torch.cuda.device_count() and SHARE_GPU()
    if backend == "nccl" and world > torch.cuda.device_count() and not shared:
        if rank == 0:
            print(json.dumps(error_line(world, "%d ranks over nccl need %d visible GPUs, this node shows %d" % (world, world, torch.cuda.device_count()))))
        raise SystemExit(2)
    if shared:      # before anything loads RCCL: see SHARE_GPU()
        os.environ["NCCL_HOSTID"] = "tfl-bench-rank%d" % rank
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        TRANSPORT["shared"] = "%d ranks share %d GPU(s), RCCL's socket transport between them (TFL_RANKS_SHARE_GPU=1): a control-flow check, not a measurement" % (world, torch.cuda.device_count())
    local = local % torch.cuda.device_count() if (backend != "nccl" or shared) else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        import datetime
        # a rank whose neighbour never shows up must end the job, not hang it: the process-group timeout is what the
        # RCCL watchdog (nccl) / DistComm.wait (gloo) enforce
        pg_timeout = datetime.timedelta(seconds=300)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)

    from fluidnet_amd import FluidNetModel, tfluids
    from fluidnet_amd.simulate import simulate_native
    model = FluidNetModel.default_3d(seed=1)
    res = args.res
    batch, mconf, step, sim = make_stepper(res, world, rank, dev, model, lambda r, lay, d: build_scene(r, r, lay, d))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed(stepfn, n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            stepfn()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(args.preroll + args.warmup):
        step()
    # EXACTLY --steps steps per timed block, barrier + synchronize on both sides; --blocks blocks back to back. The line
    # reports the MEDIAN block (box-to-box and run-to-run noise on this pool is +-5 %, a round's gains are of that order)
    # with the fastest and slowest beside it.
    block_s = sorted(timed(step, args.steps) for _ in range(max(1, args.blocks)))
    elapsed = block_s[len(block_s) // 2] if len(block_s) % 2 else 0.5 * (block_s[len(block_s) // 2 - 1] + block_s[len(block_s) // 2])
    assert bool(torch.isfinite(batch["UDiv"]).all()), "simulation blew up"
    # the two silent-failure counters of the path, read AFTER the timed region (both synchronise): activations the fp16-split
    # conv stack clamped at 65504 (a clamped run stays finite: the isfinite check above cannot see it) and back-traces that
    # hit one of calcLineTrace's invariant paths. A line with either non-zero would be a measurement of wrong results.
    range_errors = int(model.range_errors(batch["UDiv"]))
    trace_errors = int(tfluids.traceErrors(batch["UDiv"]))
    assert range_errors == 0, "fp16 range errors in the timed region: %d blocks clamped an activation" % range_errors
    assert trace_errors == 0, "line-trace invariant errors in the timed region: %d" % trace_errors

    total_cells = res ** 3
    owned_planes = res // world
    cells_per_gpu = owned_planes * res * res
    ms = elapsed / args.steps * 1e3

    # ---- per-kernel HIP-event timing over further steps (outside the timed region; rank 0's kernels) --------------
    nprof = max(3, min(10, args.steps))
    # (a recorded rank-step -- TFL_SLAB_GRAPH=1 -- is one graph launch to the profiler: these steps go through the eager call)
    prof_step = (lambda: sim.step(eager=True)) if sim is not None else step
    with tfluids.profile(batch["UDiv"]) as prof:
        for _ in range(nprof):
            prof_step()
    if sim is not None:
        sim.drain()
    kernels = {}
    for name, rec in prof.kernels.items():
        kernels[name] = {"launches_per_step": rec["calls"] / nprof, "avg_ms": rec["ms"] / rec["calls"],
                         "ms_per_step": rec["ms"] / nprof}
    lf = [2.0 * w.shape[0] * w.shape[1] * w.shape[2] ** 3 for w, _ in model.layers]   # flop per voxel per layer
    conv_path = os.environ.get("TFL_CONV_PATH", "mfma16")
    conv_flops_per_voxel = {"k_conv_direct": sum(lf), "k_conv3_mfma_in": lf[0], "k_conv3_mfma": lf[1],
                            "k_conv3_mfma_tail": sum(lf[2:]), "k_conv3_in": lf[0], "k_conv3_mid": lf[1], "k_conv3_tail": sum(lf[2:])}

    def cells_of(name):
        """cells one rank's launches of this kernel cover per step: the owned planes plus the slab step's extra planes"""
        below, above = SLAB_EXTRA_PLANES.get(name, (0, 0)) if world > 1 else (0, 0)
        return (owned_planes + (below if rank > 0 else 0) + (above if rank < world - 1 else 0)) * res * res

    # round 5: below 6 M cells per item pass B of advectVel also adds the buoyancy force (k_add_buoyancy is gone from the step):
    # the fused launch reads the advected density too -- fwd3, U3, flags, rho -> U3 = 44 B/cell (the two launches: 40 + 32)
    alg_bytes = dict(ALG_BYTES_PER_CELL)
    buoy_folded = world == 1 and "k_add_buoyancy" not in kernels and float(mconf.get("buoyancyScale", 0) or 0) > 0
    if buoy_folded:
        alg_bytes["k_vel_bwd"] = 44
    for name, k in kernels.items():
        if name in alg_bytes:
            per_step = alg_bytes[name] * cells_of(name)
            k["alg_bytes_per_cell"] = alg_bytes[name]
            k["bound"], k["achieved"], k["unit"] = "hbm", per_step / (k["ms_per_step"] * 1e-3) / 1e9, "GB/s"
            k["frac"] = k["achieved"] / HBM_PEAK_GBS
        elif name.startswith("k_conv"):
            # useful (algorithmic) conv flops of the layers this kernel name executes in one step. What runs behind the
            # name: the default 3-D path is Winograd F(2,3) along x on the VECTOR ALUs (conv_valu.hip; no MFMA issued;
            # 2/3 of a k3 layer's algorithmic MACs are issued), TFL_CONV_PATH=mfma the fp32-MFMA implicit GEMM
            wino = conv_path == "winograd" and name in ("k_conv3_in", "k_conv3_mid", "k_conv3_tail")
            m16 = conv_path == "mfma16" and name in M16_ISSUED_FLOP_PER_VOXEL
            k["unit"] = "TFLOP/s"
            alg = conv_flops_per_voxel.get(name, 0.0) * cells_of(name) / (k["ms_per_step"] * 1e-3) / 1e12
            if m16:
                # the matrix pipe's own roofline: MFMA flops the kernel issues against the dense fp16 MFMA peak. The layer's
                # ALGORITHMIC (fp32-equivalent) flops are reported beside it; they are 3/16 of the issued ones.
                k["bound"] = "mfma"
                k["operands"] = "fp32 values as fp16 hi/lo pairs (2 x 11 bit + round-to-nearest: 23+ bit), fp32 accumulate"
                k["achieved"] = M16_ISSUED_FLOP_PER_VOXEL[name] * cells_of(name) / (k["ms_per_step"] * 1e-3) / 1e12
                k["peak"] = F16_MFMA_PEAK_TFLOPS
                k["frac"] = k["achieved"] / F16_MFMA_PEAK_TFLOPS
                k["algorithmic_TFLOPs"] = alg
                k["algorithmic_frac_of_fp32_peak"] = alg / FP32_PEAK_TFLOPS
                k["algorithmic_frac_of_f16_mfma_peak"] = alg / F16_MFMA_PEAK_TFLOPS
                continue
            k["bound"] = "fp32-valu-winograd" if wino else ("mfma-f32" if "mfma" in name or conv_path == "mfma" else "fp32-valu")
            k["achieved"] = alg
            k["peak"] = FP32_PEAK_TFLOPS
            k["frac"] = k["achieved"] / FP32_PEAK_TFLOPS
            if wino:   # flops the kernel really issues: x-taps 4 multiplies per 2 outputs instead of 6; the 1x1x1 layers in full
                issued = {"k_conv3_in": lf[0] * 2 / 3, "k_conv3_mid": lf[1] * 2 / 3, "k_conv3_tail": lf[2] * 2 / 3 + sum(lf[3:])}[name]
                k["issued_frac"] = issued * cells_of(name) / (k["ms_per_step"] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS
    dom = max(kernels, key=lambda n: kernels[n]["ms_per_step"])
    dk = kernels[dom]
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if world == 1 and res == 128 and os.path.exists(tpath):
        try:
            from fluidnet_amd import _kernels
            tj = json.load(open(tpath))
            meta = tj.get("_meta", {})
            fname, sha_now = _kernels.source_sha(dom, conv_path)
            rec = (meta.get("source_sha") or {}).get(dom)
            if dom not in tj:
                traffic_source = "null: %s is not in profiles/pmc_traffic.json" % dom
            elif not rec or rec[1] is None or rec[1] != sha_now:
                # a PMC figure describes the code it was measured on: refuse it when the kernel's source file has changed
                traffic_source = ("null: profiles/pmc_traffic.json (commit %s) was measured on another version of %s "
                                  "(git blob %s then, %s now); re-run tools/pmc_bench.sh" %
                                  (meta.get("commit", "?"), fname, (rec or [None, None])[1], sha_now))
            else:
                traffic = tj[dom]
                traffic_source = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench at commit %s, "
                                  "%s unchanged since (git blob %s)" % (meta.get("commit", "?"), fname, sha_now[:12]))
        except Exception as e:      # noqa: BLE001
            traffic, traffic_source = None, "null: %r" % (e,)
    # ---- the roofline that binds an advection kernel in practice: VALU issue. From the committed SQ
    # counter pass (profiles/pmc_sq.json, tools/pmc_sq.py; refused like the traffic figure when the kernel's source changed):
    # VALU instructions per wave x waves per SIMD x CLK_PER_VALU / the launch's clocks. CLK_PER_VALU = 4: what a stream of
    # fma-class wave64 instructions costs a SIMD (tools/ubench/valu_rate.hip, profiles/r03_ubench_valu_rate.txt; simple adds /
    # moves issue in ~2.5, transcendentals in ~7) -- a fraction near 1 says the vector pipe, not HBM, is what the kernel runs on.
    CLK_PER_VALU = 4.0
    sq_issue, sq_source = {}, None
    spath = os.path.join(ROOT, "profiles", "pmc_sq.json")
    if world == 1 and res == 128 and os.path.exists(spath):
        try:
            from fluidnet_amd import _kernels
            sj = json.load(open(spath))
            smeta = sj.get("_meta", {})
            for name, rec in sj.items():
                if name.startswith("_") or name not in kernels:
                    continue
                _, sha_now = _kernels.source_sha(name, conv_path)
                was = (smeta.get("source_sha") or {}).get(name)
                if not was or was[1] != sha_now:
                    continue
                per_simd = rec["waves"] / float(smeta.get("simds", 1024))
                sq_issue[name] = {"valu_per_wave": rec["valu_per_wave"], "waves_per_simd": per_simd,
                                  "valu_issue_frac": rec["valu_per_wave"] * per_simd * CLK_PER_VALU / rec["clocks"],
                                  "mfma_util": rec.get("mfma_util"), "wait_any": rec.get("wait_any")}
                kernels[name]["valu_issue_frac"] = sq_issue[name]["valu_issue_frac"]
                if rec.get("mfma_util"):
                    kernels[name]["mfma_util_pmc"] = rec["mfma_util"]
            sq_source = "profiles/pmc_sq.json: rocprofv3 --pmc SQ_* pass of this bench at commit %s (kernels whose source changed since are left out)" % smeta.get("commit", "?")
        except Exception as e:      # noqa: BLE001
            sq_source = "null: %r" % (e,)
    hbm_meas = measured_hbm_GBps(dev)
    sum_kernel_ms = sum(k["ms_per_step"] for k in kernels.values())
    roofline = {"kernel": dom, "bound": dk.get("bound"), "achieved": dk.get("achieved"),
                "peak": HBM_PEAK_GBS if dk.get("bound") == "hbm" else dk.get("peak", FP32_PEAK_TFLOPS), "unit": dk.get("unit"),
                "frac": dk.get("frac"), "issued_frac": dk.get("issued_frac"),
                "frac_of_measured_hbm": (dk["achieved"] / hbm_meas if dk.get("bound") == "hbm" else None),
                "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": dk["avg_ms"], "launches_per_step": dk["launches_per_step"]}
    for extra in ("operands", "algorithmic_TFLOPs", "algorithmic_frac_of_fp32_peak", "algorithmic_frac_of_f16_mfma_peak"):
        if extra in dk:
            roofline[extra] = dk[extra]
    roofline["valu_issue_frac"] = (sq_issue.get(dom) or {}).get("valu_issue_frac")
    roofline["valu_issue_source"] = sq_source
    if dk.get("bound") == "hbm" and roofline["valu_issue_frac"] is not None and roofline["valu_issue_frac"] > 0.7:
        roofline["bound_in_practice"] = ("vector-ALU issue: %.2f of the SIMDs' issue slots are VALU instructions (%d per wave x %.0f waves per SIMD x %.0f clk); "
                                         "`frac` is the HBM fraction the north-star asks for, this is the roofline the kernel actually runs on"
                                         % (roofline["valu_issue_frac"], sq_issue[dom]["valu_per_wave"], sq_issue[dom]["waves_per_simd"], CLK_PER_VALU))
    if dk.get("bound") == "mfma":
        roofline["note"] = ("achieved / peak / frac = MFMA flops ISSUED against the dense fp16 MFMA peak (matrix-pipe utilisation); "
                            "the layer's algorithmic fp32 flops are 27/112 of them in the K-packed 8->8 layers (4 fp16 products per fp32 product, 27 of 28 K slices used), 3/16 in the first layer")
    headline = {}
    if "k_vel_fwd" in kernels and "k_vel_bwd" in kernels:   # the north-star's "advection kernel" figure
        t = kernels["k_vel_fwd"]["ms_per_step"] + kernels["k_vel_bwd"]["ms_per_step"]
        by = 28 * cells_of("k_vel_fwd") + alg_bytes["k_vel_bwd"] * cells_of("k_vel_bwd")
        headline = {"op": "advectVel (k_vel_fwd + k_vel_bwd)" + (" + addBuoyancy (folded into pass B)" if buoy_folded else ""),
                    "algorithmic_bytes_per_cell": 28 + alg_bytes["k_vel_bwd"], "ms": t,
                    "achieved_GBps": by / (t * 1e-3) / 1e9, "frac_of_hbm_peak": by / (t * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "frac_of_measured_hbm": by / (t * 1e-3) / 1e9 / hbm_meas,
                    "bound_in_practice": "instruction issue (DESIGN.md 3.7, profiles/r01_r04_where_the_time_went.md): ~2.7 clocks per instruction of any kind per SIMD",
                    "valu_issue_frac": {n: (sq_issue.get(n) or {}).get("valu_issue_frac") for n in ("k_vel_fwd", "k_vel_bwd")}}
    redundancy = None
    if world > 1:
        # Redundant compute of an INTERIOR rank (two neighbours), from this rank's measured time per plane of each kernel:
        # planes it computes beyond its own, weighted by what a plane of that kernel costs.
        per_plane = {n: k["ms_per_step"] / (cells_of(n) / (res * res)) for n, k in kernels.items()}
        own = sum(per_plane[n] * owned_planes for n in kernels)
        red = sum(per_plane[n] * sum(SLAB_EXTRA_PLANES.get(n, (0, 0))) for n in kernels)
        redundancy = {"redundant_compute_frac": red / (own + red), "halo_planes_stored_per_neighbour": 4,
                      "halo_planes_recomputed_per_rank": red / (own / owned_planes),
                      "messages_per_step": 3, "allreduces_per_step": 1}

    # ---- N > 1: the same grid un-split on rank 0's GPU (after the timed region, the other ranks wait at the next barrier):
    # speed-up and parallel efficiency of THIS run against a single-GPU step measured in the same job on the same box
    single = None
    if world > 1:
        if rank == 0:
            b1, m1, step1, _ = make_stepper(res, 1, 0, dev, model, lambda r, lay, d: build_scene(r, r, lay, d))
            for _ in range(args.preroll + args.warmup):
                step1()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step1()
            torch.cuda.synchronize()
            t1 = (time.perf_counter() - t0) / args.steps
            del b1, step1
            single = {"ms_per_step_single_gpu": t1 * 1e3, "speedup": t1 / (elapsed / args.steps),
                      "parallel_efficiency": t1 / (elapsed / args.steps) / world,
                      "note": "single-GPU step of the same %d^3 grid timed on rank 0's GPU in this job (strong scaling)" % res}
        barrier()

    # ---- the same step with the strict-fp32 conv stack (TFL_CONV_PATH=winograd: fp32 Winograd F(2,3) on the vector ALUs, no
    # fp16 anywhere), so that the `f32` label of the default line can be audited against an exact-fp32 figure ---------------
    conv_exact = None
    if world == 1 and conv_path == "mfma16" and not args.no_configs:      # (--no-configs: the profiling passes want the default kernels only)
        prev = os.environ.get("TFL_CONV_PATH")
        os.environ["TFL_CONV_PATH"] = "winograd"
        try:
            model_w = FluidNetModel.default_3d(seed=1)        # the path is chosen when the device handle is created
            nw = max(10, args.steps // 2)
            for _ in range(3):
                simulate_native(None, mconf, batch, model_w)
            elw = timed(lambda: simulate_native(None, mconf, batch, model_w), nw)
            conv_exact = {"path": "winograd (conv_valu.hip: fp32 Winograd F(2,3) along x on the vector ALUs)", "ms_per_step": elw / nw * 1e3,
                          "steps": nw, "mcells_per_s": total_cells * nw / elw / 1e6}
            del model_w
        finally:
            if prev is None:
                os.environ.pop("TFL_CONV_PATH", None)
            else:
                os.environ["TFL_CONV_PATH"] = prev

    sim_info = None
    if sim is not None:
        sim_info = {"overlap": int(sim.slab.overlap), "check_reach": int(sim.slab.check_reach),
                    "issued_as": ("HIP graph of %d nodes (tfl_slab_graph_step)" % sim.graph_nodes) if sim.graph is not None else
                                 ("eager tfl_simulate_step_slab" + (" (recording refused: %s)" % sim.graph_error if sim.graph_error else ""))}
    # ---- BASELINE config 5: 256^3 cut into `world` z-slabs (after the timed region; its own short timing) -------------
    config5 = None
    if not args.no_config5 and res == 128 and 256 % world == 0:
        del batch, step, sim
        torch.cuda.empty_cache()
        b5, m5, step5, sim5 = make_stepper(256, world, rank, dev, model, config5_scene)
        for _ in range(6):
            step5()
        n5 = 20
        el5 = timed(step5, n5)
        if sim5 is not None:
            sim5.drain()
        assert bool(torch.isfinite(b5["UDiv"]).all())
        config5 = {"workload": "BASELINE config 5: 3-D 256^3 plume, MacCormack + ConvNet projection, %s"
                               % ("one GPU, un-sharded" if world == 1 else "%d z-slabs of %d planes, transport: %s" % (world, 256 // world, TRANSPORT["name"])),
                   "steps_per_s": n5 / el5, "ms_per_step": el5 / n5 * 1e3, "mcells_per_s": 256 ** 3 * n5 / el5 / 1e6,
                   "steps": n5, "n_gpus": world}

    out = {
        "metric": "simulate_mcells_per_s", "value": total_cells * args.steps / elapsed / 1e6, "unit": "Mcells/s",
        "steps_per_s": args.steps / elapsed, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "ms_per_step_min": block_s[0] / args.steps * 1e3, "ms_per_step_max": block_s[-1] / args.steps * 1e3,
        "timed_blocks": len(block_s), "sum_kernel_ms": sum_kernel_ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "dtype_note": "every operator computes in fp32; the 3-D conv stack multiplies fp32 values held as fp16 hi/lo pairs on the "
                      "matrix cores with fp32 accumulation (conv_mfma16.hip); evidence in this line: conv_witness_ratio (error vs an "
                      "fp64 convolution / PyTorch-fp32's, cpu_baseline.conv_witness), range_errors = 0, and conv_exact_fp32 = the same "
                      "step with the strict-fp32 stack" if conv_path == "mfma16" else "fp32 throughout",
        "data": "synthetic",
        "config": {"workload": "BASELINE config 4 / the metric's 128^3 series: 3-D %d^3 plume + voxel obstacle (procedural "
                               "stand-in), MacCormack(Ours) advection, buoyancy, vorticity confinement, ConvNet projection "
                               "(3-D default topology, seeded weights); %s" % (res, "single GPU" if world == 1 else
                               "strong scaling: %d z-slabs of %d planes" % (world, owned_planes)),
                   "grid_zyx": [res, res, res], "per_gpu_grid_zyx": [owned_planes, res, res],
                   "decomposition": "single GPU" if world == 1 else "z-slabs, %d ranks, transport: %s%s" % (
                       world, TRANSPORT["name"], ("; " + TRANSPORT["shared"]) if TRANSPORT.get("shared") else ""),
                   "rank_step": None if sim_info is None else sim_info,
                   "preroll_steps": args.preroll, "slab": redundancy, "strong_scaling": single},
        "range_errors": range_errors, "trace_errors": trace_errors, "conv_exact_fp32": conv_exact,
        "roofline": roofline, "advection_headline": headline, "hbm_measured_peak_GBps": hbm_meas,
        "config5_256": config5, "configs": other_configs(dev) if (world == 1 and not (args.no_configs or args.no_config5)) else None, "kernels": kernels,
    }
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        b1, m1 = build_scene(res, res, None, dev)
        for _ in range(args.preroll):
            from fluidnet_amd.simulate import simulate_native
            simulate_native(None, m1, b1, model)
        out["cpu_baseline"] = cpu_baseline(b1, m1, model.layers, model=model, dev=dev)
        out["conv_witness_ratio"] = (out["cpu_baseline"].get("conv_witness") or {}).get("ratio")
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
