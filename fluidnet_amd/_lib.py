"""ctypes binding of libtfluids_hip.so (C ABI: include/tfluids_hip.h).

The library is the product: there is NO fallback. If it is missing or a HIP call fails, every
operator raises -- the oracle under oracle/ is test infrastructure and is never imported here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtfluids_hip.so")
# Development aid: TFL_LIBRARY points at another build of the SAME library (A/B timing of two kernel versions
# inside one GPU session -- box-to-box variance on the pool is ~10%). It is still the HIP library; no fallback.
LIB_PATH = os.environ.get("TFL_LIBRARY", LIB_PATH)


class TfluidsError(RuntimeError):
    """Raised where the reference raises luaL_error / THError (torch/tfluids/init.lua asserts)."""


class tfl_model_opts(ctypes.Structure):
    """include/tfluids_hip.h tfl_model_opts"""
    _fields_ = [(n, ctypes.c_int32) for n in ("in_pDiv", "in_UDiv", "in_div", "normalize", "norm_chan", "norm_func",
                                              "nonlin", "pressure_skip")]


class tfl_tensor(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("B", ctypes.c_int32), ("C", ctypes.c_int32),
                ("Z", ctypes.c_int32), ("Y", ctypes.c_int32), ("X", ctypes.c_int32)]


_T = ctypes.POINTER(tfl_tensor)
_F3 = ctypes.POINTER(ctypes.c_float)
_c = ctypes

# name -> (restype, argtypes) for every symbol include/tfluids_hip.h declares.
class tfl_sim_params(_c.Structure):
    """include/tfluids_hip.h tfl_sim_params (mconf of lib/simulate.lua)."""
    _fields_ = [("dt", _c.c_float), ("advectionMethod", _c.c_char_p), ("maccormackStrength", _c.c_float),
                ("buoyancyScale", _c.c_double), ("gravityScale", _c.c_double), ("gravity", _c.c_float * 3),
                ("vorticityConfinementAmp", _c.c_double), ("simMethod", _c.c_char_p), ("maxIter", _c.c_int32),
                ("pcgPrecond", _c.c_char_p), ("outputDiv", _c.c_int32)]


class tfl_sim_state(_c.Structure):
    """include/tfluids_hip.h tfl_sim_state."""
    _fields_ = [("p", _c.POINTER(tfl_tensor)), ("U", _c.POINTER(tfl_tensor)), ("flags", _c.POINTER(tfl_tensor)),
                ("n_density", _c.c_int32), ("density", _c.POINTER(tfl_tensor) * 8), ("pBC", _c.c_void_p),
                ("UBC", _c.c_void_p), ("densityBC", _c.c_void_p * 8), ("model", _c.c_void_p)]


class tfl_slab(_c.Structure):
    """include/tfluids_hip.h tfl_slab."""
    _fields_ = [("z_total", _c.c_int32), ("z_first", _c.c_int32), ("own_lo", _c.c_int32), ("own_hi", _c.c_int32),
                ("reach", _c.c_int32), ("overlap", _c.c_int32), ("check_reach", _c.c_int32), ("in_flight", _c.c_int32)]


COMM_START = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                          _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64)
COMM_WAIT = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.c_int)
COMM_ALLREDUCE = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int64)


class tfl_comm_chunk(_c.Structure):
    _fields_ = [("ptr", _c.c_void_p), ("n", _c.c_int64)]


COMM_START_V = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.POINTER(tfl_comm_chunk), _c.POINTER(tfl_comm_chunk),
                            _c.c_int, _c.POINTER(tfl_comm_chunk), _c.POINTER(tfl_comm_chunk))


class tfl_comm(_c.Structure):
    """include/tfluids_hip.h tfl_comm: the transport callbacks of the z-slab step (exchange_start_v: optional, NULL here
    unless a transport sets it -- the Python transports move one staged buffer per neighbour)."""
    _fields_ = [("size", _c.c_int32), ("user", _c.c_void_p), ("exchange_start", COMM_START), ("exchange_wait", COMM_WAIT),
                ("allreduce_sum", COMM_ALLREDUCE), ("exchange_start_v", COMM_START_V), ("capturable", _c.c_int32)]


SIGNATURES = {
    "tfl_abi_version": (_c.c_int, []),
    "tfl_create": (_c.c_void_p, [_c.c_int]),
    "tfl_destroy": (None, [_c.c_void_p]),
    "tfl_set_stream": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "tfl_last_error": (_c.c_char_p, [_c.c_void_p]),
    "tfl_synchronize": (_c.c_int, [_c.c_void_p]),
    "tfl_trace_errors": (_c.c_int64, [_c.c_void_p]),
    "tfl_profile_begin": (_c.c_int, [_c.c_void_p]),
    "tfl_profile_end": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_int64]),
    "tfl_advectScalar": (_c.c_int, [_c.c_void_p, _c.c_float, _T, _T, _T, _T, _T, _c.c_int,
                                    _c.c_char_p, _T, _T, _c.c_int, _c.c_int, _c.c_float, _T]),
    "tfl_advectVel": (_c.c_int, [_c.c_void_p, _c.c_float, _T, _T, _T, _T, _c.c_int, _c.c_char_p,
                                 _c.c_int, _c.c_float, _T]),
    "tfl_setWallBcsForward": (_c.c_int, [_c.c_void_p, _T, _T, _c.c_int]),
    "tfl_velocityDivergenceForward": (_c.c_int, [_c.c_void_p, _T, _T, _T, _c.c_int]),
    "tfl_velocityUpdateForward": (_c.c_int, [_c.c_void_p, _T, _T, _T, _c.c_int]),
    "tfl_vorticityConfinement": (_c.c_int, [_c.c_void_p, _T, _T, _c.c_float, _T, _T, _T, _T,
                                            _c.c_int]),
    "tfl_vorticityConfinementFrom": (_c.c_int, [_c.c_void_p, _T, _T, _T, _c.c_float, _T, _T, _c.c_int]),
    "tfl_addBuoyancy": (_c.c_int, [_c.c_void_p, _T, _T, _T, _F3, _c.c_void_p, _c.c_float,
                                   _c.c_int]),
    "tfl_addBuoyancyFrom": (_c.c_int, [_c.c_void_p, _T, _T, _T, _T, _F3, _c.c_float, _c.c_int]),
    "tfl_addGravity": (_c.c_int, [_c.c_void_p, _T, _T, _F3, _c.c_float, _c.c_int, _c.c_void_p]),
    "tfl_emptyDomain": (_c.c_int, [_c.c_void_p, _T, _c.c_int, _c.c_int]),
    "tfl_flagsToOccupancy": (_c.c_int, [_c.c_void_p, _T, _T]),
    "tfl_rectangularBlur": (_c.c_int, [_c.c_void_p, _T, _c.c_int, _c.c_int, _T, _T]),
    "tfl_signedDistanceField": (_c.c_int, [_c.c_void_p, _T, _c.c_int, _c.c_int, _T]),
    "tfl_solveLinearSystemJacobi": (_c.c_int, [_c.c_void_p, _T, _T, _T, _T, _T, _T, _c.c_int,
                                               _c.c_float, _c.c_int, _c.c_int,
                                               _c.POINTER(_c.c_float)]),
    "tfl_model_create": (_c.c_void_p, [_c.c_void_p, _c.c_int, _c.c_int, _c.POINTER(_c.c_int32),
                                       _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32),
                                       _c.POINTER(_c.POINTER(_c.c_float)),
                                       _c.POINTER(_c.POINTER(_c.c_float))]),
    "tfl_model_create_ex": (_c.c_void_p, [_c.c_void_p, _c.c_int, _c.c_int, _c.POINTER(_c.c_int32),
                                          _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32),
                                          _c.POINTER(_c.c_int32), _c.POINTER(_c.POINTER(_c.c_float)),
                                          _c.POINTER(_c.POINTER(_c.c_float))]),
    "tfl_model_create_opts": (_c.c_void_p, [_c.c_void_p, _c.c_int, _c.c_int, _c.POINTER(_c.c_int32),
                                            _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32),
                                            _c.POINTER(_c.c_int32), _c.POINTER(_c.POINTER(_c.c_float)),
                                            _c.POINTER(_c.POINTER(_c.c_float)), _c.c_void_p]),
    "tfl_model_destroy": (None, [_c.c_void_p, _c.c_void_p]),
    "tfl_model_range_errors": (_c.c_int64, [_c.c_void_p, _c.c_void_p]),
    "tfl_model_range_flag": (_c.c_int64, [_c.c_void_p, _c.c_void_p]),
    "tfl_model_workspace_floats": (_c.c_int64, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "tfl_model_forward": (_c.c_int, [_c.c_void_p, _c.c_void_p, _T, _T, _T, _T, _T, _c.c_void_p,
                                     _c.c_int64, _T, _T, _c.c_int, _c.c_float, _c.c_float]),
    "tfl_model_begin": (_c.c_int, [_c.c_void_p, _c.c_void_p, _T, _T, _T, _c.c_void_p, _c.c_int64, _c.c_int,
                                   _c.c_int, _c.c_void_p]),
    "tfl_model_finish": (_c.c_int, [_c.c_void_p, _c.c_void_p, _T, _T, _T, _T, _c.c_void_p, _c.c_int64,
                                    _c.c_void_p, _c.c_double, _T, _T, _c.c_int, _c.c_float, _c.c_float]),
    "tfl_set_dx_override": (_c.c_int, [_c.c_void_p, _c.c_float]),
    "tfl_velocityDivergenceBackward": (_c.c_int, [_c.c_void_p, _T, _T, _T, _c.c_int, _T]),
    "tfl_velocityUpdateBackward": (_c.c_int, [_c.c_void_p, _T, _T, _T, _T, _c.c_int, _T]),
    "tfl_volumetricUpSamplingNearestForward": (_c.c_int, [_c.c_void_p, _c.c_int, _T, _T]),
    "tfl_volumetricUpSamplingNearestBackward": (_c.c_int, [_c.c_void_p, _c.c_int, _T, _T, _T]),
    "tfl_packPlanes": (_c.c_int, [_c.c_void_p, _c.c_int, _c.POINTER(_T), _c.c_int, _c.c_int, _c.c_void_p, _c.c_int]),
    "tfl_applyBCsIndexed": (_c.c_int, [_c.c_void_p, _T, _T, _T, _c.c_void_p, _c.c_int64]),
    "tfl_normalize_workspace_floats": (_c.c_int64, [_c.c_int32, _c.c_int32, _c.c_int32]),
    "tfl_normalizePressureMean": (_c.c_int, [_c.c_void_p, _T, _T, _c.c_int, _c.c_void_p, _c.c_int64]),
    "tfl_pcg_workspace_floats": (_c.c_int64, [_c.c_int32, _c.c_int32, _c.c_int32]),
    "tfl_solveLinearSystemPCG": (_c.c_int, [_c.c_void_p, _T, _T, _T, _c.c_int, _c.c_char_p, _c.c_float, _c.c_int, _c.c_int,
                                            _c.c_void_p, _c.c_int64, _c.c_void_p]),
    "tfl_getDx": (_c.c_double, [_c.c_void_p, _T]),
    "tfl_copy": (_c.c_int, [_c.c_void_p, _T, _T]),
    "tfl_stream_copy": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64]),
    "tfl_bc_plan_create": (_c.c_void_p, [_c.c_void_p, _T, _T]),
    "tfl_bc_plan_destroy": (None, [_c.c_void_p, _c.c_void_p]),
    "tfl_wall_plan_create": (_c.c_void_p, [_c.c_void_p, _T]),
    "tfl_wall_plan_destroy": (None, [_c.c_void_p, _c.c_void_p]),
    "tfl_wall_plan_retire": (None, [_c.c_void_p]),
    "tfl_simulate_workspace_floats": (_c.c_int64, [_c.c_void_p, _c.POINTER(tfl_sim_params), _c.POINTER(tfl_sim_state)]),
    "tfl_simulate_step": (_c.c_int, [_c.c_void_p, _c.POINTER(tfl_sim_params), _c.POINTER(tfl_sim_state), _c.c_void_p,
                                     _c.c_int64]),
    "tfl_applyBCsIndexedMulti": (_c.c_int, [_c.c_void_p, _c.c_int, _c.POINTER(_T), _c.POINTER(_T), _c.POINTER(_T),
                                            _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_int64)]),
    "tfl_applyBCs": (_c.c_int, [_c.c_void_p, _T, _T, _T, _c.c_int, _c.c_float, _c.c_float]),
    "tfl_setWallBcsBackward": (_c.c_int, [_c.c_void_p, _T, _T, _c.c_int, _T]),
    "tfl_set_z_window": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "tfl_set_stages": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "tfl_set_advect_mode": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "tfl_get_advect_mode": (_c.c_int, [_c.c_void_p]),
    "tfl_set_z_origin": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int]),
    "tfl_model_div": (_c.c_void_p, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    "tfl_slab_halo": (_c.c_int32, [_c.c_int32]),
    "tfl_simulate_slab_workspace_floats": (_c.c_int64, [_c.c_void_p, _c.POINTER(tfl_sim_params), _c.POINTER(tfl_sim_state),
                                                        _c.POINTER(tfl_slab)]),
    "tfl_simulate_step_slab": (_c.c_int, [_c.c_void_p, _c.POINTER(tfl_sim_params), _c.POINTER(tfl_sim_state),
                                          _c.POINTER(tfl_slab), _c.POINTER(tfl_comm), _c.c_void_p, _c.c_int64]),
    "tfl_rccl_available": (_c.c_int, [_c.c_void_p]),
    "tfl_rccl_comm_origin": (_c.c_char_p, [_c.c_void_p]),
    "tfl_rccl_get_unique_id": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "tfl_rccl_comm_create": (_c.c_void_p, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int]),
    "tfl_rccl_comm_wrap": (_c.c_void_p, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int]),
    "tfl_rccl_comm_callbacks": (_c.POINTER(tfl_comm), [_c.c_void_p]),
    "tfl_rccl_comm_destroy": (None, [_c.c_void_p, _c.c_void_p]),
    "tfl_rccl_comm_set_inline": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "tfl_slab_drain": (_c.c_int, [_c.c_void_p, _c.POINTER(tfl_sim_state), _c.POINTER(tfl_slab), _c.POINTER(tfl_comm),
                                  _c.c_void_p, _c.c_int64]),
    "tfl_slab_needed_reach": (_c.c_int32, [_c.c_void_p]),
    "tfl_slab_exchange_floats": (_c.c_int64, [_c.c_int, _c.POINTER(_c.POINTER(tfl_tensor)), _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32),
                                               _c.POINTER(tfl_slab)]),
    "tfl_slab_exchange": (_c.c_int, [_c.c_void_p, _c.c_int, _c.POINTER(_c.POINTER(tfl_tensor)), _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32),
                                     _c.POINTER(tfl_slab), _c.POINTER(tfl_comm), _c.c_void_p, _c.c_int64]),
    "tfl_slab_graph_create": (_c.c_void_p, [_c.c_void_p, _c.POINTER(tfl_sim_params), _c.POINTER(tfl_sim_state), _c.POINTER(tfl_slab),
                                            _c.POINTER(tfl_comm), _c.c_void_p, _c.c_int64]),
    "tfl_slab_graph_step": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "tfl_slab_graph_nodes": (_c.c_int64, [_c.c_void_p]),
    "tfl_slab_graph_destroy": (None, [_c.c_void_p, _c.c_void_p]),
}

_lib = None


def load():
    """dlopen the HIP library (after torch, so both share one libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TfluidsError(
            "libtfluids_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C fluidnet_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    import torch  # noqa: F401  (loads the HIP runtime the process will use)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI drift between header and library
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
