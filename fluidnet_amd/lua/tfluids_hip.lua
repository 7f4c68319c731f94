-- tfluids_hip.lua -- LuaJIT FFI binding of libtfluids_hip.so (include/tfluids_hip.h).
--
-- Drop-in for the native half of torch/tfluids: after `require('tfluids')`, calling
--   local hip = require('tfluids_hip').install(tfluids)
-- (1) replaces the per-tensor-type C tables that torch/tfluids/init.lua dispatches through
--     (`X.tfluids.<op>(...)`, registered by generic/tfluids.cc:927-957 / generic/tfluids.cu:1932-1962) with functions of
--     the SAME positional signatures that forward to the MI355X library, so init.lua's wrappers (argument checks,
--     getTempStorage, copy-back) and lib/simulate.lua run unchanged;
-- (2) replaces tfluids.normalizePressureMean (init.lua:747-764, a host round trip in the reference) and
--     tfluids.simulate (lib/simulate.lua:175-327) by the device / one-call forms (tfl_normalizePressureMean,
--     tfl_simulate_step);
-- (3) offers hip.Model(gmodule): a table with the call shape of the nngraph model -- model:forward({pDiv, UDiv, flags})
--     -> {p, U} (lib/simulate.lua:262-272) -- backed by tfl_model_forward.
--
-- The ffi.cdef block below is GENERATED from include/tfluids_hip.h (tools/gen_lua_cdef.py) and the CPU test-suite
-- (tests/test_lua_binding.py) regenerates and diffs it, and checks every C call in this file against the declared
-- prototypes (name + argument count), so the binding cannot drift from the header.
-- NOTE: LuaJIT / Torch7 are not available in the build container, so this file is not EXECUTED by the test-suite;
-- the executed host mirror of the same calls is fluidnet_amd/{tfluids,simulate,model}.py. Tensors must be contiguous
-- float tensors whose storage is device memory (cutorch CudaTensor on a ROCm build of cutorch).
local ffi = require('ffi')
local bit = require('bit')

-- BEGIN generated cdef (tools/gen_lua_cdef.py)
ffi.cdef[[
typedef enum tfl_status {
  TFL_OK = 0,
  TFL_EINVAL = -1,
  TFL_EHIP = -2,
  TFL_EUNSUPPORTED = -3,
  TFL_EREACH = -5,
  TFL_ERANGE = -4
} tfl_status;
typedef struct tfl_tensor {
  float* data;
  int32_t B, C, Z, Y, X;
} tfl_tensor;
enum {
  TFL_TypeNone = 0, TFL_TypeFluid = 1, TFL_TypeObstacle = 2, TFL_TypeEmpty = 4,
  TFL_TypeInflow = 8, TFL_TypeOutflow = 16, TFL_TypeOpen = 32, TFL_TypeStick = 128
};
typedef struct tfl_ctx tfl_ctx;
tfl_ctx* tfl_create(int device);
void tfl_destroy(tfl_ctx* ctx);
int tfl_set_stream(tfl_ctx* ctx, void* hip_stream);
const char* tfl_last_error(const tfl_ctx* ctx);
int tfl_abi_version(void);
enum { TFL_ADVECT_EXACT = 0, TFL_ADVECT_FAST = 1 };
int tfl_set_advect_mode(tfl_ctx* ctx, int mode);
int tfl_get_advect_mode(const tfl_ctx* ctx);
int tfl_synchronize(tfl_ctx* ctx);
int64_t tfl_trace_errors(tfl_ctx* ctx);
int tfl_profile_begin(tfl_ctx* ctx);
int tfl_profile_end(tfl_ctx* ctx, char* buf, int64_t cap);
int tfl_advectScalar(tfl_ctx* ctx, float dt, const tfl_tensor* s, const tfl_tensor* U,
                     const tfl_tensor* flags, const tfl_tensor* fwd, const tfl_tensor* bwd,
                     int is3D, const char* method, const tfl_tensor* fwdPos,
                     const tfl_tensor* bwdPos, int boundaryWidth, int sampleOutsideFluid,
                     float maccormackStrength, const tfl_tensor* sDst);
int tfl_advectVel(tfl_ctx* ctx, float dt, const tfl_tensor* U, const tfl_tensor* flags,
                  const tfl_tensor* fwd, const tfl_tensor* bwd, int is3D, const char* method,
                  int boundaryWidth, float maccormackStrength, const tfl_tensor* UDst);
int tfl_setWallBcsForward(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags, int is3D);
int tfl_velocityDivergenceForward(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                                  const tfl_tensor* UDiv, int is3D);
int tfl_velocityUpdateForward(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                              const tfl_tensor* p, int is3D);
int tfl_vorticityConfinement(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                             float strength, const tfl_tensor* centered, const tfl_tensor* curl,
                             const tfl_tensor* curlNorm, const tfl_tensor* force, int is3D);
int tfl_vorticityConfinementFrom(tfl_ctx* ctx, const tfl_tensor* USrc, const tfl_tensor* U, const tfl_tensor* flags,
                                 float strength, const tfl_tensor* curl, const tfl_tensor* curlNorm, int is3D);
int tfl_addBuoyancy(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                    const tfl_tensor* density, const float gravity[3], float* strengthTmp,
                    float dt, int is3D);
int tfl_addBuoyancyFrom(tfl_ctx* ctx, const tfl_tensor* USrc, const tfl_tensor* U, const tfl_tensor* flags,
                        const tfl_tensor* density, const float gravity[3], float dt, int is3D);
int tfl_addGravity(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                   const float gravity[3], float dt, int is3D, float* forceTmp);
int tfl_emptyDomain(tfl_ctx* ctx, const tfl_tensor* flags, int is3D, int bnd);
int tfl_flagsToOccupancy(tfl_ctx* ctx, const tfl_tensor* flags, const tfl_tensor* occupancy);
int tfl_rectangularBlur(tfl_ctx* ctx, const tfl_tensor* src, int blurRad, int is3D, const tfl_tensor* dst,
                        const tfl_tensor* tmp);
int tfl_signedDistanceField(tfl_ctx* ctx, const tfl_tensor* flags, int searchRad, int is3D, const tfl_tensor* dst);
int64_t tfl_pcg_workspace_floats(int32_t Z, int32_t Y, int32_t X);
int tfl_solveLinearSystemPCG(tfl_ctx* ctx, const tfl_tensor* p, const tfl_tensor* flags, const tfl_tensor* div,
                             int is3D, const char* precondType, float tol, int maxIter, int verbose,
                             float* workspace, int64_t workspace_floats, float* residual);
int64_t tfl_normalize_workspace_floats(int32_t Z, int32_t Y, int32_t X);
int tfl_normalizePressureMean(tfl_ctx* ctx, const tfl_tensor* p, const tfl_tensor* flags, int is3D, float* workspace,
                              int64_t workspace_floats);
int tfl_solveLinearSystemJacobi(tfl_ctx* ctx, const tfl_tensor* p, const tfl_tensor* flags,
                                const tfl_tensor* div, const tfl_tensor* pPrev,
                                const tfl_tensor* pDelta, const tfl_tensor* pDeltaNorm, int is3D,
                                float pTol, int maxIter, int verbose, float* residual);
int tfl_velocityDivergenceBackward(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                                   const tfl_tensor* gradOutput, int is3D, const tfl_tensor* gradU);
int tfl_velocityUpdateBackward(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* p,
                               const tfl_tensor* gradOutput, int is3D, const tfl_tensor* gradP);
int tfl_setWallBcsBackward(tfl_ctx* ctx, const tfl_tensor* flags, const tfl_tensor* gradOutput, int is3D,
                           const tfl_tensor* gradU);
int tfl_volumetricUpSamplingNearestForward(tfl_ctx* ctx, int ratio, const tfl_tensor* input,
                                           const tfl_tensor* output);
int tfl_volumetricUpSamplingNearestBackward(tfl_ctx* ctx, int ratio, const tfl_tensor* input,
                                            const tfl_tensor* gradOutput, const tfl_tensor* gradInput);
typedef struct tfl_model tfl_model;
tfl_model* tfl_model_create(tfl_ctx* ctx, int is3D, int nlayers, const int32_t* cin, const int32_t* cout,
                            const int32_t* ksize, const float* const* weights,
                            const float* const* biases);
tfl_model* tfl_model_create_ex(tfl_ctx* ctx, int is3D, int nlayers, const int32_t* cin, const int32_t* cout,
                               const int32_t* ksize, const int32_t* pool, const int32_t* up,
                               const float* const* weights, const float* const* biases);
enum { TFL_NORM_UDIV = 0, TFL_NORM_PDIV = 1, TFL_NORM_DIV = 2 };
enum { TFL_NORMFUNC_STD = 0, TFL_NORMFUNC_L2 = 1 };
enum { TFL_NONLIN_RELU = 0, TFL_NONLIN_RELU6 = 1, TFL_NONLIN_SIGMOID = 2 };
typedef struct tfl_model_opts {
  int32_t in_pDiv, in_UDiv, in_div;
  int32_t normalize;
  int32_t norm_chan;
  int32_t norm_func;
  int32_t nonlin;
  int32_t pressure_skip;
} tfl_model_opts;
tfl_model* tfl_model_create_opts(tfl_ctx* ctx, int is3D, int nlayers, const int32_t* cin, const int32_t* cout,
                                 const int32_t* ksize, const int32_t* pool, const int32_t* up,
                                 const float* const* weights, const float* const* biases, const tfl_model_opts* opts);
void tfl_model_destroy(tfl_ctx* ctx, tfl_model* model);
int64_t tfl_model_range_errors(tfl_ctx* ctx, tfl_model* model);
int64_t tfl_model_range_flag(tfl_ctx* ctx, tfl_model* model);
int64_t tfl_model_workspace_floats(const tfl_model* model, int B, int Z, int Y, int X);
int tfl_model_forward(tfl_ctx* ctx, tfl_model* model, const tfl_tensor* pDiv, const tfl_tensor* UDiv,
                      const tfl_tensor* flags, const tfl_tensor* pOut, const tfl_tensor* UOut,
                      float* workspace, int64_t workspace_floats, const tfl_tensor* UBC,
                      const tfl_tensor* UBCInvMask, int doClamp, float lo, float hi);
int tfl_model_begin(tfl_ctx* ctx, tfl_model* model, const tfl_tensor* UDiv, const tfl_tensor* flags,
                    const tfl_tensor* UOut, float* workspace, int64_t workspace_floats, int zlo, int zhi,
                    double* stats);
int tfl_model_finish(tfl_ctx* ctx, tfl_model* model, const tfl_tensor* pDiv, const tfl_tensor* flags,
                     const tfl_tensor* pOut, const tfl_tensor* UOut, float* workspace,
                     int64_t workspace_floats, const double* stats, double count, const tfl_tensor* UBC,
                     const tfl_tensor* UBCInvMask, int doClamp, float lo, float hi);
int tfl_set_dx_override(tfl_ctx* ctx, float dx);
int tfl_applyBCs(tfl_ctx* ctx, const tfl_tensor* x, const tfl_tensor* bc, const tfl_tensor* invMask,
                 int doClamp, float lo, float hi);
int tfl_applyBCsIndexed(tfl_ctx* ctx, const tfl_tensor* x, const tfl_tensor* bc, const tfl_tensor* invMask,
                        const int32_t* idx, int64_t n);
int tfl_applyBCsIndexedMulti(tfl_ctx* ctx, int count, const tfl_tensor* const* x, const tfl_tensor* const* bc,
                             const tfl_tensor* const* invMask, const int32_t* const* idx, const int64_t* n);
double tfl_getDx(tfl_ctx* ctx, const tfl_tensor* flags);
int tfl_copy(tfl_ctx* ctx, const tfl_tensor* dst, const tfl_tensor* src);
int tfl_stream_copy(tfl_ctx* ctx, float* dst, const float* src, int64_t n);
typedef struct tfl_bc_plan tfl_bc_plan;
tfl_bc_plan* tfl_bc_plan_create(tfl_ctx* ctx, const tfl_tensor* bc, const tfl_tensor* invMask);
void tfl_bc_plan_destroy(tfl_ctx* ctx, tfl_bc_plan* plan);
typedef struct tfl_wall_plan tfl_wall_plan;
tfl_wall_plan* tfl_wall_plan_create(tfl_ctx* ctx, const tfl_tensor* flags);
void tfl_wall_plan_destroy(tfl_ctx* ctx, tfl_wall_plan* plan);
void tfl_wall_plan_retire(tfl_wall_plan* plan);
typedef struct tfl_sim_params {
  float dt;
  const char* advectionMethod;
  float maccormackStrength;
  double buoyancyScale;
  double gravityScale;
  float gravity[3];
  double vorticityConfinementAmp;
  const char* simMethod;
  int32_t maxIter;
  const char* pcgPrecond;
  int32_t outputDiv;
} tfl_sim_params;
typedef struct tfl_sim_state {
  const tfl_tensor* p;
  const tfl_tensor* U;
  const tfl_tensor* flags;
  int32_t n_density;
  const tfl_tensor* density[8];
  const tfl_bc_plan* pBC;
  const tfl_bc_plan* UBC;
  const tfl_bc_plan* densityBC[8];
  tfl_model* model;
} tfl_sim_state;
int64_t tfl_simulate_workspace_floats(tfl_ctx* ctx, const tfl_sim_params* params, const tfl_sim_state* state);
int tfl_simulate_step(tfl_ctx* ctx, const tfl_sim_params* params, const tfl_sim_state* state, float* workspace,
                      int64_t workspace_floats);
int tfl_packPlanes(tfl_ctx* ctx, int n, const tfl_tensor* const* fields, int zlo, int zhi, float* buf,
                   int unpack);
int tfl_set_z_window(tfl_ctx* ctx, int a0, int a1, int b0, int b1);
int tfl_set_z_origin(tfl_ctx* ctx, int z_first, int z_total);
int tfl_set_stages(tfl_ctx* ctx, int mask);
float* tfl_model_div(const tfl_model* model, int B, int Z, int Y, int X, float* workspace);
typedef struct tfl_slab {
  int32_t z_total;
  int32_t z_first;
  int32_t own_lo, own_hi;
  int32_t reach;
  int32_t overlap;
  int32_t check_reach;
  int32_t in_flight;
} tfl_slab;
typedef struct tfl_comm_chunk {
  float* ptr;
  int64_t n;
} tfl_comm_chunk;
typedef struct tfl_comm {
  int32_t size;
  void* user;
  int (*exchange_start)(void* user, int tag, const float* send_lo, int64_t n_send_lo, float* recv_lo, int64_t n_recv_lo,
                        const float* send_hi, int64_t n_send_hi, float* recv_hi, int64_t n_recv_hi);
  int (*exchange_wait)(void* user, int tag);
  int (*allreduce_sum)(void* user, double* dev, int64_t n);
  int (*exchange_start_v)(void* user, int tag, int n_lo, const tfl_comm_chunk* send_lo, const tfl_comm_chunk* recv_lo,
                          int n_hi, const tfl_comm_chunk* send_hi, const tfl_comm_chunk* recv_hi);
  int32_t capturable;
} tfl_comm;
int32_t tfl_slab_halo(int32_t reach);
int64_t tfl_simulate_slab_workspace_floats(tfl_ctx* ctx, const tfl_sim_params* params, const tfl_sim_state* state,
                                           const tfl_slab* slab);
int tfl_simulate_step_slab(tfl_ctx* ctx, const tfl_sim_params* params, const tfl_sim_state* state, tfl_slab* slab,
                           const tfl_comm* comm, float* workspace, int64_t workspace_floats);
int32_t tfl_slab_needed_reach(const tfl_ctx* ctx);
int64_t tfl_slab_exchange_floats(int n, const tfl_tensor* const* fields, const int32_t* below, const int32_t* above, const tfl_slab* slab);
int tfl_slab_exchange(tfl_ctx* ctx, int n, const tfl_tensor* const* fields, const int32_t* below, const int32_t* above,
                      const tfl_slab* slab, const tfl_comm* comm, float* scratch, int64_t scratch_floats);
int tfl_slab_drain(tfl_ctx* ctx, const tfl_sim_state* state, tfl_slab* slab, const tfl_comm* comm, float* workspace,
                   int64_t workspace_floats);
typedef struct tfl_slab_graph tfl_slab_graph;
tfl_slab_graph* tfl_slab_graph_create(tfl_ctx* ctx, const tfl_sim_params* params, const tfl_sim_state* state, tfl_slab* slab,
                                      const tfl_comm* comm, float* workspace, int64_t workspace_floats);
int tfl_slab_graph_step(tfl_ctx* ctx, tfl_slab_graph* graph);
int64_t tfl_slab_graph_nodes(const tfl_slab_graph* graph);
void tfl_slab_graph_destroy(tfl_ctx* ctx, tfl_slab_graph* graph);
typedef struct tfl_rccl_comm tfl_rccl_comm;
int tfl_rccl_available(tfl_ctx* ctx);
const char* tfl_rccl_comm_origin(tfl_ctx* ctx);
int tfl_rccl_get_unique_id(tfl_ctx* ctx, void* id);
tfl_rccl_comm* tfl_rccl_comm_create(tfl_ctx* ctx, const void* id, int rank, int world);
tfl_rccl_comm* tfl_rccl_comm_wrap(tfl_ctx* ctx, void* nccl_comm, int rank, int world);
const tfl_comm* tfl_rccl_comm_callbacks(tfl_rccl_comm* comm);
int tfl_rccl_comm_set_inline(tfl_rccl_comm* comm, int on);
void tfl_rccl_comm_destroy(tfl_ctx* ctx, tfl_rccl_comm* comm);
]]
-- END generated cdef

local M = {}
local lib, ctx

local function check(rc)
  if rc ~= 0 then error(ffi.string(lib.tfl_last_error(ctx)), 3) end
end

-- 5-D (or lower-rank: leading sizes padded with 1) contiguous float tensor -> tfl_tensor
local function T(t)
  assert(t:isContiguous(), 'Input is not contiguous')
  local d = ffi.new('tfl_tensor')
  d.data = ffi.cast('float*', torch.data(t))
  local n = t:dim()
  local sz = {1, 1, 1, 1, 1}
  for i = 1, n do sz[5 - n + i] = t:size(i) end
  d.B, d.C, d.Z, d.Y, d.X = sz[1], sz[2], sz[3], sz[4], sz[5]
  return d
end

local function vec3(t)  -- gravity arrives as a 3-element tensor (init.lua:455-458); the ABI wants host floats
  local h = t:float()
  return ffi.new('float[3]', h[1], h[2], h[3])
end

local function b2i(b) return b and 1 or 0 end

-- One grow-only device scratch (the analogue of init.lua:35-64's shared buffer) for the entry points that take a
-- caller workspace (PCG, normalizePressureMean, the model, the whole step).
local scratch
local function workspace(like, nfloats)
  nfloats = tonumber(nfloats)
  if scratch == nil or scratch:nElement() < nfloats then scratch = like.new():resize(nfloats) end
  return ffi.cast('float*', torch.data(scratch)), nfloats
end

-- ---- the native operator table (same positional signatures as tfluids_CudaMain_*) ----------------------------------
local ops = {}
function ops.advectScalar(dt, s, U, flags, fwd, bwd, is3D, method, fwdPos, bwdPos, bnd, outside, strength, sDst)
  check(lib.tfl_advectScalar(ctx, dt, T(s), T(U), T(flags), T(fwd), T(bwd), b2i(is3D), method,
                             T(fwdPos), T(bwdPos), bnd, b2i(outside), strength, T(sDst)))
end
function ops.advectVel(dt, U, flags, fwd, bwd, is3D, method, bnd, strength, UDst)
  check(lib.tfl_advectVel(ctx, dt, T(U), T(flags), T(fwd), T(bwd), b2i(is3D), method, bnd, strength, T(UDst)))
end
function ops.setWallBcsForward(U, flags, is3D)
  check(lib.tfl_setWallBcsForward(ctx, T(U), T(flags), b2i(is3D)))
end
function ops.velocityDivergenceForward(U, flags, UDiv, is3D)
  check(lib.tfl_velocityDivergenceForward(ctx, T(U), T(flags), T(UDiv), b2i(is3D)))
end
function ops.velocityUpdateForward(U, flags, p, is3D)
  check(lib.tfl_velocityUpdateForward(ctx, T(U), T(flags), T(p), b2i(is3D)))
end
function ops.vorticityConfinement(U, flags, strength, centered, curl, curlNorm, force, is3D)
  check(lib.tfl_vorticityConfinement(ctx, T(U), T(flags), strength, T(centered), T(curl), T(curlNorm), T(force), b2i(is3D)))
end
function ops.addBuoyancy(U, flags, density, gravity, strengthTmp, dt, is3D)
  check(lib.tfl_addBuoyancy(ctx, T(U), T(flags), T(density), vec3(gravity), nil, dt, b2i(is3D)))
end
function ops.addGravity(U, flags, gravity, dt, is3D, forceTmp)
  check(lib.tfl_addGravity(ctx, T(U), T(flags), vec3(gravity), dt, b2i(is3D), nil))
end
function ops.emptyDomain(flags, is3D, bnd)
  check(lib.tfl_emptyDomain(ctx, T(flags), b2i(is3D), bnd))
end
function ops.flagsToOccupancy(flags, occupancy)
  check(lib.tfl_flagsToOccupancy(ctx, T(flags), T(occupancy)))
end
function ops.rectangularBlur(src, blurRad, is3D, dst, tmp)
  check(lib.tfl_rectangularBlur(ctx, T(src), blurRad, b2i(is3D), T(dst), T(tmp)))
end
function ops.signedDistanceField(flags, searchRad, is3D, dst)
  check(lib.tfl_signedDistanceField(ctx, T(flags), searchRad, b2i(is3D), T(dst)))
end
function ops.velocityDivergenceBackward(U, flags, gradOutput, is3D, gradU)
  check(lib.tfl_velocityDivergenceBackward(ctx, T(U), T(flags), T(gradOutput), b2i(is3D), T(gradU)))
end
function ops.velocityUpdateBackward(U, flags, p, gradOutput, is3D, gradP)
  check(lib.tfl_velocityUpdateBackward(ctx, T(U), T(flags), T(p), T(gradOutput), b2i(is3D), T(gradP)))
end
-- not a native entry of the reference (tfluids.SetWallBcs:updateGradInput does it with a mask tensor,
-- set_wall_bcs.lua:50-66); exposed for a module override: gradInput[1] = hip.setWallBcsBackward(flags, gradOutput)
function M.setWallBcsBackward(flags, gradOutput, gradU)
  gradU = gradU or gradOutput.new():resizeAs(gradOutput)
  check(lib.tfl_setWallBcsBackward(ctx, T(flags), T(gradOutput), b2i(gradOutput:size(2) == 3), T(gradU)))
  return gradU
end
function ops.volumetricUpSamplingNearestForward(ratio, input, output)
  check(lib.tfl_volumetricUpSamplingNearestForward(ctx, ratio, T(input), T(output)))
end
function ops.volumetricUpSamplingNearestBackward(ratio, input, gradOutput, gradInput)
  check(lib.tfl_volumetricUpSamplingNearestBackward(ctx, ratio, T(input), T(gradOutput), T(gradInput)))
end
function ops.solveLinearSystemJacobi(p, flags, div, pPrev, pDelta, pDeltaNorm, is3D, pTol, maxIter, verbose)
  local res = ffi.new('float[1]')
  check(lib.tfl_solveLinearSystemJacobi(ctx, T(p), T(flags), T(div), T(pPrev), T(pDelta), T(pDeltaNorm),
                                        b2i(is3D), pTol, maxIter, b2i(verbose), res))
  return res[0]
end
-- init.lua:674-676: p.tfluids.solveLinearSystemPCG(tfluids._tmpPCG, p, flags, div, is3D, precondType, tol, maxIter,
-- verbose) -> residual. The reference's cache table of cuSPARSE temporaries is ignored: the workspace is ours.
function ops.solveLinearSystemPCG(tmpPCG, p, flags, div, is3D, precondType, tol, maxIter, verbose)
  local ws, n = workspace(p, lib.tfl_pcg_workspace_floats(flags:size(3), flags:size(4), flags:size(5)))
  local res = ffi.new('float[1]')
  check(lib.tfl_solveLinearSystemPCG(ctx, T(p), T(flags), T(div), b2i(is3D), precondType, tol, maxIter,
                                     b2i(verbose), ws, n, res))
  return res[0]
end

-- ---- tfluids.normalizePressureMean on the device (init.lua:747-764 copies to the host and back) ---------------------
function M.normalizePressureMean(p, flags, is3D)
  local ws, n = workspace(p, lib.tfl_normalize_workspace_floats(flags:size(3), flags:size(4), flags:size(5)))
  check(lib.tfl_normalizePressureMean(ctx, T(p), T(flags), b2i(is3D), ws, n))
end

-- ---- the projection ConvNet: hip.Model(gmodule) has the call shape of the nngraph model ------------------------------
-- Walks nn.gModule.forwardnodes in order and collects the convolution layers (cudnn / nn {Spatial,Volumetric}Convolution
-- built by lib/model_utils.lua:80-116), exactly what fluidnet_amd/torch7.py:conv_layers does for the Python binding.
local Model = {}
Model.__index = Model

-- mconf (optional): the model's configuration table (torch.loadModel returns it next to the graph); its fields
-- inputChannels / normalizeInput* / nonlinType / addPressureSkip select the forward graph as lib/model.lua:27-160 does.
function M.Model(gmodule, mconf)
  local cin, cout, ks, ws, bs, keep = {}, {}, {}, {}, {}, {}
  local is3D = false
  for _, node in ipairs(gmodule.forwardnodes) do
    local m = node.data.module
    local tn = m and torch.type(m) or ''
    if tn:find('Convolution') then
      is3D = tn:find('Volumetric') ~= nil
      local w, b = m.weight:float():contiguous(), m.bias:float():contiguous()   -- [nOut][nIn][k(z)][k(y)][k(x)]
      keep[#keep + 1] = w; keep[#keep + 1] = b
      cin[#cin + 1] = m.nInputPlane; cout[#cout + 1] = m.nOutputPlane; ks[#ks + 1] = m.kW
      ws[#ws + 1] = ffi.cast('const float*', torch.data(w)); bs[#bs + 1] = ffi.cast('const float*', torch.data(b))
    end
  end
  local n = #cin
  assert(n > 0, 'no convolution layers found in the model')
  local self = setmetatable({is3D = is3D}, Model)
  local opts = nil
  if mconf ~= nil then
    local ic = mconf.inputChannels or {pDiv = true, UDiv = false, div = true, flags = true}
    assert(ic.flags ~= false, 'Are you sure you dont want flags?')
    local chan = ({UDiv = lib.TFL_NORM_UDIV, pDiv = lib.TFL_NORM_PDIV, div = lib.TFL_NORM_DIV})[mconf.normalizeInputChan or 'UDiv']
    local func = ({std = lib.TFL_NORMFUNC_STD, norm = lib.TFL_NORMFUNC_L2})[mconf.normalizeInputFunc or 'std']
    local nonlin = ({relu = lib.TFL_NONLIN_RELU, relu6 = lib.TFL_NONLIN_RELU6, sigmoid = lib.TFL_NONLIN_SIGMOID})[mconf.nonlinType or 'relu']
    assert(chan and func and nonlin, 'bad normalizeInputChan / normalizeInputFunc / nonlinType')
    opts = ffi.new('tfl_model_opts', {b2i(ic.pDiv), b2i(ic.UDiv), b2i(ic.div), b2i(mconf.normalizeInput ~= false), chan, func,
                                      nonlin, b2i(mconf.addPressureSkip == true)})
  end
  local c_cin, c_cout, c_ks = ffi.new('int32_t[?]', n, cin), ffi.new('int32_t[?]', n, cout), ffi.new('int32_t[?]', n, ks)
  local c_ws, c_bs = ffi.new('const float*[?]', n, ws), ffi.new('const float*[?]', n, bs)
  if opts == nil then
    self.handle = lib.tfl_model_create(ctx, b2i(is3D), n, c_cin, c_cout, c_ks, c_ws, c_bs)
  else
    self.handle = lib.tfl_model_create_opts(ctx, b2i(is3D), n, c_cin, c_cout, c_ks, nil, nil, c_ws, c_bs, opts)
  end
  if self.handle == nil then error(ffi.string(lib.tfl_last_error(ctx)), 2) end
  ffi.gc(self.handle, function(h) lib.tfl_model_destroy(ctx, h) end)
  return self
end

-- model:forward({pDiv, UDiv, flags}) -> {p, U} (lib/model.lua:398, 421-450). The outputs are fresh tensors, like the
-- module outputs of the nngraph model; lib/simulate.lua:270-272 copies them into the state.
function Model:forward(input)
  local pDiv, UDiv, flags = input[1], input[2], input[3]
  self.p = self.p or pDiv.new(); self.U = self.U or UDiv.new()
  self.p:resizeAs(pDiv); self.U:resizeAs(UDiv)
  local ws, n = workspace(pDiv, lib.tfl_model_workspace_floats(self.handle, flags:size(1), flags:size(3), flags:size(4),
                                                               flags:size(5)))
  check(lib.tfl_model_forward(ctx, self.handle, T(pDiv), T(UDiv), T(flags), T(self.p), T(self.U), ws, n, nil, nil, 0, 0, 0))
  self.output = {self.p, self.U}
  return self.output
end
function Model:evaluate() return self end
function Model:cuda() return self end
-- The default 3-D conv path multiplies fp32 values as fp16 hi / lo pairs: an activation beyond 65504 (a blown-up simulation)
-- is clamped and counted, and the NEXT forward / simulate step is refused (check() raises, TFL_ERANGE) until the count has
-- been read. model:rangeErrors() reads and resets it (synchronises); model:rangeFlag() peeks at what the device has reported
-- so far (no synchronisation). TFL_CONV_PATH=winograd in the environment before hip.Model(...) is the strict-fp32 stack.
function Model:rangeErrors() return tonumber(lib.tfl_model_range_errors(ctx, self.handle)) end
function Model:rangeFlag() return tonumber(lib.tfl_model_range_flag(ctx, self.handle)) end

-- ---- tfluids.simulate as ONE native call (lib/simulate.lua:175-327 = fluidnet_amd/csrc/simulate.cpp) -----------------
local plans = setmetatable({}, {__mode = 'k'})    -- BC tensor -> {mask tensor, tfl_bc_plan*}: created once per pair
local function plan(bc, mask)
  if bc == nil or mask == nil then return nil end
  local hit = plans[bc]
  if hit == nil or hit[1] ~= mask then
    local h = lib.tfl_bc_plan_create(ctx, T(bc), T(mask))
    if h == nil then error('tfl_bc_plan_create failed', 3) end
    -- (the finalizer ends in hipFree: a host that records HIP graphs ITSELF must keep collections out of its capture -- the
    -- Python binding learnt that in round 6; Slab:record() captures inside one library call, where no collection can run)
    ffi.gc(h, function(q) lib.tfl_bc_plan_destroy(ctx, q) end)
    hit = {mask, h}
    plans[bc] = hit
  end
  return hit[2]
end
--- Call after editing BC tensors in place (the 2-D demo's interactive edits): plans are keyed by tensor identity.
function M.invalidateBCs() plans = setmetatable({}, {__mode = 'k'}) end

-- the scene's flags as one code byte per cell (include/tfluids_hip.h tfl_wall_plan, round 6): registered once with the context, found
-- again by the flags' address inside tfl_model_begin. lib/simulate.lua never writes batch.flags; call M.invalidateFlags() after
-- editing them (a moving obstacle, the 2-D demo's mouse)
local wallPlans = setmetatable({}, {__mode = 'k'})    -- flags tensor -> tfl_wall_plan*
local function wallPlan(flags)
  local seen = wallPlans[flags]
  if seen == nil then
    wallPlans[flags] = false                           -- first sighting: only remembered (a plan costs an allocation and a device
  elseif seen == false then                            -- synchronisation: not for flags that come fresh with every call)
    local h = lib.tfl_wall_plan_create(ctx, T(flags))
    if h == nil then return end                        -- not fatal: the step decodes the flag words itself
    wallPlans[flags] = ffi.gc(h, function(q) lib.tfl_wall_plan_destroy(ctx, q) end)
  end
end
function M.invalidateFlags() wallPlans = setmetatable({}, {__mode = 'k'}); collectgarbage() end

-- `model` may be a hip.Model, or an nn.gModule (wrapped on first use), or nil for the Jacobi / PCG projections.
local wrapped = setmetatable({}, {__mode = 'k'})
local function sim_args(mconf, batch, model, outputDiv)
  local keep = {}
  local function D(t) local d = T(t); keep[#keep + 1] = d; return d end
  local prm = ffi.new('tfl_sim_params')
  prm.dt = mconf.dt
  prm.advectionMethod = mconf.advectionMethod          -- nil = maccormackOurs
  prm.maccormackStrength = mconf.maccormackStrength or 0.75
  prm.buoyancyScale = mconf.buoyancyScale or 0
  prm.gravityScale = mconf.gravityScale or 0
  local g = mconf.gravity or {0, 1, 0}                  -- lib/simulate.lua:204-211
  if torch.isTensor(g) then g = g:float():totable() end
  prm.gravity[0], prm.gravity[1], prm.gravity[2] = g[1], g[2], g[3]
  prm.vorticityConfinementAmp = mconf.vorticityConfinementAmp or 0
  prm.simMethod = mconf.simMethod                       -- nil = convnet
  prm.maxIter = mconf.maxIter or 0
  prm.pcgPrecond = mconf.pcgPrecond
  prm.outputDiv = b2i(outputDiv)
  local st = ffi.new('tfl_sim_state')
  st.p, st.U, st.flags = D(batch.pDiv), D(batch.UDiv), D(batch.flags)
  if model ~= nil and (mconf.simMethod or 'convnet') == 'convnet' then wallPlan(batch.flags) end
  local dens = batch.density
  local chans = (dens == nil) and {} or (torch.isTensor(dens) and {dens} or dens)   -- RGB table in the 2-D demo
  st.n_density = #chans
  for i, c in ipairs(chans) do st.density[i - 1] = D(c) end
  st.pBC = plan(batch.pBC, batch.pBCInvMask)
  st.UBC = plan(batch.UBC, batch.UBCInvMask)
  if batch.densityBC ~= nil then
    local bcs = torch.isTensor(batch.densityBC) and {batch.densityBC} or batch.densityBC
    local mks = torch.isTensor(batch.densityBCInvMask) and {batch.densityBCInvMask} or batch.densityBCInvMask
    assert(#bcs == #chans, 'density / densityBC channel mismatch')
    for i = 1, #chans do st.densityBC[i - 1] = plan(bcs[i], mks[i]) end
  end
  if model ~= nil and getmetatable(model) ~= Model then
    wrapped[model] = wrapped[model] or M.Model(model)
    model = wrapped[model]
  end
  st.model = model and model.handle or nil
  return prm, st, keep
end

function M.simulate(conf, mconf, batch, model, outputDiv)
  local prm, st, keep = sim_args(mconf, batch, model, outputDiv)
  local ws, n = workspace(batch.UDiv, lib.tfl_simulate_workspace_floats(ctx, prm, st))
  check(lib.tfl_simulate_step(ctx, prm, st, ws, n))
  return keep ~= nil
end

-- ---- z-slab ranks: one LuaJIT process per GPU, grid cut along z, halos over RCCL inside the library -----------------
-- (no counterpart in the reference, which is single-GPU; BASELINE config 5). Rank 0 calls M.rcclUniqueId() and hands
-- the 128-byte string to the other ranks (a file, the launcher's environment, a socket); every rank then builds
--   local slab = M.Slab{zTotal = 256, zFirst = lo, ownLo = c0, ownHi = c1, id = idString, rank = r, world = n}
-- on its LOCAL extended tensors (owned planes + tfl_slab_halo(reach) planes next to each neighbour) and steps with
--   slab:simulate(conf, mconf, batch, model)       -- tfluids.simulate on the slab, bit-equal on the owned planes
--   slab:drain()                                   -- before reading halo planes / at the end
function M.rcclUniqueId()
  local id = ffi.new('char[128]')
  check(lib.tfl_rccl_get_unique_id(ctx, id))
  return ffi.string(id, 128)
end

local Slab = {}
Slab.__index = Slab
function M.Slab(o)
  local self = setmetatable({}, Slab)
  self.desc = ffi.new('tfl_slab')
  self.desc.z_total, self.desc.z_first = o.zTotal, o.zFirst
  self.desc.own_lo, self.desc.own_hi = o.ownLo, o.ownHi
  self.desc.reach, self.desc.overlap = o.reach or 1, b2i(o.overlap)
  -- checkReach: true (default) = a violation is reported by the NEXT step; 'exact' = found before the step, collectively,
  -- nothing written (tfl_slab.check_reach = 2: the error names the reach to lay the slab out for); false = unchecked
  self.desc.check_reach, self.desc.in_flight = (o.checkReach == 'exact') and 2 or b2i(o.checkReach ~= false), 0
  if (o.world or 1) > 1 then
    assert(type(o.id) == 'string' and #o.id == 128, 'Slab: id = the 128 bytes of M.rcclUniqueId() of rank 0')
    self.comm = lib.tfl_rccl_comm_create(ctx, o.id, o.rank, o.world)
    if self.comm == nil then error('tfluids_hip: ' .. ffi.string(lib.tfl_last_error(ctx))) end
    self.comm = ffi.gc(self.comm, function(c) lib.tfl_rccl_comm_destroy(ctx, c) end)
    self.callbacks = lib.tfl_rccl_comm_callbacks(self.comm)
    -- a slab without the strip / interior split has nothing for a transfer to overlap with: RCCL on the step's own stream
    -- (an event hop between two streams costs 12-15 us on the device, eight of them per step)
    if not o.overlap then check(lib.tfl_rccl_comm_set_inline(self.comm, 1)) end
  end
  return self
end
-- Record the rank-step into a HIP graph (tfl_slab_graph_create): call after a few slab:simulate() steps; from then on
-- slab:simulate() is ONE hipGraphLaunch with mconf / batch / model frozen as they were. false + the reason when the step
-- cannot be recorded (the steps stay eager).
function Slab:record()
  if self.st == nil then return false, 'call slab:simulate() at least once first' end
  local g = lib.tfl_slab_graph_create(ctx, self.prm, self.st, self.desc, self.callbacks, self.ws, self.n)
  if g == nil then return false, ffi.string(lib.tfl_last_error(ctx)) end
  self.graph = ffi.gc(g, function(h) lib.tfl_slab_graph_destroy(ctx, h) end)
  return true
end
function Slab:simulate(conf, mconf, batch, model)
  if self.graph ~= nil then check(lib.tfl_slab_graph_step(ctx, self.graph)); return end
  local prm, st, keep = sim_args(mconf, batch, model, false)
  if self.ws == nil then   -- the SAME buffer on every call: messages started by one step are consumed by the next
    self.n = tonumber(lib.tfl_simulate_slab_workspace_floats(ctx, prm, st, self.desc))
    self.buf = batch.UDiv.new():resize(self.n):zero()     -- private: the shared scratch may be re-allocated by other operators
    self.ws = ffi.cast('float*', torch.data(self.buf))
  end
  self.prm, self.st, self.keep = prm, st, keep
  local rc = lib.tfl_simulate_step_slab(ctx, prm, st, self.desc, self.callbacks, self.ws, self.n)
  if rc == -5 then      -- TFL_EREACH (checkReach = 'exact'): every rank is here, nothing has been written
    error(string.format('tfluids_hip: the flow needs a back-trace reach of %d planes: re-cut the local tensors with tfl_slab_halo(%d) halo planes ' ..
                        '(tfl_slab_exchange fetches them) and create the Slab again with reach = %d', lib.tfl_slab_needed_reach(ctx),
                        lib.tfl_slab_needed_reach(ctx), lib.tfl_slab_needed_reach(ctx)))
  end
  check(rc)
end
function Slab:drain()
  if self.st ~= nil and bit.band(self.desc.in_flight, 15) ~= 0 then      -- (bits 0-3: messages in flight)
    check(lib.tfl_slab_drain(ctx, self.st, self.desc, self.callbacks, self.ws, self.n))
  end
end

--- Arithmetic of the LDS-tiled 3-D advection kernels (include/tfluids_hip.h tfl_set_advect_mode): 'exact' (default; bit-equal
--- to the reference CPU path) or 'fast' (the tolerance mode: rel-L2 ~2e-8 against the reference, bar 1e-5).
function M.setAdvectMode(mode)
  assert(mode == 'exact' or mode == 'fast', "advect mode must be 'exact' or 'fast'")
  check(lib.tfl_set_advect_mode(ctx, mode == 'fast' and 1 or 0))
end

--- Route torch.CudaTensor's `.tfluids` method table, tfluids.normalizePressureMean and tfluids.simulate to the MI355X
--- library.
-- @param tfluids the table returned by require('tfluids')
-- @param opts {lib = path to libtfluids_hip.so, device = 0-based HIP device, stream = hipStream_t cdata, advectMode = 'exact' | 'fast',
--              keepLuaSimulate = true to leave lib/simulate.lua's tfluids.simulate in place}
function M.install(tfluids, opts)
  opts = opts or {}
  lib = ffi.load(opts.lib or 'tfluids_hip')
  assert(lib.tfl_abi_version() == 4, 'libtfluids_hip.so / tfluids_hip.lua ABI version mismatch')
  ctx = lib.tfl_create(opts.device or (cutorch.getDevice() - 1))
  assert(ctx ~= nil, 'tfl_create failed')
  if opts.stream then check(lib.tfl_set_stream(ctx, opts.stream)) end
  if opts.advectMode then M.setAdvectMode(opts.advectMode) end
  local mt = getmetatable(torch.CudaTensor)  -- luaT_registeratname(L, tbl, "tfluids") put the table here
  mt.tfluids = mt.tfluids or {}
  for name, fn in pairs(ops) do mt.tfluids[name] = fn end
  rawset(tfluids, 'normalizePressureMean', M.normalizePressureMean)
  if not opts.keepLuaSimulate then rawset(tfluids, 'simulate', M.simulate) end
  tfluids.withHIP = true
  return M
end

function M.synchronize() check(lib.tfl_synchronize(ctx)) end
function M.traceErrors() return tonumber(lib.tfl_trace_errors(ctx)) end

M.ops = ops
return M
