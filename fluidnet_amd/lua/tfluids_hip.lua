-- tfluids_hip.lua -- LuaJIT FFI binding of libtfluids_hip.so (include/tfluids_hip.h).
--
-- Drop-in for the native half of torch/tfluids: after `require('tfluids')`, calling
--   require('tfluids_hip').install(tfluids)
-- replaces the per-tensor-type C tables that torch/tfluids/init.lua dispatches through
-- (`X.tfluids.<op>(...)`, registered by generic/tfluids.cc:927-957 / generic/tfluids.cu:1932-1962)
-- with functions of the SAME positional signatures that forward to the MI355X library. init.lua's
-- wrappers (argument checks, getTempStorage, copy-back) and lib/simulate.lua then run unchanged.
--
-- NOTE: LuaJIT / Torch7 are not available in the build container (SURVEY.md section 7), so this file
-- is reviewed against include/tfluids_hip.h but not executed by the test-suite; the tested host mirror
-- of the same calls is fluidnet_amd/tfluids.py. Tensors must be contiguous float tensors whose
-- storage is device memory (cutorch CudaTensor on a ROCm build of cutorch).
local ffi = require('ffi')

ffi.cdef[[
typedef struct tfl_tensor { float* data; int32_t B, C, Z, Y, X; } tfl_tensor;
typedef struct tfl_ctx tfl_ctx;
tfl_ctx* tfl_create(int device);
void tfl_destroy(tfl_ctx* ctx);
int tfl_set_stream(tfl_ctx* ctx, void* hip_stream);
const char* tfl_last_error(const tfl_ctx* ctx);
int tfl_synchronize(tfl_ctx* ctx);
int tfl_advectScalar(tfl_ctx*, float dt, const tfl_tensor* s, const tfl_tensor* U, const tfl_tensor* flags,
                     const tfl_tensor* fwd, const tfl_tensor* bwd, int is3D, const char* method,
                     const tfl_tensor* fwdPos, const tfl_tensor* bwdPos, int boundaryWidth,
                     int sampleOutsideFluid, float maccormackStrength, const tfl_tensor* sDst);
int tfl_advectVel(tfl_ctx*, float dt, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* fwd,
                  const tfl_tensor* bwd, int is3D, const char* method, int boundaryWidth,
                  float maccormackStrength, const tfl_tensor* UDst);
int tfl_setWallBcsForward(tfl_ctx*, const tfl_tensor* U, const tfl_tensor* flags, int is3D);
int tfl_velocityDivergenceForward(tfl_ctx*, const tfl_tensor* U, const tfl_tensor* flags,
                                  const tfl_tensor* UDiv, int is3D);
int tfl_velocityUpdateForward(tfl_ctx*, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* p,
                              int is3D);
int tfl_vorticityConfinement(tfl_ctx*, const tfl_tensor* U, const tfl_tensor* flags, float strength,
                             const tfl_tensor* centered, const tfl_tensor* curl, const tfl_tensor* curlNorm,
                             const tfl_tensor* force, int is3D);
int tfl_addBuoyancy(tfl_ctx*, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* density,
                    const float gravity[3], float* strengthTmp, float dt, int is3D);
int tfl_addGravity(tfl_ctx*, const tfl_tensor* U, const tfl_tensor* flags, const float gravity[3], float dt,
                   int is3D, float* forceTmp);
int tfl_emptyDomain(tfl_ctx*, const tfl_tensor* flags, int is3D, int bnd);
int tfl_flagsToOccupancy(tfl_ctx*, const tfl_tensor* flags, const tfl_tensor* occupancy);
int tfl_velocityDivergenceBackward(tfl_ctx*, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* gradOutput,
                                   int is3D, const tfl_tensor* gradU);
int tfl_velocityUpdateBackward(tfl_ctx*, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* p,
                               const tfl_tensor* gradOutput, int is3D, const tfl_tensor* gradP);
int tfl_volumetricUpSamplingNearestForward(tfl_ctx*, int ratio, const tfl_tensor* input, const tfl_tensor* output);
int tfl_volumetricUpSamplingNearestBackward(tfl_ctx*, int ratio, const tfl_tensor* input, const tfl_tensor* gradOutput,
                                            const tfl_tensor* gradInput);
int tfl_solveLinearSystemJacobi(tfl_ctx*, const tfl_tensor* p, const tfl_tensor* flags, const tfl_tensor* div,
                                const tfl_tensor* pPrev, const tfl_tensor* pDelta,
                                const tfl_tensor* pDeltaNorm, int is3D, float pTol, int maxIter, int verbose,
                                float* residual);
/* the whole step in one call (csrc/simulate.cpp) */
typedef struct tfl_bc_plan tfl_bc_plan;
typedef struct tfl_model tfl_model;
tfl_bc_plan* tfl_bc_plan_create(tfl_ctx*, const tfl_tensor* bc, const tfl_tensor* invMask);
void tfl_bc_plan_destroy(tfl_ctx*, tfl_bc_plan*);
typedef struct tfl_sim_params { float dt; const char* advectionMethod; float maccormackStrength, buoyancyScale,
  gravityScale; float gravity[3]; float vorticityConfinementAmp; const char* simMethod; int32_t maxIter;
  const char* pcgPrecond; int32_t outputDiv; } tfl_sim_params;
typedef struct tfl_sim_state { const tfl_tensor *p, *U, *flags; int32_t n_density; const tfl_tensor* density[8];
  const tfl_bc_plan *pBC, *UBC; const tfl_bc_plan* densityBC[8]; tfl_model* model; } tfl_sim_state;
int64_t tfl_simulate_workspace_floats(tfl_ctx*, const tfl_sim_params*, const tfl_sim_state*);
int tfl_simulate_step(tfl_ctx*, const tfl_sim_params*, const tfl_sim_state*, float* workspace, int64_t workspace_floats);
]]

local M = {}
local lib, ctx

local function check(rc)
  if rc ~= 0 then error(ffi.string(lib.tfl_last_error(ctx)), 3) end
end

-- 5-D (or 1-D for the batch-norm scratch) contiguous float tensor -> tfl_tensor
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

local ops = {}
function ops.advectScalar(dt, s, U, flags, fwd, bwd, is3D, method, fwdPos, bwdPos, bnd, outside, strength, sDst)
  check(lib.tfl_advectScalar(ctx, dt, T(s), T(U), T(flags), T(fwd), T(bwd), is3D and 1 or 0, method,
                             T(fwdPos), T(bwdPos), bnd, outside and 1 or 0, strength, T(sDst)))
end
function ops.advectVel(dt, U, flags, fwd, bwd, is3D, method, bnd, strength, UDst)
  check(lib.tfl_advectVel(ctx, dt, T(U), T(flags), T(fwd), T(bwd), is3D and 1 or 0, method, bnd, strength,
                          T(UDst)))
end
function ops.setWallBcsForward(U, flags, is3D)
  check(lib.tfl_setWallBcsForward(ctx, T(U), T(flags), is3D and 1 or 0))
end
function ops.velocityDivergenceForward(U, flags, UDiv, is3D)
  check(lib.tfl_velocityDivergenceForward(ctx, T(U), T(flags), T(UDiv), is3D and 1 or 0))
end
function ops.velocityUpdateForward(U, flags, p, is3D)
  check(lib.tfl_velocityUpdateForward(ctx, T(U), T(flags), T(p), is3D and 1 or 0))
end
function ops.vorticityConfinement(U, flags, strength, centered, curl, curlNorm, force, is3D)
  check(lib.tfl_vorticityConfinement(ctx, T(U), T(flags), strength, T(centered), T(curl), T(curlNorm),
                                     T(force), is3D and 1 or 0))
end
function ops.addBuoyancy(U, flags, density, gravity, strengthTmp, dt, is3D)
  check(lib.tfl_addBuoyancy(ctx, T(U), T(flags), T(density), vec3(gravity), nil, dt, is3D and 1 or 0))
end
function ops.addGravity(U, flags, gravity, dt, is3D, forceTmp)
  check(lib.tfl_addGravity(ctx, T(U), T(flags), vec3(gravity), dt, is3D and 1 or 0, nil))
end
function ops.emptyDomain(flags, is3D, bnd)
  check(lib.tfl_emptyDomain(ctx, T(flags), is3D and 1 or 0, bnd))
end
function ops.flagsToOccupancy(flags, occupancy)
  check(lib.tfl_flagsToOccupancy(ctx, T(flags), T(occupancy)))
end
function ops.velocityDivergenceBackward(U, flags, gradOutput, is3D, gradU)
  check(lib.tfl_velocityDivergenceBackward(ctx, T(U), T(flags), T(gradOutput), is3D and 1 or 0, T(gradU)))
end
function ops.velocityUpdateBackward(U, flags, p, gradOutput, is3D, gradP)
  check(lib.tfl_velocityUpdateBackward(ctx, T(U), T(flags), T(p), T(gradOutput), is3D and 1 or 0, T(gradP)))
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
                                        is3D and 1 or 0, pTol, maxIter, verbose and 1 or 0, res))
  return res[0]
end

--- Route torch.CudaTensor's `.tfluids` method table to the MI355X library.
-- @param tfluids the table returned by require('tfluids')
-- @param opts {lib = path to libtfluids_hip.so, device = 0-based HIP device, stream = hipStream_t cdata}
function M.install(tfluids, opts)
  opts = opts or {}
  lib = ffi.load(opts.lib or 'tfluids_hip')
  ctx = lib.tfl_create(opts.device or (cutorch.getDevice() - 1))
  assert(ctx ~= nil, 'tfl_create failed')
  if opts.stream then lib.tfl_set_stream(ctx, opts.stream) end
  local mt = getmetatable(torch.CudaTensor)  -- luaT_registeratname(L, tbl, "tfluids") put the table here
  mt.tfluids = mt.tfluids or {}
  for name, fn in pairs(ops) do mt.tfluids[name] = fn end
  tfluids.withHIP = true
  return M
end

M.ops = ops
return M
