/*
 * tfluids_hip.h -- C ABI of the MI355X-native tfluids hot path (libtfluids_hip.so).
 *
 * This is the drop-in boundary for FluidNet's native layer. In the reference every operator is a
 * Lua-C function `static int tfluids_<Real>Main_<op>(lua_State*)` registered on the tensor
 * metatable (torch/tfluids/generic/tfluids.cc:927-957, generic/tfluids.cu:1932-1962) and called
 * from torch/tfluids/init.lua as `X.tfluids.<op>(positional args)`. Each tfl_<op> below takes the
 * SAME positional arguments in the SAME order (the reference call site is cited per function),
 * with THTensor* replaced by `const tfl_tensor*` (raw device pointer + the five sizes; contiguous
 * [B][C][Z][Y][X] fp32, x fastest -- third_party/grid.h:68-78) and lua numbers/booleans replaced by
 * float/int. No torch, Lua or C++ types cross this boundary.
 *
 * Semantics shared by every entry point (mirroring the reference's CUDA path):
 *   - asynchronous: work is enqueued on the context's HIP stream and the call returns without
 *     synchronising (generic/tfluids.cu:106 uses THCState_getCurrentStream the same way); the
 *     only exception is tfl_solveLinearSystemJacobi with pTol > 0, which must read the residual
 *     back every iteration exactly like the reference (generic/tfluids.cu:1886);
 *   - the caller owns every buffer; nothing is allocated, freed, resized or retained;
 *   - "temp" tensors (fwd, bwd, fwdPos, bwdPos, centered, curl, ...) are accepted because the
 *     reference wrappers pass them (init.lua:35-64 getTempStorage); their contents are undefined
 *     on entry AND on exit, exactly as in the reference. This implementation fuses sweeps and does
 *     not touch all of them;
 *   - return value: 0 on success, a negative tfl_status on error; tfl_last_error(ctx) returns a
 *     message. Errors never longjmp/throw across the ABI (the reference raises luaL_error /
 *     THError: init.lua asserts, third_party/tfluids.cc:441-466).
 */
#ifndef TFLUIDS_HIP_H_
#define TFLUIDS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFL_ABI_VERSION 4   /* 3: tfl_comm starts with its own size; tfl_set_advect_mode, tfl_stream_copy. 4: tfl_comm.capturable,
                              tfl_slab_graph_* (the rank-step as one HIP-graph launch) */

typedef enum tfl_status {
  TFL_OK = 0,
  TFL_EINVAL = -1,       /* bad argument (shape / channel / null pointer / unknown method) */
  TFL_EHIP = -2,         /* HIP runtime error (message carries hipGetErrorString) */
  TFL_EUNSUPPORTED = -3, /* valid in the reference but not built here */
  TFL_EREACH = -5,       /* tfl_simulate_step_slab with check_reach = 2: the flow is faster than the slab's halo allows; NOTHING of
                            the step has been written -- lay the slab out for tfl_slab_needed_reach() and call again */
  TFL_ERANGE = -4        /* an earlier forward pass of the model left the fp16 range of the default 3-D conv path (see
                            tfl_model_range_errors): refused until that count has been read */
} tfl_status;

/* A contiguous fp32 5-D tensor resident in HBM: [B][C][Z][Y][X], x fastest.
 * Supported sizes: the elements of ONE batch item fit a signed 32-bit index, C*Z*Y*X < 2^31 (element strides are
 * int32 inside the kernels; the batch offset is applied in 64 bits), e.g. a 3-channel velocity up to 894^3. Larger
 * tensors are refused with TFL_EUNSUPPORTED, never computed wrongly. The LDS-tiled 3-D advection kernels further use
 * 24-bit plane strides (4*X*Y < 2^24, i.e. X*Y <= 2047^2) and 32-bit byte offsets over the three velocity channels
 * (12*Z*Y*X < 2^32, i.e. up to ~710^3) and fall back to the gather kernels beyond that. */
typedef struct tfl_tensor {
  float* data;
  int32_t B, C, Z, Y, X;
} tfl_tensor;

/* Manta cell types stored as floats in `flags` (third_party/cell_type.h:22-33; exported to Lua as
 * tfluids.CellType by init.cu:108-120). */
enum {
  TFL_TypeNone = 0, TFL_TypeFluid = 1, TFL_TypeObstacle = 2, TFL_TypeEmpty = 4,
  TFL_TypeInflow = 8, TFL_TypeOutflow = 16, TFL_TypeOpen = 32, TFL_TypeStick = 128
};

typedef struct tfl_ctx tfl_ctx;

/* ---- context ------------------------------------------------------------------------------ */
/* Replaces THCState (device + current stream). `device` is a HIP device ordinal. */
tfl_ctx* tfl_create(int device);
void tfl_destroy(tfl_ctx* ctx);
/* `hip_stream` is a hipStream_t (NULL = the legacy default stream). */
int tfl_set_stream(tfl_ctx* ctx, void* hip_stream);
const char* tfl_last_error(const tfl_ctx* ctx);
int tfl_abi_version(void);
/* Arithmetic of the LDS-tiled 3-D advection kernels (advectVel / advectScalar, methods eulerOurs and maccormackOurs):
 *   TFL_ADVECT_EXACT (default)  every operation in the reference's order and rounding: results are bit-equal to the
 *                               reference CPU path (third_party/tfluids.cc:152-234,594-774), which the golden tests pin;
 *   TFL_ADVECT_FAST             the tolerance mode: trace end point = centre + displacement (no normalise / rescale round
 *                               trip), trilinear samples as contracted a + t (b - a), MacCormack correction in fp32.
 *                               Differs from the reference by a few ulp per cell; held to the north-star's rel-L2 <= 1e-5
 *                               by tests/test_hip_fullsize.py. Lanes off the fast path (obstacle neighbours, fast flow)
 *                               run the exact generic code in both modes.
 * A context starts in the mode named by the environment variable TFL_ADVECT_MODE ("fast" | "exact"; unset = exact). */
enum { TFL_ADVECT_EXACT = 0, TFL_ADVECT_FAST = 1 };
int tfl_set_advect_mode(tfl_ctx* ctx, int mode);
int tfl_get_advect_mode(const tfl_ctx* ctx);
/* Blocks until the context's stream is idle (cutorch.synchronize(), simulate.lua:258). */
int tfl_synchronize(tfl_ctx* ctx);
/* Number of back-traces that hit one of calcLineTrace's invariant-violation paths since the last
 * call (the reference CPU build THErrors there, calc_line_trace.cc:325-330,410,421; its CUDA build
 * silently substitutes). Synchronises the stream. Diagnostic only. */
int64_t tfl_trace_errors(tfl_ctx* ctx);

/* Built-in per-kernel timing, the analogue of tfluids.profilePressure (lib/simulate.lua:254-260,
 * 306-318) at kernel granularity: between begin and end every kernel this library launches from the
 * calling thread is bracketed by HIP events on its launch stream. tfl_profile_end synchronises,
 * writes a JSON object {"kernel": {"calls": n, "ms": total}, ...} into buf (truncated to cap) and
 * returns the number of distinct kernels, or a negative tfl_status. */
int tfl_profile_begin(tfl_ctx* ctx);
int tfl_profile_end(tfl_ctx* ctx, char* buf, int64_t cap);

/* ---- operators (one per row of SURVEY.md section 8b) ------------------------------------- */

/* init.lua:142-144 -> third_party/tfluids.cc:415-588 | tfluids.cu:524-633.
 * method: "euler" | "maccormack" | "eulerOurs" | "rk2Ours" | "rk3Ours" | "maccormackOurs"
 * (generic/advect_type.cc:18-37). boundaryWidth is parsed and ignored like the reference
 * (tfluids.cc:436,467: bnd = 1). s_dst must not alias s. */
int tfl_advectScalar(tfl_ctx* ctx, float dt, const tfl_tensor* s, const tfl_tensor* U,
                     const tfl_tensor* flags, const tfl_tensor* fwd, const tfl_tensor* bwd,
                     int is3D, const char* method, const tfl_tensor* fwdPos,
                     const tfl_tensor* bwdPos, int boundaryWidth, int sampleOutsideFluid,
                     float maccormackStrength, const tfl_tensor* sDst);

/* init.lua:212-213 -> third_party/tfluids.cc:776-920 | tfluids.cu:876-963.
 * rk2Ours / rk3Ours map to maccormackOurs like the reference (tfluids.cc:799-802). */
int tfl_advectVel(tfl_ctx* ctx, float dt, const tfl_tensor* U, const tfl_tensor* flags,
                  const tfl_tensor* fwd, const tfl_tensor* bwd, int is3D, const char* method,
                  int boundaryWidth, float maccormackStrength, const tfl_tensor* UDst);

/* init.lua:246 -> third_party/tfluids.cc:926-1002 | tfluids.cu:969-1046 (in place on U). */
int tfl_setWallBcsForward(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags, int is3D);

/* init.lua:278 -> third_party/tfluids.cc:1008-1066 | tfluids.cu:1052-1105. */
int tfl_velocityDivergenceForward(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                                  const tfl_tensor* UDiv, int is3D);

/* init.lua:346 -> third_party/tfluids.cc:1072-1156 | tfluids.cu:1111-1195 (in place on U). */
int tfl_velocityUpdateForward(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                              const tfl_tensor* p, int is3D);

/* init.lua:428-429 -> third_party/tfluids.cc:1341-1458 | tfluids.cu:1355-1497 (in place on U).
 * curl always has 3 channels (tfluids.cc:1363). */
int tfl_vorticityConfinement(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                             float strength, const tfl_tensor* centered, const tfl_tensor* curl,
                             const tfl_tensor* curlNorm, const tfl_tensor* force, int is3D);
/* The same operator out of place: U = USrc + confinement(USrc), every cell of U (of the z-window) written; U must not alias
 * USrc. On a 3-D grid of 2 M cells per batch item or more (64+ planes deep) this is ONE fused z-marched launch that keeps the centred velocities,
 * curl and |curl| in LDS (vorticity.hip k_vort_pipe, or k_vort_fused where the device cannot hold its block: 72 -> ~30 bytes
 * per cell of HBM traffic; TFL_VORT_FUSED=1|0 in the environment forces the route), bit-equal to tfl_vorticityConfinement;
 * smaller and 2-D grids run the two launches reading USrc and writing U (or, misaligned, copy first), for which curl (3
 * channels) / curlNorm are the scratch -- under a z-window that route needs its curl on a window one plane wider each way,
 * which tfl_simulate_step_slab arranges with tfl_set_stages; a host of its own should window only the fused route.
 * tfl_simulate_step uses the entry to fold simulate()'s `U:copy(advected)` and the confinement into one pass. No reference
 * counterpart (the reference op is in place). */
int tfl_vorticityConfinementFrom(tfl_ctx* ctx, const tfl_tensor* USrc, const tfl_tensor* U, const tfl_tensor* flags,
                                 float strength, const tfl_tensor* curl, const tfl_tensor* curlNorm, int is3D);

/* init.lua:469 -> third_party/tfluids.cc:1162-1233 | tfluids.cu:1201-1273 (in place on U).
 * gravity: 3 floats in HOST memory (the Lua wrapper passes a 3-element tensor, init.lua:455-458);
 * strengthTmp (a 3-float device scratch in the reference, tfluids.cu:1261-1265) is accepted and
 * unused: the strength travels as kernel arguments. */
int tfl_addBuoyancy(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                    const tfl_tensor* density, const float gravity[3], float* strengthTmp,
                    float dt, int is3D);

/* tfl_addBuoyancy out of place: U = USrc + buoyancy(flags, density), every cell of U written. simulate() uses it
 * with USrc = advectVel's output buffer, which makes the reference's `U:copy(UDst)` after the advection
 * (init.lua:216-218) part of this pass instead of a sweep of its own. USrc == U is tfl_addBuoyancy. */
int tfl_addBuoyancyFrom(tfl_ctx* ctx, const tfl_tensor* USrc, const tfl_tensor* U, const tfl_tensor* flags,
                        const tfl_tensor* density, const float gravity[3], float dt, int is3D);

/* init.lua:505 -> third_party/tfluids.cc:1239-1306 | tfluids.cu:1279-1349 (in place on U). */
int tfl_addGravity(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                   const float gravity[3], float dt, int is3D, float* forceTmp);

/* init.lua:552 -> generic/tfluids.cc:136-167 | generic/tfluids.cu:314-353. */
int tfl_emptyDomain(tfl_ctx* ctx, const tfl_tensor* flags, int is3D, int bnd);

/* init.lua:574 -> generic/tfluids.cc:173-210 | generic/tfluids.cu:355-397. Cells that are neither
 * exactly Fluid nor Obstacle become -1 (the CUDA behaviour); the CPU reference raises there. */
int tfl_flagsToOccupancy(tfl_ctx* ctx, const tfl_tensor* flags, const tfl_tensor* occupancy);

/* init.lua:583-595 `tfluids.rectangularBlur(src, blurRad, is3D, dst)` -> generic/tfluids.cc:642-760 (CPU only in the
 * reference): separable box blur of radius blurRad with clamped edges over every [B][C] field, each line a running sum
 * in the reference's order (bit-exact). tmp: scratch of src's size (the Lua wrapper's getTempStorage). */
int tfl_rectangularBlur(tfl_ctx* ctx, const tfl_tensor* src, int blurRad, int is3D, const tfl_tensor* dst,
                        const tfl_tensor* tmp);
/* init.lua:603-613 `tfluids.signedDistanceField(flags, searchRad, is3D, dst)` -> generic/tfluids.cc:766-821 |
 * generic/tfluids.cu:690-747: distance to the nearest obstacle cell within a (2 searchRad + 1)^dim window, clamped to
 * searchRad, 0 inside obstacles (the loss weighting of lib/modules/fluid_criterion.lua:149). */
int tfl_signedDistanceField(tfl_ctx* ctx, const tfl_tensor* flags, int searchRad, int is3D, const tfl_tensor* dst);

/* init.lua:645-677 `tfluids.solveLinearSystemPCG(p, flags, div, is3D, tol, maxIter, precondType, verbose)` ->
 * tfluids_CudaMain_solveLinearSystemPCG, generic/tfluids.cu:1257-1759 (CUDA only in the reference; the baseline
 * "exact" pressure solver and the accuracy yard-stick of the paper). p is zeroed, then every connected fluid
 * component (generic/find_connected_fluid_components.cc) of every batch element is solved by (P)CG from x = 0 with
 * the Laplacian of setupLaplacian (generic/tfluids.cu:904-1093), the component's mean is removed and the result
 * scattered into p; components of one cell are skipped, of fewer than five use no preconditioner.
 * precondType: "none" | "ilu0" | "ic0" (init.lua default "ic0"); tol (default 1e-6) on ||r||; maxIter (default
 * 1000); *residual = max over the solves of the final ||r|| (-inf if there was nothing to solve).
 * The reference keeps its temporaries in the tfluids._tmpPCG table; here the caller passes a workspace of
 * tfl_pcg_workspace_floats(Z, Y, X) floats (8-byte aligned). Synchronises the stream (as the reference does after
 * every dot product). On 3-D grids the IC(0) / ILU(0) triangular solves run as pipelined wavefronts whose sub-boxes wait
 * for each other (all of them must be resident on the GPU at once); if one times out (~1 s: something else holds part of
 * the GPU) the solve is repeated with one launch per hyperplane. TFL_PCG_HYPERPLANES=1 in the environment selects that
 * schedule from the start. Returns TFL_EINVAL for a fluid cell on the domain border (the reference raises). */
int64_t tfl_pcg_workspace_floats(int32_t Z, int32_t Y, int32_t X);
int tfl_solveLinearSystemPCG(tfl_ctx* ctx, const tfl_tensor* p, const tfl_tensor* flags, const tfl_tensor* div,
                             int is3D, const char* precondType, float tol, int maxIter, int verbose,
                             float* workspace, int64_t workspace_floats, float* residual);

/* init.lua:747-764 `tfluids.normalizePressureMean(p, flags, is3D)` -> tfluids_<Real>Main_normalizePressureMean,
 * generic/tfluids.cc:845-925 (CPU only in the reference: CUDA tensors are copied to the host and back): subtracts
 * from p, per batch element, the mean of p over each connected component of fluid cells (non-fluid cells untouched).
 * Runs on the device here (the component labelling of tfl_solveLinearSystemPCG). The reference's `inds` IntTensor
 * temp becomes a workspace of tfl_normalize_workspace_floats(Z, Y, X) floats, 8-byte aligned. */
int64_t tfl_normalize_workspace_floats(int32_t Z, int32_t Y, int32_t X);
int tfl_normalizePressureMean(tfl_ctx* ctx, const tfl_tensor* p, const tfl_tensor* flags, int is3D, float* workspace,
                              int64_t workspace_floats);

/* init.lua:726-727 -> generic/tfluids.cu:1765-1927 (the reference has no CPU version).
 * p is overwritten (initial guess is zero like the reference, :1869-1872). pPrev is scratch;
 * pDelta / pDeltaNorm are accepted and unused (the residual is reduced on the fly). The final
 * residual max_b ||p - pPrev||_2 is written to *residual when non-NULL (one host sync at the
 * end; every iteration when pTol > 0, like the reference's maxall at :1886). */
int tfl_solveLinearSystemJacobi(tfl_ctx* ctx, const tfl_tensor* p, const tfl_tensor* flags,
                                const tfl_tensor* div, const tfl_tensor* pPrev,
                                const tfl_tensor* pDelta, const tfl_tensor* pDeltaNorm, int is3D,
                                float pTol, int maxIter, int verbose, float* residual);

/* ---- training-side callers and the multi-resolution resampler --------------------- */
/* init.lua:310-313 -> generic/tfluids.cc:49-130 | generic/tfluids.cu:224-291. gradU is overwritten. */
int tfl_velocityDivergenceBackward(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags,
                                   const tfl_tensor* gradOutput, int is3D, const tfl_tensor* gradU);
/* init.lua:358-383 -> generic/tfluids.cc:216-344 | generic/tfluids.cu:407-513. gradP is overwritten; each word
 * gathers its contributions in the reference's serial order (deterministic; the reference scatters atomically). */
int tfl_velocityUpdateBackward(tfl_ctx* ctx, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* p,
                               const tfl_tensor* gradOutput, int is3D, const tfl_tensor* gradP);
/* tfluids.SetWallBcs:updateGradInput (tfluids/set_wall_bcs.lua:50-66): gradU = mask * gradOutput with mask =
 * setWallBcsForward(ones, flags), i.e. the forward operator applied to the incoming gradient (the module's own comment:
 * "copy the gradOutput and treat it as an input velocity and call the forward function"); the gradient w.r.t. flags is
 * zero. gradU may alias gradOutput. */
int tfl_setWallBcsBackward(tfl_ctx* ctx, const tfl_tensor* flags, const tfl_tensor* gradOutput, int is3D,
                           const tfl_tensor* gradU);
/* init.lua:618-627 -> generic/tfluids.cc:509-633 | generic/tfluids.cu:516-686. */
int tfl_volumetricUpSamplingNearestForward(tfl_ctx* ctx, int ratio, const tfl_tensor* input,
                                           const tfl_tensor* output);
int tfl_volumetricUpSamplingNearestBackward(tfl_ctx* ctx, int ratio, const tfl_tensor* input,
                                            const tfl_tensor* gradOutput, const tfl_tensor* gradInput);

/* ---- the pressure-projection ConvNet (lib/model.lua `default` model, forward only) ---------- */
/* In the reference the projection is `model:forward({pDiv, UDiv, flags})` on an nngraph of cudnn
 * convolutions and tfluids nn.Modules (lib/simulate.lua:262-272, lib/model.lua:27-401), not a
 * tfluids op, so the boundary here is one "model" object + one forward call. */
typedef struct tfl_model tfl_model;

/* Builds the `default` topology: nlayers convolutions (stride 1, zero pad (k-1)/2, cross-correlation,
 * lib/model_utils.lua:80-116), ReLU after all but the last; input channels {pDiv/scale, div/scale,
 * occupancy} (cin[0] must be 3; other input sets: tfl_model_create_opts), cout[nlayers-1] must be 1; at most 64 channels per layer. weights[l] is HOST memory laid out like
 * cudnn.{Spatial,Volumetric}Convolution.weight: [cout][cin][k(z)][k(y)][k(x)] (no z for 2-D);
 * biases[l] is [cout]. The model copies and re-lays-out the weights; the caller keeps ownership.
 * Returns NULL on error (see tfl_last_error). */
tfl_model* tfl_model_create(tfl_ctx* ctx, int is3D, int nlayers, const int32_t* cin, const int32_t* cout,
                            const int32_t* ksize, const float* const* weights,
                            const float* const* biases);
/* The same with the two extra per-layer knobs of lib/model.lua's layer tables (:163-178, :211-218): pool[l] = psize
 * (2: cudnn 2x average pooling after the layer's ReLU, model_utils.lua:184-208) and up[l] = usize (2: the layer is an
 * nn.{Spatial,Volumetric}ConvolutionUpsample, lib/modules/{spatial,volumetric}_convolution_upsample.lua -- weights[l] then has
 * cout[l] * 2^dim output channels, channel index = o * 2^dim + sub-position, and the result is pixel-shuffled to twice
 * the resolution). NULL = all 1. This covers the `tog` model types (2-D: 16,32,32,64,64,32,1 with k 5,5,5,5,1,1,3;
 * 3-D: 16,16,16,16,32,32,1 with k 3,3,3,3,1,1,3); such models run through the shape-generic kernels, and the grid
 * must be divisible by the product of the pooling factors. */
tfl_model* tfl_model_create_ex(tfl_ctx* ctx, int is3D, int nlayers, const int32_t* cin, const int32_t* cout,
                               const int32_t* ksize, const int32_t* pool, const int32_t* up,
                               const float* const* weights, const float* const* biases);
/* The mconf switches of lib/model.lua:27-160, 356-387 that change the FORWARD graph (fields of torch/lib/default_conf.lua:
 * 60-100 with the same meaning; the defaults there are what tfl_model_create[_ex] builds and what NULL means here):
 *   in_pDiv / in_UDiv / in_div  inputChannels (flags always feed the net, model.lua:81); the net input is the join, in this
 *                               order, of pDiv/scale, SetWallBcs(UDiv)/scale (2 or 3 channels), div/scale, occupancy
 *                               (model.lua:130-148), so cin[0] = in_pDiv + in_UDiv*(2|3) + in_div + 1
 *   normalize, norm_chan, norm_func  normalizeInput / normalizeInputChan / normalizeInputFunc: scale = per-sample 'std'
 *                               (n-1) or 'norm' (l2) of SetWallBcs(UDiv), pDiv or div; 0 = no scaling
 *   nonlin                      nonlinType after every layer but the last (model_utils.lua:20-34)
 *   pressure_skip               addPressureSkip: pDiv/scale joins the input of the LAST layer as its last channel
 *                               (model.lua:356-360): cin[nlayers-1] = cout[nlayers-2] + 1
 * Models with non-default options run through the shape-generic kernels and are not available to the z-slab step. */
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
/* The 3-D default topology's convolution stack runs as a split-operand fp16 MFMA implicit GEMM (csrc/conv_mfma16.hip):
 * every fp32 operand travels as two fp16 halves, which ends at |x| = 65504. An activation above that is clamped and the
 * forward pass counts it; this returns the number of thread blocks that clamped since the last call and resets the
 * count (synchronises the context's stream). 0 for every working simulation: the net's input is divided by the
 * velocity's standard deviation. Always 0 on the fp32 paths (TFL_CONV_PATH=winograd|mfma|direct). -1 on error. */
int64_t tfl_model_range_errors(tfl_ctx* ctx, tfl_model* model);
/* The same count WITHOUT a stream synchronisation, as far as the device has reported it: the projection kernel at the end of
 * a forward pass copies a non-zero count into pinned host memory, and this reads that word (not reset; tfl_model_range_errors
 * resets it). While it is non-zero, tfl_model_forward / tfl_model_begin / tfl_simulate_step[_slab] return TFL_ERANGE instead
 * of stepping on from a clamped pressure -- the reference's fp32 cuDNN would have carried values up to 3e38; create the model
 * under TFL_CONV_PATH=winograd for a strict-fp32 stack without a range limit. 0 on the fp32 paths. -1 on error. */
int64_t tfl_model_range_flag(tfl_ctx* ctx, tfl_model* model);
/* Scratch floats tfl_model_forward needs for a [B][.][Z][Y][X] grid. */
int64_t tfl_model_workspace_floats(const tfl_model* model, int B, int Z, int Y, int X);
/* {pOut, UOut} = model:forward({pDiv, UDiv, flags}) (lib/model.lua:398, 421-450). Inputs are not
 * modified; pOut may alias pDiv and UOut may alias UDiv (simulate.lua:270-272 copies the prediction
 * back into the state, which this makes free). When UBC/UBCInvMask are non-NULL the tail of
 * simulate() -- U = U*UBCInvMask + UBC and, if doClamp, clamp(U, lo, hi) (simulate.lua:321-326) -- is
 * fused into the last kernel. */
int tfl_model_forward(tfl_ctx* ctx, tfl_model* model, const tfl_tensor* pDiv, const tfl_tensor* UDiv,
                      const tfl_tensor* flags, const tfl_tensor* pOut, const tfl_tensor* UOut,
                      float* workspace, int64_t workspace_floats, const tfl_tensor* UBC,
                      const tfl_tensor* UBCInvMask, int doClamp, float lo, float hi);

/* The same forward in two halves, for z-slab decomposition (BASELINE config 5): `begin` applies the
 * wall BCs, computes the divergence and reduces sum(u), sum(u^2) of SetWallBcs(UDiv) over the owned
 * z-planes [zlo, zhi) into stats[2*B] (device doubles; NULL = the model's internal buffer); the caller
 * all-reduces stats across ranks; `finish` runs the rest with `count` = the GLOBAL number of velocity
 * samples per batch item (C*Z*Y*X of the whole grid). tfl_model_forward == begin(0, Z) + finish.
 * Models created with non-default tfl_model_opts (another normaliser channel / function, normalisation off) form their
 * scale from statistics the model computes itself inside `finish`: pass stats = NULL there (a caller-supplied `stats`
 * is refused with TFL_EUNSUPPORTED rather than silently ignored), and `count` is not used. */
int tfl_model_begin(tfl_ctx* ctx, tfl_model* model, const tfl_tensor* UDiv, const tfl_tensor* flags,
                    const tfl_tensor* UOut, float* workspace, int64_t workspace_floats, int zlo, int zhi,
                    double* stats);
int tfl_model_finish(tfl_ctx* ctx, tfl_model* model, const tfl_tensor* pDiv, const tfl_tensor* flags,
                     const tfl_tensor* pOut, const tfl_tensor* UOut, float* workspace,
                     int64_t workspace_floats, const double* stats, double count, const tfl_tensor* UBC,
                     const tfl_tensor* UBCInvMask, int doClamp, float lo, float hi);

/* A z-slab rank holds only part of the grid, but tfluids.getDx = 1/max(X,Y,Z) (grid.cc:37-40) is a
 * property of the WHOLE grid: dx > 0 overrides the value addBuoyancy / addGravity derive from the
 * local tensor sizes; 0 restores the default. */
int tfl_set_dx_override(tfl_ctx* ctx, float dx);

/* x = x*invMask + bc (both NULL: skip), then clamp to [lo, hi] if doClamp: one fused launch for
 * setConstVals' cmul+add pairs and the final U:clamp (lib/simulate.lua:130-160, 326), which the
 * reference issues as separate THC elementwise kernels. */
int tfl_applyBCs(tfl_ctx* ctx, const tfl_tensor* x, const tfl_tensor* bc, const tfl_tensor* invMask,
                 int doClamp, float lo, float hi);

/* tfl_applyBCs restricted to the element indices idx[0..n) (device int32, indices into the flattened
 * tensors): x[e] = x[e]*invMask[e] + bc[e]. For BC tensors that are the identity (invMask 1, bc 0)
 * almost everywhere -- the plume BCs touch 4 y-rows -- the host caches the non-identity indices once and
 * every setConstVals becomes O(|BC cells|) instead of three dense sweeps per field. */
int tfl_applyBCsIndexed(tfl_ctx* ctx, const tfl_tensor* x, const tfl_tensor* bc, const tfl_tensor* invMask,
                        const int32_t* idx, int64_t n);

/* count <= 8 tfl_applyBCsIndexed calls in ONE launch (setConstVals touches p, U and every density channel
 * back to back, lib/simulate.lua:130-160; each list is a few thousand cells, so the launches themselves were
 * the cost). x[i], bc[i], invMask[i] as in tfl_applyBCsIndexed; idx[i] / n[i] the device index list of pair i. */
int tfl_applyBCsIndexedMulti(tfl_ctx* ctx, int count, const tfl_tensor* const* x, const tfl_tensor* const* bc,
                             const tfl_tensor* const* invMask, const int32_t* const* idx, const int64_t* n);

/* init.lua:560-564 `tfluids.getDx(flags)`: 1 / max(X, Y, Z) as a double (a Lua number), or the override of
 * tfl_set_dx_override. */
double tfl_getDx(tfl_ctx* ctx, const tfl_tensor* flags);

/* `dst:copy(src)` on the context's stream (same element count). */
int tfl_copy(tfl_ctx* ctx, const tfl_tensor* dst, const tfl_tensor* src);
/* A plain streaming copy of n floats (16-byte aligned, n % 4 == 0) by the library's own float4 kernel: the measured
 * HBM yard-stick of bench.py (timed through tfl_profile_begin / _end as "k_stream_copy"). No reference counterpart. */
int tfl_stream_copy(tfl_ctx* ctx, float* dst, const float* src, int64_t n);

/* ---- the whole step as one call: lib/simulate.lua:175-327 in native code -------------------------------------
 * For hosts that are not Python (the LuaJIT binding): tfl_simulate_step runs tfluids.simulate() -- advectScalar
 * per density channel, advectVel, setConstVals, addBuoyancy / addGravity / vorticityConfinement, setConstVals, the
 * pressure projection (ConvNet | Jacobi | PCG), setConstVals, U:clamp(+-1e6) -- with the same launch-saving
 * orchestration as fluidnet_amd/simulate.py: index-list BCs, idempotent BC pairs skipped on fields nothing has
 * written, the U:copy after advectVel folded into addBuoyancy, setConstVals(U) + clamp fused into the projection.
 * State tensors are updated in place like the reference's batchGPU.
 *
 * tfl_bc_plan: the precomputed form of one (BC, BCInvMask) pair of lib/simulate.lua:130-160 (the list of cells where
 * the pair is not the identity; whether it is sparse; whether it is idempotent). Create once per pair, re-create
 * when the BC tensors change (createPlumeBCs, the 2-D demo's interactive edits). The plan keeps the two pointers. */
typedef struct tfl_bc_plan tfl_bc_plan;
tfl_bc_plan* tfl_bc_plan_create(tfl_ctx* ctx, const tfl_tensor* bc, const tfl_tensor* invMask);
void tfl_bc_plan_destroy(tfl_ctx* ctx, tfl_bc_plan* plan);

/* tfl_wall_plan (round 6, optional): what the projection asks of a scene's flags as one 16-bit code per cell (the setWallBcs
 * decisions: zero u_x / u_y / u_z; is-fluid, is-open, and the same of the three minus-neighbours), computed once. The flags of a simulation are set when the scene is built (lib/simulate.lua never writes them);
 * the projection's first and last kernel otherwise re-derive all that from up to ten rows of flag words per cell row, every step.
 * Create one per flags array ([B][1][Z][Y][X]) on the context that steps it: tfl_model_begin / tfl_model_forward /
 * tfl_simulate_step[_slab] find it again by the array's address and shape and read the bytes instead -- the same decisions,
 * the same bits out. Destroy it (and create a new one) whenever the flags change, and before the array is freed. */
typedef struct tfl_wall_plan tfl_wall_plan;
tfl_wall_plan* tfl_wall_plan_create(tfl_ctx* ctx, const tfl_tensor* flags);
void tfl_wall_plan_destroy(tfl_ctx* ctx, tfl_wall_plan* plan);
/* Unregister without freeing (no HIP call: safe from a garbage collector's finalizer, where a hipFree could invalidate a graph
 * capture in progress); tfl_wall_plan_destroy still has to follow. */
void tfl_wall_plan_retire(tfl_wall_plan* plan);

typedef struct tfl_sim_params {      /* mconf of lib/simulate.lua (defaults of lib/default_conf.lua in brackets) */
  float dt;
  const char* advectionMethod;       /* NULL = "maccormackOurs" */
  float maccormackStrength;
  double buoyancyScale;              /* 0 = off. Lua numbers: (dx/4)*scale is formed in double and rounded once, */
  double gravityScale;               /* 0 = off.  exactly like simulate.lua:204-239 does on its float tensors   */
  float gravity[3];                  /* direction, simulate.lua:204-211 [0, 1, 0] */
  double vorticityConfinementAmp;    /* 0 = off */
  const char* simMethod;             /* NULL | "convnet" | "jacobi" | "pcg" */
  int32_t maxIter;                   /* jacobi / pcg; <= 0 -> 100 */
  const char* pcgPrecond;            /* NULL = "ic0" like simulate.lua:283; "none" | "ilu0" | "ic0". The unpreconditioned
                                        solve is ~1.7x faster per call at 128^3 here (slower at 256^3) and takes ~3x the iterations */
  int32_t outputDiv;                 /* 1: return before the projection (simulate.lua:241-245) */
} tfl_sim_params;

typedef struct tfl_sim_state {
  const tfl_tensor* p;               /* batch.pDiv   [B][1][Z][Y][X] */
  const tfl_tensor* U;               /* batch.UDiv   [B][C][Z][Y][X] */
  const tfl_tensor* flags;
  int32_t n_density;                 /* 0..8 scalar channels (the RGB table of the 2-D demo = 3) */
  const tfl_tensor* density[8];
  const tfl_bc_plan* pBC;            /* NULL = no such BC pair */
  const tfl_bc_plan* UBC;
  const tfl_bc_plan* densityBC[8];
  tfl_model* model;                  /* needed for simMethod convnet */
} tfl_sim_state;

int64_t tfl_simulate_workspace_floats(tfl_ctx* ctx, const tfl_sim_params* params, const tfl_sim_state* state);
int tfl_simulate_step(tfl_ctx* ctx, const tfl_sim_params* params, const tfl_sim_state* state, float* workspace,
                      int64_t workspace_floats);

/* ---- z-slab decomposition across the GPUs of a node (BASELINE config 5; no counterpart in the reference, which is
 * single-GPU). A rank holds planes [z_first, z_first + Zlocal) of a z_total-deep grid: the planes it OWNS,
 * [own_lo, own_hi) in local indices, plus `halo` planes of its neighbours on each side that has one (none at the domain
 * ends). Building blocks first, the assembled step last. ------------------------------------------------------- */

/* Gather planes [zlo, zhi) of n <= 8 fields (each with its own B and C; all with the same Z, Y, X) into one contiguous
 * buffer laid out [field][b][c][plane][Y][X] (unpack = 0), or scatter such a buffer back (unpack = 1): what RCCL
 * send/recv moves over xGMI. */
int tfl_packPlanes(tfl_ctx* ctx, int n, const tfl_tensor* const* fields, int zlo, int zhi, float* buf,
                   int unpack);

/* Compute window: until changed, advectScalar, advectVel, addBuoyancy[From], addGravity, vorticityConfinement,
 * tfl_model_begin and tfl_model_finish compute ONLY the z-planes [a0, a1) and [b0, b1) of their tensors (the two runs
 * go out as one launch; an empty second run is b0 == b1); reads still address the whole array. All zero = every plane
 * (the default). The other operators ignore the window. A slab rank uses it to run each phase of the step on exactly
 * the planes whose inputs are valid, and to split a phase into boundary strips (computed first, so their halo message
 * can leave) and interior (computed while the message is in flight). */
int tfl_set_z_window(tfl_ctx* ctx, int a0, int a1, int b0, int b1);

/* Where the tensors of the next advectScalar / advectVel calls sit inside the whole grid: local plane 0 is global plane
 * z_first of z_total. The back-traces then form their positions in GLOBAL z, so that a slab rounds every position
 * exactly like the unsplit grid does (a slab-relative coordinate lands in another binade and rounds differently);
 * the domain walls of the trace are those of the whole grid. (0, 0) = the tensors are the whole grid (default). */
int tfl_set_z_origin(tfl_ctx* ctx, int z_first, int z_total);

/* Pass selection for the multi-pass operators (0 = all passes, the default), so that each pass can get its own
 * window: advectScalar 1 = 3^dim min/max grid, 2 = pass A (forward), 4 = pass B (backward + correct + clamp);
 * advectVel 2 / 4 likewise; vorticityConfinement 2 = curl, 4 = confinement force; tfl_model_begin 2 = wall BCs +
 * divergence + partial sums, 4 = reduction over [zlo, zhi); tfl_model_finish (3-D default topology only) 1 / 2 / 4 =
 * first / second / third(+1x1x1) conv layer, 8 = velocity update + un-scale + wall BCs. */
int tfl_set_stages(tfl_ctx* ctx, int mask);

/* The divergence plane tfl_model_begin wrote inside `workspace` (a [B][1][Z][Y][X] field): a slab rank exchanges its
 * halo planes before the first conv layer. */
float* tfl_model_div(const tfl_model* model, int B, int Z, int Y, int X, float* workspace);

typedef struct tfl_slab {
  int32_t z_total;        /* planes of the whole grid */
  int32_t z_first;        /* global index of local plane 0 */
  int32_t own_lo, own_hi; /* owned planes [own_lo, own_hi), local indices; own_lo = 0 at the lower domain end, else the
                             stored halo depth (>= tfl_slab_halo(reach)); likewise above */
  int32_t reach;          /* R: the step is exact while max|u_z|*dt < R cells (the back-trace of the advection);
                             0 = 1. Halo depth needed = max(4, 2R + 1) */
  int32_t overlap;        /* 1: split the phases that feed a message into boundary strips + interior so that the
                             transfer overlaps compute (worth it when a slab holds >~ 1M cells); 0: one launch each */
  int32_t check_reach;    /* 1: every step reduces max|u_z| on the device into a sticky maximum and copies it to pinned
                             memory; a violation by the state step n starts from is reported by the call for step n+2
                             at the latest (that call waits for step n's copy, which has long landed unless the host is
                             more than two steps ahead of the device: round 6 -- waiting for step n+1's cost 60 us per
                             step); messages in flight are drained before the error is returned. The report is per
                             rank: stop every rank when one reports (or use 2). Where the planes fill the projection
                             kernel's blocks (X % 128 == 0, Y % 8 == 0, ConvNet projection) the maximum is taken by that
                             kernel over the planes it writes -- the rank's OWNED planes; its halo planes are their owners'
                             to report -- and the step launches no reduction of its own from the second step on; the report
                             then comes one call later still (round 6).
                             2 (round 6, "exact"): the check comes BEFORE the step's advection and is collective -- max|u_z|
                             of the state the step starts from, the reach it needs all-reduced over the ranks (8 doubles
                             through tfl_comm.allreduce_sum), one host synchronisation. A step that needs more than `reach`
                             returns TFL_EREACH on EVERY rank with nothing written; the host widens the halos
                             (tfl_slab_needed_reach, tfl_slab_exchange) and calls again, so the cut run stays the
                             un-cut run whatever the flow does (fluidnet_amd/dist.py SlabSimulation does it by itself).
                             The same all-reduce carries every rank's fp16 range word, so TFL_ERANGE is collective too
                             in this mode; with check_reach 0 / 1 a slab that has neighbours never returns TFL_ERANGE on
                             its own (the others would wait for its halos): poll tfl_model_range_flag on the host.
                             Costs the host its lead over the device (~10-30 us per step): opt-in */
  int32_t in_flight;      /* OUT/IN, initialise to 0 for a new run: bits 0-3 = halo messages started by the previous call and
                             not yet consumed (tfl_simulate_step_slab finishes them; tfl_slab_drain does so explicitly);
                             bit 8 = this run has reset the context's reach word (library-internal) */
} tfl_slab;

/* Transport supplied by the host (RCCL send/recv through torch.distributed in fluidnet_amd/dist.py; any MPI-like
 * layer works). All buffers are device memory inside the step's workspace. Every callback is called from the thread
 * that called tfl_simulate_step_slab, after the library has enqueued the packing kernels on the context's stream:
 *   exchange_start: begin sending send_lo[0..n_send_lo) to rank-1 and receiving recv_lo from it, send_hi / recv_hi with
 *                   rank+1 (n == 0 / NULL: no such neighbour). Must order the transfer after the work enqueued so far
 *                   on the stream; should not block the host.
 *   exchange_wait:  make the stream wait until the transfers of exchange_start(tag) have landed.
 *   allreduce_sum:  in-place sum of n device doubles over all ranks, ordered on the stream.
 * Return 0, or non-zero to abort the step (reported as TFL_EINVAL with "comm callback failed"). */
typedef struct tfl_comm_chunk {   /* n contiguous floats of device memory */
  float* ptr;
  int64_t n;
} tfl_comm_chunk;

typedef struct tfl_comm {
  int32_t size;   /* sizeof(tfl_comm) as the HOST compiled it: the library reads no member that lies beyond it, so a host built
                     against a header without the optional trailing callbacks keeps working (0 is refused) */
  void* user;
  int (*exchange_start)(void* user, int tag, const float* send_lo, int64_t n_send_lo, float* recv_lo, int64_t n_recv_lo,
                        const float* send_hi, int64_t n_send_hi, float* recv_hi, int64_t n_recv_hi);
  int (*exchange_wait)(void* user, int tag);
  int (*allreduce_sum)(void* user, double* dev, int64_t n);
  /* Optional (NULL: not offered). The same exchange WITHOUT staging buffers: the halo planes of a [C][Z][Y][X] field are
   * one contiguous run per (batch item, channel), so a message is a short list of chunks that a transport able to move
   * several pieces per neighbour (RCCL: several ncclSend / ncclRecv in one group) takes straight out of, and delivers
   * straight into, the fields -- the step then launches no pack / unpack kernels (6 of a middle rank's 22 launches per
   * step). Chunk i of send_hi pairs with chunk i of the upper neighbour's recv_lo (same order, same sizes); n_lo / n_hi = 0:
   * no such neighbour. The received planes are written while the transfer runs, i.e. any time between this call and
   * exchange_wait(tag); the step never reads them in that window. */
  int (*exchange_start_v)(void* user, int tag, int n_lo, const tfl_comm_chunk* send_lo, const tfl_comm_chunk* recv_lo,
                          int n_hi, const tfl_comm_chunk* send_hi, const tfl_comm_chunk* recv_hi);
  /* Optional (ABI 4; 0 or absent: no). Non-zero: every callback only ENQUEUES stream operations ordered against the
   * context's stream (kernels, copies, event record / wait, RCCL calls) and never waits on the host or reads device
   * results -- so a whole rank-step can be recorded into a HIP graph (tfl_slab_graph_create). The native RCCL transport
   * sets it; a transport that runs host code per message (the torch.distributed ones of fluidnet_amd/dist.py) must not. */
  int32_t capturable;
} tfl_comm;

/* Halo depth a slab must store next to each neighbour for reach R (>= 4). */
int32_t tfl_slab_halo(int32_t reach);
int64_t tfl_simulate_slab_workspace_floats(tfl_ctx* ctx, const tfl_sim_params* params, const tfl_sim_state* state,
                                           const tfl_slab* slab);

/* tfl_simulate_step on one z-slab: the owned planes of p, U and density come out as the single-GPU step would compute
 * them (bit for bit, except for the summation order of the std normaliser's all-reduce). State tensors are the LOCAL
 * extended arrays; BC plans are made on the local BC tensors; flags halos are static (filled by the caller once).
 * Preconditions: 3-D, maccormackOurs, ConvNet projection with the 3-D default topology, at most one density channel,
 * B*C layout as in tfl_simulate_step; on the first call every halo plane holds valid data (the caller cut its arrays
 * out of a global initial state); `workspace` must be the SAME buffer on every call (halo messages of p and U started
 * at the end of one step are consumed by the next).
 * Per step: three neighbour exchanges + one 2*B-double all-reduce --
 *   U(max(R+1, 2R) planes) + p(4 below, 3 above)   ONE message started at the end of the previous step, consumed at the start
 *   advected U(3 below, 4 above) + density(max(4, 2R+1)) after MacCormack pass B, overlapped with its interior
 *   divergence(4 below, 3 above)          overlapped with the interior of the first conv layer
 * and every phase runs under the narrowest z-window that keeps the owned planes exact, so the redundant compute is a
 * few planes per phase instead of a fixed wide halo. */
int tfl_simulate_step_slab(tfl_ctx* ctx, const tfl_sim_params* params, const tfl_sim_state* state, tfl_slab* slab,
                           const tfl_comm* comm, float* workspace, int64_t workspace_floats);

/* The reach the last TFL_EREACH of this context asked for: the smallest R with max|u_z|*dt < R over ALL ranks (0: none yet). */
int32_t tfl_slab_needed_reach(const tfl_ctx* ctx);

/* One halo exchange of n <= 4 fields outside a step (what a host needs to widen a slab's halos after TFL_EREACH, or to
 * fill them at start-up without a global array): planes [own_lo - below[i], own_lo) of field i are received from the lower
 * neighbour and [own_hi, own_hi + above[i]) from the upper one; the matching owned planes are sent. Every field has the
 * slab's local depth (own_hi + halo). Staged through `scratch` (tfl_slab_exchange_floats of them); collective; returns with
 * the transfers ordered on the context's stream. */
int64_t tfl_slab_exchange_floats(int n, const tfl_tensor* const* fields, const int32_t* below, const int32_t* above, const tfl_slab* slab);
int tfl_slab_exchange(tfl_ctx* ctx, int n, const tfl_tensor* const* fields, const int32_t* below, const int32_t* above,
                      const tfl_slab* slab, const tfl_comm* comm, float* scratch, int64_t scratch_floats);

/* Finish the messages a previous tfl_simulate_step_slab left in flight: afterwards the halo planes of U and p are
 * valid too (call before reading halos on the host, or before freeing the workspace). */
int tfl_slab_drain(tfl_ctx* ctx, const tfl_sim_state* state, tfl_slab* slab, const tfl_comm* comm, float* workspace,
                   int64_t workspace_floats);

/* ---- the rank-step as ONE host call (round 6). tfl_simulate_step_slab costs the host a dozen kernel launches plus the
 * transport's calls per step -- on a 16-plane slab about as long as the GPU needs to run them. tfl_slab_graph_create records
 * one step (same arguments; they, the tensors they point to and the workspace must stay where they are while the graph lives)
 * on a stream of its own into an executable HIP graph: kernels, the transport's sends / receives / all-reduce and the
 * stream forks between them. tfl_slab_graph_step replays it on the context's stream: it makes the two host-side checks of the
 * eager call first (the reach word and the fp16 range word the previous step left in pinned memory; same error codes), then
 * ONE hipGraphLaunch. Differences from the eager step: the U / p message is started AND finished inside the step (no message
 * in flight between calls: slab->in_flight stays 0, tfl_slab_drain is a no-op), and the scalar parameters are frozen.
 * Needs tfl_comm.capturable (or a slab without neighbours); call it after at least one eager tfl_simulate_step_slab on
 * the same arguments (allocations, RCCL's connection set-up). NULL + tfl_last_error when the step cannot be recorded --
 * the caller keeps stepping eagerly. Results are those of the eager step bit for bit. */
typedef struct tfl_slab_graph tfl_slab_graph;
tfl_slab_graph* tfl_slab_graph_create(tfl_ctx* ctx, const tfl_sim_params* params, const tfl_sim_state* state, tfl_slab* slab,
                                      const tfl_comm* comm, float* workspace, int64_t workspace_floats);
int tfl_slab_graph_step(tfl_ctx* ctx, tfl_slab_graph* graph);
int64_t tfl_slab_graph_nodes(const tfl_slab_graph* graph);   /* nodes of the recorded graph (kernels, copies, events) */
void tfl_slab_graph_destroy(tfl_ctx* ctx, tfl_slab_graph* graph);

/* ---- native transport for the z-slab step: RCCL send/recv over xGMI inside the library (csrc/comm_rccl.cpp) ------------
 * For hosts without a communication layer of their own (the LuaJIT loop, plain C): one process per GPU, ranks ordered
 * along z (rank r's upper neighbour is r+1). Rank 0 obtains a unique id and hands the 128 bytes to the other ranks by
 * any means it has (a file, a socket, MPI, the launcher's environment); every rank then creates its communicator and
 * passes tfl_rccl_comm_callbacks() as the `comm` of tfl_simulate_step_slab / tfl_slab_drain. RCCL is dlopen'ed on first
 * use ($TFL_RCCL_LIBRARY, else a copy already loaded in the process, else librccl.so.1): the library itself links
 * only the HIP runtime. Transfers run on a communication stream of the communicator, ordered against the context's
 * stream by events; no call blocks the host. The context must outlive the communicator and keep its device. */
#define TFL_RCCL_UNIQUE_ID_BYTES 128
typedef struct tfl_rccl_comm tfl_rccl_comm;
/* 1 when an RCCL could be bound (the reason is in tfl_last_error otherwise). */
int tfl_rccl_available(tfl_ctx* ctx);
/* which library was bound ("librccl.so.1 (already loaded)", a path, ...) */
const char* tfl_rccl_comm_origin(tfl_ctx* ctx);
/* ncclGetUniqueId: fills id[0..128). */
int tfl_rccl_get_unique_id(tfl_ctx* ctx, void* id);
/* ncclCommInitRank on the context's device (collective: every rank of `world` must call it). NULL on failure. */
tfl_rccl_comm* tfl_rccl_comm_create(tfl_ctx* ctx, const void* id, int rank, int world);
/* The same around a communicator the host already has (an ncclComm_t); it is not destroyed by tfl_rccl_comm_destroy. */
tfl_rccl_comm* tfl_rccl_comm_wrap(tfl_ctx* ctx, void* nccl_comm, int rank, int world);
const tfl_comm* tfl_rccl_comm_callbacks(tfl_rccl_comm* comm);
/* 1: issue every send / receive / all-reduce on the CONTEXT's stream instead of the communication stream (no events between the
 * two). For slabs run with tfl_slab.overlap = 0 -- nothing there for a transfer to overlap with, and an event hop between
 * two streams costs 12-15 us of device-side latency on this stack, eight of them per eager step. 0 (default): the
 * communication stream. Call between steps with no message in flight (after tfl_slab_drain). A recorded step
 * (tfl_slab_graph_create) does not care: inside a graph the hops are dependencies, not events. */
int tfl_rccl_comm_set_inline(tfl_rccl_comm* comm, int on);
void tfl_rccl_comm_destroy(tfl_ctx* ctx, tfl_rccl_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* TFLUIDS_HIP_H_ */
