// tfl_ctx.hpp -- the context object behind the opaque tfl_ctx* of include/tfluids_hip.h (private to the library:
// abi.cpp owns it, simulate.cpp reads the stream and the z-slab reach-check words).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "tfl_host.hpp"

// the wall codes of one flags array (include/tfluids_hip.h tfl_wall_plan): found again by the array's address and shape
struct tfl_wall_plan {
  const float* flags = nullptr;
  unsigned short* code = nullptr;             // 16 bits per cell (model.hip k_wall_code)
  int B = 0, Z = 0, Y = 0, X = 0;
  bool is3d = false;
  struct tfl_ctx* owner = nullptr;            // the context it is registered with (cleared by tfl_destroy: the plan may outlive it)
};

struct tfl_ctx {
  std::vector<tfl_wall_plan*> wall_plans;     // registered by tfl_wall_plan_create, looked up by tfl_model_begin
  std::mutex wall_mu;                         // (a binding may retire a plan from a finalizer thread while this context steps)
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  unsigned long long* d_trace_err = nullptr;  // device word: traces that hit an invariant path
  double* d_resid = nullptr;                  // Jacobi residual accumulators [kMaxBatch]
  double* h_resid = nullptr;                  // pinned mirror
  float dx_override = 0.0f;                   // > 0: use instead of 1/max(X,Y,Z) (z-slab ranks: global dx)
  int dx_dim = 0;                             // > 0: dx = 1/dx_dim formed exactly like getDx does (tfl_simulate_step_slab)
  tfl::ZWin zwin = {0, 0, 0, 0};              // tfl_set_z_window: planes the next operators compute (all zero = all)
  tfl::ZOrigin zorigin = {0, 0};              // tfl_set_z_origin: where the arrays sit in the whole grid (z-slab ranks)
  int advect_fast = 0;                        // tfl_set_advect_mode (initialised from TFL_ADVECT_MODE by tfl_create)
  int stages = 0;                             // tfl_set_stages: which passes of a multi-pass operator run (0 = all)
  float* d_reach = nullptr;                   // z-slab reach check: max|u_z| of the current step (device word)
  float* h_reach = nullptr;                   // pinned mirror (TWO words: the maximum, and the publication count), read by later calls
  // check_reach = 1 (round 6): the device word is a STICKY maximum (never reset by a step: a violation cannot be overwritten before
  // the host has seen it). The step's LAST kernel (k_project) copies it into the mapped pinned mirror -- an async 4-byte D2H copy
  // BLOCKS the host on this stack -- and, behind a system-scope fence, the count of such publications (a device word it
  // increments) beside it. The call for step n + 1 waits, spinning on that pinned count, for the publication of step n - 1:
  // there unless the host is more than two steps ahead, so the wait bounds the host's lead and costs nothing. No API call
  // and no event on either side: the hipEventRecord per step of the first form cost the device 4 us of a 0.1 ms rank-step.
  float* d_reach_host = nullptr;              // the device address of h_reach
  unsigned* d_reach_tick = nullptr;           // device word: publications made so far
  bool reach_sink = false;                    // the projection launches of THIS step publish (set by the slab step, cleared by its guard)
  bool reach_folded = false;                  // the last projection launch of a slab step folded max|u_z| of the planes it wrote into d_reach: the next step's k_absmax is not needed
  unsigned reach_issued = 0;                  // publishing launches enqueued so far (host side)
  unsigned reach_hist[2] = {0, 0};            // reach_issued as it stood at the end of the step before last / of the last step
  double* h_reach_flags = nullptr;            // pinned [kReachFlags]: the all-reduced "reach >= r" counts of check_reach = 2 (created on first use)
  int needed_reach = 0;                       // what the last TFL_EREACH asked for (tfl_slab_needed_reach)
  tfl::BcFoldArg fold = {nullptr, 0u, 0u};    // tfl_simulate_step: a setConstVals pair (device descriptor + gate) the next operator may apply to its output
  bool fold_done = false;                     // ... and whether a launcher did (tfl_host.hpp BcFold)
  tfl::BuoyFold buoy = {nullptr, 0.0f, 0.0f, 0.0f};   // tfl_simulate_step: the buoyancy force the next advectVel may add itself (tfl_host.hpp BuoyFold)
  bool buoy_done = false;
  bool vort_from_two_launch = false;          // tfl_simulate_step: the next tfl_vorticityConfinementFrom skips the fused kernel (grid below its size)
  int wf_skip = 0;                            // > 0: a pipelined PCG sweep timed out on this context: the next wf_skip solves go straight to
                                              // hyperplane sweeps, then the pipelined form is tried again (a transient stall must not latch for good)
  int wf_timeouts = 0;                        // how often that happened (the back-off doubles, one warning per latch)
  bool defer_stats = false;                   // tfl_model_forward: the first conv layer sums k_bcs_div_stats' partials itself (no k_reduce_stats launch)
  bool in_step = false;                       // inside tfl_simulate_step[_slab]: the fp16 range gate was taken at the step's entry, tfl_model_begin /
                                              // tfl_model_forward do not take it again mid-step (a half-stepped state otherwise)
  bool capturing = false;                     // tfl_slab_graph_create is recording the step on `stream`: no host waits, no host reads
};
