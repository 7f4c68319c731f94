// conv.hip -- convolution layers of the projection ConvNet (gfx950).
//
// Replaces cudnn.SpatialConvolution / cudnn.VolumetricConvolution forward as used by
// torch/lib/model_utils.lua:80-116: stride 1, zero padding (k-1)/2, cross-correlation, bias, with the
// following nn.ReLU fused into the epilogue. Activations are channel-planar fp32 [B][C][Z][Y][X].
//
// k_conv_direct is the shape-generic path (any C_in, k; C_out in {1, 2, 4, 8, 16, 32, 64}): one thread
// per voxel holding all C_out accumulators in registers; the weights are re-laid out on the host as
// [tap][c_in][c_out] so every weight address is wave-uniform and travels through the scalar cache
// (s_load), leaving the vector memory path to the activations. The fmaf chain is exact fp32.
//
// The `tog` topologies (lib/model.lua:163-178, 211-218) add two things, both here: 2x average pooling after a layer
// (cudnn.{Spatial,Volumetric}AveragePooling(2,..,2), model_utils.lua:184-208) = k_avg_pool2, and
// nn.{Spatial,Volumetric}ConvolutionUpsample (lib/modules/spatial_convolution_upsample.lua:24-93,
// volumetric_convolution_upsample.lua): a convolution to up^dim * C_out channels whose result is pixel-shuffled,
// out[b][o][z*u+a][y*u+b][x*u+c] = conv[b][((o*u + a)*u + b)*u + c][z][y][x]. The shuffle costs nothing here: the conv
// is launched once per sub-position (a, b, c) with that sub-position's weight slice and a strided store (ConvUp).
#include "tfl_device.hpp"
#include "tfl_host.hpp"

namespace tfl {

struct ConvUp {      // output placement: voxel (i, j, k) of the conv grid goes to (i*u + a, j*u + b, k*uz + c) of `dout`
  int u, uz, a, b, c;
  int oX, oY, oZ;    // size of the output grid
  int act;           // epilogue: 0 none, 1 ReLU, 2 ReLU6, 3 sigmoid (model_utils.lua:20-34)
  int och;           // channel planes per batch item of `out` (>= COUT: room for a joined skip channel)
};
__device__ __forceinline__ float conv_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.0f);
  if (act == 2) return fminf(fmaxf(v, 0.0f), 6.0f);
  if (act == 3) return 1.0f / (1.0f + expf(-v));     // nn.Sigmoid (THNN: 1 / (1 + exp(-x)))
  return v;
}

// CPT = output channels per thread: COUT for large grids (each activation is loaded once), COUT/4 for small
// ones where the grid would otherwise leave most CUs idle (2-D 128^2 = 64 blocks of 256 threads).
template <bool IS3D, int COUT, int CPT>
__global__ __launch_bounds__(256) void k_conv_direct(Dom d, int cin, int ksz, const float* __restrict__ in,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ out, ConvUp up) {
  constexpr int G = COUT / CPT;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int zg = blockIdx.z / G, co0 = (blockIdx.z - zg * G) * CPT;
  const int b = zg / d.Z, k = zg - b * d.Z;
  if (i >= d.X || j >= d.Y) return;
  const long long cells = d.sc;
  const long long ocells = (long long)up.oX * up.oY * up.oZ;
  in += b * cells * cin; out += b * ocells * up.och;
  float acc[CPT];
#pragma unroll
  for (int c = 0; c < CPT; c++) acc[c] = bias[co0 + c];
  const int r = (ksz - 1) / 2;
  const int rz = IS3D ? r : 0;
  int tap = 0;
  for (int dz = -rz; dz <= rz; dz++) {
    for (int dy = -r; dy <= r; dy++) {
      for (int dx = -r; dx <= r; dx++, tap++) {
        const int x = i + dx, y = j + dy, z = k + dz;
        const bool ok = x >= 0 && x < d.X && y >= 0 && y < d.Y && z >= 0 && z < d.Z;
        const int o = ok ? TFL_AT(d, x, y, z) : 0;
        const float* wt = w + (long long)tap * cin * COUT;
        // 8 independent activation loads in flight per batch (a one-load-one-fma loop is a chain of
        // L1 latencies: 144 taps x ~200 cycles made the 2-D 16-channel layers 27 us at 128^2)
        int c = 0;
        for (; c + 8 <= cin; c += 8) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; q++) v[q] = ok ? in[o + (c + q) * d.sc] : 0.0f;
#pragma unroll
          for (int q = 0; q < 8; q++)
#pragma unroll
            for (int co = 0; co < CPT; co++) acc[co] = fmaf(v[q], wt[(c + q) * COUT + co0 + co], acc[co]);
        }
        for (; c < cin; c++) {
          const float v = ok ? in[o + c * d.sc] : 0.0f;
#pragma unroll
          for (int co = 0; co < CPT; co++) acc[co] = fmaf(v, wt[c * COUT + co0 + co], acc[co]);
        }
      }
    }
  }
  const long long o = (i * up.u + up.a) + (long long)up.oX * ((j * up.u + up.b) + (long long)up.oY * (k * up.uz + up.c));
#pragma unroll
  for (int c = 0; c < CPT; c++) out[o + (co0 + c) * ocells] = conv_act(acc[c], up.act);
}

// cudnn average pooling, window = stride = 2, no padding: out size floor(n / 2) per pooled axis (z only in 3-D)
template <bool IS3D>
__global__ __launch_bounds__(256) void k_avg_pool2(int rows, int Zo, int Yo, int Xo, int Z, int Y, int X,
                                                   const float* __restrict__ in, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int r = blockIdx.z / Zo, k = blockIdx.z - r * Zo;      // r = b*C + c
  if (i >= Xo || j >= Yo) return;
  const float* p = in + ((long long)r * Z + (IS3D ? 2 * k : 0)) * Y * X + (long long)(2 * j) * X + 2 * i;
  float s = (p[0] + p[1]) + (p[X] + p[X + 1]);
  if (IS3D) {
    const float* q = p + (long long)Y * X;
    s += (q[0] + q[1]) + (q[X] + q[X + 1]);
  }
  out[((long long)r * Zo + k) * Yo * Xo + (long long)j * Xo + i] = s * (IS3D ? 0.125f : 0.25f);
}

template <bool IS3D, int COUT>
static void launch_direct(hipStream_t st, const Dom& d, int B, int cin, int ksz, const float* in,
                          const float* w, const float* bias, float* out, const ConvUp& up) {
  const dim3 blk(64, 4, 1);
  const unsigned nxy = ((d.X + 63) / 64) * ((d.Y + 3) / 4);
  constexpr int CPT_SMALL = COUT >= 4 ? COUT / 4 : COUT;
  const bool split = COUT >= 4 && (long long)nxy * d.Z * B < 1024;   // fewer than ~4 blocks per CU: split channels
  TFL_TIMED("k_conv_direct", st);
  if (split) {
    const dim3 grd((d.X + 63) / 64, (d.Y + 3) / 4, (unsigned)(d.Z * B * (COUT / CPT_SMALL)));
    k_conv_direct<IS3D, COUT, CPT_SMALL><<<grd, blk, 0, st>>>(d, cin, ksz, in, w, bias, out, up);
  } else {
    const dim3 grd((d.X + 63) / 64, (d.Y + 3) / 4, (unsigned)(d.Z * B));
    k_conv_direct<IS3D, COUT, COUT><<<grd, blk, 0, st>>>(d, cin, ksz, in, w, bias, out, up);
  }
}

// w: device, [tap][cin][cout]. act: 0 none | 1 ReLU | 2 ReLU6 | 3 sigmoid. out_ch: channel planes per batch item of
// `out` (0 = cout). Returns false when cout has no instantiation.
bool conv_direct(hipStream_t st, bool is3d, int B, int Z, int Y, int X, int cin, int cout, int ksz, int act,
                 const float* in, const float* w, const float* bias, float* out, int upf, int sub, int out_ch) {
  Dom d = make_dom(Z, Y, X);
  d.w0 = 0; d.n0 = Z; d.w1 = 0; d.nw = Z;       // the shape-generic path always covers the whole grid
  ConvUp up;
  up.u = upf; up.uz = is3d ? upf : 1;
  up.a = sub % upf; up.b = (sub / upf) % upf; up.c = is3d ? sub / (upf * upf) : 0;
  up.oX = X * up.u; up.oY = Y * up.u; up.oZ = Z * up.uz;
  up.act = act; up.och = out_ch > 0 ? out_ch : cout;
#define TFL_CASE(N)                                                                        \
  case N:                                                                                  \
    if (is3d) launch_direct<true, N>(st, d, B, cin, ksz, in, w, bias, out, up);          \
    else launch_direct<false, N>(st, d, B, cin, ksz, in, w, bias, out, up);              \
    return true;
  switch (cout) {
    TFL_CASE(1) TFL_CASE(2) TFL_CASE(4) TFL_CASE(8) TFL_CASE(16) TFL_CASE(32) TFL_CASE(64)
    default: return false;
  }
#undef TFL_CASE
}

void avg_pool2(hipStream_t st, bool is3d, int rows, int Z, int Y, int X, const float* in, float* out) {
  const int Zo = is3d ? Z / 2 : Z, Yo = Y / 2, Xo = X / 2;
  const dim3 blk(64, 4, 1), grd((Xo + 63) / 64, (Yo + 3) / 4, (unsigned)(rows * Zo));
  TFL_TIMED("k_avg_pool2", st);
  if (is3d) k_avg_pool2<true><<<grd, blk, 0, st>>>(rows, Zo, Yo, Xo, Z, Y, X, in, out);
  else k_avg_pool2<false><<<grd, blk, 0, st>>>(rows, Zo, Yo, Xo, Z, Y, X, in, out);
}

}  // namespace tfl
