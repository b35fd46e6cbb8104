// Backward-pass kernels of the DDPM-UNet training step
// (/root/reference/models/diffusion/ddpm.py:111-121,142-144: loss.backward() + Adam).
// Correctness-first round: every reduction has a fixed order (no atomics), the heavy
// contraction (weight gradient) runs on the fp32 matrix cores, the rest are small
// HBM/latency-bound kernels.  Data gradients of the convolutions reuse the forward
// implicit-GEMM kernel (cm_conv.hip) with transposed / flipped packed weights.
#include "cm_kernels.h"

#include <cstdint>

namespace cm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// --------------------------------------------------------------------------------
// Weight gradient of a convolution:  dW[tap][co][ci] = sum_{b,v} dy[b,v,co] * a[b, v*stride+tap-pad, ci]
// where a = pm * silu(gn(x)) is the conv's ACTUAL input, recomputed on the fly from the raw
// producer tensor exactly as the forward staging does (it was never materialised).
// One workgroup owns a (32-co block, 32-ci block) pair and a strided subset of the voxel
// tiles; per tile it stages the dy rows and the input halo into LDS, and its 4 waves split
// the taps.  MFMA 32x32x2 with the VOXEL index as the contraction dimension:
//   A operand = dy[m][co] (lane: co = l&31, k = l>>5),  B operand = a[m+tap][ci].
// Partials go to part[g][cb][kb][tap][32][32]; wgrad_reduce sums them in order.
// --------------------------------------------------------------------------------
template <bool ZS>
__global__ __launch_bounds__(256) void wgrad_kernel(const ConvArgs a, const float *__restrict__ dy, int dy_cs,
                                                    float *__restrict__ part, int G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // parity form of an upsample conv (a.par): blockIdx.x = tile group * 8 + parity class; the class (pz,py,px) reads
  // dY at the output voxels 2i + p and the LOW-resolution input at i + e + p - 1, e in {0,1}^3 (the forward's own
  // 2x2x2 form, cm_conv.hip) -- 8 taps instead of 27 on the materialised upsampled tensor: 3.4x fewer FLOPs
  const int par = a.par ? (int)(blockIdx.x & 7) : 0;
  const int pz = (par >> 2) & 1, py = (par >> 1) & 1, px = par & 1, os = a.par ? 2 : 1;
  const int g = a.par ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, cb = blockIdx.y, kb = blockIdx.z;
  const int td = a.td, pad = (td == 3) ? 1 : 0;
  const int HZ = (a.bz - 1) * a.stride + td, HY = (a.by - 1) * a.stride + td, HX = (a.bx - 1) * a.stride + td;
  const int HV = HZ * HY * HX;
  const int nbox = a.bz * a.by * a.bx;
  const int TM = ZS ? 64 : (nbox + 31) & ~31;   // z-split: one 32-row block per plane
  const int ntile = a.nts * a.ntz * a.nty * a.ntx;
  const int Ctot = a.C0 + a.C1;

  int *rowhv = reinterpret_cast<int *>(lds);  // [TM] halo index of each output row (box-relative)
  int *rowoff = rowhv + TM;                   // [TM] output voxel index of the row in the current tile, or -1
  float *dyt = lds + 2 * TM;                  // [TM][32]
  float *at = dyt + TM * 32;                  // [HV][32]

  for (int m = tid; m < TM; m += 256) {
    int hv = 0;
    const int pk = (m < nbox || ZS) ? a.mtab[m] : -1;
    if (pk >= 0) hv = ((((pk >> 18) & 255) * a.stride) * HY + ((pk >> 9) & 511) * a.stride) * HX + (pk & 511) * a.stride;
    rowhv[m] = hv;
  }
  // input channels of this block in the concatenated channel space (a 4-channel group never
  // straddles the concat boundary: C0 is a multiple of 8)
  const int ci0 = kb * 32;
  const int ntaps = a.ntaps;
  // taps per wave: 27 = 7+7+7+6 (tap t = wave + 4 ti); z-split: slot ti = 3 dz + k holds in-plane tap q = wave + 4 k
  // (k < 2) of z tap dz, slot k = 2 the ninth in-plane tap (q = 8) on wave dz only -- the z tap of a slot is then the
  // same on every wave and the skipped ones drop out at compile time
  constexpr int TW = ZS ? 9 : 7;
  f32x16 acc[TW];
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  // 1x1x1 convs have a single tap: the four waves then split the voxel pairs of a tile instead (wave w takes pairs
  // w, w + 4, ...) and write four pseudo-tap partials that the reduce pass sums -- otherwise three waves idle
  const bool vsplit = ntaps == 1 && !(a.dbg & 8192);
  int tapoff[TW];
#pragma unroll
  for (int ti = 0; ti < TW; ++ti) {
    int t = vsplit ? 0 : min(wave + 4 * ti, ntaps - 1);
    if constexpr (ZS) t = (ti / 3) * 9 + ((ti % 3) < 2 ? wave + 4 * (ti % 3) : 8);
    const int dz = t / (td * td), rem = t - dz * td * td, dyy = rem / td, dx = rem - dyy * td;
    tapoff[ti] = ((dz * HY + dyy) * HX + dx) * 32;
  }
  const int Zc = a.Zs << a.ups, Yc = a.Ys << a.ups, Xc = a.Xs << a.ups;
  for (int tile = g; tile < ntile; tile += G) {
    int tt = tile;
    const int tx = tt % a.ntx; tt /= a.ntx;
    const int ty = tt % a.nty; tt /= a.nty;
    const int tz = tt % a.ntz;
    const int b = tt / a.ntz;
    const int z0 = tz * a.bz, y0 = ty * a.by, x0 = tx * a.bx;
    __syncthreads();
    // rows of this tile
    for (int m = tid; m < TM; m += 256) {
      int off = -1;
      const int pk = (m < nbox || ZS) ? a.mtab[m] : -1;
      if (pk >= 0) {
        const int oz = os * (z0 + ((pk >> 18) & 255)) + pz, oy = os * (y0 + ((pk >> 9) & 511)) + py, ox = os * (x0 + (pk & 511)) + px;
        if (oz < a.Zo && oy < a.Yo && ox < a.Xo) off = ((b * a.Zo + oz) * a.Yo + oy) * a.Xo + ox;
      }
      rowoff[m] = off;
    }
    __syncthreads();
    // dy tile
    for (int i = tid; i < TM * 8; i += 256) {
      const int m = i >> 3, q = i & 7;
      const int off = rowoff[m];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (off >= 0) {
        const int co = cb * 32 + 4 * q;
        const float *p = dy + (size_t)off * dy_cs + co;
        v[0] = co + 0 < a.Co ? p[0] : 0.f; v[1] = co + 1 < a.Co ? p[1] : 0.f;
        v[2] = co + 2 < a.Co ? p[2] : 0.f; v[3] = co + 3 < a.Co ? p[3] : 0.f;
      }
      *reinterpret_cast<f32x4 *>(&dyt[m * 32 + 4 * q]) = v;
    }
    // input halo (same transform as the forward staging)
    const int cz0 = z0 * a.stride + (a.par ? pz - 1 : -pad), cy0 = y0 * a.stride + (a.par ? py - 1 : -pad),
              cx0 = x0 * a.stride + (a.par ? px - 1 : -pad);
    for (int i = tid; i < HV * 8; i += 256) {
      const int hv = i >> 3, q = i & 7;
      const int pk = a.hvtab[hv];
      const int cx = cx0 + (pk & 511), cy = cy0 + ((pk >> 9) & 511), cz = cz0 + ((pk >> 18) & 255);
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
      const int c = ci0 + 4 * q;
      if (c < Ctot && cz >= 0 && cz < Zc && cy >= 0 && cy < Yc && cx >= 0 && cx < Xc) {
        const int off = ((b * a.Zs + (cz >> a.ups)) * a.Ys + (cy >> a.ups)) * a.Xs + (cx >> a.ups);
        w = (c < a.C0) ? *reinterpret_cast<const f32x4 *>(a.src0 + (size_t)off * a.C0 + c)
                       : *reinterpret_cast<const f32x4 *>(a.src1 + (size_t)off * a.C1 + (c - a.C0));
        if (a.gn) {
          const float *gp = a.gn + (size_t)b * 2 * Ctot + ci0 + 4 * q;
          w = w * *reinterpret_cast<const f32x4 *>(gp) + *reinterpret_cast<const f32x4 *>(gp + Ctot);
          if (a.silu) { w[0] *= sigmoid_f(w[0]); w[1] *= sigmoid_f(w[1]); w[2] *= sigmoid_f(w[2]); w[3] *= sigmoid_f(w[3]); }
        }
        if (a.pm) w = w * *reinterpret_cast<const f32x4 *>(a.pm + (size_t)b * a.pm_stride + ci0 + 4 * q);
      }
      *reinterpret_cast<f32x4 *>(&at[hv * 32 + 4 * q]) = w;
    }
    __syncthreads();
    // voxel pairs outermost, the wave's taps innermost: one dy value and one halo index feed up to
    // seven independent accumulators, so the LDS reads of a pair are issued together
    if (vsplit) {
#pragma unroll 4
      for (int m0 = 2 * wave; m0 < TM; m0 += 8) {
        const float av = dyt[(m0 + h) * 32 + r];
        const float bv0 = at[rowhv[m0 + h] * 32 + r];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv0, acc[0], 0, 0, 0);
      }
      continue;
    }
    if constexpr (ZS) {
      // rows [0,32) lie in plane 0 (their dz = 0 taps read the padding plane: exact zeros, skipped), [32,64) in plane 1
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int dzs = half == 0 ? 1 : 0;    // the two z taps this plane uses: dzs, dzs + 1
        const bool x0 = wave == dzs, x1 = wave == dzs + 1;   // this wave holds the ninth in-plane tap of that z tap
#pragma unroll 2
        for (int m0 = 32 * half; m0 < 32 * half + 32; m0 += 2) {
          const float av = dyt[(m0 + h) * 32 + r];
          const int hb = rowhv[m0 + h] * 32 + r;
          const float b00 = at[hb + tapoff[3 * dzs]], b01 = at[hb + tapoff[3 * dzs + 1]];
          const float b10 = at[hb + tapoff[3 * dzs + 3]], b11 = at[hb + tapoff[3 * dzs + 4]];
          acc[3 * dzs] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b00, acc[3 * dzs], 0, 0, 0);
          acc[3 * dzs + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b01, acc[3 * dzs + 1], 0, 0, 0);
          acc[3 * dzs + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b10, acc[3 * dzs + 3], 0, 0, 0);
          acc[3 * dzs + 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b11, acc[3 * dzs + 4], 0, 0, 0);
          if (x0) acc[3 * dzs + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, at[hb + tapoff[3 * dzs + 2]], acc[3 * dzs + 2], 0, 0, 0);
          if (x1) acc[3 * dzs + 5] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, at[hb + tapoff[3 * dzs + 5]], acc[3 * dzs + 5], 0, 0, 0);
        }
      }
      continue;
    }
#pragma unroll 2
    for (int m0 = 0; m0 < TM; m0 += 2) {
      const float av = dyt[(m0 + h) * 32 + r];
      const int hb = rowhv[m0 + h] * 32 + r;
      float bv[TW];
#pragma unroll
      for (int ti = 0; ti < TW; ++ti) bv[ti] = at[hb + tapoff[ti]];
#pragma unroll
      for (int ti = 0; ti < TW; ++ti)
        if (wave + 4 * ti < ntaps) acc[ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[ti], acc[ti], 0, 0, 0);
    }
  }
  if (vsplit) {   // pseudo-tap = wave: part[g][cb][kb][4][32][32]
    float *p = part + ((((size_t)blockIdx.x * gridDim.y + cb) * gridDim.z + kb) * 4 + wave) * 1024;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int co = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      p[co * 32 + r] = acc[0][reg];
    }
    return;
  }
#pragma unroll
  for (int ti = 0; ti < TW; ++ti) {
    int t = wave + 4 * ti;
    if constexpr (ZS) t = (ti % 3) < 2 ? (ti / 3) * 9 + wave + 4 * (ti % 3) : (wave == ti / 3 ? (ti / 3) * 9 + 8 : ntaps);
    if (t < ntaps) {
      float *p = part + ((((size_t)blockIdx.x * gridDim.y + cb) * gridDim.z + kb) * ntaps + t) * 1024;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int co = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        p[co * 32 + r] = acc[ti][reg];
      }
    }
  }
}

hipError_t launch_wgrad(const ConvArgs &a, int MB, const float *dy, int dy_cs, float *part, int G, int ncb, int nkb,
                        hipStream_t st) {
  if (a.bs != 1 || (a.par && (a.td != 2 || a.ntaps != 8 || a.stride != 1 || a.ups))) return hipErrorInvalidValue;
  const int HV = ((a.bz - 1) * a.stride + a.td) * ((a.by - 1) * a.stride + a.td) * ((a.bx - 1) * a.stride + a.td);
  const int TM = a.zsplit ? 64 : 32 * MB;
  if (a.zsplit && (a.bz != 2 || a.by * a.bx > 32 || a.ntaps != 27 || a.stride != 1 || a.Zo != 2 || a.Zs != 2 || a.par || a.ups)) return hipErrorInvalidValue;
  const size_t lds = ((size_t)2 * TM + (size_t)TM * 32 + (size_t)HV * 32) * 4;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set[dev & 63]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev & 63] = true;
  }
  ConvArgs aa = a;
  aa.dbg = conv_dbg_flags();
  if (a.zsplit) hipLaunchKernelGGL(wgrad_kernel<true>, dim3(G, ncb, nkb), dim3(256), lds, st, aa, dy, dy_cs, part, G);
  else hipLaunchKernelGGL(wgrad_kernel<false>, dim3(a.par ? G * 8 : G, ncb, nkb), dim3(256), lds, st, aa, dy, dy_cs, part, G);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// Two-plane (quarter-resolution) weight gradient (see launch_wgrad_zs).  grid (G, ncb, ceil(nkb / 2)).
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void wgrad_zs_kernel(const ConvArgs a, const float *__restrict__ dy, int dy_cs,
                                                          float *__restrict__ part, int G, int nkb) {
  constexpr int NKB = 2, XS = 32 * NKB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7: in-plane tap q = wave of every z tap
  const int r = lane & 31, h = lane >> 5;
  const int g = blockIdx.x, cb = blockIdx.y, kg = blockIdx.z;
  const int HY = a.by + 2, HX = a.bx + 2, PL = HY * HX, HV = 2 * PL;
  const int nrow = a.by * a.bx;               // valid rows per plane (<= 32)
  int *rowhv = reinterpret_cast<int *>(lds);   // [64] in-plane halo offset (x XS) of a row's own voxel
  float *dyt = lds + 64;                       // [64][32]: rows [0,32) plane 0, [32,64) plane 1
  float *at = dyt + 64 * 32;                   // [2][HY][HX][XS]
  if (tid < 64) {
    const int rr = tid & 31;
    const int y = rr < nrow ? rr / a.bx : 0, x = rr < nrow ? rr % a.bx : 0;
    rowhv[tid] = (y * HX + x) * XS;
  }
  const int Ctot = a.C0 + a.C1;
  const int qy = wave / 3, qx = wave - 3 * qy;               // in-plane tap (dy, dx) = (q / 3, q % 3) of q = wave
  const int off_main = (qy * HX + qx) * XS, off_x = (2 * HX + 2) * XS;   // q = 8: (2, 2)
  f32x16 acc[3][NKB], accx[NKB];               // [z tap][input block]; the extra tap (q = 8) of z tap dz = wave
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int k = 0; k < NKB; ++k)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[d][k][j] = 0.f;
#pragma unroll
  for (int k = 0; k < NKB; ++k)
#pragma unroll
    for (int j = 0; j < 16; ++j) accx[k][j] = 0.f;
  const int ntile = a.B * a.nty * a.ntx;
  // Global loads of a tile are issued one tile AHEAD (registers), so that they complete under the previous tile's
  // matrix phase -- one workgroup per CU (8 accumulator blocks per wave), nothing else would cover their latency.
  constexpr int NH = 5;                        // halo float4 per thread (host: HV * 8 * NKB <= 512 * NH)
  f32x4 dyv, hvv[NH];
  auto issue = [&](int tile) {
    int tt = tile;
    const int tx = tt % a.ntx; tt /= a.ntx;
    const int ty = tt % a.nty;
    const int b = tt / a.nty;
    const int y0 = ty * a.by, x0 = tx * a.bx;
    {
      const int m = tid >> 3, q = tid & 7;       // 64 rows x 8 channel quads = 512 items
      const int z = m >> 5, rr = m & 31;
      dyv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rr < nrow) {
        const int oy = y0 + rr / a.bx, ox = x0 + rr % a.bx, co = cb * 32 + 4 * q;
        if (oy < a.Yo && ox < a.Xo && co < a.Co) {
          const float *p = dy + ((((size_t)b * 2 + z) * a.Yo + oy) * a.Xo + ox) * dy_cs + co;
          if (co + 3 < a.Co) dyv = *reinterpret_cast<const f32x4 *>(p);
          else { dyv[0] = p[0]; dyv[1] = co + 1 < a.Co ? p[1] : 0.f; dyv[2] = co + 2 < a.Co ? p[2] : 0.f; }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const int i = tid + 512 * u;
      hvv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < HV * 8 * NKB) {
        const int hv = i / (8 * NKB), q = i - hv * (8 * NKB);
        const int hz = hv / PL, rem = hv - hz * PL, hy = rem / HX, hx = rem - hy * HX;
        const int cy = y0 + hy - 1, cx = x0 + hx - 1;
        const int c = kg * XS + 4 * q;
        if (c < Ctot && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs) {
          const size_t off = (((size_t)b * 2 + hz) * a.Ys + cy) * a.Xs + cx;
          hvv[u] = (c < a.C0) ? *reinterpret_cast<const f32x4 *>(a.src0 + off * a.C0 + c)
                             : *reinterpret_cast<const f32x4 *>(a.src1 + off * a.C1 + (c - a.C0));
        }
      }
    }
  };
  // activation of the staged input as the forward applies it, then the LDS image; out-of-grid / padding entries stay zero
  auto commit = [&](int tile) {
    int tt = tile;
    const int tx = tt % a.ntx; tt /= a.ntx;
    const int ty = tt % a.nty;
    const int b = tt / a.nty;
    const int y0 = ty * a.by, x0 = tx * a.bx;
    *reinterpret_cast<f32x4 *>(&dyt[(tid >> 3) * 32 + 4 * (tid & 7)]) = dyv;
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const int i = tid + 512 * u;
      if (i < HV * 8 * NKB) {
        const int hv = i / (8 * NKB), q = i - hv * (8 * NKB);
        const int rem = hv % PL, hy = rem / HX, hx = rem - hy * HX;
        const int cy = y0 + hy - 1, cx = x0 + hx - 1;
        const int c = kg * XS + 4 * q;
        f32x4 w = hvv[u];
        if (c < Ctot && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs) {
          if (a.gn) {
            const float *gp = a.gn + (size_t)b * 2 * Ctot + c;
            w = w * *reinterpret_cast<const f32x4 *>(gp) + *reinterpret_cast<const f32x4 *>(gp + Ctot);
            if (a.silu) { w[0] *= sigmoid_f(w[0]); w[1] *= sigmoid_f(w[1]); w[2] *= sigmoid_f(w[2]); w[3] *= sigmoid_f(w[3]); }
          }
          if (a.pm) w = w * *reinterpret_cast<const f32x4 *>(a.pm + (size_t)b * a.pm_stride + c);
        } else {
          w = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        *reinterpret_cast<f32x4 *>(&at[hv * XS + 4 * q]) = w;
      }
    }
  };
  if (g < ntile) issue(g);
  for (int tile = g; tile < ntile; tile += G) {
    __syncthreads();                           // previous tile's operands have been read
    commit(tile);
    __syncthreads();
    if (tile + G < ntile) issue(tile + G);     // in flight under this tile's matrix phase
    // plane z uses the z taps dz = 1 - z, 2 - z ... i.e. z = 0: dz in {1, 2} (source planes 0, 1), z = 1: dz in {0, 1}
#pragma unroll
    for (int z = 0; z < 2; ++z) {
      const int dzs = 1 - z;                     // first of the two z taps of this plane; its source plane is 0, the next one 1
      const bool x0w = wave == dzs, x1w = wave == dzs + 1;
#pragma unroll 2
      for (int m0 = 32 * z; m0 < 32 * z + 32; m0 += 2) {
        const float av = dyt[(m0 + h) * 32 + r];
        const int hb = rowhv[m0 + h] + r;
        float b0[NKB], b1[NKB];
#pragma unroll
        for (int k = 0; k < NKB; ++k) { b0[k] = at[hb + off_main + k * 32]; b1[k] = at[hb + PL * XS + off_main + k * 32]; }
#pragma unroll
        for (int k = 0; k < NKB; ++k) {
          acc[dzs][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[k], acc[dzs][k], 0, 0, 0);
          acc[dzs + 1][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1[k], acc[dzs + 1][k], 0, 0, 0);
        }
        if (x0w) {
#pragma unroll
          for (int k = 0; k < NKB; ++k) accx[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, at[hb + off_x + k * 32], accx[k], 0, 0, 0);
        }
        if (x1w) {
#pragma unroll
          for (int k = 0; k < NKB; ++k) accx[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, at[hb + PL * XS + off_x + k * 32], accx[k], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NKB; ++k) {
    const int kb = kg * NKB + k;
    if (kb >= nkb) continue;
    float *pb = part + (((size_t)g * gridDim.y + cb) * nkb + kb) * 27 * 1024;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float *p = pb + (size_t)(d * 9 + wave) * 1024;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) p[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = acc[d][k][reg];
    }
    if (wave < 3) {
      float *p = pb + (size_t)(wave * 9 + 8) * 1024;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) p[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = accx[k][reg];
    }
  }
}

bool wgrad_zs_ok(const ConvArgs &a) {
  return a.ntaps == 27 && a.stride == 1 && !a.par && !a.ups && a.Zo == 2 && a.Zs == 2 && a.by * a.bx <= 32 && a.Ys == a.Yo && a.Xs == a.Xo &&
         !(a.C0 & 3) && !(a.C1 & 3) && (size_t)2 * (a.by + 2) * (a.bx + 2) * 16 <= 512 * 5 &&
         ((size_t)64 + 64 * 32 + (size_t)2 * (a.by + 2) * (a.bx + 2) * 64) * 4 <= 64 * 1024;
}

hipError_t launch_wgrad_zs(const ConvArgs &a, const float *dy, int dy_cs, float *part, int G, int ncb, int nkb, hipStream_t st) {
  if (!wgrad_zs_ok(a)) return hipErrorInvalidValue;
  if (a.ntaps != 27 || a.stride != 1 || a.par || a.ups || a.Zo != 2 || a.Zs != 2 || a.by * a.bx > 32 || a.Ys != a.Yo || a.Xs != a.Xo ||
      G < 1 || (a.C0 & 3) || (a.C1 & 3))
    return hipErrorInvalidValue;
  const size_t lds = ((size_t)64 + 64 * 32 + (size_t)2 * (a.by + 2) * (a.bx + 2) * 64) * 4;
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  if ((size_t)2 * (a.by + 2) * (a.bx + 2) * 16 > 512 * 5) return hipErrorInvalidValue;   // halo items per thread (NH)
  hipLaunchKernelGGL(wgrad_zs_kernel, dim3((unsigned)G, (unsigned)ncb, (unsigned)((nkb + 1) / 2)), dim3(512), lds, st, a, dy, dy_cs, part, G, nkb);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// Parity-form weight gradient of the upsample convs (see launch_wgrad_par).  grid (G * 8, ncb, nkb).
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_par_kernel(const ConvArgs a, const float *__restrict__ dy, int dy_cs,
                                                        float *__restrict__ part, int G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int par = blockIdx.x & 7, g = blockIdx.x >> 3, cb = blockIdx.y, kb = blockIdx.z;
  const int pz = (par >> 2) & 1, py = (par >> 1) & 1, px = par & 1;
  const int HZ = a.bz + 1, HY = a.by + 1, HX = a.bx + 1, HV = HZ * HY * HX;
  const int nbox = a.bz * a.by * a.bx, TM = (nbox + 1) & ~1;
  int *rowhv = reinterpret_cast<int *>(lds);   // [TM] halo index (x 32) of a row's tap (0,0,0)
  float *dyt = lds + ((TM + 3) & ~3);          // [TM][32]
  float *at = dyt + TM * 32;                   // [HV][32]
  for (int m = tid; m < TM; m += 256) {
    const int mm = m < nbox ? m : 0;
    const int x = mm % a.bx, q = mm / a.bx, y = q % a.by, z = q / a.by;
    rowhv[m] = ((z * HY + y) * HX + x) * 32;
  }
  int tapoff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) tapoff[e] = ((((e >> 2) & 1) * HY + ((e >> 1) & 1)) * HX + (e & 1)) * 32;
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const int ntile = a.B * a.ntz * a.nty * a.ntx;
  const int ci0 = kb * 32, Ctot = a.C0 + a.C1;
  for (int tile = g; tile < ntile; tile += G) {
    int tt = tile;
    const int tx = tt % a.ntx; tt /= a.ntx;
    const int ty = tt % a.nty; tt /= a.nty;
    const int tz = tt % a.ntz;
    const int b = tt / a.ntz;
    const int z0 = tz * a.bz, y0 = ty * a.by, x0 = tx * a.bx;
    __syncthreads();
    // All global loads of the tile first (registers), then the LDS image: a load -> LDS-store loop costs one memory
    // round trip per iteration (12 of them per tile here: 25 us per tile against 4 us of MFMAs).
    constexpr int NDY = 5, NHA = 8;              // dY / halo float4 per thread (host: TM * 8 <= 256 * NDY, HV * 8 <= 256 * NHA)
    f32x4 dyv[NDY], hav[NHA];
    // dY rows of this class: output voxel 2 i + p of low-resolution voxel i
#pragma unroll
    for (int u = 0; u < NDY; ++u) {
      const int i = tid + 256 * u;
      const int m = i >> 3, q = i & 7;
      dyv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < TM * 8 && m < nbox) {
        const int x = m % a.bx, qq = m / a.bx, y = qq % a.by, z = qq / a.by;
        const int iz = z0 + z, iy = y0 + y, ix = x0 + x;
        const int co = cb * 32 + 4 * q;
        if (iz < a.Zs && iy < a.Ys && ix < a.Xs && co < a.Co) {
          const float *p = dy + ((((size_t)b * a.Zo + 2 * iz + pz) * a.Yo + 2 * iy + py) * a.Xo + 2 * ix + px) * dy_cs + co;
          if (co + 3 < a.Co) dyv[u] = *reinterpret_cast<const f32x4 *>(p);
          else { dyv[u][0] = p[0]; dyv[u][1] = co + 1 < a.Co ? p[1] : 0.f; dyv[u][2] = co + 2 < a.Co ? p[2] : 0.f; }
        }
      }
    }
    // low-resolution input halo, origin i0 + p - 1 (zero outside the grid = the padding of the upsampled tensor)
#pragma unroll
    for (int u = 0; u < NHA; ++u) {
      const int i = tid + 256 * u;
      const int hv = i >> 3, q = i & 7;
      hav[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < HV * 8) {
        const int hx = hv % HX, qq = hv / HX, hy = qq % HY, hz = qq / HY;
        const int cz = z0 + hz + pz - 1, cy = y0 + hy + py - 1, cx = x0 + hx + px - 1;
        const int c = ci0 + 4 * q;
        if (c < Ctot && cz >= 0 && cz < a.Zs && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs) {
          const size_t off = (((size_t)b * a.Zs + cz) * a.Ys + cy) * a.Xs + cx;
          hav[u] = (c < a.C0) ? *reinterpret_cast<const f32x4 *>(a.src0 + off * a.C0 + c)
                             : *reinterpret_cast<const f32x4 *>(a.src1 + off * a.C1 + (c - a.C0));
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NDY; ++u) {
      const int i = tid + 256 * u;
      if (i < TM * 8) *reinterpret_cast<f32x4 *>(&dyt[(i >> 3) * 32 + 4 * (i & 7)]) = dyv[u];
    }
#pragma unroll
    for (int u = 0; u < NHA; ++u) {
      const int i = tid + 256 * u;
      if (i < HV * 8) *reinterpret_cast<f32x4 *>(&at[(i >> 3) * 32 + 4 * (i & 7)]) = hav[u];
    }
    __syncthreads();
#pragma unroll 2
    for (int m0 = 2 * wave; m0 < TM; m0 += 8) {
      const float av = dyt[(m0 + h) * 32 + r];
      const int hb = rowhv[m0 + h] + r;
      float bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = at[hb + tapoff[e]];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[e], acc[e], 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) lds[((wave - 1) * 16 + reg) * 64 + lane] = acc[e][reg];
    }
    __syncthreads();
    if (wave == 0) {
      float *p = part + ((((size_t)blockIdx.x * gridDim.y + cb) * gridDim.z + kb) * 8 + e) * 1024;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        float v = acc[e][reg];
#pragma unroll
        for (int w = 0; w < 3; ++w) v += lds[(w * 16 + reg) * 64 + lane];
        p[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = v;
      }
    }
  }
}

hipError_t launch_wgrad_par(const ConvArgs &a, const float *dy, int dy_cs, float *part, int G, int ncb, int nkb, hipStream_t st) {
  if (!a.par || a.gn || a.pm || a.ups || G < 1 || (a.C0 & 3) || (a.C1 & 3) || a.Zo != 2 * a.Zs || a.Yo != 2 * a.Ys || a.Xo != 2 * a.Xs)
    return hipErrorInvalidValue;
  const int nbox = a.bz * a.by * a.bx, TM = (nbox + 1) & ~1, HV = (a.bz + 1) * (a.by + 1) * (a.bx + 1);
  if (TM > 160 || HV > 256) return hipErrorInvalidValue;   // staging registers: 5 dY + 8 halo float4 per thread
  const size_t lds = std::max<size_t>(((size_t)((TM + 3) & ~3) + (size_t)TM * 32 + (size_t)HV * 32) * 4, (size_t)3 * 1024 * 4);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(wgrad_par_kernel, dim3((unsigned)G * 8, (unsigned)ncb, (unsigned)nkb), dim3(256), lds, st, a, dy, dy_cs, part, G);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// 1x1x1 weight gradient over flat rows (see launch_wgrad_1x1).  grid (G, ncb, nkb / NKB).
// --------------------------------------------------------------------------------
template <int NKB>
__global__ __launch_bounds__(256) void wgrad_1x1_kernel(const ConvArgs a, const float *__restrict__ dy, int dy_cs, long long V,
                                                        float *__restrict__ part, int G, int nkb) {
  constexpr int TM = 128, XS = 32 * NKB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *dyt = lds;                 // [TM][32]
  float *xt = lds + TM * 32;        // [TM][32 * NKB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int cb = blockIdx.y, kg = blockIdx.z;
  const int Ctot = a.C0 + a.C1;
  const long long N = (long long)a.B * V;
  const long long nchunk = (N + TM - 1) / TM;
  f32x16 acc[NKB];
#pragma unroll
  for (int i = 0; i < NKB; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (long long ch = blockIdx.x; ch < nchunk; ch += G) {
    const long long n0 = ch * TM;
    // every global load of the chunk first (registers), then activation + LDS image (a load -> store loop would pay one
    // memory round trip per iteration: 4 + 4 NKB of them)
    f32x4 dyv[4], xv[4 * NKB];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u;
      const int m = i >> 3, q = i & 7;
      const long long n = n0 + m;
      dyv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (n < N) {
        const int co = cb * 32 + 4 * q;
        const float *p = dy + (size_t)n * dy_cs + co;
        if (co + 3 < a.Co) dyv[u] = *reinterpret_cast<const f32x4 *>(p);
        else { dyv[u][0] = co < a.Co ? p[0] : 0.f; dyv[u][1] = co + 1 < a.Co ? p[1] : 0.f; dyv[u][2] = co + 2 < a.Co ? p[2] : 0.f; }
      }
    }
#pragma unroll
    for (int u = 0; u < 4 * NKB; ++u) {
      const int i = tid + 256 * u;
      const int m = i / (8 * NKB), q = i - m * (8 * NKB);
      const long long n = n0 + m;
      const int c = kg * XS + 4 * q;
      xv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (n < N && c < Ctot)
        xv[u] = (c < a.C0) ? *reinterpret_cast<const f32x4 *>(a.src0 + (size_t)n * a.C0 + c)
                           : *reinterpret_cast<const f32x4 *>(a.src1 + (size_t)n * a.C1 + (c - a.C0));
    }
    __syncthreads();                             // previous chunk's tiles consumed
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u;
      *reinterpret_cast<f32x4 *>(&dyt[(i >> 3) * 32 + 4 * (i & 7)]) = dyv[u];
    }
#pragma unroll
    for (int u = 0; u < 4 * NKB; ++u) {
      const int i = tid + 256 * u;
      const int m = i / (8 * NKB), q = i - m * (8 * NKB);
      const long long n = n0 + m;
      const int c = kg * XS + 4 * q;
      f32x4 w = xv[u];
      if (n < N && c < Ctot) {
        const int b = (int)(n / V);
        if (a.gn) {
          const float *gp = a.gn + (size_t)b * 2 * Ctot + c;
          w = w * *reinterpret_cast<const f32x4 *>(gp) + *reinterpret_cast<const f32x4 *>(gp + Ctot);
          if (a.silu) { w[0] *= sigmoid_f(w[0]); w[1] *= sigmoid_f(w[1]); w[2] *= sigmoid_f(w[2]); w[3] *= sigmoid_f(w[3]); }
        }
        if (a.pm) w = w * *reinterpret_cast<const f32x4 *>(a.pm + (size_t)b * a.pm_stride + c);
      }
      *reinterpret_cast<f32x4 *>(&xt[m * XS + 4 * q]) = w;
    }
    __syncthreads();
#pragma unroll 4
    for (int m0 = 2 * wave; m0 < TM; m0 += 8) {
      const float av = dyt[(m0 + h) * 32 + r];
      float bv[NKB];
#pragma unroll
      for (int k = 0; k < NKB; ++k) bv[k] = xt[(m0 + h) * XS + k * 32 + r];
#pragma unroll
      for (int k = 0; k < NKB; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[k], acc[k], 0, 0, 0);
    }
  }
#pragma unroll
  for (int k = 0; k < NKB; ++k) {
    __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) lds[((wave - 1) * 16 + reg) * 64 + lane] = acc[k][reg];
    }
    __syncthreads();
    const int kb = kg * NKB + k;
    if (wave == 0 && kb < nkb) {
      float *p = part + (((size_t)blockIdx.x * gridDim.y + cb) * nkb + kb) * 1024;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        float v = acc[k][reg];
#pragma unroll
        for (int w = 0; w < 3; ++w) v += lds[(w * 16 + reg) * 64 + lane];
        p[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = v;
      }
    }
  }
}

hipError_t launch_wgrad_1x1(const ConvArgs &a, const float *dy, int dy_cs, long long V, float *part, int G, int ncb, int nkb, int NKB,
                            hipStream_t st) {
  if (a.ntaps != 1 || a.stride != 1 || a.par || a.ups || G < 1 || (a.C0 & 3) || (a.C1 & 3)) return hipErrorInvalidValue;
  const dim3 grid((unsigned)G, (unsigned)ncb, (unsigned)((nkb + NKB - 1) / NKB));
  const size_t lds = (size_t)128 * 32 * (1 + NKB) * 4;
#define CM_W1(K)                                                                                          \
  if (NKB == K) {                                                                                         \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_1x1_kernel<K>),               \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);           \
    if (e != hipSuccess) return e;                                                                        \
    hipLaunchKernelGGL((wgrad_1x1_kernel<K>), grid, dim3(256), lds, st, a, dy, dy_cs, V, part, G, nkb);    \
    return hipGetLastError();                                                                             \
  }
  CM_W1(1) CM_W1(2) CM_W1(3) CM_W1(4)
#undef CM_W1
  return hipErrorInvalidValue;
}

// --------------------------------------------------------------------------------
// Packed-column weight gradient (see WgradPackArgs): the four waves split the voxel pairs of a tile, each keeps all
// NBLK accumulator blocks; a lane's column (tap, narrow channel) is a fixed offset into the staged narrow halo.
// --------------------------------------------------------------------------------
template <int CN>
__global__ __launch_bounds__(256) void wgrad_pack_kernel(const WgradPackArgs a, int G) {
  constexpr int NBLK = (27 * CN + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int HZ = a.bz + 2, HY = a.by + 2, HX = a.bx + 2, HV = HZ * HY * HX;
  const int TM = a.bz * a.by * a.bx;
  int *rowhv = reinterpret_cast<int *>(lds);   // [TM] halo index of a row's own voxel at tap (0,0,0)
  float *wt = lds + ((TM + 3) & ~3);           // [TM][32]
  float *nt = wt + TM * 32;                    // [HV][CN]
  for (int m = tid; m < TM; m += 256) {
    const int x = m % a.bx, q = m / a.bx, y = q % a.by, z = q / a.by;
    rowhv[m] = ((z * HY + y) * HX + x) * CN;
  }
  int coloff[NBLK];
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    const int col = blk * 32 + r, tap = col / CN, ch = col - tap * CN;
    int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
    if (a.mirror) { dz = 2 - dz; dy = 2 - dy; dx = 2 - dx; }
    coloff[blk] = tap < 27 ? ((dz * HY + dy) * HX + dx) * CN + ch : 0;   // (columns beyond 27 taps: ignored by the reduce)
  }
  f32x16 acc[NBLK];
#pragma unroll
  for (int i = 0; i < NBLK; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const int ntile = a.B * a.ntz * a.nty * a.ntx;
  for (int tile = blockIdx.x; tile < ntile; tile += G) {
    int tt = tile;
    const int tx = tt % a.ntx; tt /= a.ntx;
    const int ty = tt % a.nty; tt /= a.nty;
    const int tz = tt % a.ntz;
    const int b = tt / a.ntz;
    const int z0 = tz * a.bz, y0 = ty * a.by, x0 = tx * a.bx;
    // all global loads of the tile first (registers), then activation + LDS image (see wgrad_par_kernel)
    constexpr int NWD = 8, NNA = 5;              // wide / narrow float4 per thread (host: TM <= 256, HV * CN / 4 <= 256 * NNA)
    f32x4 wv[NWD], nv[NNA];
#pragma unroll
    for (int u = 0; u < NWD; ++u) {
      const int i = tid + 256 * u;
      const int m = i >> 3, q = i & 7;
      wv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < TM * 8) {
        const int x = m % a.bx, qq = m / a.bx, y = qq % a.by, z = qq / a.by;
        const int cz = z0 + z, cy = y0 + y, cx = x0 + x;
        if (cz < a.Z && cy < a.Y && cx < a.X)
          wv[u] = *reinterpret_cast<const f32x4 *>(a.wide + ((((size_t)b * a.Z + cz) * a.Y + cy) * a.X + cx) * a.wide_cs + 4 * q);
      }
    }
#pragma unroll
    for (int u = 0; u < NNA; ++u) {
      const int i = tid + 256 * u;
      nv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < HV * (CN / 4)) {
        const int hv = i / (CN / 4), q = i - hv * (CN / 4);
        const int hx = hv % HX, qq = hv / HX, hy = qq % HY, hz = qq / HY;
        const int cz = z0 + hz - 1, cy = y0 + hy - 1, cx = x0 + hx - 1;
        if (cz >= 0 && cz < a.Z && cy >= 0 && cy < a.Y && cx >= 0 && cx < a.X) {
          const float *p = a.narrow + ((((size_t)b * a.Z + cz) * a.Y + cy) * a.X + cx) * a.narrow_cs + 4 * q;
#pragma unroll
          for (int e = 0; e < 4; ++e) nv[u][e] = 4 * q + e < a.cn_valid ? p[e] : 0.f;
        }
      }
    }
    __syncthreads();                           // previous tile consumed (and rowhv written)
#pragma unroll
    for (int u = 0; u < NWD; ++u) {
      const int i = tid + 256 * u;
      if (i < TM * 8) {
        const int m = i >> 3, q = i & 7;
        const int x = m % a.bx, qq = m / a.bx, y = qq % a.by, z = qq / a.by;
        f32x4 w = wv[u];
        if (a.gn && z0 + z < a.Z && y0 + y < a.Y && x0 + x < a.X) {   // (rows outside the grid stay zero)
          const float *gp = a.gn + (size_t)b * 64 + 4 * q;
          w = w * *reinterpret_cast<const f32x4 *>(gp) + *reinterpret_cast<const f32x4 *>(gp + 32);
          if (a.silu) { w[0] *= sigmoid_f(w[0]); w[1] *= sigmoid_f(w[1]); w[2] *= sigmoid_f(w[2]); w[3] *= sigmoid_f(w[3]); }
        }
        *reinterpret_cast<f32x4 *>(&wt[m * 32 + 4 * q]) = w;
      }
    }
#pragma unroll
    for (int u = 0; u < NNA; ++u) {
      const int i = tid + 256 * u;
      if (i < HV * (CN / 4)) *reinterpret_cast<f32x4 *>(&nt[(i / (CN / 4)) * CN + 4 * (i % (CN / 4))]) = nv[u];
    }
    __syncthreads();
#pragma unroll 2
    for (int m0 = 2 * wave; m0 < TM; m0 += 8) {
      const float av = wt[(m0 + h) * 32 + r];
      const int hb = rowhv[m0 + h];
      float bv[NBLK];
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk) bv[blk] = nt[hb + coloff[blk]];
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[blk], acc[blk], 0, 0, 0);
    }
  }
  // cross-wave sum through LDS, block by block, in wave order (the staging buffers are dead); wave 0 stores
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) lds[((wave - 1) * 16 + reg) * 64 + lane] = acc[blk][reg];
    }
    __syncthreads();
    if (wave == 0) {
      float *p = a.part + ((size_t)blockIdx.x * NBLK + blk) * 1024;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        float v = acc[blk][reg];
#pragma unroll
        for (int w = 0; w < 3; ++w) v += lds[(w * 16 + reg) * 64 + lane];
        p[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = v;
      }
    }
  }
}

hipError_t launch_wgrad_pack(const WgradPackArgs &a, int CN, int G, hipStream_t st) {
  const int TM = a.bz * a.by * a.bx, HV = (a.bz + 2) * (a.by + 2) * (a.bx + 2);
  if (TM > 256 || (TM & 1) || (CN != 4 && CN != 8) || a.cn_valid > CN || G < 1 || HV * (CN / 4) > 256 * 5) return hipErrorInvalidValue;
  const size_t lds = std::max<size_t>(((size_t)((TM + 3) & ~3) + (size_t)TM * 32 + (size_t)HV * CN) * 4, (size_t)3 * 1024 * 4);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  if (CN == 4) hipLaunchKernelGGL(wgrad_pack_kernel<4>, dim3(G), dim3(256), lds, st, a, G);
  else hipLaunchKernelGGL(wgrad_pack_kernel<8>, dim3(G), dim3(256), lds, st, a, G);
  return hipGetLastError();
}

// Threads walk the partial layout (column fastest, coalesced); four lanes per element take the partials
// p = lane, lane + 4, ... and are merged in lane order: a fixed summation order.  Internal tap (dz,dy,dx) =
// reference [kH=dy][kW=dx][kL=dz] as in wgrad_reduce_kernel.
__global__ __launch_bounds__(256) void wgrad_pack_reduce_kernel(const float *__restrict__ part, int G, int CN, int cn_valid, int mirror,
                                                                int Co, int Ci, float *__restrict__ dW) {
  __shared__ float sh[256];
  const int NBLK = (27 * CN + 31) / 32;
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), gl = threadIdx.x >> 6;
  const int total = NBLK * 1024;
  float s = 0.f;
  if (e < total)
    for (int p = gl; p < G; p += 4) s += part[(size_t)p * total + e];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (gl != 0 || e >= total) return;
  const float t = ((sh[threadIdx.x] + sh[threadIdx.x + 64]) + sh[threadIdx.x + 128]) + sh[threadIdx.x + 192];
  const int blk = e >> 10, i = (e >> 5) & 31, j = e & 31;
  const int col = blk * 32 + j, tap = col / CN, ch = col - tap * CN;
  if (tap >= 27 || ch >= cn_valid) return;
  const int co = mirror ? ch : i, ci = mirror ? i : ch;
  const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
  if (co < Co && ci < Ci) dW[((size_t)co * Ci + ci) * 27 + (dy * 3 + dx) * 3 + dz] = t;
}

hipError_t launch_wgrad_pack_reduce(const float *part, int G, int CN, int cn_valid, int mirror, int Co, int Ci, float *dW, hipStream_t st) {
  const int total = ((27 * CN + 31) / 32) * 1024;
  hipLaunchKernelGGL(wgrad_pack_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, st, part, G, CN, cn_valid, mirror, Co, Ci, dW);
  return hipGetLastError();
}

// Parity-form partials part[g * 8 + p][cb][kb][e (8)][32 co][32 ci] -> dW[co][ci][27 taps] (reference layout):
// the forward's parity weights are W_p[e] = sum of the taps d that land on source offset e for parity p, so
// dW[d] = sum_p dW_p[e(p, d)]  with  e = emap(p_axis, d_axis) per axis:  p = 0: d0 -> 0, d1,d2 -> 1;  p = 1: d0,d1 -> 0, d2 -> 1.
__global__ __launch_bounds__(256) void wgrad_reduce_par_kernel(const float *__restrict__ part, int G, int ncb, int nkb, int Co, int Ci,
                                                               float *__restrict__ dW) {
  const long long n = (long long)ncb * nkb * 27 * 1024;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int cil = (int)(i & 31), col = (int)((i >> 5) & 31);
  long long q = i >> 10;
  const int d = (int)(q % 27); q /= 27;
  const int kb = (int)(q % nkb);
  const int cb = (int)(q / nkb);
  const int co = cb * 32 + col, ci = kb * 32 + cil;
  if (co >= Co || ci >= Ci) return;
  const int dz = d / 9, dy = (d / 3) % 3, dx = d % 3;
  const long long per_gp = (long long)ncb * nkb * 8 * 1024;
  float s = 0.f;
  for (int g = 0; g < G; ++g)
    for (int p = 0; p < 8; ++p) {
      const int pz = (p >> 2) & 1, py = (p >> 1) & 1, px = p & 1;
      const int ez = pz == 0 ? (dz == 0 ? 0 : 1) : (dz == 2 ? 1 : 0);
      const int ey = py == 0 ? (dy == 0 ? 0 : 1) : (dy == 2 ? 1 : 0);
      const int ex = px == 0 ? (dx == 0 ? 0 : 1) : (dx == 2 ? 1 : 0);
      const int e = (ez * 2 + ey) * 2 + ex;
      s += part[(size_t)(g * 8 + p) * per_gp + ((((size_t)cb * nkb + kb) * 8 + e) << 10) + col * 32 + cil];
    }
  dW[((size_t)co * Ci + ci) * 27 + (dy * 3 + dx) * 3 + dz] = s;
}

hipError_t launch_wgrad_reduce_par(const float *part, int G, int ncb, int nkb, int Co, int Ci, float *dW, hipStream_t st) {
  const long long n = (long long)ncb * nkb * 27 * 1024;
  hipLaunchKernelGGL(wgrad_reduce_par_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, G, ncb, nkb, Co, Ci, dW);
  return hipGetLastError();
}

// part[g][cb][kb][tap][32 co][32 ci] summed over g (fixed order) -> reference weight layout
// [Co][Ci][kH][kW][kL] (internal tap (dz,dy,dx) = reference [kH=dy][kW=dx][kL=dz]); 1 tap -> [Co][Ci].
// Threads walk the PARTIAL layout (ci fastest) so the G reads per output are coalesced.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int G, int ncb, int nkb,
                                                           int ntaps, int Co, int Ci, float *__restrict__ dW, int nvs) {
  if (ntaps == 1 && nvs == 4) {   // 1x1x1: the kernel wrote four pseudo-tap (per-wave) partials per group
    const long long n1 = (long long)ncb * nkb * 1024;
    const long long i1 = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i1 >= n1) return;
    const int cil1 = (int)(i1 & 31), col1 = (int)((i1 >> 5) & 31);
    const long long blk = i1 >> 10;
    const int co1 = (int)(blk / nkb) * 32 + col1, ci1 = (int)(blk % nkb) * 32 + cil1;
    if (co1 >= Co || ci1 >= Ci) return;
    float s = 0.f;
    for (int g = 0; g < G; ++g)
      for (int w = 0; w < 4; ++w) s += part[((size_t)g * ncb * nkb + blk) * 4096 + (size_t)w * 1024 + col1 * 32 + cil1];
    dW[(size_t)co1 * Ci + ci1] = s;
    return;
  }
  const long long per_g = (long long)ncb * nkb * ntaps * 1024;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_g) return;
  const int cil = (int)(i & 31), col = (int)((i >> 5) & 31);
  long long q = i >> 10;
  const int t = (int)(q % ntaps); q /= ntaps;
  const int kb = (int)(q % nkb);
  const int cb = (int)(q / nkb);
  const int co = cb * 32 + col, ci = kb * 32 + cil;
  if (co >= Co || ci >= Ci) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int g = 0;
  for (; g + 3 < G; g += 4) {
    s0 += part[(size_t)g * per_g + i];
    s1 += part[(size_t)(g + 1) * per_g + i];
    s2 += part[(size_t)(g + 2) * per_g + i];
    s3 += part[(size_t)(g + 3) * per_g + i];
  }
  for (; g < G; ++g) s0 += part[(size_t)g * per_g + i];
  int tref = 0;
  if (ntaps == 27) { const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3; tref = (dy * 3 + dx) * 3 + dz; }
  dW[((size_t)co * Ci + ci) * ntaps + tref] = (s0 + s1) + (s2 + s3);
}

hipError_t launch_wgrad_reduce(const float *part, int G, int ncb, int nkb, int ntaps, int Co, int Ci, float *dW,
                               hipStream_t st, int force_nvs) {
  const long long total = (long long)ncb * nkb * ntaps * 1024;
  const int nvs = force_nvs ? force_nvs : (ntaps == 1 && !(conv_dbg_flags() & 8192)) ? 4 : 1;   // per-wave partials of the 1x1x1 voxel split
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, G, ncb, nkb,
                     ntaps, Co, Ci, dW, nvs);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// Per-(sample, channel) sums over the voxels of a channels-last tensor: out[b][c] = sum_v x[b][v][c].
// Gives the bias gradient (summed over b afterwards) and the gradient of the broadcast
// time-embedding term h += dense_1(...)[:, :, None, None, None] (layers.py:62).
// grid B, 256 threads, fixed summation order.
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void voxel_sum_kernel(const float *__restrict__ x, int V, int C, int cs,
                                                        float *__restrict__ part, int ostride, int B, int S) {
  __shared__ float sh[256];
  const int b = blockIdx.x, sl = blockIdx.z, tid = threadIdx.x;
  const int c = blockIdx.y * 32 + (tid & 31), vl = tid >> 5;  // 8 voxel lanes
  const int vs = (V + S - 1) / S, v0 = sl * vs, v1 = min(V, v0 + vs);
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float *p = x + (size_t)b * V * cs + c;
    int v = v0 + vl;
    for (; v + 8 < v1; v += 16) { s0 += p[(size_t)v * cs]; s1 += p[(size_t)(v + 8) * cs]; }
    if (v < v1) s0 += p[(size_t)v * cs];
  }
  sh[tid] = s0 + s1;
  __syncthreads();
  if (tid < 32 && c < C) {
    float t = 0.f;
    for (int l = 0; l < 8; ++l) t += sh[l * 32 + tid];
    part[((size_t)sl * B + b) * ostride + c] = t;
  }
}

// out[b][c] = sum_s part[s][b][c]  (fixed order)
__global__ void slice_sum_kernel(const float *__restrict__ part, int S, int B, int C, int stride, float *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float t = 0.f;
  for (int s = 0; s < S; ++s) t += part[((size_t)s * B + b) * stride + c];
  out[(size_t)b * stride + c] = t;
}

hipError_t launch_voxel_sum(const float *x, int B, int V, int C, int cs, float *out, int ostride, float *scratch,
                            hipStream_t st) {
  // enough workgroups for the chip: voxel slices when (samples x channel blocks) alone are too few
  const int blocks = B * ((C + 31) / 32);
  int S = 1;
  while (S < 16 && blocks * S < 1024 && V / (S * 2) >= 64) S *= 2;
  if (S == 1) {
    hipLaunchKernelGGL(voxel_sum_kernel, dim3(B, (C + 31) / 32, 1), dim3(256), 0, st, x, V, C, cs, out, ostride, B, 1);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(voxel_sum_kernel, dim3(B, (C + 31) / 32, S), dim3(256), 0, st, x, V, C, cs, scratch, ostride, B, S);
  hipLaunchKernelGGL(slice_sum_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, scratch, S, B, C, ostride, out);
  return hipGetLastError();
}

// out[c] (+)= sum_b in[b][c]; one workgroup per 32 channels, 8 sample lanes, fixed-order tree
__global__ __launch_bounds__(256) void batch_sum_kernel(const float *__restrict__ in, int B, int C, int stride,
                                                        float *__restrict__ out, int accumulate) {
  __shared__ float sh[256];
  const int tid = threadIdx.x;
  const int c = blockIdx.x * 32 + (tid & 31), bl = tid >> 5;
  float s = 0.f;
  if (c < C)
    for (int b = bl; b < B; b += 8) s += in[(size_t)b * stride + c];
  sh[tid] = s;
  __syncthreads();
  if (tid < 32 && c < C) {
    float t = 0.f;
    for (int l = 0; l < 8; ++l) t += sh[l * 32 + tid];
    out[c] = accumulate ? out[c] + t : t;
  }
}

__global__ __launch_bounds__(256) void copy_jobs_kernel(const CopyJob *__restrict__ jobs) {
  const CopyJob j = jobs[blockIdx.y];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < j.n; i += (long long)gridDim.x * 256) j.dst[i] = j.src[i];
}

hipError_t launch_copy_jobs(const CopyJob *jobs, int njobs, long long max_n, hipStream_t st) {
  if (njobs <= 0) return hipSuccess;
  const unsigned gx = (unsigned)std::min<long long>(64, (max_n + 255) / 256);
  hipLaunchKernelGGL(copy_jobs_kernel, dim3(gx, (unsigned)njobs), dim3(256), 0, st, jobs);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void voxel_sum_jobs_kernel(const VsumJob *__restrict__ jobs, int ncb_max) {
  __shared__ float sh[256];
  const VsumJob j = jobs[blockIdx.y];
  const int b = blockIdx.x / ncb_max, cbk = blockIdx.x - b * ncb_max;
  if (cbk * 32 >= j.C) return;                 // (workgroup-uniform)
  const int tid = threadIdx.x;
  const int c = cbk * 32 + (tid & 31), vl = tid >> 5;   // 8 voxel lanes, four running sums each (fixed order)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < j.C) {
    const float *p = j.x + (size_t)b * j.V * j.cs + c;
    int v = vl;
    for (; v + 24 < j.V; v += 32) {
      s0 += p[(size_t)v * j.cs]; s1 += p[(size_t)(v + 8) * j.cs]; s2 += p[(size_t)(v + 16) * j.cs]; s3 += p[(size_t)(v + 24) * j.cs];
    }
    for (; v < j.V; v += 8) s0 += p[(size_t)v * j.cs];
  }
  sh[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (tid < 32 && c < j.C) {
    float t = 0.f;
    for (int l = 0; l < 8; ++l) t += sh[l * 32 + tid];
    j.out[(size_t)b * j.ostride + c] = t;
    if (j.out2) j.out2[(size_t)b * j.ostride2 + c] = t;
  }
}

hipError_t launch_voxel_sum_jobs(const VsumJob *jobs, int njobs, int B, int maxC, hipStream_t st) {
  if (njobs <= 0) return hipSuccess;
  const int ncb = (maxC + 31) / 32;
  hipLaunchKernelGGL(voxel_sum_jobs_kernel, dim3((unsigned)(B * ncb), (unsigned)njobs), dim3(256), 0, st, jobs, ncb);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void batch_sum_jobs_kernel(const BsumJob *__restrict__ jobs, int B) {
  __shared__ float sh[256];
  const BsumJob j = jobs[blockIdx.y];
  if ((int)blockIdx.x * 32 >= j.C) return;    // (workgroup-uniform)
  const int tid = threadIdx.x;
  const int c = blockIdx.x * 32 + (tid & 31), bl = tid >> 5;
  float s = 0.f;
  if (c < j.C)
    for (int b = bl; b < B; b += 8) s += j.in[(size_t)b * j.stride + c];
  sh[tid] = s;
  __syncthreads();
  if (tid < 32 && c < j.C) {
    float t = 0.f;
    for (int l = 0; l < 8; ++l) t += sh[l * 32 + tid];
    j.out[c] = t;
  }
}

hipError_t launch_batch_sum_jobs(const BsumJob *jobs, int njobs, int B, int maxC, hipStream_t st) {
  if (njobs <= 0) return hipSuccess;
  hipLaunchKernelGGL(batch_sum_jobs_kernel, dim3((maxC + 31) / 32, njobs), dim3(256), 0, st, jobs, B);
  return hipGetLastError();
}

hipError_t launch_batch_sum(const float *in, int B, int C, int stride, float *out, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(batch_sum_kernel, dim3((C + 31) / 32), dim3(256), 0, st, in, B, C, stride, out, accumulate);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// GroupNorm (+SiLU, +Dropout3d multiplier) backward, three small kernels.
//   forward:  xh = (x - mu_g) * rstd_g ;  y = gamma*xh + beta ;  a = pm * silu(y)      (silu optional)
//   given dA: dy = dA * pm * silu'(y)
//             dgamma_c = sum_{b,v} dy*xh ; dbeta_c = sum_{b,v} dy
//             dx = rstd*gamma*dy - (rstd/N) * (S1_g + xh * S2_g),  S1 = sum_g gamma*dy, S2 = sum_g gamma*dy*xh
// x may be the channel concatenation of two tensors (groups may straddle the boundary).
// --------------------------------------------------------------------------------

__device__ __forceinline__ float gnb_dy(const GnbArgs &a, int b, int c, int Ctot, float x, float dA, float &xh) {
  const float mu = a.mr[((size_t)b * 2) * Ctot + c], rstd = a.mr[((size_t)b * 2 + 1) * Ctot + c];
  xh = (x - mu) * rstd;
  float d = dA;
  if (a.pm) d *= a.pm[(size_t)b * a.pm_stride + c];
  if (a.silu) {
    const float y = a.gn[((size_t)b * 2) * Ctot + c] * x + a.gn[((size_t)b * 2 + 1) * Ctot + c];
    const float s = sigmoid_f(y);
    d *= s * (1.0f + y * (1.0f - s));
  }
  return d;
}

__global__ __launch_bounds__(256) void gnb_reduce_kernel(const GnbArgs a) {
  // thread = (channel quad of a 32-channel block, one of 32 voxel lanes): float4 loads, the per-(sample, channel) rows in
  // registers; lanes merged through LDS in lane order
  __shared__ f32x4 sh[2][256];
  const int sl = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int Ctot = a.C0 + a.C1;
  const int vs = (a.V + a.nsl - 1) / a.nsl, v0 = sl * vs, v1 = min(a.V, v0 + vs);
  const int q = tid & 7, vl = tid >> 3;
  for (int cbase = 0; cbase < Ctot; cbase += 32) {
    const int c = cbase + 4 * q;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (c < Ctot) {
      const float *xp; int Cx, cc;
      if (c < a.C0) { xp = a.x0; Cx = a.C0; cc = c; } else { xp = a.x1; Cx = a.C1; cc = c - a.C0; }
      const float *mrp = a.mr + (size_t)b * 2 * Ctot + c;
      const f32x4 mu = *reinterpret_cast<const f32x4 *>(mrp), rstd = *reinterpret_cast<const f32x4 *>(mrp + Ctot);
      f32x4 pm = {1.f, 1.f, 1.f, 1.f}, gs = {0.f, 0.f, 0.f, 0.f}, gh = {0.f, 0.f, 0.f, 0.f};
      if (a.pm) pm = *reinterpret_cast<const f32x4 *>(a.pm + (size_t)b * a.pm_stride + c);
      if (a.silu) {
        const float *gnp = a.gn + (size_t)b * 2 * Ctot + c;
        gs = *reinterpret_cast<const f32x4 *>(gnp); gh = *reinterpret_cast<const f32x4 *>(gnp + Ctot);
      }
      for (int v = v0 + vl; v < v1; v += 32) {
        const size_t row = (size_t)b * a.V + v;
        const f32x4 x = *reinterpret_cast<const f32x4 *>(xp + row * Cx + cc);
        f32x4 d = *reinterpret_cast<const f32x4 *>(a.dA + row * a.dA_cs + c);
        const f32x4 xh = (x - mu) * rstd;
        if (a.pm) d = d * pm;
        if (a.silu) {
          const f32x4 y = gs * x + gh;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float sg = sigmoid_f(y[e]); d[e] *= sg * (1.0f + y[e] * (1.0f - sg)); }
        }
        s1 += d;
        s2 += d * xh;
      }
    }
    sh[0][tid] = s1; sh[1][tid] = s2;
    __syncthreads();
    if (tid < 8 && c < Ctot) {
      f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
      for (int l = 0; l < 32; ++l) { t1 += sh[0][l * 8 + tid]; t2 += sh[1][l * 8 + tid]; }
      float *p = a.part + (((size_t)b * a.nsl + sl) * Ctot + c) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { p[2 * e] = t1[e]; p[2 * e + 1] = t2[e]; }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void gnb_finalize_kernel(const GnbArgs a) {
  extern __shared__ float sm[];  // [Ctot] s1, [Ctot] s2, [groups] S1, [groups] S2
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Ctot = a.C0 + a.C1;
  float *s1 = sm, *s2 = sm + Ctot, *S1 = sm + 2 * Ctot, *S2 = S1 + a.groups;
  for (int c = tid; c < Ctot; c += 256) {
    float t1 = 0.f, t2 = 0.f;
    for (int s = 0; s < a.nsl; ++s) {
      const float *p = a.part + (((size_t)b * a.nsl + s) * Ctot + c) * 2;
      t1 += p[0]; t2 += p[1];
    }
    s1[c] = t1; s2[c] = t2;
    a.dgb[((size_t)b * 2) * Ctot + c] = t2;       // dgamma contribution of this sample
    a.dgb[((size_t)b * 2 + 1) * Ctot + c] = t1;   // dbeta
  }
  __syncthreads();
  const int cg = Ctot / a.groups;
  if (tid < a.groups) {
    float t1 = 0.f, t2 = 0.f;
    for (int i = 0; i < cg; ++i) { const int c = tid * cg + i; t1 += a.gamma[c] * s1[c]; t2 += a.gamma[c] * s2[c]; }
    S1[tid] = t1; S2[tid] = t2;
  }
  __syncthreads();
  const float invN = 1.0f / ((float)cg * (float)a.V);
  for (int c = tid; c < Ctot; c += 256) {
    const int gidx = c / cg;
    const float rstd = a.mr[((size_t)b * 2 + 1) * Ctot + c];
    a.coef[((size_t)b * 3 + 0) * Ctot + c] = rstd * a.gamma[c];
    a.coef[((size_t)b * 3 + 1) * Ctot + c] = rstd * S1[gidx] * invN;
    a.coef[((size_t)b * 3 + 2) * Ctot + c] = rstd * S2[gidx] * invN;
  }
}

__global__ __launch_bounds__(256) void gnb_apply_kernel(const GnbArgs a) {
  // one channel QUAD of one row per thread (float4 loads / stores; 32-bit index arithmetic): the scalar form spent most
  // of its instructions on a 64-bit division and ten 4-byte loads per element
  const int Ctot = a.C0 + a.C1, Q = Ctot >> 2;
  const unsigned total = (unsigned)a.B * (unsigned)a.V * (unsigned)Q;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned row = i / (unsigned)Q;
  const int c = (int)(i - row * (unsigned)Q) * 4;
  const int b = (int)(row / (unsigned)a.V);
  const float *xp; float *gp; int Cx, cc, accf;
  if (c < a.C0) { xp = a.x0; gp = a.g0; Cx = a.C0; cc = c; accf = a.acc0; }
  else { xp = a.x1; gp = a.g1; Cx = a.C1; cc = c - a.C0; accf = a.acc1; }
  const f32x4 x = *reinterpret_cast<const f32x4 *>(xp + (size_t)row * Cx + cc);
  f32x4 d = *reinterpret_cast<const f32x4 *>(a.dA + (size_t)row * a.dA_cs + c);
  const float *mrp = a.mr + (size_t)b * 2 * Ctot + c;
  const f32x4 mu = *reinterpret_cast<const f32x4 *>(mrp), rstd = *reinterpret_cast<const f32x4 *>(mrp + Ctot);
  const f32x4 xh = (x - mu) * rstd;
  if (a.pm) d = d * *reinterpret_cast<const f32x4 *>(a.pm + (size_t)b * a.pm_stride + c);
  if (a.silu) {
    const float *gnp = a.gn + (size_t)b * 2 * Ctot + c;
    const f32x4 y = *reinterpret_cast<const f32x4 *>(gnp) * x + *reinterpret_cast<const f32x4 *>(gnp + Ctot);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float sg = sigmoid_f(y[e]); d[e] *= sg * (1.0f + y[e] * (1.0f - sg)); }
  }
  const float *cf = a.coef + (size_t)b * 3 * Ctot + c;
  const f32x4 dx = *reinterpret_cast<const f32x4 *>(cf) * d - *reinterpret_cast<const f32x4 *>(cf + Ctot) -
                   xh * *reinterpret_cast<const f32x4 *>(cf + 2 * Ctot);
  f32x4 *o = reinterpret_cast<f32x4 *>(gp + (size_t)row * Cx + cc);
  *o = accf ? *o + dx : dx;
}

hipError_t launch_gn_backward(const GnbArgs &a, hipStream_t st) {
  const int Ctot = a.C0 + a.C1;
  hipLaunchKernelGGL(gnb_reduce_kernel, dim3(a.nsl, a.B), dim3(256), 0, st, a);
  hipLaunchKernelGGL(gnb_finalize_kernel, dim3(a.B), dim3(256), (size_t)(2 * Ctot + 2 * a.groups) * sizeof(float), st, a);
  const long long total = (long long)a.B * a.V * (Ctot / 4);
  if ((a.C0 & 3) || (a.C1 & 3) || (a.dA_cs & 3) || total >= (1ll << 31)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gnb_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// Elementwise helpers of the backward graph
// --------------------------------------------------------------------------------
// dst[b][v][c] (+)= src[b][v][c]   (channels-last, possibly different channel strides)
__global__ void add_into_kernel(float *__restrict__ dst, int dcs, const float *__restrict__ src, int scs, int C,
                                long long rows, int accumulate, int vec) {
  // vec: one channel quad of one row per thread (float4, 32-bit index arithmetic; the host checked strides, size and
  // pointer alignment); scalar form otherwise
  if (vec) {
    const unsigned Q = (unsigned)(C >> 2);
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (unsigned)rows * Q) return;
    const unsigned row = i / Q, c = (i - row * Q) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (size_t)row * scs + c);
    f32x4 *o = reinterpret_cast<f32x4 *>(dst + (size_t)row * dcs + c);
    *o = accumulate ? *o + v : v;
    return;
  }
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C) return;
  const int c = (int)(i % C);
  const long long row = i / C;
  const float v = src[row * scs + c];
  float *o = dst + row * dcs + c;
  *o = accumulate ? *o + v : v;
}

hipError_t launch_add_into(float *dst, int dcs, const float *src, int scs, int C, long long rows, int accumulate,
                           hipStream_t st) {
  const bool vec = !((C | dcs | scs) & 3) && rows * (C >> 2) < (1ll << 31) && !(reinterpret_cast<uintptr_t>(dst) & 15) &&
                   !(reinterpret_cast<uintptr_t>(src) & 15);
  const long long n = vec ? rows * (C >> 2) : rows * C;
  hipLaunchKernelGGL(add_into_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, dcs, src, scs, C, rows, accumulate, vec ? 1 : 0);
  return hipGetLastError();
}

// nearest x2 upsample of a channels-last tensor (materialised only for the weight gradient of
// the upsample conv) and its adjoint (2x2x2 sum pooling).
__global__ void upsample2_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int Z, int Y, int X, int C) {
  const long long total = (long long)B * 8 * Z * Y * X * C;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  long long q = i / C;
  const int ux = (int)(q % (2 * X)); q /= 2 * X;
  const int uy = (int)(q % (2 * Y)); q /= 2 * Y;
  const int uz = (int)(q % (2 * Z));
  const int b = (int)(q / (2 * Z));
  y[i] = x[((((size_t)b * Z + (uz >> 1)) * Y + (uy >> 1)) * X + (ux >> 1)) * C + c];
}

__global__ void sumpool2_kernel(const float *__restrict__ y, float *__restrict__ x, int B, int Z, int Y, int X, int C,
                                int accumulate) {
  const long long total = (long long)B * Z * Y * X * C;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  long long q = i / C;
  const int xx = (int)(q % X); q /= X;
  const int yy = (int)(q % Y); q /= Y;
  const int zz = (int)(q % Z);
  const int b = (int)(q / Z);
  float s = 0.f;
  for (int dz = 0; dz < 2; ++dz)
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx)
        s += y[((((size_t)b * 2 * Z + 2 * zz + dz) * 2 * Y + 2 * yy + dy) * 2 * X + 2 * xx + dx) * C + c];
  x[i] = accumulate ? x[i] + s : s;
}

hipError_t launch_upsample2(const float *x, float *y, int B, int Z, int Y, int X, int C, hipStream_t st) {
  const long long total = (long long)B * 8 * Z * Y * X * C;
  hipLaunchKernelGGL(upsample2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, y, B, Z, Y, X, C);
  return hipGetLastError();
}

hipError_t launch_sumpool2(const float *y, float *x, int B, int Z, int Y, int X, int C, int accumulate, hipStream_t st) {
  const long long total = (long long)B * Z * Y * X * C;
  hipLaunchKernelGGL(sumpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, y, x, B, Z, Y, X, C,
                     accumulate);
  return hipGetLastError();
}

// Adjoint of "take every second voxel" used by the stride-2 conv data gradient: y (Z,Y,X) is
// written into the even positions of a zero tensor of size (2Z,2Y,2X).
__global__ void zero_stuff2_kernel(const float *__restrict__ y, float *__restrict__ u, int B, int Z, int Y, int X, int C) {
  const long long total = (long long)B * 8 * Z * Y * X * C;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  long long q = i / C;
  const int ux = (int)(q % (2 * X)); q /= 2 * X;
  const int uy = (int)(q % (2 * Y)); q /= 2 * Y;
  const int uz = (int)(q % (2 * Z));
  const int b = (int)(q / (2 * Z));
  float v = 0.f;
  if (!((ux | uy | uz) & 1)) v = y[((((size_t)b * Z + (uz >> 1)) * Y + (uy >> 1)) * X + (ux >> 1)) * C + c];
  u[i] = v;
}

hipError_t launch_zero_stuff2(const float *y, float *u, int B, int Z, int Y, int X, int C, hipStream_t st) {
  const long long total = (long long)B * 8 * Z * Y * X * C;
  hipLaunchKernelGGL(zero_stuff2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, y, u, B, Z, Y, X, C);
  return hipGetLastError();
}

// d(loss)/d(eps_hat) of mean((eps_hat - eps)^2) written into the channels-last gradient of the
// final conv output [B][L][H][W][8]: zero for the past frames (unet.py:166 slices them away)
// and for the padding channels.
__global__ void mse_grad_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                float *__restrict__ g, int B, int C, int H, int W, int P, int F) {
  const int L = P + F;
  const long long total = (long long)B * L * H * W * 8;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i & 7);
  long long q = i >> 3;
  const int w = (int)(q % W); q /= W;
  const int hh = (int)(q % H); q /= H;
  const int l = (int)(q % L);
  const int b = (int)(q / L);
  float v = 0.f;
  if (c < C && l >= P) {
    const size_t ref = ((((size_t)b * C + c) * H + hh) * W + w) * F + (l - P);
    v = 2.0f * (pred[ref] - target[ref]) / (float)((long long)B * C * H * W * F);
  }
  g[i] = v;
}

hipError_t launch_mse_grad(const float *pred, const float *target, float *g, int B, int C, int H, int W, int P, int F,
                           hipStream_t st) {
  const long long total = (long long)B * (P + F) * H * W * 8;
  hipLaunchKernelGGL(mse_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pred, target, g, B, C,
                     H, W, P, F);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// Attention core backward per (sample, head): P = softmax(q k^T * s);  O = P v
//   dV = P^T dO ; dP = dO V^T ; dS = P o (dP - rowsum(dP o P)) ; dQ = s * dS K ; dK = s * dS^T Q
// qkv / dqkv channels-last [B][S][3E]; dO [B][S][E].  S <= 256 (rows handled by threads).
// --------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float *__restrict__ qkv, const float *__restrict__ dO,
                                                       float *__restrict__ dqkv, int S, int E, float *__restrict__ big) {
  // Q K V dO [S][D + 1] (odd row stride: a lane per key row reads conflict-free), P [S][S], dS [S][S]
  // `big` != null (S x S matrices too large for LDS, e.g. 216 tokens on the doubled grid): P and dS of this
  // (sample, head) live in a global scratch slab instead; same code, slower, exact same arithmetic
  constexpr int DP = D + 1;
  extern __shared__ float sm[];
  const int hd = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  float *Q = sm, *K = Q + S * DP, *Vv = K + S * DP, *dOs = Vv + S * DP;
  float *Pm = big ? big + ((size_t)b * gridDim.x + hd) * 2 * S * S : dOs + S * DP, *dSm = Pm + S * S;
  const float *base = qkv + (size_t)b * S * 3 * E;
  for (int i = tid; i < S * D; i += 256) {
    const int s = i / D, d = i % D;
    Q[s * DP + d] = base[(size_t)s * 3 * E + hd * D + d];
    K[s * DP + d] = base[(size_t)s * 3 * E + E + hd * D + d];
    Vv[s * DP + d] = base[(size_t)s * 3 * E + 2 * E + hd * D + d];
    dOs[s * DP + d] = dO[((size_t)b * S + s) * E + hd * D + d];
  }
  __syncthreads();
  const float scale = rsqrtf((float)D);
  // scores and dP = dO V^T, one (row, key) pair per thread and round
  for (int idx = tid; idx < S * S; idx += 256) {
    const int row = idx / S, j = idx - row * S;
    float sc = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      sc = fmaf(Q[row * DP + d], K[j * DP + d], sc);
      dp = fmaf(dOs[row * DP + d], Vv[j * DP + d], dp);
    }
    Pm[idx] = sc * scale;
    dSm[idx] = dp;
  }
  if (big) __threadfence_block();               // (global scratch written by other threads of this workgroup)
  __syncthreads();
  // row softmax and dS = P o (dP - rowsum(dP o P)): four adjacent lanes per row, merged by shuffles in a fixed order
  for (int r0 = 0; r0 < S; r0 += 64) {
    const int row = r0 + (tid >> 2), l4 = tid & 3;
    const bool ok = row < S;
    float mx = -3.0e38f;
    if (ok) for (int j = l4; j < S; j += 4) mx = fmaxf(mx, Pm[row * S + j]);
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    float l = 0.f;
    if (ok) for (int j = l4; j < S; j += 4) { const float p = __expf(Pm[row * S + j] - mx); Pm[row * S + j] = p; l += p; }
    l += __shfl_xor(l, 1);
    l += __shfl_xor(l, 2);
    const float inv = 1.0f / l;
    float dot = 0.f;
    if (ok) for (int j = l4; j < S; j += 4) { const float p = Pm[row * S + j] * inv; Pm[row * S + j] = p; dot = fmaf(dSm[row * S + j], p, dot); }
    dot += __shfl_xor(dot, 1);
    dot += __shfl_xor(dot, 2);
    if (ok) for (int j = l4; j < S; j += 4) dSm[row * S + j] = Pm[row * S + j] * (dSm[row * S + j] - dot);
  }
  if (big) __threadfence_block();
  __syncthreads();
  float *ob = dqkv + (size_t)b * S * 3 * E;
  for (int i = tid; i < S * D; i += 256) {
    const int s = i / D, d = i % D;
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int j = 0; j < S; ++j) {
      dq = fmaf(dSm[s * S + j], K[j * DP + d], dq);
      dk = fmaf(dSm[j * S + s], Q[j * DP + d], dk);
      dv = fmaf(Pm[j * S + s], dOs[j * DP + d], dv);
    }
    ob[(size_t)s * 3 * E + hd * D + d] = dq * scale;
    ob[(size_t)s * 3 * E + E + hd * D + d] = dk * scale;
    ob[(size_t)s * 3 * E + 2 * E + hd * D + d] = dv;
  }
}

size_t attn_bwd_scratch_floats(int B, int S, int E, int heads) {
  const int D = E / heads;
  const size_t smem = ((size_t)4 * S * (D + 1) + (size_t)2 * S * S) * sizeof(float);
  return smem > 160 * 1024 ? (size_t)B * heads * 2 * S * S : 0;
}

hipError_t launch_attn_bwd(const float *qkv, const float *dO, float *dqkv, int B, int S, int E, int heads, float *big, hipStream_t st) {
  const int D = E / heads;
  size_t smem = ((size_t)4 * S * (D + 1) + (size_t)2 * S * S) * sizeof(float);
  if (smem > 160 * 1024) {
    if (!big) return hipErrorInvalidValue;
    smem = (size_t)4 * S * (D + 1) * sizeof(float);
    if (smem > 160 * 1024) return hipErrorInvalidValue;
  } else {
    big = nullptr;
  }
#define CM_AB(DD)                                                                                         \
  if (D == DD) {                                                                                          \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_kernel<DD>),               \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);           \
    if (e != hipSuccess) return e;                                                                        \
    hipLaunchKernelGGL((attn_bwd_kernel<DD>), dim3(heads, B), dim3(256), smem, st, qkv, dO, dqkv, S, E, big);    \
    return hipGetLastError();                                                                             \
  }
  CM_AB(8) CM_AB(16) CM_AB(32) CM_AB(64)
#undef CM_AB
  return hipErrorInvalidValue;
}

// --------------------------------------------------------------------------------
// Time-embedding path, forward for the B timesteps of a training batch and backward.
//   e = table[t] ; z1 = W1 e + b1 ; h1 = silu(z1) ; te = W2 h1 + b2 ; s = silu(te) ; proj = Wd s + bd
// One workgroup; B <= 256.  The forward writes proj rows [B][nproj] (the conv epilogues then
// index row b); the backward consumes dproj [B][nproj] and produces all weight gradients.
// --------------------------------------------------------------------------------

// forward recompute, one workgroup per sample
__global__ __launch_bounds__(256) void time_fwd_rows_kernel(const TimeBwdArgs a) {
  extern __shared__ float sm[];  // e[te], h1[tx]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int B = a.B, te = a.te, tx = a.tx;
  float *e = a.ws, *h1 = e + (size_t)B * te, *z1 = h1 + (size_t)B * tx, *tev = z1 + (size_t)B * tx, *sv = tev + (size_t)B * tx;
  float *es = sm, *hs = sm + te;
  for (int k = tid; k < te; k += 256) { const float v = a.table[(size_t)a.t[b] * te + k]; es[k] = v; e[(size_t)b * te + k] = v; }
  __syncthreads();
  for (int o = tid; o < tx; o += 256) {
    float acc = a.b1[o];
    for (int k = 0; k < te; ++k) acc = fmaf(a.W1[(size_t)o * te + k], es[k], acc);
    z1[(size_t)b * tx + o] = acc;
    const float h = acc * sigmoid_f(acc);
    hs[o] = h; h1[(size_t)b * tx + o] = h;
  }
  __syncthreads();
  for (int o = tid; o < tx; o += 256) {
    float acc = a.b2[o];
    for (int k = 0; k < tx; ++k) acc = fmaf(a.W2[(size_t)o * tx + k], hs[k], acc);
    tev[(size_t)b * tx + o] = acc;
    sv[(size_t)b * tx + o] = acc * sigmoid_f(acc);
  }
}

// C[m][n] = sum_k A[m*sam + k*sak] * Bm[k*sbk + n]   (* silu'(pre[m][n]) when pre != null); one thread per
// output, n fastest: B reads coalesced, A reads broadcast within a row of threads.
__global__ __launch_bounds__(256) void small_gemm_kernel(const float *__restrict__ A, long long sam, long long sak,
                                                         const float *__restrict__ Bm, long long sbk,
                                                         float *__restrict__ Cm, int M, int N, int K,
                                                         const float *__restrict__ pre) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)M * N) return;
  const int n = (int)(i % N);
  const long long m = i / N;
  const float *ap = A + m * sam;
  const float *bp = Bm + n;
  // eight loads of each operand in flight per round (the operands are L2-resident: one ~1 us round trip per round
  // instead of one per k pair), eight running sums merged in a fixed order
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 7 < K; k += 8) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { av[u] = ap[(long long)(k + u) * sak]; bv[u] = bp[(long long)(k + u) * sbk]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = fmaf(av[u], bv[u], acc[u]);
  }
  for (; k < K; ++k) acc[0] = fmaf(ap[(long long)k * sak], bp[(long long)k * sbk], acc[0]);
  float v = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  if (pre) { const float y = pre[i], sg = sigmoid_f(y); v *= sg * (1.0f + y * (1.0f - sg)); }
  Cm[i] = v;
}

static void small_gemm(const float *A, long long sam, long long sak, const float *Bm, long long sbk, float *Cm, int M,
                       int N, int K, const float *pre, hipStream_t st) {
  const long long total = (long long)M * N;
  hipLaunchKernelGGL(small_gemm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, A, sam, sak, Bm, sbk, Cm,
                     M, N, K, pre);
}

hipError_t launch_time_bwd(const TimeBwdArgs &a, hipStream_t st) {
  const int B = a.B, te = a.te, tx = a.tx, np = a.nproj;
  float *e = a.ws, *h1 = e + (size_t)B * te, *z1 = h1 + (size_t)B * tx, *tev = z1 + (size_t)B * tx,
        *sv = tev + (size_t)B * tx, *dte = sv + (size_t)B * tx, *dz1 = dte + (size_t)B * tx;
  hipLaunchKernelGGL(time_fwd_rows_kernel, dim3(B), dim3(256), (size_t)(te + tx) * sizeof(float), st, a);
  // dWd[o][k] = sum_b dproj[b][o] * s[b][k] ; dbd[o] = sum_b dproj[b][o]
  small_gemm(a.dproj, 1, np, sv, tx, a.dWd, np, tx, B, nullptr, st);
  hipError_t err = launch_batch_sum(a.dproj, B, np, np, a.dbd, 0, st);
  if (err != hipSuccess) return err;
  // dte[b][k] = (sum_o dproj[b][o] Wd[o][k]) * silu'(te[b][k])
  small_gemm(a.dproj, np, 1, a.Wd, tx, dte, B, tx, np, tev, st);
  // dW2[o][k] = sum_b dte[b][o] h1[b][k] ; db2
  small_gemm(dte, 1, tx, h1, tx, a.dW2, tx, tx, B, nullptr, st);
  err = launch_batch_sum(dte, B, tx, tx, a.db2, 0, st);
  if (err != hipSuccess) return err;
  // dz1[b][k] = (sum_o dte[b][o] W2[o][k]) * silu'(z1[b][k])
  small_gemm(dte, tx, 1, a.W2, tx, dz1, B, tx, tx, z1, st);
  // dW1[o][k] = sum_b dz1[b][o] e[b][k] ; db1
  small_gemm(dz1, 1, tx, e, te, a.dW1, tx, te, B, nullptr, st);
  err = launch_batch_sum(dz1, B, tx, tx, a.db1, 0, st);
  if (err != hipSuccess) return err;
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// Adam with coupled L2 (torch.optim.Adam(weight_decay=wd), ddpm.py:53-56), elementwise.
// --------------------------------------------------------------------------------
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gg = g[i] + wd * p[i];
  const float mm = b1 * m[i] + (1.0f - b1) * gg;
  const float vv = b2 * v[i] + (1.0f - b2) * gg * gg;
  m[i] = mm;
  v[i] = vv;
  p[i] -= (lr / bc1) * mm / (sqrtf(vv) / sqrtf(bc2) + eps);
}

hipError_t launch_adam(float *p, const float *g, float *m, float *v, long long n, float lr, float b1, float b2,
                       float eps, float wd, int step, hipStream_t st) {
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps,
                     wd, bc1, bc2);
  return hipGetLastError();
}

// packed[i] = sum_{k<8} W[idx[i][k]]  (idx < 0: skipped) -- re-packs a master weight tensor into
// an MFMA fragment layout (forward, data-gradient or parity layout) after an optimizer step.
__global__ void gather_pack_kernel(const float *__restrict__ W, const int *__restrict__ idx, int nk,
                                   float *__restrict__ packed, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nk; ++k) {
    const int j = idx[i * nk + k];
    if (j >= 0) s += W[j];
  }
  packed[i] = s;
}

// All re-packs of one optimizer step in ONE launch: job j covers the global element range
// [start[j], start[j+1]); a thread finds its job by bisection over the (few hundred) starts.
__global__ void gather_pack_jobs_kernel(const float *__restrict__ W, const PackJob *__restrict__ jobs, const int *__restrict__ blk2job) {
  const PackJob jb = jobs[blk2job[blockIdx.x]];
  const long long e = ((long long)blockIdx.x - jb.blk0) * 256 + threadIdx.x;
  if (e >= jb.n) return;
  float s = 0.f;
  for (int k = 0; k < jb.nk; ++k) {
    const int j = jb.idx[e * jb.nk + k];
    if (j >= 0) s += jb.coef ? jb.coef[e * jb.nk + k] * W[j] : W[j];   // Winograd weights: G g G^T is a weighted sum of 9 taps
  }
  jb.dst[e] = s;
}

hipError_t launch_gather_pack_jobs(const float *W, const PackJob *jobs, const int *blk2job, long long nblocks, hipStream_t st) {
  if (nblocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_pack_jobs_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, W, jobs, blk2job);
  return hipGetLastError();
}

__global__ void wino_pack_jobs_kernel(const float *__restrict__ W, const WinoPackJob *__restrict__ jobs, const int *__restrict__ blk2job) {
  // one thread per (output channel, input channel, z tap) of the PADDED ranges: 9 taps in, the 16 components out
  // (ci fastest across lanes: the 27-index rows of neighbouring lanes are adjacent, the stores of a component cover
  // 16-byte runs of the packed layout)
  const WinoPackJob jb = jobs[blk2job[blockIdx.x]];
  const long long t = ((long long)blockIdx.x - jb.blk0) * 256 + threadIdx.x;
  if (t >= jb.n) return;
  const int nch = jb.Ci_pad / 16;
  const int ci = (int)(t % jb.Ci_pad);
  const int dz = (int)((t / jb.Ci_pad) % 3);
  const int co = (int)(t / ((long long)jb.Ci_pad * 3));
  float g[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = 0.f;
  if (co < jb.Co && ci < jb.Ci) {
    const int *ip = jb.idx27 + ((size_t)co * jb.Ci + ci) * 27 + dz * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) { const int j = ip[k]; g[k] = j >= 0 ? W[j] : 0.f; }
  }
  const int nt = co >> 5, r = co & 31, chunk = ci >> 4, k8 = (ci >> 3) & 1, hh = (ci >> 2) & 1, jj = ci & 3;
  float *base = jb.dst + ((((size_t)nt * nch + chunk) * 4) * 24 * 64 + hh * 32 + r) * 4 + jj;
  // G = (1,0,0), (1/2,1/2,1/2), (1/2,-1/2,1/2), (0,0,1); terms in (dy, dx) order with the zero coefficients skipped
  // (the summation order of the generic gather this replaces)
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
#pragma unroll
  for (int xy = 0; xy < 4; ++xy)
#pragma unroll
    for (int xx = 0; xx < 4; ++xx) {
      float s = 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float c = G[xy][dy] * G[xx][dx];
          if (c != 0.f) s += c * g[dy * 3 + dx];
        }
      base[((size_t)xy * 24 + (dz * 2 + k8) * 4 + xx) * 64 * 4] = s;
    }
}

hipError_t launch_wino_pack_jobs(const float *W, const WinoPackJob *jobs, const int *blk2job, long long nblocks, hipStream_t st) {
  if (nblocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(wino_pack_jobs_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, W, jobs, blk2job);
  return hipGetLastError();
}

hipError_t launch_gather_pack(const float *W, const int *idx, int nk, float *packed, long long n, hipStream_t st) {
  hipLaunchKernelGGL(gather_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, idx, nk, packed, n);
  return hipGetLastError();
}

}  // namespace cm
