// 3x3x3 convolution with very few output channels (the UNet's last layer, unet.py:118-122:
// GroupNorm -> SiLU -> Conv3d(base -> C), C = 3 or 4).  On the matrix cores a 32-wide N tile would
// carry 28 padding columns, i.e. 8x the useful work; here every thread owns one output voxel and its
// NCO accumulators on the vector ALUs instead:
//   - the halo box is staged into LDS exactly like the implicit-GEMM kernel does (GroupNorm affine
//     and SiLU applied on the way in), row stride CK + 4 floats;
//   - the weights of the chunk sit in LDS as [tap][ci][NCO] and are read as wave-uniform
//     (broadcast) vectors;
//   - the 27 taps are split over the 256 / TM thread groups of the tile and the partial sums are
//     merged through LDS in a fixed order.
// Bound: VALU (4 * NCO FMAs per 16-byte LDS read), ~3 us of math per 128-voxel tile against
// ~17 us of MFMA time for the padded tile.
#include "cm_kernels.h"

namespace cm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_s(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// BZ x BY x BX != 0: compile-time tile box with 32-channel chunks (the tuned shape of the reference grids):
// halo extents, strides and the coordinate tables fold into immediates.
template <int NCO, int BZ = 0, int BY = 0, int BX = 0>
__global__ __launch_bounds__(256) void conv_smalln_kernel(const ConvArgs a, const float *__restrict__ wsm, int TM_rt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool SPEC = BZ != 0;
  const int TM = SPEC ? 32 * ((BZ * BY * BX + 31) / 32) : TM_rt;
  const int a_bz = SPEC ? BZ : a.bz, a_by = SPEC ? BY : a.by, a_bx = SPEC ? BX : a.bx, a_CK = SPEC ? 32 : a.CK;
  const int tid = threadIdx.x;
  int tile = blockIdx.x;
  if (!(gridDim.x & 7) && !(a.dbg & 4096)) tile = (tile & 7) * (int)(gridDim.x >> 3) + (tile >> 3);  // XCD-aware order (see cm_conv.hip)
  const int tx = tile % a.ntx; tile /= a.ntx;
  const int ty = tile % a.nty; tile /= a.nty;
  const int tz = tile % a.ntz;
  const int b = tile / a.ntz;
  const int z0 = tz * a_bz, y0 = ty * a_by, x0 = tx * a_bx;
  const int HZ = a_bz + 2, HY = a_by + 2, HX = a_bx + 2;
  const int HV = HZ * HY * HX;
  const int S = a_CK + 4;
  auto mtab_at = [&](int m) -> int {
    if constexpr (SPEC) {
      if (m >= BZ * BY * BX) return -1;
      const int z = m / (BY * BX), rem = m - z * (BY * BX), y = rem / BX, x = rem - y * BX;
      return (z << 18) | (y << 9) | x;
    } else {
      return a.mtab[m];
    }
  };
  auto hvtab_at = [&](int hv) -> int {
    if constexpr (SPEC) {
      constexpr int cHY = BY + 2, cHX = BX + 2;
      const int hz = hv / (cHY * cHX), rem = hv - hz * (cHY * cHX), hy = rem / cHX, hx = rem - hy * cHX;
      return (hz << 18) | (hy << 9) | hx;
    } else {
      return a.hvtab[hv];
    }
  };
  const int Ctot = a.C0 + a.C1;
  const int nchunks = a.nch0 + a.nch1;
  const int NG = 256 / TM;                     // tap groups (threads per output voxel)

  float *A = lds;                              // [HV][S]
  float *W = A + (size_t)HV * S;               // [27][CK][NCO]
  float *red = W + 27 * a_CK * NCO;            // [NG - 1][TM][NCO]

  // this thread's output voxel
  const int m = tid % TM, grp = tid / TM;
  const int pk = mtab_at(m);
  int off = -1, hbase = 0;
  if (pk >= 0) {
    const int x = pk & 511, y = (pk >> 9) & 511, z = (pk >> 18) & 255;
    const int oz = z0 + z, oy = y0 + y, ox = x0 + x;
    if (oz < a.Zo && oy < a.Yo && ox < a.Xo) off = ((b * a.Zo + oz) * a.Yo + oy) * a.Xo + ox;
    hbase = ((z * HY + y) * HX + x) * S;
  }
  // Specialised variant (TM = 64: one tap group per wave): the weights are wave-uniform, so they are read
  // straight from global memory with scalar loads (s_load -> SGPR operands of the FMAs) instead of as
  // broadcast LDS vectors -- the LDS pipe then only carries the A reads (it was the limiter: 4 weight reads
  // per 16 FMAs).
  constexpr bool WS = SPEC && ((BZ * BY * BX + 31) / 32) * 32 % 64 == 0;
  const int grp_u = WS ? __builtin_amdgcn_readfirstlane(grp) : grp;
  const int t0 = grp_u * 27 / NG, t1 = (grp_u + 1) * 27 / NG;

  float acc[NCO];
#pragma unroll
  for (int i = 0; i < NCO; ++i) acc[i] = 0.f;

  const int K4 = a_CK >> 2;
  for (int ch = 0; ch < nchunks; ++ch) {
    const float *src;
    int Cs, c0, cg0;
    if (ch < a.nch0) { src = a.src0; Cs = a.C0; c0 = ch * a_CK; cg0 = c0; }
    else { src = a.src1; Cs = a.C1; c0 = (ch - a.nch0) * a_CK; cg0 = a.C0 + c0; }
    __syncthreads();
    if constexpr (!WS)
      for (int i = tid; i < 27 * a_CK * NCO; i += 256) W[i] = wsm[(size_t)ch * 27 * a_CK * NCO + i];
    if constexpr (SPEC) {
      // all of this thread's halo loads first, then the activations: one memory round trip per chunk instead of one per
      // item (the loop below issues load -> normalise -> LDS write per item: 7 dependent round trips, ~10 us per workgroup)
      constexpr int cHV = (BZ + 2) * (BY + 2) * (BX + 2), NI = (cHV * 8 + 255) / 256;
      const int q = tid & 7;                       // K4 = 8: the channel quad is the same for every item of a thread
      f32x4 v[NI];
      unsigned okm = 0;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int hv = (tid >> 3) + 32 * k;
        const int hp = hvtab_at(hv < cHV ? hv : cHV - 1);
        const int cx = x0 - 1 + (hp & 511), cy = y0 - 1 + ((hp >> 9) & 511), cz = z0 - 1 + ((hp >> 18) & 255);
        const bool ok = hv < cHV && cz >= 0 && cz < a.Zs && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs;
        const size_t so = ok ? ((size_t)(b * a.Zs + cz) * a.Ys + cy) * a.Xs + cx : 0;
        if (a.h16 & (ch < a.nch0 ? 1 : 2)) {        // f16 source tensor
          const cm_f32x2_t two = *reinterpret_cast<const cm_f32x2_t *>(reinterpret_cast<const _Float16 *>(src) + so * Cs + c0 + 4 * q);
          const cm_f16x4_t hv4 = __builtin_bit_cast(cm_f16x4_t, two);
          v[k] = f32x4{(float)hv4[0], (float)hv4[1], (float)hv4[2], (float)hv4[3]};
        } else {
          v[k] = *reinterpret_cast<const f32x4 *>(src + so * Cs + c0 + 4 * q);
        }
        okm |= (ok ? 1u : 0u) << k;
      }
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, pm = {1.f, 1.f, 1.f, 1.f};
      if (a.gn) {
        const float *g = a.gn + (size_t)b * 2 * Ctot + cg0 + 4 * q;
        sc = *reinterpret_cast<const f32x4 *>(g);
        sh = *reinterpret_cast<const f32x4 *>(g + Ctot);
      }
      if (a.pm) pm = *reinterpret_cast<const f32x4 *>(a.pm + (size_t)b * a.pm_stride + cg0 + 4 * q);
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int hv = (tid >> 3) + 32 * k;
        f32x4 w = v[k];
        if (a.gn) {
          w = w * sc + sh;
          if (a.silu) { w[0] = silu_s(w[0]); w[1] = silu_s(w[1]); w[2] = silu_s(w[2]); w[3] = silu_s(w[3]); }
        }
        if (a.pm) w = w * pm;
        if (!((okm >> k) & 1u)) w = f32x4{0.f, 0.f, 0.f, 0.f};
        if (hv < cHV) *reinterpret_cast<f32x4 *>(&A[hv * S + 4 * q]) = w;
      }
    } else
    for (int i = tid; i < HV * K4; i += 256) {
      const int hv = i / K4, q = i - hv * K4;
      const int hp = hvtab_at(hv);
      const int cx = x0 - 1 + (hp & 511), cy = y0 - 1 + ((hp >> 9) & 511), cz = z0 - 1 + ((hp >> 18) & 255);
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
      if (cz >= 0 && cz < a.Zs && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs) {
        const size_t so = ((size_t)(b * a.Zs + cz) * a.Ys + cy) * a.Xs + cx;
        if (a.h16 & (ch < a.nch0 ? 1 : 2)) {
          const cm_f32x2_t two = *reinterpret_cast<const cm_f32x2_t *>(reinterpret_cast<const _Float16 *>(src) + so * Cs + c0 + 4 * q);
          const cm_f16x4_t hv4 = __builtin_bit_cast(cm_f16x4_t, two);
          w = f32x4{(float)hv4[0], (float)hv4[1], (float)hv4[2], (float)hv4[3]};
        } else {
          w = *reinterpret_cast<const f32x4 *>(src + so * Cs + c0 + 4 * q);
        }
        if (a.gn) {
          const float *g = a.gn + (size_t)b * 2 * Ctot + cg0 + 4 * q;
          w = w * *reinterpret_cast<const f32x4 *>(g) + *reinterpret_cast<const f32x4 *>(g + Ctot);
          if (a.silu) { w[0] = silu_s(w[0]); w[1] = silu_s(w[1]); w[2] = silu_s(w[2]); w[3] = silu_s(w[3]); }
        }
        if (a.pm) w = w * *reinterpret_cast<const f32x4 *>(a.pm + (size_t)b * a.pm_stride + cg0 + 4 * q);
      }
      *reinterpret_cast<f32x4 *>(&A[hv * S + 4 * q]) = w;
    }
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
      const int dz = t / 9, rem = t - dz * 9, dy = rem / 3, dx = rem - dy * 3;
      const float *ap = A + hbase + ((dz * HY + dy) * HX + dx) * S;
      const float *wp = WS ? wsm + ((size_t)ch * 27 + t) * a_CK * NCO : W + (size_t)t * a_CK * NCO;
      for (int q = 0; q < K4; ++q) {
        const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float *wr = wp + (4 * q + j) * NCO;
#pragma unroll
          for (int n4 = 0; n4 < NCO; n4 += 4) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(wr + n4);
            acc[n4 + 0] = fmaf(av[j], wv[0], acc[n4 + 0]);
            acc[n4 + 1] = fmaf(av[j], wv[1], acc[n4 + 1]);
            acc[n4 + 2] = fmaf(av[j], wv[2], acc[n4 + 2]);
            acc[n4 + 3] = fmaf(av[j], wv[3], acc[n4 + 3]);
          }
        }
      }
    }
  }
  // merge the tap groups in order 0, 1, ... and write
  if (grp > 0) {
#pragma unroll
    for (int i = 0; i < NCO; ++i) red[((grp - 1) * TM + m) * NCO + i] = acc[i];
  }
  __syncthreads();
  if (grp == 0 && off >= 0) {
    for (int g = 1; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < NCO; ++i) acc[i] += red[((g - 1) * TM + m) * NCO + i];
    float *o = a.out + (size_t)off * a.out_cs;
#pragma unroll
    for (int i = 0; i < NCO; ++i)
      if (i < a.Co) o[i] = acc[i] + a.bias[i];
  }
}

bool conv_smalln_ok(const ConvArgs &a, int MB) {
  const int TM = 32 * MB;
  return a.ntaps == 27 && a.td == 3 && a.stride == 1 && !a.par && !a.ups && a.bs == 1 && a.Co <= 8 && !a.temb &&
         !a.resid && !a.stat_part && a.ks <= 1 && a.out_cs >= a.Co && (TM == 32 || TM == 64 || TM == 128 || TM == 256);
}

size_t conv_smalln_lds(const ConvArgs &a, int MB) {
  const int TM = 32 * MB, nco = a.Co <= 4 ? 4 : 8;
  const size_t HV = (size_t)(a.bz + 2) * (a.by + 2) * (a.bx + 2);
  return (HV * (a.CK + 4) + (size_t)27 * a.CK * nco + (size_t)(256 / TM) * TM * nco) * sizeof(float);
}

hipError_t launch_conv_smalln(const ConvArgs &a_in, int MB, const float *wsm, hipStream_t st) {
  ConvArgs a = a_in;
  a.dbg = conv_dbg_flags();
  if (!conv_smalln_ok(a, MB)) return hipErrorInvalidValue;
  const int TM = 32 * MB;
  const size_t lds = conv_smalln_lds(a, MB);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const dim3 grid((unsigned)(a.B * a.ntz * a.nty * a.ntx));
  static bool attr_set[64][2] = {{false}};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int v = a.Co <= 4 ? 0 : 1;
  if (!attr_set[dev & 63][v]) {
    hipError_t e = v == 0 ? hipFuncSetAttribute(reinterpret_cast<const void *>(conv_smalln_kernel<4>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                          : hipFuncSetAttribute(reinterpret_cast<const void *>(conv_smalln_kernel<8>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev & 63][v] = true;
  }
  if (v == 0 && a.CK == 32 && a.bz == 4 && a.by == 4 && a.bx == 4) {
    static bool spec_set[64] = {false};
    if (!spec_set[dev & 63]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_smalln_kernel<4, 4, 4, 4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      spec_set[dev & 63] = true;
    }
    hipLaunchKernelGGL((conv_smalln_kernel<4, 4, 4, 4>), grid, dim3(256), lds, st, a, wsm, TM);
    return hipGetLastError();
  }
  if (v == 0) hipLaunchKernelGGL(conv_smalln_kernel<4>, grid, dim3(256), lds, st, a, wsm, TM);
  else hipLaunchKernelGGL(conv_smalln_kernel<8>, grid, dim3(256), lds, st, a, wsm, TM);
  return hipGetLastError();
}

}  // namespace cm
