// Implicit-GEMM 3-D convolution for gfx950 (MI355X), exact fp32 on the matrix cores.
//
// Replaces every nn.Conv3d of the reference UNet
// (/root/reference/models/backbones/layers.py:32,43,46,84,94; unet.py:32,121)
// plus, through the 1x1x1 mode, the packed in/out projections of
// nn.MultiheadAttention (layers.py:10).
//
// Structure (one 256-thread workgroup = 4 waves, one output tile of 32*MB voxels
// x 32*NB channels):
//   * the input halo box of the tile is staged ONCE per channel chunk into LDS,
//     channels-last with row stride CK+4 dwords (conflict-free ds_read_b128);
//     GroupNorm affine + SiLU of the producer are applied while staging, zero
//     padding / nearest-upsample / stride-2 / channel-concat are pure index math;
//     nothing is im2col-materialised;
//   * the K dimension (27 taps x CK channels) is split over the 4 waves: every
//     wave owns the whole output tile and a quarter of the (tap, 8-channel) steps;
//     its weight fragments stream global->VGPR in pre-packed fragment order (1 KiB
//     coalesced per wave-load, no LDS, no redundancy between waves);
//   * v_mfma_f32_32x32x2_f32: bit-exact fp32 FMA chains at the fp32 peak rate;
//   * partial accumulators are reduced across waves through LDS, then the owner
//     wave applies bias + time-embedding + residual and stores channels-last.
#include "cm_kernels.h"

namespace cm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

template <int MB, int NB>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TM = 32 * MB;
  constexpr int TN = 32 * NB;
  constexpr int NBLK = MB * NB;
  constexpr int RB = NBLK < 4 ? NBLK : 4;  // accumulator blocks reduced per round

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  int tile = blockIdx.x;
  const int tx = tile % a.ntx; tile /= a.ntx;
  const int ty = tile % a.nty; tile /= a.nty;
  const int tz = tile % a.ntz;
  const int ts = tile / a.ntz;
  const int nt = blockIdx.y;
  const int b0 = ts * a.bs, z0 = tz * a.bz, y0 = ty * a.by, x0 = tx * a.bx;

  const int pad = (a.ntaps == 27) ? 1 : 0;
  const int HZ = (a.bz - 1) * a.stride + 1 + 2 * pad;
  const int HY = (a.by - 1) * a.stride + 1 + 2 * pad;
  const int HX = (a.bx - 1) * a.stride + 1 + 2 * pad;
  const int HV1 = HZ * HY * HX;
  const int HV = a.bs * HV1;
  const int HVp = (HV + 3) & ~3;
  const int S = a.CK + 4;
  const int nbox = a.bs * a.bz * a.by * a.bx;

  int *outoff = reinterpret_cast<int *>(lds);  // [TM] output voxel index or -1
  int *srcoff = outoff + TM;                   // [HVp] source voxel index or -1
  float *A = lds + TM + HVp;                   // [HV][S] staged halo tile / reduction scratch

  // ---- index tables -----------------------------------------------------
  for (int m = tid; m < TM; m += 256) {
    int off = -1;
    if (m < nbox) {
      const int x = m % a.bx;
      int q = m / a.bx;
      const int y = q % a.by; q /= a.by;
      const int z = q % a.bz;
      const int s = q / a.bz;
      const int b = b0 + s, oz = z0 + z, oy = y0 + y, ox = x0 + x;
      if (b < a.B && oz < a.Zo && oy < a.Yo && ox < a.Xo) off = ((b * a.Zo + oz) * a.Yo + oy) * a.Xo + ox;
    }
    outoff[m] = off;
  }
  {
    const int Zc = a.Zs << a.ups, Yc = a.Ys << a.ups, Xc = a.Xs << a.ups;
    for (int hv = tid; hv < HVp; hv += 256) {
      int off = -1;
      if (hv < HV) {
        const int hx = hv % HX;
        int q = hv / HX;
        const int hy = q % HY; q /= HY;
        const int hz = q % HZ;
        const int s = q / HZ;
        const int b = b0 + s;
        const int cz = z0 * a.stride - pad + hz, cy = y0 * a.stride - pad + hy, cx = x0 * a.stride - pad + hx;
        if (b < a.B && cz >= 0 && cz < Zc && cy >= 0 && cy < Yc && cx >= 0 && cx < Xc)
          off = ((b * a.Zs + (cz >> a.ups)) * a.Ys + (cy >> a.ups)) * a.Xs + (cx >> a.ups);
      }
      srcoff[hv] = off;
    }
  }

  // ---- per-lane LDS row base of each of this wave's MB row blocks -----------
  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mb * 32 + r;
    int hv = 0;
    if (m < nbox) {
      const int x = m % a.bx;
      int q = m / a.bx;
      const int y = q % a.by; q /= a.by;
      const int z = q % a.bz;
      const int s = q / a.bz;
      hv = ((s * HZ + z * a.stride) * HY + y * a.stride) * HX + x * a.stride;
    }
    abase[mb] = hv * S + 4 * h;
  }

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.0f;

  const int K4 = a.CK >> 2, K8 = a.CK >> 3;
  const int nsteps = a.ntaps * K8;
  const int nchunks = a.nch0 + a.nch1;
  const int Ctot = a.C0 + a.C1;
  const f32x4 *wtile = reinterpret_cast<const f32x4 *>(a.wfrag) + (size_t)nt * nchunks * nsteps * NB * 64 + lane;

  const int q4 = tid % K4, v0 = tid / K4, vstep = 256 / K4;

  for (int ch = 0; ch < nchunks; ++ch) {
    const float *src;
    int Cs, c0, cg0;
    if (ch < a.nch0) { src = a.src0; Cs = a.C0; c0 = ch * a.CK; cg0 = c0; }
    else { src = a.src1; Cs = a.C1; c0 = (ch - a.nch0) * a.CK; cg0 = a.C0 + c0; }
    __syncthreads();  // tables ready (first pass) / previous chunk fully consumed
    // ---- stage the halo tile of this channel chunk ---------------------------
    for (int hv = v0; hv < HV; hv += vstep) {
      const int off = srcoff[hv];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (off >= 0) {
        v = *reinterpret_cast<const f32x4 *>(src + (size_t)off * Cs + c0 + 4 * q4);
        if (a.gn) {
          const int b = b0 + (a.bs == 1 ? 0 : hv / HV1);
          const float *g = a.gn + (size_t)b * 2 * Ctot + cg0 + 4 * q4;
          const f32x4 sc = *reinterpret_cast<const f32x4 *>(g);
          const f32x4 sh = *reinterpret_cast<const f32x4 *>(g + Ctot);
          v = v * sc + sh;
          if (a.silu) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
        }
      }
      *reinterpret_cast<f32x4 *>(&A[hv * S + 4 * q4]) = v;
    }
    __syncthreads();
    // ---- this wave's share of the (tap, 8-channel) steps ---------------------
    const f32x4 *wch = wtile + (size_t)ch * nsteps * NB * 64;
    for (int s = wave; s < nsteps; s += 4) {
      const int t = s / K8, j = s - t * K8;
      int tapoff = 0;
      if (a.ntaps == 27) {
        const int dz = t / 9, rem = t - dz * 9, dy = rem / 3, dx = rem - dy * 3;
        tapoff = (dz * HY + dy) * HX + dx;
      }
      const int aoff = tapoff * S + j * 8;
      f32x4 bf[NB], af[MB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bf[nb] = wch[(size_t)(s * NB + nb) * 64];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) af[mb] = *reinterpret_cast<const f32x4 *>(&A[abase[mb] + aoff]);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mb][jj], bf[nb][jj], acc[mb][nb], 0, 0, 0);
    }
  }

  // ---- cross-wave reduction (rounds of RB blocks) + epilogue -------------------
  const int vox_out = a.Zo * a.Yo * a.Xo;
#pragma unroll
  for (int g0 = 0; g0 < NBLK; g0 += RB) {
    __syncthreads();  // A tile (or previous round's scratch) no longer read
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int blk = g0 + i;
      if (blk < NBLK) {
        const int owner = blk & 3;
        if (wave != owner) {
          const int slot = wave - (wave > owner ? 1 : 0);
          float *dst = A + ((i * 3 + slot) * 16) * 64 + lane;
          const int mb = blk / NB, nb = blk % NB;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) dst[reg * 64] = acc[mb][nb][reg];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int blk = g0 + i;
      if (blk < NBLK && wave == (blk & 3)) {
        const int mb = blk / NB, nb = blk % NB;
        f32x16 v = acc[mb][nb];
#pragma unroll
        for (int slot = 0; slot < 3; ++slot) {
          const float *sp = A + ((i * 3 + slot) * 16) * 64 + lane;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) v[reg] += sp[reg * 64];
        }
        const int n = nt * TN + nb * 32 + r;
        if (n < a.Co) {
          const float bias = a.bias[n];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int m = mb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const int off = outoff[m];
            if (off >= 0) {
              float o = v[reg] + bias;
              if (a.temb) {
                const int b = off / vox_out;
                o += a.temb[(size_t)a.tidx[b] * a.temb_stride + n];
              }
              if (a.resid) o += a.resid[(size_t)off * a.res_cs + n];
              a.out[(size_t)off * a.out_cs + n] = o;
            }
          }
        }
      }
    }
  }
}

size_t conv_lds_bytes(const ConvArgs &a, int MB, int NB) {
  const int pad = (a.ntaps == 27) ? 1 : 0;
  const int HZ = (a.bz - 1) * a.stride + 1 + 2 * pad;
  const int HY = (a.by - 1) * a.stride + 1 + 2 * pad;
  const int HX = (a.bx - 1) * a.stride + 1 + 2 * pad;
  const int HV = a.bs * HZ * HY * HX;
  const int HVp = (HV + 3) & ~3;
  const int S = a.CK + 4;
  const int nblk = MB * NB;
  const int rb = nblk < 4 ? nblk : 4;
  size_t tile = (size_t)HV * S;
  size_t red = (size_t)rb * 3 * 16 * 64;
  size_t words = (size_t)32 * MB + HVp + (tile > red ? tile : red);
  return words * 4;
}

#define CM_CONV_VARIANTS(X) \
  X(1, 1) X(2, 1) X(3, 1) X(4, 1) X(5, 1) X(6, 1) X(7, 1) X(8, 1) \
  X(1, 2) X(2, 2) X(3, 2) X(4, 2) \
  X(1, 4) X(2, 4)

bool conv_variant_exists(int MB, int NB) {
#define X(m, n) if (MB == m && NB == n) return true;
  CM_CONV_VARIANTS(X)
#undef X
  return false;
}

hipError_t launch_conv(const ConvArgs &a, int MB, int NB, hipStream_t st) {
  const size_t lds = conv_lds_bytes(a, MB, NB);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int TN = 32 * NB;
  dim3 grid((unsigned)(a.nts * a.ntz * a.nty * a.ntx), (unsigned)((a.Co + TN - 1) / TN));
#define X(m, n)                                                                              \
  if (MB == m && NB == n) {                                                                  \
    static bool attr_set[64] = {false};                                                      \
    int dev = 0;                                                                             \
    (void)hipGetDevice(&dev);                                                                \
    if (!attr_set[dev & 63]) {                                                               \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_mfma_kernel<m, n>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return e;                                                         \
      attr_set[dev & 63] = true;                                                             \
    }                                                                                        \
    hipLaunchKernelGGL((conv_mfma_kernel<m, n>), grid, dim3(256), lds, st, a);                 \
    return hipGetLastError();                                                                \
  }
  CM_CONV_VARIANTS(X)
#undef X
  return hipErrorInvalidValue;
}

}  // namespace cm
