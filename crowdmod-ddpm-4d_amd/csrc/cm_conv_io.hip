// The UNet's first convolution (reference: /root/reference/models/backbones/unet.py:32,142 --
// nn.Conv3d(C, base, 3, padding=1) on cat(past, future), C = 3 or 4 input channels).
//
// K = 27 taps x 4 (or 8) channels is tiny and N = base = 32 is exactly one MFMA tile, so the shape
// wants the opposite of the generic kernel's split: the WHOLE weight set of a 32-channel output tile
// lives in registers (54 or 108 VGPRs per lane, loaded once per wave), the waves of a workgroup split
// the output VOXELS (32-row blocks), and nothing is reduced across waves.  The generic kernel padded
// the 4 channels to an 8-channel K step (2x the matrix work) and ran its per-chunk machinery for one
// chunk: 51.6 us against an MFMA floor of 9.7 us and a write floor of ~5 us (28 MB).
//
//   * tile = bz x by planes of FULL x-rows (bx = X): the 32 voxels of an accumulator block are then
//     consecutive along x up to one row break, i.e. consecutive in the LDS halo image -> conflict-free
//     A reads (the box tiles of the generic kernel scatter a block over 8 rows);
//   * LDS image: two planes [half][voxel][CIN/2] so that lane half hh (the MFMA's k index) reads its own
//     contiguous plane: ds_read_b64 (CIN = 4) / ds_read_b128 (CIN = 8) per tap, no selects;
//   * MFMA (tap t, step p) contracts k = {channel (CIN/2)*hh + p : hh = 0, 1}; the weights are packed on the
//     host in exactly that order, one register per step;
//   * epilogue: bias, channels-last store (128 B per voxel row), GroupNorm statistics of the output per
//     32-row block in the slot format gn_finalize reads (same arithmetic as the generic conv epilogue).
#include "cm_kernels.h"

namespace cm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CIN>
__global__ __launch_bounds__(256) void conv_first_kernel(const ConvArgs a, const float *__restrict__ wpk, int nblk) {
  constexpr int NS = 27 * CIN / 2;           // MFMA steps = weight registers per lane
  constexpr int HC = CIN / 2;                // channels per lane half
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  int tile = blockIdx.x;
  if (!(gridDim.x & 7) && !(a.dbg & 4096)) tile = (tile & 7) * (int)(gridDim.x >> 3) + (tile >> 3);  // XCD-aware order (cm_conv.hip)
  const int ty = tile % a.nty; tile /= a.nty;
  const int tz = tile % a.ntz;
  const int b = tile / a.ntz;
  const int nt = blockIdx.y;
  const int X = a.Xo, bz = a.bz, by = a.by;
  const int z0 = tz * bz, y0 = ty * by;
  const int HY = by + 2, HX = X + 2;
  const int HV = (bz + 2) * HY * HX;
  const int nbox = bz * by * X;

  int *outoff = reinterpret_cast<int *>(lds);            // [32 * nblk]
  float *P0 = lds + 32 * nblk;                           // plane hh at P0 + hh * HV * HC

  // weights of this output tile: one coalesced dword per step, resident for the whole kernel
  float wreg[NS];
  {
    const float *wp = wpk + (size_t)nt * NS * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) wreg[s] = wp[s * 64];
  }
  // ---- stage the halo image (zero padding outside the grid) ----------------------------
  // (four items per pass, loads first: one memory round trip per pass instead of one per item)
  constexpr int SU = 4;
  for (int hv0 = tid; hv0 < HV; hv0 += 256 * SU) {
    f32x4 v0[SU], v1[CIN == 8 ? SU : 1];
    bool okv[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int hv = min(hv0 + 256 * u, HV - 1);
      const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
      const int cz = z0 + hz - 1, cy = y0 + hy - 1, cx = hx - 1;
      const bool ok = cz >= 0 && cz < a.Zs && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs;
      const float *sp = a.src0 + ((size_t)((b * a.Zs + (ok ? cz : 0)) * a.Ys + (ok ? cy : 0)) * a.Xs + (ok ? cx : 0)) * a.C0;
      v0[u] = *reinterpret_cast<const f32x4 *>(sp);
      if constexpr (CIN == 8) v1[u] = *reinterpret_cast<const f32x4 *>(sp + 4);
      okv[u] = ok;
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int hv = hv0 + 256 * u;
      if (hv >= HV) continue;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 w0 = okv[u] ? v0[u] : z4;
      if constexpr (CIN == 4) {
        *reinterpret_cast<f32x2 *>(P0 + hv * 2) = f32x2{w0[0], w0[1]};
        *reinterpret_cast<f32x2 *>(P0 + (HV + hv) * 2) = f32x2{w0[2], w0[3]};
      } else {
        const f32x4 w1 = okv[u] ? v1[u] : z4;
        *reinterpret_cast<f32x4 *>(P0 + hv * 4) = w0;
        *reinterpret_cast<f32x4 *>(P0 + (HV + hv) * 4) = w1;
      }
    }
  }
  for (int m = tid; m < 32 * nblk; m += 256) {
    int off = -1;
    if (m < nbox) {
      const int z = m / (by * X), rem = m - z * (by * X), y = rem / X, x = rem - y * X;
      const int oz = z0 + z, oy = y0 + y;
      if (oz < a.Zo && oy < a.Yo && b < a.B) off = ((b * a.Zo + oz) * a.Yo + oy) * a.Xo + x;
    }
    outoff[m] = off;
  }
  __syncthreads();

  const float *Ph = P0 + (size_t)hh * HV * HC;
  const int n = nt * 32 + r;
  const bool nok = n < a.Co;
  const float bias = a.bias[nok ? n : 0];
  for (int blk = wave; blk < nblk; blk += 4) {
    const int m = min(blk * 32 + r, nbox - 1);
    const int z = m / (by * X), rem = m - z * (by * X), y = rem / X, x = rem - y * X;
    const float *ap = Ph + (size_t)((z * HY + y) * HX + x) * HC;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
      const int toff = ((dz * HY + dy) * HX + dx) * HC;
      if constexpr (CIN == 4) {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(ap + toff);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], wreg[2 * t], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[1], wreg[2 * t + 1], acc, 0, 0, 0);
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(ap + toff);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[jj], wreg[4 * t + jj], acc, 0, 0, 0);
      }
    }
    // ---- epilogue of this 32-row block: gathers first, then the stores (counted waits) ----
    int offs[16];
    float rs[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) offs[reg] = outoff[blk * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) rs[reg] = acc[reg] + bias;
    if (a.h16 & 4) {                              // f16 output tensor (reduced-precision plan)
      _Float16 *oh = reinterpret_cast<_Float16 *>(a.out);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (nok && offs[reg] >= 0) oh[(size_t)offs[reg] * a.out_cs + n] = (_Float16)rs[reg];
    } else {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (nok && offs[reg] >= 0) a.out[(size_t)offs[reg] * a.out_cs + n] = rs[reg];
    }
    if (a.stat_part || a.astat) {
      // GroupNorm statistics of the block (two-pass on registers; the two lane halves hold disjoint rows)
      float s1 = 0.f, cnt = 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (offs[reg] >= 0) { s1 += rs[reg]; cnt += 1.f; }
      s1 += __shfl_xor(s1, 32);
      cnt += __shfl_xor(cnt, 32);
      const float mean = cnt > 0.f ? s1 / cnt : 0.f;
      float q = 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (offs[reg] >= 0) { const float d = rs[reg] - mean; q += d * d; }
      q += __shfl_xor(q, 32);
      const int slot = (tz * a.nty + ty) * nblk + blk;
      if (a.astat) {
        if (hh == 0 && nok && b < a.B && cnt > 0.f) cm_stat_atomic(a.astat + ((size_t)b * a.astat_C + n) * 3, s1, mean, q);
      } else {
        if (hh == 0 && nok && b < a.B) {
          float *sp = a.stat_part + (((size_t)b * a.stat_ns + slot) * a.stat_C + n) * 2;
          sp[0] = mean;
          sp[1] = q;
        }
        if (lane == 0 && n == 0 && b < a.B) a.stat_cnt[(size_t)b * a.stat_ns + slot] = cnt;
      }
    }
  }
}

int conv_first_blocks(const ConvArgs &a) { return (a.bz * a.by * a.Xo + 31) / 32; }

size_t conv_first_lds(const ConvArgs &a, int cin) {
  const size_t HV = (size_t)(a.bz + 2) * (a.by + 2) * (a.Xo + 2);
  return ((size_t)32 * conv_first_blocks(a) + HV * cin) * sizeof(float);
}

bool conv_first_ok(const ConvArgs &a, int cin) {
  return a.ntaps == 27 && a.td == 3 && a.stride == 1 && !a.par && !a.ups && !a.gn && !a.temb && !a.resid && !a.src1 &&
         a.ks <= 1 && (cin == 4 || cin == 8) && a.C0 >= cin && a.C0 % 4 == 0 && a.bx == a.Xo && a.Xs == a.Xo &&
         conv_first_lds(a, cin) <= 64 * 1024;
}

hipError_t launch_conv_first(const ConvArgs &a_in, int cin, const float *wpk, hipStream_t st) {
  ConvArgs a = a_in;
  a.dbg = conv_dbg_flags();
  if (!conv_first_ok(a, cin)) return hipErrorInvalidValue;
  const int nblk = conv_first_blocks(a);
  const dim3 grid((unsigned)(a.B * a.ntz * a.nty), (unsigned)((a.Co + 31) / 32));
  const size_t lds = conv_first_lds(a, cin);
  if (cin == 4) hipLaunchKernelGGL(conv_first_kernel<4>, grid, dim3(256), lds, st, a, wpk, nblk);
  else hipLaunchKernelGGL(conv_first_kernel<8>, grid, dim3(256), lds, st, a, wpk, nblk);
  return hipGetLastError();
}

}  // namespace cm
