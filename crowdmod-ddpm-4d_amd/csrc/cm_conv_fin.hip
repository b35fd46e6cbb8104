// The UNet's LAST convolution (/root/reference/models/backbones/unet.py:118-122: GroupNorm -> SiLU -> Conv3d(base -> C, 3x3x3),
// C <= 4 output channels) on the matrix core -- round 4.  conv_smalln_kernel (cm_conv_small.hip) runs it on the vector ALUs: 3456
// multiply-adds per voxel and thread, 49 us per ATC step (83 us on the 24x72 grid), because a 32-wide matrix tile with 4 valid
// output columns wastes 7/8 of the instruction.  Here the 27 TAPS are packed into the columns instead:
//
//     P[v][(t, co)] = sum_ci act(x)[v][ci] * W[co][ci][t]          one GEMM per halo plane: rows = the plane's halo voxels,
//                                                                   K = 32 input channels, N = 27 * 4 = 108 columns (4 blocks)
//     out[u][co]    = bias[co] + sum_t P[u + off(t)][(t, co)]       27 shifted reads per output voxel
//
// 96 matrix instructions of the six-term form per 36 output voxels instead of 324 per 32.  One 256-thread workgroup = one
// (sample, BY x BX in-plane tile), looping over the z planes:
//   * per input plane: the (BY + 2) x (BX + 2) <= 64 halo voxels are activated (GroupNorm affine + SiLU), split into three
//     bf16 terms and written to LDS as A[k half][row][term][4 dwords] (the direct kernel's conflict-free layout); the next
//     plane's loads are in flight under this plane's work;
//   * wave w owns column block w (taps 8w .. 8w + 7) and both 32-row blocks; its weight fragments (2 k steps x 3 terms) stay in
//     24 registers for the whole workgroup; 2 x 2 x 6 = 24 matrix instructions per plane and wave;
//   * P goes to LDS ([64 rows][128 columns] fp32); thread (u, dz) gathers the nine (dy, dx) entries of its z tap and adds them to
//     the running sum of output plane z - dz + 1, kept in LDS ([3 planes][voxel][4]) -- for one output the three z taps arrive in
//     the order dz = 0, 1, 2 from three consecutive planes, so the summation order is fixed; the thread that adds dz = 2
//     finishes the voxel (bias, store) and clears the slot.
// Zero padding: out-of-grid halo voxels are zero rows (P = 0); the padding planes z = -1 and Z are never staged.
// F16 (reduced-precision plan): one v_mfma_f32_32x32x16_f16 per (row block, k step) on f16 operands; the source tensor may be
// stored as f16 (ConvArgs::h16 bit 0).
#include "cm_kernels.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace cm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f2(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int FIN_ROWS = 64;                     // halo voxels of a plane tile (two 32-row blocks), >= (BY + 2)(BX + 2)
constexpr int FIN_PCOLS = 132;                   // P row stride in floats (128 columns + 4: 16-byte gathers of a row group stay apart)

// wfin: [nb 4][k step 2][term NTM][lane 64] 16 B, column n = 32 nb + lane % 32 = 4 t + co, k = 16 ks + 8 (lane / 32) + j
template <int MODE>                                // 0: six bf16 cross terms, 1: f16 operands, 2: three bf16 cross terms (relaxed plan), 3: three f16 cross terms (h2)
__global__ __launch_bounds__(256, 2) void conv_fin_kernel(const ConvArgs a, const float *__restrict__ wfin, int ntx) {
  constexpr bool F16 = MODE == 1, H2 = MODE == 3;
  constexpr int U0 = MODE >= 2 ? 3 : 0, NTW = MODE >= 2 ? 2 : 3;
  constexpr int NTM = F16 ? 1 : 3;
  constexpr int RW = F16 ? 4 : 12;               // LDS dwords per (k half, row): terms x 4
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *A = lds;                                // [2 k steps][2 hh][FIN_ROWS][RW]
  float *P = A + 4 * FIN_ROWS * RW;              // [FIN_ROWS][FIN_PCOLS]   (A: two k steps x two k halves)
  float *OACC = P + FIN_ROWS * FIN_PCOLS;        // [3][BY * BX (<= 64)][4] running sums of three output planes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int BY = a.by, BX = a.bx, HX = BX + 2, HV = (BY + 2) * HX, NV = BY * BX;
  int tile = blockIdx.x;
  const int ntp = a.nty * ntx;
  const int b = tile / ntp, p = tile - b * ntp;
  const int ty = p / ntx, tx = p - ty * ntx;
  const int y0 = ty * BY, x0 = tx * BX;
  const size_t Vp = (size_t)a.Ys * a.Xs;         // voxels per plane
  const int C = a.C0;                            // 32

  // ---- this thread's two staging items: halo voxel hv = tid / 8 + 32 k, channel quad q = tid & 7 -------------------------
  const int q = tid & 7;
  int soff[2];
  bool sok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int hv = (tid >> 3) + 32 * k;
    const int hy = hv / HX, hx = hv - hy * HX;
    const int cy = y0 - 1 + hy, cx = x0 - 1 + hx;
    sok[k] = hv < HV && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs;
    soff[k] = sok[k] ? cy * a.Xs + cx : 0;
  }
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (a.gn) {
    const float *g = a.gn + (size_t)b * 2 * C + 4 * q;
    sc = *reinterpret_cast<const f32x4 *>(g);
    sh = *reinterpret_cast<const f32x4 *>(g + C);
  }
  const bool h16 = (a.h16 & 1) != 0;
  f32x4 ld[2];
  auto issue = [&](int z) {
    if (h16) {
      const _Float16 *sp = reinterpret_cast<const _Float16 *>(a.src0) + ((size_t)(b * a.Zs + z) * Vp) * C + 4 * q;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const cm_f32x2_t two = *reinterpret_cast<const cm_f32x2_t *>(sp + (size_t)soff[k] * C);
        const f16x4 hv4 = __builtin_bit_cast(f16x4, two);
        ld[k] = f32x4{(float)hv4[0], (float)hv4[1], (float)hv4[2], (float)hv4[3]};
      }
    } else {
      const float *sp = a.src0 + ((size_t)(b * a.Zs + z) * Vp) * C + 4 * q;
#pragma unroll
      for (int k = 0; k < 2; ++k) ld[k] = *reinterpret_cast<const f32x4 *>(sp + (size_t)soff[k] * C);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int hv = (tid >> 3) + 32 * k;
      f32x4 w = ld[k];
      if (a.gn) {
        w = w * sc + sh;
        if (a.silu) { w[0] = silu_f2(w[0]); w[1] = silu_f2(w[1]); w[2] = silu_f2(w[2]); w[3] = silu_f2(w[3]); }
      }
      if (!sok[k]) w = f32x4{0.f, 0.f, 0.f, 0.f};
      // k = 4 q .. 4 q + 3 of the 32 channels: k step q >> 2, k half (q >> 1) & 1, dwords 2 (q & 1) of the half's 4
      float *dst = A + (size_t)(((q >> 1) & 1) * FIN_ROWS + hv) * RW + 2 * (q & 1) + (q >> 2) * (2 * FIN_ROWS * RW);
      if constexpr (F16) {
        const f16x4 hv4 = {(_Float16)w[0], (_Float16)w[1], (_Float16)w[2], (_Float16)w[3]};
        *reinterpret_cast<f16x4 *>(dst) = hv4;
      } else {
        cm_u32x2_t t3[3];
        if constexpr (H2) cm_split2_f16(w, t3);
        else cm_split3_bf16<NTW>(w, t3);
#pragma unroll
        for (int tm = 0; tm < NTW; ++tm) *reinterpret_cast<cm_u32x2_t *>(dst + 4 * tm) = t3[tm];
      }
    }
  };
  // (A holds two images, one per k step of 16 channels: ks * (2 FIN_ROWS RW) + (hh FIN_ROWS + row) RW)

  // weight fragments of this wave's column block: registers for the whole workgroup
  f32x4 bw[2][NTM];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int tm = 0; tm < NTM; ++tm) bw[ks][tm] = reinterpret_cast<const f32x4 *>(wfin)[((size_t)(wave * 2 + ks) * NTM + tm) * 64 + lane];

  // gather role of this thread: output voxel u = tid % 64 (< NV), z tap dzg = tid / 64 (< 3)
  const int gu = tid & 63, dzg = tid >> 6;
  const bool gath = gu < NV && dzg < 3;
  const int guy = gu / BX, gux = gu - guy * BX;
  const int grow = guy * HX + gux;                // halo row of tap (dy, dx) = (0, 0)
  const int oy = y0 + guy, ox = x0 + gux;
  const bool oin = gath && oy < a.Yo && ox < a.Xo;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (gath && dzg >= 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) bias4[c] = c < a.Co ? a.bias[c] : 0.f;
  }
  for (int i = tid; i < 3 * 64 * 4; i += 256) OACC[i] = 0.f;

  const int Z = a.Zs;
  constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
  issue(0);
  stage();
  if (Z > 1) issue(1);
  __syncthreads();                                // A of plane 0 complete, the running sums cleared
  for (int z = 0; z < Z; ++z) {
    // ---- P^T = W^T x A^T for this wave's 32 columns, both row blocks: the WEIGHTS are the row operand, so that a lane ends up
    //      with 4 taps x 4 output channels of ITS voxel -- four 16-byte stores per block instead of sixteen 4-byte ones ---------
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float *ap = A + (size_t)ks * (2 * FIN_ROWS * RW) + (size_t)(hh * FIN_ROWS + 32 * j + r) * RW;
        if constexpr (F16) {
          const f32x4 af = *reinterpret_cast<const f32x4 *>(ap);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bw[ks][0]), __builtin_bit_cast(f16x8, af), acc[j], 0, 0, 0);
        } else {
          f32x4 af[3];
#pragma unroll
          for (int tm = 0; tm < 3; ++tm) af[tm] = *reinterpret_cast<const f32x4 *>(ap + 4 * tm);
#pragma unroll
          for (int u = U0; u < 6; ++u) {
            if constexpr (H2)
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bw[ks][TB[u]]), __builtin_bit_cast(f16x8, af[TA[u]]), acc[j], 0, 0, 0);
            else
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw[ks][TB[u]]), __builtin_bit_cast(bf16x8, af[TA[u]]), acc[j], 0, 0, 0);
          }
        }
      }
    // lane (r, hh), registers 4 g .. 4 g + 3 of block j: voxel row 32 j + r, columns 4 (8 wave + 2 g + hh) + {0 .. 3} = tap 8 wave + 2 g + hh
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 pv = f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
        if constexpr (H2) pv = pv * a.h2_oscale;   // (weights were packed as w * 2^k)
        *reinterpret_cast<f32x4 *>(P + (size_t)(32 * j + r) * FIN_PCOLS + 4 * (8 * wave + 2 * g + hh)) = pv;
      }
    __syncthreads();                              // P complete; every wave has read A
    // ---- the next plane's image (A is free) beside this plane's gathers (P) -------------------------------------------------
    if (z + 1 < Z) {
      stage();
      if (z + 2 < Z) issue(z + 2);
    }
    // gather: input plane z feeds output plane zo = z + 1 - dz through z tap dz; the taps of one output arrive from three
    // consecutive planes in the order dz = 0, 1, 2, each added by one thread: a fixed summation order
    if (gath) {
      const int zo = z + 1 - dzg;
      if (zo >= 0 && zo < a.Zo) {
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
          const int dy = t9 / 3, dx = t9 - 3 * dy;
          s4 += *reinterpret_cast<const f32x4 *>(P + (size_t)(grow + dy * HX + dx) * FIN_PCOLS + 4 * (dzg * 9 + t9));
        }
        float *oa = OACC + (size_t)(((zo % 3) * 64 + gu) * 4);
        f32x4 v = *reinterpret_cast<const f32x4 *>(oa) + s4;
        // the last tap of an output: dz = 2 (from plane zo + 1), or dz = 1 for the top plane (its dz = 2 plane is zero padding) --
        // the thread that adds it finishes the voxel (bias, store) and clears the slot for output plane zo + 3
        const bool last = dzg == 2 || (dzg == 1 && z == Z - 1);
        if (last) {
          if (oin) {
            float *o = a.out + ((size_t)((b * a.Zo + zo) * a.Yo + oy) * a.Xo + ox) * a.out_cs;
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < a.Co) o[c] = v[c] + bias4[c];
          }
          v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        *reinterpret_cast<f32x4 *>(oa) = v;
      }
    }
    __syncthreads();                              // A of plane z + 1 complete; the gathers have read P; slot updates visible
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
// in-plane tile (by, bx): divides the grid, (by + 2)(bx + 2) <= 64 halo voxels, by * bx <= 64; prefers full row blocks
bool conv_fin_pick(int Y, int X, int *by, int *bx) {
  double best = 0;
  for (int y = 1; y <= Y; ++y)
    for (int x = 1; x <= X; ++x) {
      if (Y % y || X % x) continue;
      const int hv = (y + 2) * (x + 2);
      if (hv > FIN_ROWS || y * x > 64) continue;
      const double score = (double)(y * x) / FIN_ROWS;      // useful outputs per 64 staged rows
      if (score > best) { best = score; *by = y; *bx = x; }
    }
  return best > 0;
}

bool conv_fin_ok(const ConvArgs &a) {
  return a.ntaps == 27 && a.td == 3 && a.stride == 1 && !a.par && !a.ups && a.C0 == 32 && a.C1 == 0 && a.Co >= 1 && a.Co <= 4 && !a.temb &&
         !a.resid && !a.stat_part && !a.astat && !a.pm && a.ks <= 1 && a.out_cs >= a.Co && a.Zs == a.Zo && a.Ys == a.Yo && a.Xs == a.Xo && a.by > 0 &&
         a.bx > 0 && a.Yo % a.by == 0 && a.Xo % a.bx == 0 && (a.by + 2) * (a.bx + 2) <= FIN_ROWS && a.by * a.bx <= 64 && !(a.h16 & ~1);
}

size_t conv_fin_lds(bool f16) {
  return ((size_t)2 * 2 * FIN_ROWS * (f16 ? 4 : 12) + (size_t)FIN_ROWS * FIN_PCOLS + 3 * 64 * 4) * sizeof(float);
}

// mode 0: six-term products (fragments of launch_fin_pack(f16 = false)), 1: f16 operands, 2: three of the six terms on the mode-0 fragments
hipError_t launch_conv_fin(const ConvArgs &a_in, const float *wfin, int mode, hipStream_t st) {
  ConvArgs a = a_in;
  a.dbg = conv_dbg_flags();
  if (!conv_fin_ok(a) || mode < 0 || mode > 3) return hipErrorInvalidValue;
  const int nty = a.Yo / a.by, ntx = a.Xo / a.bx;
  a.nty = nty;
  const dim3 grid((unsigned)(a.B * nty * ntx));
  const size_t lds = conv_fin_lds(mode == 1);
  static bool attr_set[64][4] = {{false}};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const void *fn = mode == 1 ? reinterpret_cast<const void *>(conv_fin_kernel<1>)
                 : mode == 2 ? reinterpret_cast<const void *>(conv_fin_kernel<2>)
                 : mode == 3 ? reinterpret_cast<const void *>(conv_fin_kernel<3>) : reinterpret_cast<const void *>(conv_fin_kernel<0>);
  if (!attr_set[dev & 63][mode]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev & 63][mode] = true;
  }
  if (mode == 1) hipLaunchKernelGGL(conv_fin_kernel<1>, grid, dim3(256), lds, st, a, wfin, ntx);
  else if (mode == 2) hipLaunchKernelGGL(conv_fin_kernel<2>, grid, dim3(256), lds, st, a, wfin, ntx);
  else if (mode == 3) hipLaunchKernelGGL(conv_fin_kernel<3>, grid, dim3(256), lds, st, a, wfin, ntx);
  else hipLaunchKernelGGL(conv_fin_kernel<0>, grid, dim3(256), lds, st, a, wfin, ntx);
  return hipGetLastError();
}

// Fragments from the layer's fp32 weights in the REFERENCE layout [Co][32][kH][kW][kL] (internal tap (dz, dy, dx) = element
// [dy][dx][dz]): [nb 4][k step 2][term][lane 64][8 x 16 bit]; column n = 32 nb + lane % 32 = 4 t + co (t = (dz * 3 + dy) * 3 + dx; columns
// >= 108 and co >= Co are zero), k = 16 ks + 8 (lane / 32) + j.  f16 = 0: three bf16 terms (exact split), 1: one f16 term.
// One thread per (nb, ks, lane, j); run at load time and after every optimizer step (the same kernel: one definition).
__global__ __launch_bounds__(256) void fin_pack_kernel(const float *__restrict__ w, unsigned short *__restrict__ out, int Co, int f16, float wscale) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= 4 * 2 * 64 * 8) return;
  const int j = o & 7, lane = (o >> 3) & 63, ks = (o >> 9) & 1, nb = o >> 10;
  const int n = 32 * nb + (lane & 31), t = n >> 2, co = n & 3, ci = 16 * ks + 8 * (lane >> 5) + j;
  float v = 0.f;
  if (t < 27 && co < Co) {
    const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
    v = w[((size_t)co * 32 + ci) * 27 + (dy * 3 + dx) * 3 + dz];
  }
  const int ntm = f16 == 1 ? 1 : 3;
  const size_t base = (((size_t)(nb * 2 + ks) * ntm) * 64 + lane) * 8 + j;
  if (f16 == 2) {                                  // h2: f16 hi / mid of w * 2^k in the three-term slots (the third stays zero)
    const float vs = v * wscale;
    const _Float16 h = (_Float16)vs;
    const _Float16 md = (_Float16)(vs - (float)h);
    out[base] = __builtin_bit_cast(unsigned short, h);
    out[base + (size_t)64 * 8] = __builtin_bit_cast(unsigned short, md);
    out[base + (size_t)2 * 64 * 8] = 0;
  } else if (f16) {
    const _Float16 h = (_Float16)v;
    out[base] = __builtin_bit_cast(unsigned short, h);
  } else {
    float rem = v;
#pragma unroll
    for (int tm = 0; tm < 3; ++tm) {
      const __bf16 h = (__bf16)rem;
      out[base + (size_t)tm * 64 * 8] = __builtin_bit_cast(unsigned short, h);
      rem -= (float)h;
    }
  }
}

hipError_t launch_fin_pack(const float *w_ref, float *wfin, int Co, int mode, hipStream_t st, float wscale) {
  hipLaunchKernelGGL(fin_pack_kernel, dim3(16), dim3(256), 0, st, w_ref, reinterpret_cast<unsigned short *>(wfin), Co, mode, wscale);
  return hipGetLastError();
}

}  // namespace cm
