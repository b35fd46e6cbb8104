// Whole-sample 3x3x3 convolution for the lowest UNet resolution (two z planes, <= 64 voxels per plane): the ten
// quarter-resolution nn.Conv3d layers of the reference's encoder level 2 / bottleneck / decoder
// (/root/reference/models/backbones/layers.py:30-43,57-75 -- GroupNorm -> SiLU -> conv, time row, residual / 1x1x1
// skip conv -- as wired at unet.py:57-115).  Inference plan only.
//
// Why (round 3): this resolution is 13 % of the FLOPs and was a third of the step -- 25 dependent launches, each conv a
// K-split launch (512 single-round workgroups that run prologue -> stage -> matrix phase -> cross-wave reduction -> store
// in lock-step) plus a combine / GroupNorm-finalise launch.  At 54 voxels a SAMPLE is the natural tile:
//   * one 512-thread workgroup = (sample, 32 output channels); the sample's whole activated input (both planes, 1-voxel
//     in-plane halo, ALL input channels: <= 150 KB) is staged in LDS once;
//   * the GroupNorm of the input is finalised INSIDE the workgroup from the producer's per-slot (mean, M2) partials --
//     at this resolution that is 2-4 slots per channel, i.e. a 4 KB read -- so neither the K-split combine pass nor the
//     gn_finalize launch exists any more (15 launches per step);
//   * z-split as in cm_conv.hip: row block = voxels of ONE plane, so the z tap that would multiply the zero-padding plane
//     is never issued (18 of 27 taps); here both planes share their A fragments: per in-plane tap (dy, dx) and 8-channel
//     step the fragments of input plane 0 and 1 feed  out0 += A0 W(dz=1) + A1 W(dz=2),  out1 += A0 W(dz=0) + A1 W(dz=1);
//   * K (in-plane tap x 8-channel step) is split over the 8 waves in contiguous ranges, weights stream global -> VGPR in
//     consumption order (one 3 KB group per step, next group prefetched), the 8 partial accumulator sets are summed in
//     wave order through LDS (16-byte accesses), then bias / time row / residual / fused 1x1x1 skip conv, channels-last
//     store and the GroupNorm statistics of the output in the slot format (slot = row block).
// Deterministic: fixed summation order, no atomics; a sample's result does not depend on the batch it is in.
#include "cm_kernels.h"

#include <algorithm>

namespace cm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef CM_QR2_ABL
#define CM_QR2_ABL 0         // compile-time ablations of conv_qr2_kernel's split forms (experiments only; results are wrong): 1 one matrix
#endif                       // instruction per product, 2 no weight refill, 4 no activation, 8 no reduction / epilogue, 16 no GroupNorm merge
typedef __bf16 bf16x8q __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8q __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4q __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float silu_q(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// Chan et al. pairwise combination of (n, mean, M2) triples (same arithmetic as cm_misc.hip)
__device__ __forceinline__ void chan_combine_q(float &n, float &mean, float &m2, float nb, float meanb, float m2b) {
  if (nb == 0.f) return;
  const float nt = n + nb;
  const float d = meanb - mean;
  const float f = nb * __builtin_amdgcn_rcpf(nt);
  mean += d * f;
  m2 += m2b + d * d * n * f;
  n = nt;
}

// MBP = 32-row blocks per plane (1: up to 32 voxels per plane, 2: up to 64); MB = 2 MBP blocks per workgroup.
template <int MBP, bool SKIP>
__global__ __launch_bounds__(512, 2) void conv_qr_kernel(const QrArgs a) {
  constexpr int MB = 2 * MBP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.x, nt = blockIdx.y;
  const int Ci = a.C0 + a.C1, Q = Ci >> 2;
  const int Y = a.Y, X = a.X, PV = Y * X, V = 2 * PV;
  const int HX = X + 2, HYX = (Y + 2) * HX, HV = 2 * HYX;
  const int S = Ci + 4;
  // LDS: gtab [2 Ci] scale | shift, tmp [2 Ci + 32], hvinfo [HVp], red [2][MB][8][32], then A [HV][S] (later: P, the reduction buffer)
  float *gtab = lds;
  float *tmp = gtab + 2 * Ci;
  int *hvinfo = reinterpret_cast<int *>(tmp + 2 * Ci + 32);
  const int HVp = (HV + 3) & ~3;
  float *red = reinterpret_cast<float *>(hvinfo + HVp);
  float *A = red + 2 * MB * 8 * 32;

  // ---- halo table: in-sample source voxel of halo voxel hv, or -1 (in-plane zero padding) ------------------------
  if (tid < HV) {
    const int pl = tid / HYX, rem = tid - pl * HYX, hy = rem / HX, hx = rem - hy * HX;
    const bool in = hy >= 1 && hy <= Y && hx >= 1 && hx <= X;
    hvinfo[tid] = in ? (pl * Y + (hy - 1)) * X + (hx - 1) : -1;
  }
  // ---- GroupNorm of the input, finalised here from the producers' slot partials (layers.py:30,41) ----------------
  if (a.gamma) {
    if (tid < Ci) {
      const int c = tid;
      const float *p, *nn;
      int Cx, cc, ns;
      if (c < a.C0) { p = a.part0; nn = a.cnt0; Cx = a.C0; cc = c; ns = a.ns0; } else { p = a.part1; nn = a.cnt1; Cx = a.C1; cc = c - a.C0; ns = a.ns1; }
      float N = 0.f, M = 0.f, S2 = 0.f;
      for (int s = 0; s < ns; ++s) {
        const float2 q = *reinterpret_cast<const float2 *>(p + (((size_t)b * ns + s) * Cx + cc) * 2);
        chan_combine_q(N, M, S2, nn[(size_t)b * ns + s], q.x, q.y);
      }
      tmp[c] = M;
      tmp[Ci + c] = S2;
    }
    __syncthreads();
    const int cg = Ci / a.groups;
    if (tid < a.groups) {
      float N = 0.f, M = 0.f, S2 = 0.f;
      for (int i = 0; i < cg; ++i) chan_combine_q(N, M, S2, (float)V, tmp[tid * cg + i], tmp[Ci + tid * cg + i]);
      tmp[2 * Ci + tid] = M;
      tmp[2 * Ci + 16 + tid] = rsqrtf(S2 / N + a.eps);
    }
    __syncthreads();
    if (tid < Ci) {
      const int g = tid / cg;
      const float sc = tmp[2 * Ci + 16 + g] * a.gamma[tid];
      gtab[tid] = sc;
      gtab[Ci + tid] = a.beta[tid] - tmp[2 * Ci + g] * sc;
      if (a.gn_out && nt == 0) {                 // kept for consumers outside this kernel family (none in the inference plan)
        a.gn_out[((size_t)b * 2 + 0) * Ci + tid] = sc;
        a.gn_out[((size_t)b * 2 + 1) * Ci + tid] = gtab[Ci + tid];
      }
    }
  }
  __syncthreads();

  // ---- stage the whole activated input: thread = (halo voxel lane, channel quad) ---------------------------------
  {
    const int QP = 1 << a.qshift;                // power of two >= Q
    const int q = tid & (QP - 1), hv0 = tid >> a.qshift, hvstep = 512 >> a.qshift;
    const bool qok = q < Q;
    const int c = 4 * q;
    const bool from0 = c < a.C0;
    const float *sp = from0 ? a.src0 + (size_t)b * V * a.C0 + c : a.src1 + (size_t)b * V * a.C1 + (c - a.C0);
    const int Cs = from0 ? a.C0 : a.C1;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (a.gamma && qok) {
      sc = *reinterpret_cast<const f32x4 *>(gtab + c);
      sh = *reinterpret_cast<const f32x4 *>(gtab + Ci + c);
    }
    constexpr int SU = 8;
    for (int h0 = hv0; h0 < HV; h0 += hvstep * SU) {
      f32x4 v[SU];
      int off[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int hv = h0 + u * hvstep;
        off[u] = (hv < HV && qok) ? hvinfo[hv] : -1;
        v[u] = *reinterpret_cast<const f32x4 *>(sp + (size_t)(off[u] >= 0 ? off[u] : 0) * Cs);
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int hv = h0 + u * hvstep;
        f32x4 w = v[u];
        if (a.gamma) {
          w = w * sc + sh;
          if (a.silu) { w[0] = silu_q(w[0]); w[1] = silu_q(w[1]); w[2] = silu_q(w[2]); w[3] = silu_q(w[3]); }
        }
        if (off[u] < 0) w = f32x4{0.f, 0.f, 0.f, 0.f};
        if (hv < HV && qok) *reinterpret_cast<f32x4 *>(A + (size_t)hv * S + c) = w;
      }
    }
  }
  // ---- epilogue operands of this thread's output elements, requested now ------------------------------------------
  const int n = nt * 32 + r;
  const float bias_pre = a.bias[n];
  const float tv_pre = a.temb ? a.temb[(size_t)a.tidx[b] * a.temb_stride + n] : 0.f;

  // ---- per-lane A row bases: row block j of a plane holds the plane's voxels 32 j .. 32 j + 31 --------------------
  int abase[MBP];
#pragma unroll
  for (int j = 0; j < MBP; ++j) {
    const int v = min(j * 32 + r, PV - 1);
    const int y = v / X, x = v - y * X;
    abase[j] = (y * HX + x) * S + 4 * hh;        // tap (dy, dx) adds (dy HX + dx) S, input plane p adds p HYX S, step k8 adds 8 k8
  }
  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  const int K8 = Ci >> 3, ng = 9 * K8;
  const int g0 = wave * ng / 8, g1 = (wave + 1) * ng / 8;
  const f32x4 *wq = reinterpret_cast<const f32x4 *>(a.wq) + (size_t)nt * ng * 3 * 64 + lane;
  f32x4 bw[2][3];
  if (g0 < g1) {
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) bw[0][dz] = wq[((size_t)g0 * 3 + dz) * 64];
  }
  __syncthreads();                                // A complete

  // ---- matrix phase: this wave's groups (k8, dy, dx), two at a time so that the weight ring is indexed statically ----
  auto step = [&](int g, const f32x4 (&w3)[3]) {
    const int k8 = g / 9, t9 = g - 9 * k8, dy = t9 / 3, dx = t9 - 3 * dy;
    const int toff = (dy * HX + dx) * S + 8 * k8;
    f32x4 a0[MBP], a1[MBP];
#pragma unroll
    for (int j = 0; j < MBP; ++j) {
      a0[j] = *reinterpret_cast<const f32x4 *>(A + abase[j] + toff);
      a1[j] = *reinterpret_cast<const f32x4 *>(A + abase[j] + toff + HYX * S);
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int j = 0; j < MBP; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j][jj], w3[1][jj], acc[j], 0, 0, 0);              // out plane 0 <- in plane 0, dz = 1
        acc[MBP + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j][jj], w3[0][jj], acc[MBP + j], 0, 0, 0);  // out plane 1 <- in plane 0, dz = 0
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j][jj], w3[2][jj], acc[j], 0, 0, 0);              // out plane 0 <- in plane 1, dz = 2
        acc[MBP + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j][jj], w3[1][jj], acc[MBP + j], 0, 0, 0);  // out plane 1 <- in plane 1, dz = 1
      }
  };
  for (int g = g0; g < g1; g += 2) {
    if (g + 1 < g1) {
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) bw[1][dz] = wq[((size_t)(g + 1) * 3 + dz) * 64];
    }
    step(g, bw[0]);
    if (g + 1 < g1) {
      if (g + 2 < g1) {
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) bw[0][dz] = wq[((size_t)(g + 2) * 3 + dz) * 64];
      }
      step(g + 1, bw[1]);
    }
  }
  // ---- fused 1x1x1 skip convolution on the RAW block input (layers.py:46,74): 8-channel steps dealt round-robin ----
  if constexpr (SKIP) {
    const int Cs2 = a.s2C0 + a.s2C1, ngs = Cs2 >> 3;
    const f32x4 *ws = reinterpret_cast<const f32x4 *>(a.s2w) + (size_t)nt * ngs * 64 + lane;
    for (int gs = wave; gs < ngs; gs += 8) {
      const int c = 8 * gs + 4 * hh;
      const bool s0 = c < a.s2C0;
      const float *sp = s0 ? a.s2src0 + (size_t)b * V * a.s2C0 + c : a.s2src1 + (size_t)b * V * a.s2C1 + (c - a.s2C0);
      const int Cs = s0 ? a.s2C0 : a.s2C1;
      const f32x4 w4 = ws[(size_t)gs * 64];
      f32x4 av[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int pz = i / MBP, j = i % MBP;
        const int v = min(j * 32 + r, PV - 1);
        av[i] = *reinterpret_cast<const f32x4 *>(sp + (size_t)(pz * PV + v) * Cs);
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][jj], w4[jj], acc[i], 0, 0, 0);
    }
  }
  // ---- sum the 8 waves' partial accumulators in wave order through LDS -------------------------------------------------
  __syncthreads();                                // every wave is done reading A: reuse it as P[8 waves][MB][4][64] float4
  f32x4 *P = reinterpret_cast<f32x4 *>(A);
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      P[((size_t)(wave * MB + i) * 4 + q) * 64 + lane] = f32x4{acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
  __syncthreads();
  constexpr int NPW = MB * 4 / 8;                 // (block, register quad) pairs per wave: 1 (MB = 2) or 2 (MB = 4)
  float val[NPW][4];
  int oidx[NPW][4];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pi = wave * NPW + i, mb = pi >> 2, q = pi & 3;
    f32x4 s = P[((size_t)(0 * MB + mb) * 4 + q) * 64 + lane];
#pragma unroll
    for (int ws2 = 1; ws2 < 8; ++ws2) s += P[((size_t)(ws2 * MB + mb) * 4 + q) * 64 + lane];
    const int pz = mb / MBP, j = mb % MBP;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int v = j * 32 + 8 * q + 4 * hh + e;   // accumulator register 4 q + e of lane half hh: row 8 q + 4 hh + e of the block
      oidx[i][e] = v < PV ? pz * PV + v : -1;
      val[i][e] = s[e] + bias_pre + tv_pre;
    }
  }
  if (a.resid) {
    const float *rp = a.resid + (size_t)b * V * a.res_cs + n;
#pragma unroll
    for (int i = 0; i < NPW; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) val[i][e] += rp[(size_t)(oidx[i][e] >= 0 ? oidx[i][e] : 0) * a.res_cs];
  }
  {
    float *op = a.out + (size_t)b * V * a.out_cs + n;
#pragma unroll
    for (int i = 0; i < NPW; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (oidx[i][e] >= 0) op[(size_t)oidx[i][e] * a.out_cs] = val[i][e];
  }
  // ---- GroupNorm statistics of the output: per (slot = row block, channel) mean and M2 over the block's valid rows ----
  if (a.stat_part) {
    float *red1 = red, *red2 = red + MB * 8 * 32;   // [MB][8 = (q, hh)][32 channels]
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pi = wave * NPW + i, mb = pi >> 2, q = pi & 3;
      float s1 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (oidx[i][e] >= 0) s1 += val[i][e];
      red1[(mb * 8 + 2 * q + hh) * 32 + r] = s1;
    }
    __syncthreads();
    float mean[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pi = wave * NPW + i, mb = pi >> 2, q = pi & 3, j = mb % MBP;
      const float cnt = (float)max(0, min(32, PV - 32 * j));
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red1[(mb * 8 + k) * 32 + r];
      mean[i] = cnt > 0.f ? t / cnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (oidx[i][e] >= 0) { const float d = val[i][e] - mean[i]; m2 += d * d; }
      red2[(mb * 8 + 2 * q + hh) * 32 + r] = m2;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pi = wave * NPW + i, mb = pi >> 2, q = pi & 3, j = mb % MBP;
      if (q == 0 && hh == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red2[(mb * 8 + k) * 32 + r];
        float *sp2 = a.stat_part + (((size_t)b * MB + mb) * a.stat_C + n) * 2;
        sp2[0] = mean[i];
        sp2[1] = t;
        if (r == 0 && nt == 0) a.stat_cnt[(size_t)b * MB + mb] = (float)max(0, min(32, PV - 32 * j));
      }
    }
  }
}

// ---- v2: eight independent wave pipelines (Ci a multiple of 64, 8 GroupNorm groups) --------------------------------------
// v1 above runs finalise -> stage everything -> matrix phase -> reduce in lock-step over the whole chip (one workgroup per
// CU, one round): for the 128 -> 128 layers 29 us against 15 us of matrix instructions.  With K split over the waves in
// contiguous channel ranges, wave w only ever READS the channels [w Ci/8, (w+1) Ci/8) of the staged input -- and with the
// reference's 8 GroupNorm groups that range is exactly GroupNorm group w.  So every wave is a pipeline of its own:
//   finalise ITS group's statistics (lanes = the group's channels, merged redundantly by every lane through a wave-private
//   LDS row) -> per 8-channel step: halo loads of step s + 1 in flight (registers) | 9 taps x 16 MBP matrix instructions
//   on step s | GroupNorm + SiLU of step s + 1 -> wave-private LDS slice (double-buffered, row stride 12 floats)
// with no workgroup barrier between the table set-up and the final reduction: LDS operations of one wave execute in order,
// so a slice written by the wave is visible to its own later reads.  The two waves of a SIMD de-phase by themselves: one
// stages / waits for memory while the other multiplies.
// B6: fp32 products from exact three-way bf16 splits (cm_conv_ups.hip explains the arithmetic).  A step is 16 channels (the
// wave's range padded with zeros to whole steps), its slice holds three bf16 planes per voxel (single-buffered: the waves
// have one or two steps), a tap is 4 (plane, z tap) pairs x 6 v_mfma_f32_32x32x16_bf16.  The fused skip conv stays fp32.
template <int MBP, bool SKIP, int B6 = 0>         // B6: 0 fp32 matrix instruction, 1 six bf16 cross terms, 2 three (relaxed plan), 3 three f16 cross terms (h2)
__global__ __launch_bounds__(512, 2) void conv_qr2_kernel(const QrArgs a) {
  constexpr int MB = 2 * MBP;
  constexpr int SS = B6 ? 28 : 12;                 // slice row stride in floats: 8 channels + 4 pad (conflict-free 16-byte reads); B6: 3 x 8 + 4
  constexpr int NBUF = B6 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.x, nt = blockIdx.y;
  const int Ci = a.C0 + a.C1, cw = Ci >> 3;         // channels per wave = per GroupNorm group
  const int Y = a.Y, X = a.X, PV = Y * X, V = 2 * PV;
  const int HX = X + 2, HYX = (Y + 2) * HX, HV = 2 * HYX;
  const int HVp = (HV + 3) & ~3;
  // LDS: hvinfo [HVp] | gst [8 waves][2][32] | red [2][MB][8][32] | slices [8 waves][2][HVp][SS]  (later: P)
  int *hvinfo = reinterpret_cast<int *>(lds);
  float *gst = lds + HVp;
  float *red = gst + 8 * 64;
  float *slices = red + 2 * MB * 8 * 32;
  float *sl0 = slices + (size_t)wave * NBUF * HVp * SS;

  if (tid < HV) {
    const int pl = tid / HYX, rem = tid - pl * HYX, hy = rem / HX, hx = rem - hy * HX;
    const bool in = hy >= 1 && hy <= Y && hx >= 1 && hx <= X;
    hvinfo[tid] = in ? (pl * Y + (hy - 1)) * X + (hx - 1) : -1;
  }
  // ---- this wave's GroupNorm group: per-channel merge of the producers' slots (lane = channel of the group) ----------
  const int c0w = wave * cw;
  const bool norm = a.gamma != nullptr;            // false: the source enters as it is (data gradients of the training step)
  if (norm && !(CM_QR2_ABL & 16)) {
    float M = 0.f, S2 = 0.f;
    if (lane < cw) {
      const int c = c0w + lane;
      const float *p, *nn;
      int Cx, cc, ns;
      if (c < a.C0) { p = a.part0; nn = a.cnt0; Cx = a.C0; cc = c; ns = a.ns0; } else { p = a.part1; nn = a.cnt1; Cx = a.C1; cc = c - a.C0; ns = a.ns1; }
      float N = 0.f;
      for (int s4 = 0; s4 < ns; s4 += 4) {          // four slots per round trip (the loop paid one per slot: 2-4 slots here)
        float2 q[4];
        float cn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int sl = s4 + u < ns ? s4 + u : ns - 1;
          q[u] = *reinterpret_cast<const float2 *>(p + (((size_t)b * ns + sl) * Cx + cc) * 2);
          cn[u] = s4 + u < ns ? nn[(size_t)b * ns + sl] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) chan_combine_q(N, M, S2, cn[u], q[u].x, q[u].y);
      }
      gst[wave * 64 + lane] = M;
      gst[wave * 64 + 32 + lane] = S2;
    }
  }
  __syncthreads();                                // hvinfo (all waves) and the wave's own gst rows
  float gmean = 0.f, grstd = 1.f;                   // this wave's group statistics (merged below, under the first halo loads)
  // ---- staging items of a step: (halo voxel, channel quad of the step's 8 channels); item = lane + 64 k --------------------
  constexpr int NIT = B6 ? 9 : 6;                  // items per lane: 2 HV / 64 <= 6 (HV <= 192: conv_qr2_ok); B6: 4 HV / 64 <= 9
  constexpr int QSH = B6 ? 2 : 1;                  // log2 channel quads per step
  const int nit = ((HV << QSH) + 63) >> 6;
  int ioff[NIT];                                     // source voxel offset (-1: padding / beyond the box)
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int it = lane + 64 * k, hv = it >> QSH;
    ioff[k] = (k < nit && hv < HV) ? hvinfo[hv] : -1;
  }
  const int qq = lane & ((1 << QSH) - 1);           // (64 k is a multiple of the quad count: the quad of an item does not depend on k)
  const int nsteps = B6 ? (cw + 15) >> 4 : cw >> 3;
  bool cq_ok = true;                                // B6: this lane's quad of the step lies inside the wave's channel range
  f32x4 ld[NIT], ldg = {1.f, 1.f, 1.f, 1.f}, ldb = {0.f, 0.f, 0.f, 0.f};   // halo loads of the next step and its affine rows, in flight
  f32x4 ldp = {1.f, 1.f, 1.f, 1.f};                   // Dropout3d multipliers of the step's channels (training forward)
  auto issue = [&](int s) {
    const int cl = (B6 ? 16 : 8) * s + 4 * qq;
    cq_ok = cl < cw;
    const int c = c0w + (cq_ok ? cl : 0);
    if (norm) {
      ldg = *reinterpret_cast<const f32x4 *>(a.gamma + c);
      ldb = *reinterpret_cast<const f32x4 *>(a.beta + c);
    }
    if (a.pm) ldp = *reinterpret_cast<const f32x4 *>(a.pm + (size_t)b * a.pm_stride + c);
    const bool from0 = c < a.C0;
    const float *sp = from0 ? a.src0 + (size_t)b * V * a.C0 + c : a.src1 + (size_t)b * V * a.C1 + (c - a.C0);
    const int Cs = from0 ? a.C0 : a.C1;
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if (k < nit) ld[k] = *reinterpret_cast<const f32x4 *>(sp + (size_t)(ioff[k] >= 0 ? ioff[k] : 0) * Cs);
  };
  auto stage = [&](int s) {
    const f32x4 sc = ldg * grstd, sh = ldb - gmean * sc;
    float *dst = sl0 + (size_t)(B6 ? 0 : (s & 1)) * HVp * SS + (B6 ? 2 : 4) * qq;
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if (k < nit) {
        const int it = lane + 64 * k, hv = it >> QSH;
        f32x4 w = ld[k];
        if (norm && !(CM_QR2_ABL & 4)) w = w * sc + sh;
        if (a.silu && !(CM_QR2_ABL & 4)) { w[0] = silu_q(w[0]); w[1] = silu_q(w[1]); w[2] = silu_q(w[2]); w[3] = silu_q(w[3]); }
        if (a.pm) w = w * ldp;
        if (ioff[k] < 0 || !cq_ok) w = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (B6) {
          if (hv < HV) {
            constexpr int NTW = B6 >= 2 ? 2 : 3;
            cm_u32x2_t t3[3];
            if constexpr (B6 == 3) cm_split2_f16(w, t3);   // h2: f16 hi / mid (bounded input: GroupNorm + SiLU output)
            else cm_split3_bf16<NTW>(w, t3);        // hi / mid / lo planes, exact remainders
#pragma unroll
            for (int tm = 0; tm < NTW; ++tm) *reinterpret_cast<cm_u32x2_t *>(dst + (size_t)hv * SS + 8 * tm) = t3[tm];
          }
        } else {
          if (hv < HV) *reinterpret_cast<f32x4 *>(dst + (size_t)hv * SS) = w;
        }
      }
  };
  issue(0);
  // every lane merges the group's channels in channel order (same chain on all lanes: no cross-lane traffic) -- AFTER the halo
  // loads of the first step have been requested: the 16-channel chain runs under their round trip
  if (norm && !(CM_QR2_ABL & 16)) {
    float N = 0.f, M = 0.f, S2 = 0.f;
    for (int i = 0; i < cw; ++i) chan_combine_q(N, M, S2, (float)V, gst[wave * 64 + i], gst[wave * 64 + 32 + i]);
    gmean = M;
    grstd = rsqrtf(S2 / N + a.eps);
  }
  const int n = nt * 32 + r;
  const float bias_pre = a.bias[n];
  const float tv_pre = a.temb ? a.temb[(size_t)a.tidx[b] * a.temb_stride + n] : 0.f;
  int abase[MBP];
#pragma unroll
  for (int j = 0; j < MBP; ++j) {
    const int v = min(j * 32 + r, PV - 1);
    const int y = v / X, x = v - y * X;
    abase[j] = (y * HX + x) * SS + 4 * hh;
  }
  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  if constexpr (B6) {
    // split fragments [n tile][wave][step][tap 9][dz][term][lane] 16 B (pack_qr_b6)
    const int ngw = nsteps * 9;
    const f32x4 *wq = reinterpret_cast<const f32x4 *>(a.wq6) + ((size_t)(nt * 8 + wave) * ngw) * 9 * 64 + lane;
#ifndef CM_QR2_RD
#define CM_QR2_RD 1
#endif
    // (must divide 9: a slot is tap % RD in every step.  Six terms: 3 taps would be 108 registers and spill.  h2 loads two of the
    //  three term slots -- 72 registers, 213 in all -- and three taps of read-ahead are worth 9 of the ten launches' 197 us)
    constexpr int RD = MBP == 1 ? (B6 == 3 ? 3 : CM_QR2_RD) : 1;
    f32x4 bw[RD][3][3];
#pragma unroll
    for (int t = 0; t < RD; ++t)
      if (t < ngw) {
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
          for (int tm = 0; tm < 3; ++tm) bw[t][dz][tm] = wq[(((size_t)t * 3 + dz) * 3 + tm) * 64];
      }
    stage(0);
    if (nsteps > 1) issue(1);
    for (int s = 0; s < nsteps; ++s) {
      const float *Asl = sl0;
      f32x4 a0[2][MBP][3], a1[2][MBP][3];             // (current | next tap) x row block x term
#pragma unroll
      for (int j = 0; j < MBP; ++j)
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) {
          a0[0][j][tm] = *reinterpret_cast<const f32x4 *>(Asl + abase[j] + 8 * tm);
          a1[0][j][tm] = *reinterpret_cast<const f32x4 *>(Asl + abase[j] + HYX * SS + 8 * tm);
        }
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int gl = s * 9 + t9;
        if (t9 + 1 < 9) {
          const int dy = (t9 + 1) / 3, dx = (t9 + 1) - 3 * dy;
          const int toff = (dy * HX + dx) * SS;
#pragma unroll
          for (int j = 0; j < MBP; ++j)
#pragma unroll
            for (int tm = 0; tm < 3; ++tm) {
              a0[(t9 + 1) & 1][j][tm] = *reinterpret_cast<const f32x4 *>(Asl + abase[j] + toff + 8 * tm);
              a1[(t9 + 1) & 1][j][tm] = *reinterpret_cast<const f32x4 *>(Asl + abase[j] + toff + HYX * SS + 8 * tm);
            }
        }
        // (A term, B term), small products first: hi = 0, mid = 1, lo = 2
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
        auto six = [&](f32x16 &d, const f32x4 (&av)[3], const f32x4 (&wv)[3]) {
#pragma unroll
          for (int u = (B6 >= 2 ? 3 : 0); u < ((CM_QR2_ABL & 1) ? (B6 >= 2 ? 4 : 1) : 6); ++u) {
            if constexpr (B6 == 3)
              d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8q, av[TA[u]]), __builtin_bit_cast(f16x8q, wv[TB[u]]), d, 0, 0, 0);
            else
              d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8q, av[TA[u]]), __builtin_bit_cast(bf16x8q, wv[TB[u]]), d, 0, 0, 0);
          }
        };
#pragma unroll
        for (int j = 0; j < MBP; ++j) {
          six(acc[j], a0[t9 & 1][j], bw[t9 % RD][1]);
          six(acc[MBP + j], a0[t9 & 1][j], bw[t9 % RD][0]);
          six(acc[j], a1[t9 & 1][j], bw[t9 % RD][2]);
          six(acc[MBP + j], a1[t9 & 1][j], bw[t9 % RD][1]);
        }
        if (gl + RD < ngw && !(CM_QR2_ABL & 2)) {
#pragma unroll
          for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int tm = 0; tm < 3; ++tm) bw[t9 % RD][dz][tm] = wq[(((size_t)(gl + RD) * 3 + dz) * 3 + tm) * 64];
        }
        asm volatile("" ::: "memory");
      }
      if (s + 1 < nsteps) {
        stage(s + 1);
        if (s + 2 < nsteps) issue(s + 2);
      }
    }
  } else {
  const int K8 = Ci >> 3, ng = 9 * K8;
  const f32x4 *wq = reinterpret_cast<const f32x4 *>(a.wq) + ((size_t)nt * ng + (size_t)wave * nsteps * 9) * 3 * 64 + lane;
  // weight ring: RD in-plane taps deep (9 = a whole step ahead where the registers allow it); slot = tap % RD is static, a
  // slot is refilled with the group RD taps later right after its use.  One tap ahead (1 k matrix cycles) was less than the
  // L2 latency under load: the waves stalled on their weights at every tap (28.6 us for 15 us of matrix instructions).
  constexpr int RD = MBP == 1 ? 9 : 3;
  const int ngw = nsteps * 9;                      // groups of this wave
  f32x4 bw[RD][3];
#pragma unroll
  for (int t = 0; t < RD; ++t)
    if (t < ngw) {
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) bw[t][dz] = wq[((size_t)t * 3 + dz) * 64];
    }
  stage(0);
  if (nsteps > 1) issue(1);

  for (int s = 0; s < nsteps; ++s) {
    const float *Asl = sl0 + (size_t)(s & 1) * HVp * SS;
    // 9 in-plane taps of this 8-channel step; A fragments are read one tap ahead, weights RD taps ahead
    f32x4 a0[MBP], a1[MBP], a0n[MBP], a1n[MBP];
#pragma unroll
    for (int j = 0; j < MBP; ++j) {
      a0[j] = *reinterpret_cast<const f32x4 *>(Asl + abase[j]);
      a1[j] = *reinterpret_cast<const f32x4 *>(Asl + abase[j] + HYX * SS);
    }
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int gl = s * 9 + t9;                    // group index within this wave's range
      if (t9 + 1 < 9) {
        const int dy = (t9 + 1) / 3, dx = (t9 + 1) - 3 * dy;
        const int toff = (dy * HX + dx) * SS;
#pragma unroll
        for (int j = 0; j < MBP; ++j) {
          a0n[j] = *reinterpret_cast<const f32x4 *>(Asl + abase[j] + toff);
          a1n[j] = *reinterpret_cast<const f32x4 *>(Asl + abase[j] + toff + HYX * SS);
        }
      }
      const f32x4 (&w3)[3] = bw[t9 % RD];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int j = 0; j < MBP; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j][jj], w3[1][jj], acc[j], 0, 0, 0);
          acc[MBP + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j][jj], w3[0][jj], acc[MBP + j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j][jj], w3[2][jj], acc[j], 0, 0, 0);
          acc[MBP + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j][jj], w3[1][jj], acc[MBP + j], 0, 0, 0);
        }
      // refill this slot with the group RD taps later, AFTER the matrix instructions that read it (no register copies: a copy
      // of the slot per tap was 12 vector moves that the fp32 matrix pipe does not hide)
      if (gl + RD < ngw) {
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) bw[t9 % RD][dz] = wq[((size_t)(gl + RD) * 3 + dz) * 64];
      }
      asm volatile("" ::: "memory");                // (keeps the refill here: the compiler would sink it to its first use)
#pragma unroll
      for (int j = 0; j < MBP; ++j) { a0[j] = a0n[j]; a1[j] = a1n[j]; }
    }
    if (s + 1 < nsteps) {
      stage(s + 1);
      if (s + 2 < nsteps) issue(s + 2);
    }
  }
  }
  if constexpr (SKIP) {
    const int Cs2 = a.s2C0 + a.s2C1, ngs = Cs2 >> 3;
    const f32x4 *ws = reinterpret_cast<const f32x4 *>(a.s2w) + (size_t)nt * ngs * 64 + lane;
    for (int gs = wave; gs < ngs; gs += 8) {
      const int c = 8 * gs + 4 * hh;
      const bool s0 = c < a.s2C0;
      const float *sp = s0 ? a.s2src0 + (size_t)b * V * a.s2C0 + c : a.s2src1 + (size_t)b * V * a.s2C1 + (c - a.s2C0);
      const int Cs = s0 ? a.s2C0 : a.s2C1;
      f32x4 w4 = ws[(size_t)gs * 64];
      if constexpr (B6 == 3) w4 = w4 * (1.0f / a.h2_oscale);   // (the main accumulators hold 2^k times the sum: same scale, exact)
      f32x4 av[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int pz = i / MBP, j = i % MBP;
        const int v = min(j * 32 + r, PV - 1);
        av[i] = *reinterpret_cast<const f32x4 *>(sp + (size_t)(pz * PV + v) * Cs);
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][jj], w4[jj], acc[i], 0, 0, 0);
    }
  }
  // ---- sum the 8 waves' partial accumulators in wave order through LDS (as v1) ----------------------------------------
  if ((CM_QR2_ABL & 8) != 0) {                      // (ablation: no reduction / epilogue; one store keeps the accumulators alive)
    float sg = 0.f;
#pragma unroll
    for (int i = 0; i < MB; ++i) sg += acc[i][0] + acc[i][9];
    if (sg == 123.456f) a.out[0] = sg;
    return;
  }
  __syncthreads();
  f32x4 *P = reinterpret_cast<f32x4 *>(slices);
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      P[((size_t)(wave * MB + i) * 4 + q) * 64 + lane] = f32x4{acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
  __syncthreads();
  constexpr int NPW = MB * 4 / 8;
  float val[NPW][4];
  int oidx[NPW][4];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pi = wave * NPW + i, mb = pi >> 2, q = pi & 3;
    f32x4 s = P[((size_t)(0 * MB + mb) * 4 + q) * 64 + lane];
#pragma unroll
    for (int ws2 = 1; ws2 < 8; ++ws2) s += P[((size_t)(ws2 * MB + mb) * 4 + q) * 64 + lane];
    const int pz = mb / MBP, j = mb % MBP;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int v = j * 32 + 8 * q + 4 * hh + e;
      oidx[i][e] = v < PV ? pz * PV + v : -1;
      val[i][e] = (B6 == 3 ? s[e] * a.h2_oscale : s[e]) + bias_pre + tv_pre;
    }
  }
  if (a.resid) {
    const float *rp = a.resid + (size_t)b * V * a.res_cs + n;
#pragma unroll
    for (int i = 0; i < NPW; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) val[i][e] += rp[(size_t)(oidx[i][e] >= 0 ? oidx[i][e] : 0) * a.res_cs];
  }
  {
    float *op = a.out + (size_t)b * V * a.out_cs + n;
#pragma unroll
    for (int i = 0; i < NPW; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (oidx[i][e] >= 0) op[(size_t)oidx[i][e] * a.out_cs] = val[i][e];
  }
  if (a.stat_part) {
    float *red1 = red, *red2 = red + MB * 8 * 32;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pi = wave * NPW + i, mb = pi >> 2, q = pi & 3;
      float s1 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (oidx[i][e] >= 0) s1 += val[i][e];
      red1[(mb * 8 + 2 * q + hh) * 32 + r] = s1;
    }
    __syncthreads();
    float mean[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pi = wave * NPW + i, mb = pi >> 2, q = pi & 3, j = mb % MBP;
      const float cnt = (float)max(0, min(32, PV - 32 * j));
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red1[(mb * 8 + k) * 32 + r];
      mean[i] = cnt > 0.f ? t / cnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (oidx[i][e] >= 0) { const float d = val[i][e] - mean[i]; m2 += d * d; }
      red2[(mb * 8 + 2 * q + hh) * 32 + r] = m2;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pi = wave * NPW + i, mb = pi >> 2, q = pi & 3, j = mb % MBP;
      if (q == 0 && hh == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red2[(mb * 8 + k) * 32 + r];
        float *sp2 = a.stat_part + (((size_t)b * MB + mb) * a.stat_C + n) * 2;
        sp2[0] = mean[i];
        sp2[1] = t;
        if (r == 0 && nt == 0) a.stat_cnt[(size_t)b * MB + mb] = (float)max(0, min(32, PV - 32 * j));
      }
    }
  }
}

// split fragments of the six-term form from the fp32 fragments (pack_qr order [n tile][g = k8 * 9 + t9][dz][lane][4]) into
// pack_qr_b6 order; one thread per weight.  Padding slots of the last step of a wave stay zero (never written).
__global__ __launch_bounds__(256) void qr_b6_repack_kernel(const float *__restrict__ wq, unsigned short *__restrict__ w6, long long n, int Ci) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ng = 9 * (Ci >> 3), cw = Ci >> 3, nsw = (cw + 15) >> 4;
  const int jj = (int)(i & 3), lane = (int)((i >> 2) & 63);
  long long q = i >> 8;
  const int dz = (int)(q % 3); q /= 3;
  const int g = (int)(q % ng);
  const int nt = (int)(q / ng);
  const int k8 = g / 9, t9 = g - 9 * k8, ci = 8 * k8 + 4 * (lane >> 5) + jj, r = lane & 31;
  const int wv = ci / cw, cl = ci - wv * cw, st = cl >> 4, hd = (cl >> 3) & 1, j = cl & 7;
  unsigned short *dst = w6 + ((((((size_t)(nt * 8 + wv) * nsw + st) * 9 + t9) * 3 + dz) * 3) * 64 + 32 * hd + r) * 8 + j;
  float rem = wq[i];
#pragma unroll
  for (int tm = 0; tm < 3; ++tm) {
    const __bf16 hb = (__bf16)rem;
    dst[(size_t)tm * 64 * 8] = __builtin_bit_cast(unsigned short, hb);
    rem -= (float)hb;
  }
}

hipError_t launch_qr_b6_repack(const float *wq, float *wq6, long long n_floats, int Ci, hipStream_t st) {
  if (n_floats <= 0 || Ci % 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(qr_b6_repack_kernel, dim3((unsigned)((n_floats + 255) / 256)), dim3(256), 0, st, wq, reinterpret_cast<unsigned short *>(wq6),
                     n_floats, Ci);
  return hipGetLastError();
}

size_t conv_qr2_lds(const QrArgs &a, int MBP, bool b6) {
  const int HV = 2 * (a.Y + 2) * (a.X + 2), HVp = (HV + 3) & ~3, MB = 2 * MBP;
  const size_t head = (size_t)HVp + 8 * 64 + 2 * MB * 8 * 32;
  const size_t sl = b6 ? (size_t)8 * HVp * 28 : (size_t)8 * 2 * HVp * 12, pbuf = (size_t)8 * MB * 4 * 64 * 4;
  return (head + std::max(sl, pbuf)) * sizeof(float);
}

// v2 applies when the K ranges of the 8 waves are the 8 GroupNorm groups in whole 8-channel steps
bool conv_qr2_ok(const QrArgs &a) {
  const int Ci = a.C0 + a.C1;
  return (a.gamma || a.raw) && a.groups == 8 && Ci % 64 == 0 && Ci / 8 <= 32 && a.C0 % 4 == 0 && a.C1 % 4 == 0 && 2 * (a.Y + 2) * (a.X + 2) <= 192 &&
         conv_qr2_lds(a, a.Y * a.X > 32 ? 2 : 1, false) <= 160 * 1024;
}

// the six-term bf16 form of v2: split fragments present, <= 9 staging items per lane, its single-buffered slices fit
bool conv_qr2_b6_ok(const QrArgs &a) {
  // (two row blocks per plane -- HERMES-CR-120's 42-voxel planes -- take it as well: 247 registers, no spill; 2.15 -> 1.98 ms/step there)
  return a.wq6 && conv_qr2_ok(a) && 4 * 2 * (a.Y + 2) * (a.X + 2) <= 64 * 9 &&
         conv_qr2_lds(a, a.Y * a.X > 32 ? 2 : 1, true) <= 160 * 1024;
}

size_t conv_qr_lds(const QrArgs &a, int MBP) {
  const int Ci = a.C0 + a.C1, HV = 2 * (a.Y + 2) * (a.X + 2), MB = 2 * MBP;
  const size_t head = (size_t)2 * Ci + 2 * Ci + 32 + ((HV + 3) & ~3) + 2 * MB * 8 * 32;
  const size_t amat = (size_t)HV * (Ci + 4), pbuf = (size_t)8 * MB * 4 * 64 * 4;
  return (head + std::max(amat, pbuf)) * sizeof(float);
}

bool conv_qr_ok(const QrArgs &a) {
  const int Ci = a.C0 + a.C1, PV = a.Y * a.X;
  if (PV < 1 || PV > 64 || Ci % 8 || a.C0 % 4 || a.C1 % 4 || Ci > 512 || a.Co % 32 || a.groups > 16 || (a.gamma && Ci % a.groups)) return false;
  if (a.s2w && ((a.s2C0 + a.s2C1) % 8 || a.s2C0 % 4 || a.s2C1 % 4)) return false;
  if (2 * (a.Y + 2) * (a.X + 2) > 512) return false;
  return conv_qr_lds(a, PV > 32 ? 2 : 1) <= 160 * 1024;
}

hipError_t launch_conv_qr(const QrArgs &a_in, hipStream_t st) {
  QrArgs a = a_in;
  if (!conv_qr_ok(a)) return hipErrorInvalidValue;
  const int Q = (a.C0 + a.C1) >> 2;
  a.qshift = 0;
  while ((1 << a.qshift) < Q) ++a.qshift;
  if (a.qshift > 9) return hipErrorInvalidValue;
  const int MBP = a.Y * a.X > 32 ? 2 : 1;
  static const bool no_v2 = cm::diag_env("CM_NO_QR2") != nullptr;
  const bool v2 = conv_qr2_ok(a) && !no_v2;
  const bool b6 = v2 && conv_qr2_b6_ok(a);
  const size_t lds = v2 ? conv_qr2_lds(a, MBP, b6) : conv_qr_lds(a, MBP);
  const dim3 grid((unsigned)a.B, (unsigned)(a.Co / 32));
#define CM_QR_GO(KERNEL)                                                                            \
  {                                                                                                 \
    static bool attr_set[64] = {false};                                                             \
    int dev = 0;                                                                                    \
    (void)hipGetDevice(&dev);                                                                       \
    if (!attr_set[dev & 63]) {                                                                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return e;                                                                \
      attr_set[dev & 63] = true;                                                                    \
    }                                                                                               \
    hipLaunchKernelGGL(KERNEL, grid, dim3(512), lds, st, a);                                        \
    return hipGetLastError();                                                                       \
  }
  if (b6 && a.three == 2) {                           // default plan: h2 fragments, three f16 cross terms
    if (MBP == 1) {
      if (a.s2w) CM_QR_GO((conv_qr2_kernel<1, true, 3>))
      CM_QR_GO((conv_qr2_kernel<1, false, 3>))
    }
    if (a.s2w) CM_QR_GO((conv_qr2_kernel<2, true, 3>))
    CM_QR_GO((conv_qr2_kernel<2, false, 3>))
  }
  if (b6 && a.three) {                                // relaxed plan: three cross terms on the same fragments
    if (MBP == 1) {
      if (a.s2w) CM_QR_GO((conv_qr2_kernel<1, true, 2>))
      CM_QR_GO((conv_qr2_kernel<1, false, 2>))
    }
    if (a.s2w) CM_QR_GO((conv_qr2_kernel<2, true, 2>))
    CM_QR_GO((conv_qr2_kernel<2, false, 2>))
  }
  if (b6) {
    if (MBP == 1) {
      if (a.s2w) CM_QR_GO((conv_qr2_kernel<1, true, 1>))
      CM_QR_GO((conv_qr2_kernel<1, false, 1>))
    }
    if (a.s2w) CM_QR_GO((conv_qr2_kernel<2, true, 1>))
    CM_QR_GO((conv_qr2_kernel<2, false, 1>))
  }
  if (v2) {
    if (MBP == 1) {
      if (a.s2w) CM_QR_GO((conv_qr2_kernel<1, true>))
      CM_QR_GO((conv_qr2_kernel<1, false>))
    }
    if (a.s2w) CM_QR_GO((conv_qr2_kernel<2, true>))
    CM_QR_GO((conv_qr2_kernel<2, false>))
  }
  if (MBP == 1) {
    if (a.s2w) CM_QR_GO((conv_qr_kernel<1, true>))
    CM_QR_GO((conv_qr_kernel<1, false>))
  }
  if (a.s2w) CM_QR_GO((conv_qr_kernel<2, true>))
  CM_QR_GO((conv_qr_kernel<2, false>))
#undef CM_QR_GO
}

}  // namespace cm
