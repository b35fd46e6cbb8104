// Small HBM-/latency-bound kernels around the convolutions: GroupNorm statistics,
// layout changes at the ABI edge, the time-embedding MLP, the attention core and
// the sampler updates.  gfx950 only (64-wide wavefronts).
#include "cm_kernels.h"

namespace cm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

// Chan et al. pairwise combination of (n, mean, M2) triples.
__device__ __forceinline__ void chan_combine(float &n, float &mean, float &m2, float nb, float meanb, float m2b) {
  if (nb == 0.f) return;
  const float nt = n + nb;
  const float d = meanb - mean;
  const float f = nb * __builtin_amdgcn_rcpf(nt);  // hardware reciprocal (1 ulp) instead of an IEEE division
  mean += d * f;
  m2 += m2b + d * d * n * f;
  n = nt;
}

// --------------------------------------------------------------------------------
// Per-channel statistics of a channels-last tensor x[B][V][C] (GroupNorm stats are
// assembled from them in gn_finalize so that any channel grouping -- including
// groups that straddle a torch.cat boundary, unet.py:160 + layers.py:30 -- works).
// grid (nslice, B), 256 threads; thread = (channel quad, voxel lane).
// Each thread runs a shifted two-moment accumulation (pivot = its first value),
// threads are merged with Chan's formula through LDS in a fixed order.
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chan_stats_kernel(const float *__restrict__ x, int V, int C, int nslice,
                                                         float *__restrict__ part, float *__restrict__ cnt) {
  __shared__ float sh[256 * 4 * 3];
  const int b = blockIdx.y, sl = blockIdx.x;
  const int Q = C >> 2;                 // channel quads
  const int tid = threadIdx.x;
  const int vs = (V + nslice - 1) / nslice;
  const int vbeg = sl * vs, vend = min(V, vbeg + vs);
  const int nvl = 256 / Q > 0 ? 256 / Q : 1;  // voxel lanes per quad
  const int q = tid % Q, vl = tid / Q;
  float n = 0.f;
  f32x4 piv = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  if (vl < nvl && Q <= 256) {
    const float *xb = x + ((size_t)b * V) * C + 4 * q;
    bool first = true;
    for (int v = vbeg + vl; v < vend; v += nvl) {
      const f32x4 val = *reinterpret_cast<const f32x4 *>(xb + (size_t)v * C);
      if (first) { piv = val; first = false; }
      const f32x4 d = val - piv;
      s1 += d;
      s2 += d * d;
      n += 1.f;
    }
  }
  // to (n, mean, M2) per component
  float mean[4], m2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (n > 0.f) {
      const float md = s1[i] / n;
      mean[i] = piv[i] + md;
      m2[i] = s2[i] - s1[i] * md;
      if (m2[i] < 0.f) m2[i] = 0.f;
    } else { mean[i] = 0.f; m2[i] = 0.f; }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sh[(tid * 4 + i) * 3 + 0] = n;
    sh[(tid * 4 + i) * 3 + 1] = mean[i];
    sh[(tid * 4 + i) * 3 + 2] = m2[i];
  }
  __syncthreads();
  // one thread per channel merges the voxel lanes in order
  for (int c = tid; c < C; c += 256) {
    const int qq = c >> 2, i = c & 3;
    float N = 0.f, M = 0.f, S2 = 0.f;
    for (int l = 0; l < nvl; ++l) {
      const int t = l * Q + qq;
      if (t < 256) chan_combine(N, M, S2, sh[(t * 4 + i) * 3 + 0], sh[(t * 4 + i) * 3 + 1], sh[(t * 4 + i) * 3 + 2]);
    }
    float *p = part + (((size_t)b * nslice + sl) * C + c) * 2;
    p[0] = M;
    p[1] = S2;
  }
  if (tid == 0) cnt[(size_t)b * nslice + sl] = (float)max(0, vend - vbeg);
}

hipError_t launch_chan_stats(const float *x, int B, int V, int C, int nslice, float *part, float *cnt,
                             hipStream_t st) {
  if (C % 4 != 0 || C / 4 > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(chan_stats_kernel, dim3(nslice, B), dim3(256), 0, st, x, V, C, nslice, part, cnt);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// GroupNorm finalisation: nn.GroupNorm(8, C) statistics (biased variance, eps)
// over the channel-concatenation of up to two tensors (layers.py:30,41,9; unet.py:119)
// folded with the affine into one scale/shift pair per (sample, channel).
// grid B, 1024 threads: a full-resolution tensor arrives as ~100 slots per channel, and the merge
// of one channel's slots is a serial chain, so the slots are spread over 1024 / Ct lanes and
// their loads are issued four at a time.
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const float *__restrict__ p0, const float *__restrict__ n0,
                                                          int ns0, int C0, const float *__restrict__ p1,
                                                          const float *__restrict__ n1, int ns1, int C1, int V,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, int groups, float eps,
                                                          float *__restrict__ gn, float *__restrict__ mr) {
  // sm: [nl][Ct][3] partial triples, then [Ct] mean, [Ct] m2, [groups] gmean, [groups] grstd
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Ct = C0 + C1;
  const int nl = max(1, 1024 / Ct);  // slot lanes per channel
  float *tri = sm;
  float *cmean = sm + (size_t)nl * Ct * 3, *cm2 = cmean + Ct, *gmean = cm2 + Ct, *grstd = gmean + groups;
  // stage 1: thread (channel c, lane j) merges the slots j, j+nl, ... of its channel in order
  for (int idx = tid; idx < nl * Ct; idx += 1024) {
    const int c = idx % Ct, j = idx / Ct;
    const float *p, *nn;
    int Cx, cc, ns;
    if (c < C0) { p = p0; nn = n0; Cx = C0; cc = c; ns = ns0; } else { p = p1; nn = n1; Cx = C1; cc = c - C0; ns = ns1; }
    float N = 0.f, M = 0.f, S2 = 0.f;
    for (int s0 = j; s0 < ns; s0 += 4 * nl) {
      float cnt[4];
      float2 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = s0 + u * nl;
        const int sc = s < ns ? s : j;
        cnt[u] = s < ns ? nn[(size_t)b * ns + sc] : 0.f;
        q[u] = *reinterpret_cast<const float2 *>(p + (((size_t)b * ns + sc) * Cx + cc) * 2);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + u * nl < ns) chan_combine(N, M, S2, cnt[u], q[u].x, q[u].y);
    }
    tri[(j * Ct + c) * 3 + 0] = N;
    tri[(j * Ct + c) * 3 + 1] = M;
    tri[(j * Ct + c) * 3 + 2] = S2;
  }
  __syncthreads();
  for (int c = tid; c < Ct; c += 1024) {
    float N = 0.f, M = 0.f, S2 = 0.f;
    for (int j = 0; j < nl; ++j) chan_combine(N, M, S2, tri[(j * Ct + c) * 3], tri[(j * Ct + c) * 3 + 1], tri[(j * Ct + c) * 3 + 2]);
    cmean[c] = M;
    cm2[c] = S2;
  }
  __syncthreads();
  const int cg = Ct / groups;
  if (tid < groups) {
    float N = 0.f, M = 0.f, S2 = 0.f;
    for (int i = 0; i < cg; ++i) chan_combine(N, M, S2, (float)V, cmean[tid * cg + i], cm2[tid * cg + i]);
    gmean[tid] = M;
    grstd[tid] = rsqrtf(S2 / N + eps);
  }
  __syncthreads();
  for (int c = tid; c < Ct; c += 1024) {
    const int g = c / cg;
    const float sc = grstd[g] * gamma[c];
    gn[((size_t)b * 2 + 0) * Ct + c] = sc;
    gn[((size_t)b * 2 + 1) * Ct + c] = beta[c] - gmean[g] * sc;
    if (mr) {  // kept for the backward pass: group mean / rstd expanded per channel
      mr[((size_t)b * 2 + 0) * Ct + c] = gmean[g];
      mr[((size_t)b * 2 + 1) * Ct + c] = grstd[g];
    }
  }
}

hipError_t launch_gn_finalize(const float *part0, const float *cnt0, int ns0, int C0, const float *part1,
                              const float *cnt1, int ns1, int C1, int V, const float *gamma, const float *beta,
                              int groups, float eps, float *gn, float *mr, int B, hipStream_t st) {
  const int Ct = C0 + C1;
  if (Ct % groups != 0) return hipErrorInvalidValue;
  const int nl = 1024 / Ct > 0 ? 1024 / Ct : 1;
  const size_t smem = ((size_t)nl * Ct * 3 + 2 * Ct + 2 * groups) * sizeof(float);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(1024), smem, st, part0, cnt0, ns0, C0, part1, cnt1, ns1, C1, V,
                     gamma, beta, groups, eps, gn, mr);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// Second pass of a K-split convolution (tiny-spatial layers, cm_conv.hip ks > 1): sums the
// S raw partial outputs in a fixed order, applies the conv epilogue (bias, time-embedding
// row, residual) and produces the GroupNorm statistics of the result per 32-row slot.
// grid (nslots, B), 1024 threads = C channels x (1024/C) row lanes; C <= 256, so a thread owns at
// most 8 of the slot's 32 rows and keeps all their partial loads in flight together.
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ksplit_combine_kernel(const CombineArgs a) {
  __shared__ float red[1024];
  const int slot = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int nl = 1024 / a.C;                 // row lanes (>= 4)
  const int c = tid % a.C, rl = tid / a.C;
  const bool act = rl < nl;
  const int r0 = slot * 32, r1 = min(a.V, r0 + 32);
  const float *__restrict__ part = a.part;
  const float *__restrict__ resid = a.resid;
  float add = 0.f;
  if (act) {
    add = a.bias[c];
    if (a.temb) add += a.temb[(size_t)a.tidx[b] * a.temb_stride + c];
  }
  constexpr int RPT = 8;                     // rows per thread (32 / nl <= 8)
  float vals[RPT];
  size_t idx[RPT];
  bool ok[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int row = r0 + rl + i * nl;
    ok[i] = act && i * nl < 32 && row < r1;
    idx[i] = ok[i] ? ((size_t)b * a.V + row) * a.C + c : 0;
    vals[i] = 0.f;
  }
  // partial sums in the fixed order s = 0..S-1; the loads of one s for all rows go out together
  float rs[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) rs[i] = (resid && ok[i]) ? resid[(idx[i] / a.C) * a.res_cs + c] : 0.f;
  for (int s0 = 0; s0 < a.S; s0 += 4) {      // four partials per round, all their loads in flight together
    float p[4][RPT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t so = (size_t)min(s0 + u, a.S - 1) * a.stride;
#pragma unroll
      for (int i = 0; i < RPT; ++i) p[u][i] = part[so + idx[i]];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (s0 + u < a.S) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) vals[i] += p[u][i];
      }
  }
  float s1 = 0.f;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    vals[i] = ok[i] ? vals[i] + add + rs[i] : 0.f;
    if (ok[i]) { a.out[idx[i]] = vals[i]; s1 += vals[i]; }
  }
  if (!a.stat_part) return;
  // per-channel statistics of the slot: merge the row lanes through LDS (fixed order)
  red[tid] = s1;
  __syncthreads();
  float tot = 0.f;
  if (act)
    for (int l = 0; l < nl; ++l) tot += red[l * a.C + c];
  const float n = (float)(r1 - r0);
  const float mean = tot / n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < RPT; ++i)
    if (ok[i]) { const float d = vals[i] - mean; q += d * d; }
  __syncthreads();
  red[tid] = q;
  __syncthreads();
  if (act && rl == 0) {
    float m2 = 0.f;
    for (int l = 0; l < nl; ++l) m2 += red[l * a.C + c];
    float *sp = a.stat_part + (((size_t)b * a.nslots + slot) * a.C + c) * 2;
    sp[0] = mean;
    sp[1] = m2;
    if (c == 0) a.stat_cnt[(size_t)b * a.nslots + slot] = n;
  }
}

hipError_t launch_ksplit_combine(const CombineArgs &a, hipStream_t st) {
  if (a.C > 256 || a.C < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ksplit_combine_kernel, dim3(a.nslots, a.B), dim3(1024), 0, st, a);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// K-split second pass + the GroupNorm finalisation of the layer that consumes the result, in one launch: a tiny
// launch costs ~5 us on this part whatever it does, and the quarter-resolution layers paid it twice per conv
// (ksplit_combine + gn_finalize).  grid B, 1024 threads: the workgroup runs the ksplit_combine body for each of the
// sample's (at most two) 32-row slots -- same thread mapping and summation order, so the slot statistics it
// writes are the ones the two-launch path writes -- then merges the slots per channel, adds the channels of the
// concatenated second tensor from its finished partials, and folds group mean / rstd with the affine exactly as
// gn_finalize_kernel does (layers.py:30,41; unet.py:119).
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void combine_gn_kernel(const CombineArgs a) {
  __shared__ float red[2][1024];
  __shared__ float cmean[1024], cm2[1024], gmean[32], grstd[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nl = 1024 / a.C;
  const int c = tid % a.C, rl = tid / a.C;
  const bool act = rl < nl;
  const float *__restrict__ part = a.part;
  const float *__restrict__ resid = a.resid;
  const int Ct = a.C + a.fin_C1;
  // the second tensor's partials (independent of everything below): first four slots in flight from the start
  constexpr int PS = 4;
  float n1v[PS];
  float2 q1v[PS];
  const int cc1 = tid - a.C;
  const bool has1 = tid >= a.C && tid < Ct;
#pragma unroll
  for (int u = 0; u < PS; ++u) {
    const bool o = has1 && u < a.fin_ns1;
    n1v[u] = o ? a.fin_n1[(size_t)b * a.fin_ns1 + u] : 0.f;
    q1v[u] = o ? *reinterpret_cast<const float2 *>(a.fin_p1 + (((size_t)b * a.fin_ns1 + u) * a.fin_C1 + cc1) * 2) : float2{0.f, 0.f};
  }
  float add = 0.f;
  if (act) {
    add = a.bias[c];
    if (a.temb) add += a.temb[(size_t)a.tidx[b] * a.temb_stride + c];
  }
  constexpr int RPT = 8, NS = 2;
  float vals[NS][RPT];
  int idx[NS][RPT];                 // (host checks B * V * C < 2^31)
  bool ok[NS][RPT];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int row = s * 32 + rl + i * nl;
      ok[s][i] = act && s < a.nslots && i * nl < 32 && row < min(a.V, s * 32 + 32);
      idx[s][i] = ok[s][i] ? (b * a.V + row) * a.C + c : 0;
      vals[s][i] = 0.f;
    }
  // residual first, then the partials two at a time: every load of a round is in flight before the first add
  // (one memory round trip per round instead of one per partial); the sum order stays k = 0..S-1
  float rs[NS][RPT];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int i = 0; i < RPT; ++i) rs[s][i] = (resid && ok[s][i]) ? resid[(size_t)(b * a.V + s * 32 + rl + i * nl) * a.res_cs + c] : 0.f;
  for (int k0 = 0; k0 < a.S; k0 += 2) {
    float p[2][NS][RPT];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t ko = (size_t)min(k0 + u, a.S - 1) * a.stride;
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int i = 0; i < RPT; ++i) p[u][s][i] = part[ko + idx[s][i]];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (k0 + u < a.S) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int i = 0; i < RPT; ++i) vals[s][i] += p[u][s][i];
      }
  }
  // both slots go through the reductions together (two LDS rows): four barriers in all
  float mean_s[NS], m2_s[NS], n_s[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      vals[s][i] = ok[s][i] ? vals[s][i] + add + rs[s][i] : 0.f;
      if (ok[s][i]) { a.out[idx[s][i]] = vals[s][i]; s1 += vals[s][i]; }
    }
    red[s][tid] = s1;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float tot = 0.f;
    if (act)
      for (int l = 0; l < nl; ++l) tot += red[s][l * a.C + c];
    n_s[s] = (float)max(0, min(a.V, s * 32 + 32) - s * 32);
    mean_s[s] = n_s[s] > 0.f ? tot / n_s[s] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i)
      if (ok[s][i]) { const float d = vals[s][i] - mean_s[s]; q += d * d; }
    red[s][tid] = q;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float m2 = 0.f;
    if (act && rl == 0) {
      for (int l = 0; l < nl; ++l) m2 += red[s][l * a.C + c];
      if (s < a.nslots && a.stat_part) {
        float *sp = a.stat_part + (((size_t)b * a.nslots + s) * a.C + c) * 2;
        sp[0] = mean_s[s];
        sp[1] = m2;
        if (c == 0) a.stat_cnt[(size_t)b * a.nslots + s] = n_s[s];
      }
    }
    m2_s[s] = m2;
    if (s >= a.nslots) n_s[s] = 0.f;
  }
  // ---- GroupNorm finalisation of the consumer: per-channel triples, then the groups ----
  if (tid < Ct) {
    float N = 0.f, M = 0.f, S2 = 0.f;
    if (tid < a.C) {                       // (tid < C  =>  rl == 0, c == tid: this thread merged the row lanes above)
#pragma unroll
      for (int s = 0; s < NS; ++s) chan_combine(N, M, S2, n_s[s], mean_s[s], m2_s[s]);
    } else {
#pragma unroll
      for (int u = 0; u < PS; ++u) chan_combine(N, M, S2, n1v[u], q1v[u].x, q1v[u].y);
      for (int u = PS; u < a.fin_ns1; ++u)
        chan_combine(N, M, S2, a.fin_n1[(size_t)b * a.fin_ns1 + u], a.fin_p1[(((size_t)b * a.fin_ns1 + u) * a.fin_C1 + cc1) * 2],
                     a.fin_p1[(((size_t)b * a.fin_ns1 + u) * a.fin_C1 + cc1) * 2 + 1]);
    }
    cmean[tid] = M;
    cm2[tid] = S2;
  }
  __syncthreads();
  // groups: 8 lanes per group merge a strided share of its channels, lane 0 merges the 8 shares (both in a fixed
  // order); the whole step lives in the first waves, shares exchanged by shuffles
  const int cg = Ct / a.fin_groups;
  if (tid < a.fin_groups * 8) {
    const int g = tid >> 3, j = tid & 7;
    float N = 0.f, M = 0.f, S2 = 0.f;
    for (int i = j; i < cg; i += 8) chan_combine(N, M, S2, (float)a.V, cmean[g * cg + i], cm2[g * cg + i]);
    float GN = 0.f, GM = 0.f, GS = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float nk = __shfl(N, (tid & ~7) + k), mk = __shfl(M, (tid & ~7) + k), sk = __shfl(S2, (tid & ~7) + k);
      chan_combine(GN, GM, GS, nk, mk, sk);
    }
    if (j == 0) {
      gmean[g] = GM;
      grstd[g] = rsqrtf(GS / GN + a.fin_eps);
    }
  }
  __syncthreads();
  if (tid < Ct) {
    const int g = tid / cg;
    const float sc = grstd[g] * a.fin_gamma[tid];
    a.fin_gn[((size_t)b * 2 + 0) * Ct + tid] = sc;
    a.fin_gn[((size_t)b * 2 + 1) * Ct + tid] = a.fin_beta[tid] - gmean[g] * sc;
    if (a.fin_mr) {
      a.fin_mr[((size_t)b * 2 + 0) * Ct + tid] = gmean[g];
      a.fin_mr[((size_t)b * 2 + 1) * Ct + tid] = grstd[g];
    }
  }
}

bool combine_gn_ok(const CombineArgs &a) {
  const int Ct = a.C + a.fin_C1;
  return a.C >= 1 && a.C <= 256 && a.V <= 64 && (long long)a.B * a.V * a.C < (1ll << 31) && a.nslots <= 2 && Ct <= 1024 && a.fin_groups >= 1 && a.fin_groups <= 32 && a.fin_groups * 8 <= 1024 &&
         Ct % a.fin_groups == 0 && a.stat_part;
}

hipError_t launch_combine_gn(const CombineArgs &a, hipStream_t st) {
  if (!combine_gn_ok(a) || !a.fin_gn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(combine_gn_kernel, dim3(a.B), dim3(1024), 0, st, a);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// Per-(sample, channel, frame) reductions behind the reference's sampling metrics
// (utils/metrics/metricsGenerator.py:70-92,120-186,293-339): squared error, squared error and
// count under the density mask gt[0] > 1e-5, total variation of prediction and ground truth,
// plane sums, and the ground-truth min / max per (sample, channel) for the data ranges.
// pred / gt: [N, C, H, W, F] (reference layout).  One workgroup per (frame, channel, sample);
// sums in double like np.mean(..., dtype=float64), fixed order.
// out [N][C][F][8] doubles: sse, masked sse, masked count, tv_pred, tv_gt, sum_pred, sum_gt, unused
// minmax [N][C][F][2] floats: min / max of the gt plane (reduced over frames on the host).
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frame_metrics_kernel(const float *__restrict__ pred, const float *__restrict__ gt,
                                                            int C, int H, int W, int F, double *__restrict__ out,
                                                            float *__restrict__ minmax) {
  __shared__ double sh[7][256];
  __shared__ float shm[2][256];
  const int j = blockIdx.x, c = blockIdx.y, n = blockIdx.z, tid = threadIdx.x;
  const size_t plane = ((size_t)n * C + c) * H * W * F;        // element (h, w, j) at plane + (h*W + w)*F + j
  const size_t plane0 = ((size_t)n * C + 0) * H * W * F;       // density channel: the mask
  double a[7] = {0, 0, 0, 0, 0, 0, 0};
  float mn = 3.0e38f, mx = -3.0e38f;
  for (int i = tid; i < H * W; i += 256) {
    const int h = i / W, w = i - h * W;
    const float g = gt[plane + (size_t)i * F + j], p = pred[plane + (size_t)i * F + j];
    const float d = g - p;
    a[0] += (double)d * (double)d;
    if (gt[plane0 + (size_t)i * F + j] > 0.00001f) { a[1] += (double)d * (double)d; a[2] += 1.0; }
    if (h + 1 < H) {
      a[3] += (double)fabsf(pred[plane + (size_t)(i + W) * F + j] - p);
      a[4] += (double)fabsf(gt[plane + (size_t)(i + W) * F + j] - g);
    }
    if (w + 1 < W) {
      a[3] += (double)fabsf(pred[plane + (size_t)(i + 1) * F + j] - p);
      a[4] += (double)fabsf(gt[plane + (size_t)(i + 1) * F + j] - g);
    }
    a[5] += (double)p;
    a[6] += (double)g;
    mn = fminf(mn, g);
    mx = fmaxf(mx, g);
  }
  for (int k = 0; k < 7; ++k) sh[k][tid] = a[k];
  shm[0][tid] = mn; shm[1][tid] = mx;
  __syncthreads();
  if (tid < 8) {
    double *o = out + ((((size_t)n * C + c) * F + j) * 8);
    if (tid < 7) {
      double s = 0;
      for (int l = 0; l < 256; ++l) s += sh[tid][l];
      o[tid] = s;
    } else {
      float lo = 3.0e38f, hi = -3.0e38f;
      for (int l = 0; l < 256; ++l) { lo = fminf(lo, shm[0][l]); hi = fmaxf(hi, shm[1][l]); }
      o[7] = 0.0;
      minmax[(((size_t)n * C + c) * F + j) * 2 + 0] = lo;
      minmax[(((size_t)n * C + c) * F + j) * 2 + 1] = hi;
    }
  }
}

hipError_t launch_frame_metrics(const float *pred, const float *gt, int N, int C, int H, int W, int F, double *out,
                                float *minmax, hipStream_t st) {
  hipLaunchKernelGGL(frame_metrics_kernel, dim3(F, C, N), dim3(256), 0, st, pred, gt, C, H, W, F, out, minmax);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// ABI edge: reference layout [B,C,H,W,L] <-> channels-last [B][L][H][W][8].
// unet.py:138 (cat past||future on L) and unet.py:166 (keep frames >= P) are folded in.
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_input_kernel(const float *__restrict__ past,
                                                             const float *__restrict__ fut, float *__restrict__ x8,
                                                             int B, int C, int H, int W, int P, int F, int which) {
  const int L = P + F;
  const long long total = (long long)B * L * H * W;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int w = (int)(i % W);
  long long q = i / W;
  const int hh = (int)(q % H); q /= H;
  const int l = (int)(q % L);
  const int b = (int)(q / L);
  const bool is_past = l < P;
  if (is_past && !(which & 1)) return;
  if (!is_past && !(which & 2)) return;
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float val = 0.f;
    if (c < C) {
      const size_t base = (((size_t)b * C + c) * H + hh) * W + w;
      val = is_past ? past[base * P + l] : fut[base * F + (l - P)];
    }
    v[c] = val;
  }
  f32x4 *dst = reinterpret_cast<f32x4 *>(x8 + (size_t)i * 8);
  dst[0] = f32x4{v[0], v[1], v[2], v[3]};
  dst[1] = f32x4{v[4], v[5], v[6], v[7]};
}

hipError_t launch_assemble_input(const float *past, const float *future, float *x8, int B, int C, int H, int W, int P,
                                 int F, int which, hipStream_t st) {
  const long long total = (long long)B * (P + F) * H * W;
  hipLaunchKernelGGL(assemble_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, past, future, x8,
                     B, C, H, W, P, F, which);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void extract_output_kernel(const float *__restrict__ eps_cl, int cs,
                                                             float *__restrict__ out, int B, int C, int H, int W,
                                                             int P, int F) {
  const long long total = (long long)B * C * H * W * F;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int f = (int)(i % F);
  long long q = i / F;
  const int w = (int)(q % W); q /= W;
  const int hh = (int)(q % H); q /= H;
  const int c = (int)(q % C);
  const int b = (int)(q / C);
  const int L = P + F;
  out[i] = eps_cl[((((size_t)b * L + (P + f)) * H + hh) * W + w) * cs + c];
}

hipError_t launch_extract_output(const float *eps_cl, int cs, float *out, int B, int C, int H, int W, int P, int F,
                                 hipStream_t st) {
  const long long total = (long long)B * C * H * W * F;
  hipLaunchKernelGGL(extract_output_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, eps_cl, cs, out,
                     B, C, H, W, P, F);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void cl_to_ref_kernel(const float *__restrict__ x, int cs, float *__restrict__ out,
                                                        int B, int C, int Z, int Y, int X) {
  const long long total = (long long)B * C * Y * X * Z;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int z = (int)(i % Z);
  long long q = i / Z;
  const int xx = (int)(q % X); q /= X;
  const int y = (int)(q % Y); q /= Y;
  const int c = (int)(q % C);
  const int b = (int)(q / C);
  out[i] = x[((((size_t)b * Z + z) * Y + y) * X + xx) * cs + c];
}

hipError_t launch_cl_to_ref(const float *x_cl, int cs, float *out, int B, int C, int Z, int Y, int X, hipStream_t st) {
  const long long total = (long long)B * C * Y * X * Z;
  hipLaunchKernelGGL(cl_to_ref_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x_cl, cs, out, B, C,
                     Z, Y, X);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// Time embedding (embeddings.py:24-30) followed by every ResnetBlock's
// dense_1(SiLU(temb)) (layers.py:35,62), one workgroup per table row.  It depends
// only on t and the weights, so the host evaluates it once for all 1000 rows at
// load time and the conv epilogues index the result by t[b].
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_mlp_kernel(const float *__restrict__ table, const float *__restrict__ W1,
                                                       const float *__restrict__ b1, const float *__restrict__ W2,
                                                       const float *__restrict__ b2, const float *__restrict__ Wd,
                                                       const float *__restrict__ bd, int te, int tx, int nproj,
                                                       float *__restrict__ temb_raw, float *__restrict__ out,
                                                       const long long *__restrict__ rowidx) {
  extern __shared__ float sm[];  // e[te], h1[tx], h2[tx]
  float *e = sm, *h1 = sm + te, *h2 = h1 + tx;
  const int row = blockIdx.x, tid = threadIdx.x;
  const long long trow = rowidx ? rowidx[row] : row;   // training: table row = t[b]
  for (int i = tid; i < te; i += 256) e[i] = table[(size_t)trow * te + i];
  __syncthreads();
  for (int o = tid; o < tx; o += 256) {
    float acc = b1[o];
    for (int i = 0; i < te; ++i) acc = fmaf(W1[(size_t)o * te + i], e[i], acc);
    h1[o] = silu_f(acc);
  }
  __syncthreads();
  for (int o = tid; o < tx; o += 256) {
    float acc = b2[o];
    for (int i = 0; i < tx; ++i) acc = fmaf(W2[(size_t)o * tx + i], h1[i], acc);
    if (temb_raw) temb_raw[(size_t)row * tx + o] = acc;
    h2[o] = silu_f(acc);
  }
  __syncthreads();
  for (int o = tid; o < nproj; o += 256) {
    float acc = bd[o];
    for (int i = 0; i < tx; ++i) acc = fmaf(Wd[(size_t)o * tx + i], h2[i], acc);
    out[(size_t)row * nproj + o] = acc;
  }
}

hipError_t launch_time_mlp(const float *table, const float *W1, const float *b1, const float *W2, const float *b2,
                           const float *Wd, const float *bd, int te, int tx, int nproj, int nrows, float *temb_raw,
                           float *out, const long long *rowidx, hipStream_t st) {
  const size_t smem = (size_t)(te + 2 * tx) * sizeof(float);
  hipLaunchKernelGGL(time_mlp_kernel, dim3(nrows), dim3(256), smem, st, table, W1, b1, W2, b2, Wd, bd, te, tx, nproj,
                     temb_raw, out, rowidx);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// Attention core of nn.MultiheadAttention (layers.py:16): per (sample, head)
// softmax(q k^T / sqrt(d)) v with d = 32, S = H*W*L tokens at quarter resolution
// (54 / 84 / 216).  One workgroup per (head, sample): K and V live in LDS, each
// thread owns one query row (online softmax).  The in/out projections run on the
// MFMA 1x1x1 conv path.
// --------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_core_kernel(const float *__restrict__ qkv, float *__restrict__ out, int S,
                                                        int E) {
  // PARTS adjacent lanes share one query row, each owning 8 of the D head dims: the
  // q.k dot product is finished with PARTS-wide xor-shuffles, the softmax state is
  // replicated, and every lane accumulates its 8 output dims.
  constexpr int PARTS = D / 8;
  constexpr int ROWS = 256 / PARTS;
  extern __shared__ float sm[];  // K[S][D], V[S][D]
  float *Ks = sm, *Vs = sm + (size_t)S * D;
  const int hd = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float *base = qkv + (size_t)b * S * 3 * E;
  for (int i = tid; i < S * (D / 4); i += 256) {
    const int s = i / (D / 4), d4 = i % (D / 4);
    const f32x4 k = *reinterpret_cast<const f32x4 *>(base + (size_t)s * 3 * E + E + hd * D + 4 * d4);
    const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (size_t)s * 3 * E + 2 * E + hd * D + 4 * d4);
    *reinterpret_cast<f32x4 *>(Ks + s * D + 4 * d4) = k;
    *reinterpret_cast<f32x4 *>(Vs + s * D + 4 * d4) = v;
  }
  __syncthreads();
  const float scale = rsqrtf((float)D);
  const int part = tid % PARTS, rl = tid / PARTS;
  // gridDim.z > 1 (many tokens): workgroup z handles the query-row passes z, z + gridDim.z, ... so that the
  // S^2 work of one (head, sample) spreads over several CUs (each re-stages K / V: 2 * S * D floats)
  for (int row0 = blockIdx.z * ROWS; row0 < S; row0 += ROWS * gridDim.z) {
    const int row = row0 + rl;
    const int rowc = row < S ? row : S - 1;  // surplus lanes shadow the last row (shuffles stay convergent)
    const float *qp = base + (size_t)rowc * 3 * E + hd * D + 8 * part;
    const f32x4 q0 = *reinterpret_cast<const f32x4 *>(qp) * scale;
    const f32x4 q1 = *reinterpret_cast<const f32x4 *>(qp + 4) * scale;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    float mx = -3.0e38f, l = 0.f;
    for (int j = 0; j < S; ++j) {
      const f32x4 k0 = *reinterpret_cast<const f32x4 *>(Ks + j * D + 8 * part);
      const f32x4 k1 = *reinterpret_cast<const f32x4 *>(Ks + j * D + 8 * part + 4);
      float sc = q0[0] * k0[0];
      sc = fmaf(q0[1], k0[1], sc); sc = fmaf(q0[2], k0[2], sc); sc = fmaf(q0[3], k0[3], sc);
      sc = fmaf(q1[0], k1[0], sc); sc = fmaf(q1[1], k1[1], sc); sc = fmaf(q1[2], k1[2], sc); sc = fmaf(q1[3], k1[3], sc);
#pragma unroll
      for (int m = 1; m < PARTS; m <<= 1) sc += __shfl_xor(sc, m);
      const f32x4 v0 = *reinterpret_cast<const f32x4 *>(Vs + j * D + 8 * part);
      const f32x4 v1 = *reinterpret_cast<const f32x4 *>(Vs + j * D + 8 * part + 4);
      if (sc > mx) {
        const float corr = __expf(mx - sc);
        l *= corr;
        o0 *= corr;
        o1 *= corr;
        mx = sc;
      }
      const float p = __expf(sc - mx);
      l += p;
      o0 += v0 * p;
      o1 += v1 * p;
    }
    if (row < S) {
      const float inv = 1.0f / l;
      float *op = out + ((size_t)b * S + row) * E + hd * D + 8 * part;
      *reinterpret_cast<f32x4 *>(op) = o0 * inv;
      *reinterpret_cast<f32x4 *>(op + 4) = o1 * inv;
    }
  }
}

hipError_t launch_attn_core(const float *qkv, float *out, int B, int S, int E, int heads, hipStream_t st) {
  const int D = E / heads;
  const size_t smem = (size_t)2 * S * D * sizeof(float);
  if (smem > 160 * 1024) return hipErrorInvalidValue;
#define CM_ATTN(DD)                                                                                         \
  if (D == DD) {                                                                                            \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_core_kernel<DD>),                \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);             \
    if (e != hipSuccess) return e;                                                                          \
    const int rows_per_pass = 256 / (DD / 8);                                                              \
    const int nz = S > 2 * rows_per_pass ? (S + rows_per_pass - 1) / rows_per_pass : 1;                     \
    hipLaunchKernelGGL((attn_core_kernel<DD>), dim3(heads, B, nz), dim3(256), smem, st, qkv, out, S, E);     \
    return hipGetLastError();                                                                               \
  }
  CM_ATTN(8) CM_ATTN(16) CM_ATTN(32) CM_ATTN(64)
#undef CM_ATTN
  return hipErrorInvalidValue;
}

// --------------------------------------------------------------------------------
// Attention core of the reduced-precision plan (cm_model_set_precision(F16); the reference's torch.amp.autocast covers
// nn.MultiheadAttention, ddpm.py:116-120 / layers.py:16) for token counts beyond the fused block's LDS design (216 tokens on
// the 24x72 grid): softmax(q k^T / sqrt(d)) v per (head, sample) on v_mfma_f32_32x32x16_f16 -- f16 operands, fp32
// accumulation, fp32 softmax.  One 256-thread workgroup per (head, sample): K and V^T live in LDS as f16, each wave takes
// query blocks of 32 rows.  Scores are computed TRANSPOSED (S^T = K Q^T: keys in the rows, queries in the columns), so in the
// accumulator layout a lane owns ONE query and its registers are keys: the online softmax is lane-local plus one xor-32
// exchange for the running maximum, and the probabilities are already the B operand of O^T += V^T P^T -- registers
// 8u .. 8u + 7 of a lane are the keys 16u + {0..3, 8..11} + 4 hh of the block, the order V^T's fragments are read in.
// d = 32 only (the reference's 4 heads of 128 channels); the fp32 plan keeps attn_core_kernel.
// --------------------------------------------------------------------------------
typedef _Float16 f16x8m __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4m __attribute__((ext_vector_type(4)));
typedef float f32x16m __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void attn_core_f16_kernel(const float *__restrict__ qkv, float *__restrict__ out, int S, int E) {
  constexpr int D = 32, KS = D + 8;                 // K row stride in halves (80 B: 16-byte aligned, conflict-free b128 reads)
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  const int nkb = (S + 31) >> 5, Sp = nkb * 32, VS = Sp + 8;   // V^T row stride in halves (8-byte aligned)
  _Float16 *Ks = reinterpret_cast<_Float16 *>(smraw);           // [Sp][KS]
  _Float16 *Vt = Ks + (size_t)Sp * KS;                          // [D][VS]
  const int hd = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const float *base = qkv + (size_t)b * S * 3 * E;
  // ---- stage K (rows) and V (transposed) as f16; rows / columns beyond S are zero ----
  for (int i = tid; i < Sp * (D / 4); i += 256) {
    const int s = i / (D / 4), d4 = i % (D / 4);
    f32x4 k = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
    if (s < S) {
      k = *reinterpret_cast<const f32x4 *>(base + (size_t)s * 3 * E + E + hd * D + 4 * d4);
      v = *reinterpret_cast<const f32x4 *>(base + (size_t)s * 3 * E + 2 * E + hd * D + 4 * d4);
    }
    *reinterpret_cast<f16x4m *>(Ks + (size_t)s * KS + 4 * d4) = f16x4m{(_Float16)k[0], (_Float16)k[1], (_Float16)k[2], (_Float16)k[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) Vt[(size_t)(4 * d4 + e) * VS + s] = (_Float16)v[e];
  }
  __syncthreads();
  const float scale = rsqrtf((float)D);
  for (int qb = wave; qb < nkb; qb += 4) {
    const int q = qb * 32 + r, qc = q < S ? q : S - 1;
    // B operand: Q^T, lane (query r, hh): dims 16 g + 8 hh .. + 7, scaled as the fp32 kernel scales q
    f16x8m qf[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float *qp = base + (size_t)qc * 3 * E + hd * D + 16 * g + 8 * hh;
      const f32x4 q0 = *reinterpret_cast<const f32x4 *>(qp) * scale, q1 = *reinterpret_cast<const f32x4 *>(qp + 4) * scale;
      qf[g] = f16x8m{(_Float16)q0[0], (_Float16)q0[1], (_Float16)q0[2], (_Float16)q0[3],
                     (_Float16)q1[0], (_Float16)q1[1], (_Float16)q1[2], (_Float16)q1[3]};
    }
    f32x16m o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    float mx = -3.0e38f, l = 0.f;                   // running maximum (shared by the two halves of a query) and this half's sum
    for (int kb = 0; kb < nkb; ++kb) {
      // S^T block: keys 32 kb + (row), A = K rows of this lane's key
      f32x16m sc;
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[i] = 0.f;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f16x8m kf = *reinterpret_cast<const f16x8m *>(Ks + (size_t)(kb * 32 + r) * KS + 16 * g + 8 * hh);
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[g], sc, 0, 0, 0);
      }
      // this lane's 16 keys of the block: key = 32 kb + (reg & 3) + 8 (reg >> 2) + 4 hh
      float bm = -3.0e38f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int key = kb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        if (key >= S) sc[reg] = -3.0e38f;
        bm = fmaxf(bm, sc[reg]);
      }
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      const float mnew = fmaxf(mx, bm);
      const float corr = __expf(mx - mnew);
      mx = mnew;
      l *= corr;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] *= corr;
      float ps = 0.f;
      f16x8m pf[2];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int key = kb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        const float pv = key < S ? __expf(sc[reg] - mnew) : 0.f;
        ps += pv;
        pf[reg >> 3][reg & 7] = (_Float16)pv;
      }
      l += ps;
      // O^T (dims x queries) += V^T P^T over the block's 32 keys: two K = 16 groups in the register order above
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const _Float16 *vp = Vt + (size_t)r * VS + kb * 32 + 16 * u + 4 * hh;      // lane (dim r, hh): keys 16u + 4hh + {0..3, 8..11}
        const f16x4m v0 = *reinterpret_cast<const f16x4m *>(vp), v1 = *reinterpret_cast<const f16x4m *>(vp + 8);
        const f16x8m vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[u], o, 0, 0, 0);
      }
    }
    l += __shfl_xor(l, 32);
    if (q < S) {
      const float inv = 1.0f / l;
      float *op = out + ((size_t)b * S + q) * E + hd * D;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) op[(reg & 3) + 8 * (reg >> 2) + 4 * hh] = o[reg] * inv;
    }
  }
}

hipError_t launch_attn_core_f16(const float *qkv, float *out, int B, int S, int E, int heads, hipStream_t st) {
  if (heads <= 0 || E / heads != 32 || E % heads || S < 1) return hipErrorInvalidValue;
  const int Sp = (S + 31) / 32 * 32;
  const size_t smem = ((size_t)Sp * (32 + 8) + (size_t)32 * (Sp + 8)) * sizeof(_Float16);
  if (smem > 160 * 1024) return hipErrorInvalidValue;
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set[dev & 63]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_core_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev & 63] = true;
  }
  hipLaunchKernelGGL(attn_core_f16_kernel, dim3(heads, B), dim3(256), smem, st, qkv, out, S, E);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// Device RNG: Philox4x32-10 keyed by the seed, counter = (element quad, global
// sample id, step, stream).  Independent of batch sharding: sample i of the job
// draws the same numbers on 1 GPU and on 8.
// --------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned long long p0 = 0xD2511F53ull * c0;
    const unsigned long long p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float philox_normal(unsigned long long seed, long long sample, int step, long long elem) {
  unsigned r[4];
  philox4x32_10((unsigned)(elem >> 1), (unsigned)sample, (unsigned)step, (unsigned)((sample >> 32) ^ 0x5eed),
                (unsigned)seed, (unsigned)(seed >> 32), r);
  // Box-Muller on one pair; element parity picks cos / sin branch
  const float u1 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float rad = sqrtf(-2.0f * __logf(u1));
  const float ang = 6.283185307179586f * u2;
  return (elem & 1) ? rad * __sinf(ang) : rad * __cosf(ang);
}

__global__ __launch_bounds__(256) void randn_kernel(float *__restrict__ x, long long per, long long total,
                                                    unsigned long long seed, long long sample_id_base, int step) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long b = i / per, e = i - b * per;
  x[i] = philox_normal(seed, sample_id_base + b, step, e);
}

hipError_t launch_randn(float *x, int B, long long per, unsigned long long seed, long long sample_id_base, int step,
                        hipStream_t st) {
  const long long total = (long long)B * per;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, per, total, seed,
                     sample_id_base, step);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// One reverse-process update for both samplers, written as
//     x' = c_x * x + c_eps * eps_hat + c_noise * z          (+ sparsity guidance)
// DDPM.step (ddpm.py:31-37):  c_x = 1/sqrt(alpha_t), c_eps = -c_x*beta_t/sqrt(1-abar_t),
//                             c_noise = sqrt(beta_t)
// DDIM Eq.12 (ddpm.py:262-265): c_x = sab_prev/sab_t, c_eps = sqrt(1-sab_prev^2-sigma^2)
//                             - c_x*s1m_t, c_noise = sigma
// guidance (ddpm.py:223-226, guidance.py:4-8): x'[:,0] -= guid * sign(x'[:,0]).
// Besides x (reference layout) it rewrites the future frames of the channels-last
// UNet input so the next step needs no re-assembly.
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sampler_step_kernel(StepArgs a) {
  if (a.tab) {
    const int k = *a.kctr;
    const StepRow r = a.tab[k];
    a.c_x = r.c_x; a.c_eps = r.c_eps; a.c_noise = r.c_noise; a.guid = r.guid; a.draw = r.draw; a.step = r.step;
    if (a.noise) a.noise += (long long)k * a.row_stride + a.boff;
    if (a.hist) a.hist += (long long)(k + 1) * a.row_stride + a.boff;
  }
  const long long per = (long long)a.C * a.H * a.W * a.F;
  const long long total = per * a.B;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (a.t_next && i < a.B) a.t_next[i] = a.t_next_v;   // (the UNet of this step has consumed the buffer)
  if (a.zero_u64)
    for (long long z = i; z < a.zero_n; z += (long long)gridDim.x * 256) a.zero_u64[z] = 0ull;
  if (i >= total) return;
  const long long b = i / per, e = i - b * per;
  const int f = (int)(e % a.F);
  long long q = e / a.F;
  const int w = (int)(q % a.W); q /= a.W;
  const int hh = (int)(q % a.H);
  const int c = (int)(q / a.H);
  const int L = a.P + a.F;
  const size_t cl = ((((size_t)b * L + (a.P + f)) * a.H + hh) * a.W + w);
  const float eps = a.eps_cl[cl * a.cs + c];
  float z = 0.f;
  if (a.draw) z = a.noise ? a.noise[i] : philox_normal(a.seed, a.sample_id_base + b, a.step, e);
  // evaluation order mirrors the reference expression tree where it matters for rounding
  float xn = a.c_x * a.x[i] + a.c_eps * eps + a.c_noise * z;
  if (a.guid != 0.f && c == 0) xn -= a.guid * (xn > 0.f ? 1.f : (xn < 0.f ? -1.f : 0.f));
  a.x[i] = xn;
  if (a.x8) a.x8[cl * 8 + c] = xn;
  if (a.hist) a.hist[i] = xn;
}

hipError_t launch_sampler_step(const StepArgs &a, hipStream_t st) {
  const long long total = (long long)a.B * a.C * a.H * a.W * a.F;
  hipLaunchKernelGGL(sampler_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void q_sample_kernel(const float *__restrict__ x0, const long long *__restrict__ t,
                                                       const float *__restrict__ eps, const float *__restrict__ sab,
                                                       const float *__restrict__ s1m, float *__restrict__ xt,
                                                       long long per, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long b = i / per;
  const long long tt = t[b];
  xt[i] = sab[tt] * x0[i] + s1m[tt] * eps[i];
}

hipError_t launch_q_sample(const float *x0, const long long *t, const float *eps, const float *sab, const float *s1m,
                           float *xt, int B, long long per, hipStream_t st) {
  const long long total = (long long)B * per;
  hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x0, t, eps, sab, s1m,
                     xt, per, total);
  return hipGetLastError();
}

// mean squared error, F.mse_loss(reduction='mean') (ddpm.py:120): per-workgroup partial sums in
// a fixed order, final sum by one thread -> deterministic.
__global__ __launch_bounds__(256) void mse_partial_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          long long n, float *__restrict__ partial) {
  __shared__ float sh[256];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float d = a[i] - b[i];
    acc = fmaf(d, d, acc);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ void mse_final_kernel(const float *__restrict__ partial, int np, long long n, float *__restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < np; ++i) s += (double)partial[i];
    *loss = (float)(s / (double)n);
  }
}

hipError_t launch_mse_loss(const float *a, const float *b, long long n, float *partial, float *loss, hipStream_t st) {
  const int np = 64;
  hipLaunchKernelGGL(mse_partial_kernel, dim3(np), dim3(256), 0, st, a, b, n, partial);
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, st, partial, np, n, loss);
  return hipGetLastError();
}

// Dropout3d (layers.py:42,71) zeroes whole (sample, channel) volumes with probability p and
// scales the survivors by 1/(1-p).
__global__ void dropout_mask_kernel(float *__restrict__ mask, int B, int C, float p, unsigned long long seed,
                                    long long sample_id_base, int step) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  unsigned r[4];
  philox4x32_10((unsigned)c, (unsigned)(sample_id_base + b), (unsigned)step, 0xD120u, (unsigned)seed,
                (unsigned)(seed >> 32), r);
  const float u = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  mask[i] = (u >= p) ? 1.0f / (1.0f - p) : 0.0f;
}

hipError_t launch_dropout_mask(float *mask, int B, int C, float p, unsigned long long seed, long long sample_id_base,
                               int step, hipStream_t st) {
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, mask, B, C, p, seed,
                     sample_id_base, step);
  return hipGetLastError();
}

__global__ void fill_t_kernel(long long *t, int B, long long v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B) t[i] = v;
}

__global__ __launch_bounds__(256) void step_begin_kernel(long long *t, int B, const StepRow *tab, int *kctr) {
  __shared__ int ks;
  if (threadIdx.x == 0) { ks = *kctr + 1; *kctr = ks; }
  __syncthreads();
  const long long v = tab[ks].t;
  for (int i = threadIdx.x; i < B; i += 256) t[i] = v;
}

hipError_t launch_step_begin(long long *t, int B, const StepRow *tab, int *kctr, hipStream_t st) {
  hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(256), 0, st, t, B, tab, kctr);
  return hipGetLastError();
}

// Sampler-output health check (failure detection): one workgroup counts the
// elements whose exponent field is all ones.
__global__ __launch_bounds__(1024) void count_nonfinite_kernel(const float *__restrict__ x, long long n, int *__restrict__ count) {
  __shared__ int sh[1024];
  int c = 0;
  for (long long i = threadIdx.x; i < n; i += 1024) c += ((__float_as_uint(x[i]) & 0x7f800000u) == 0x7f800000u) ? 1 : 0;
  sh[threadIdx.x] = c;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = sh[0];
}

hipError_t launch_count_nonfinite(const float *x, long long n, int *count, hipStream_t st) {
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3(1), dim3(1024), 0, st, x, n, count);
  return hipGetLastError();
}

hipError_t launch_fill_t(long long *t, int B, long long value, hipStream_t st) {
  hipLaunchKernelGGL(fill_t_kernel, dim3((B + 255) / 256), dim3(256), 0, st, t, B, value);
  return hipGetLastError();
}

// ---- fall-backs of the accumulator statistics (ConvArgs::astat / gs0) ------------------------------------------------------
// gn rows [B][2][C0 + C1] from the accumulators: for a consumer kernel that cannot finalise them in its own prologue
__global__ __launch_bounds__(256) void gn_from_sums_kernel(const ConvArgs a, int V, float *__restrict__ rows) {
  extern __shared__ __attribute__((aligned(16))) double gsum[];   // [C0 + C1][2]
  const int b = blockIdx.x;
  cm_gn_rows_from_sums(a, b, V, rows + (size_t)b * 2 * (a.C0 + a.C1), gsum, threadIdx.x, 256);
}
hipError_t launch_gn_from_sums(const ConvArgs &a, int V, float *gn_rows, hipStream_t st) {
  hipLaunchKernelGGL(gn_from_sums_kernel, dim3((unsigned)a.B), dim3(256), (size_t)(a.C0 + a.C1) * 2 * sizeof(double), st, a, V, gn_rows);
  return hipGetLastError();
}
// accumulators from slot partials: for a producer kernel that wrote (mean, M2, count) slots instead of adding to them
__global__ __launch_bounds__(256) void slots_to_sums_kernel(const float *__restrict__ part, const float *__restrict__ cnt, int nslots, int C,
                                                            unsigned long long *__restrict__ astat, int astat_C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256)
    for (int s2 = 0; s2 < nslots; ++s2) {
      const float n = cnt[(size_t)b * nslots + s2];
      if (!(n > 0.f)) continue;
      const float *sp = part + (((size_t)b * nslots + s2) * C + c) * 2;
      cm_stat_atomic(astat + ((size_t)b * astat_C + c) * 3, sp[0] * n, sp[0], sp[1]);
    }
}
hipError_t launch_slots_to_sums(const float *part, const float *cnt, int nslots, int C, int B, unsigned long long *astat, int astat_C, hipStream_t st) {
  hipLaunchKernelGGL(slots_to_sums_kernel, dim3((unsigned)B), dim3(256), 0, st, part, cnt, nslots, C, astat, astat_C);
  return hipGetLastError();
}

}  // namespace cm
