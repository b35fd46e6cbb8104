// Fused AttentionBlock (reference: /root/reference/models/backbones/layers.py:5-18):
//     x + MHA(GroupNorm(8, E)(x))          over S = Z*Y*X tokens, E channels, 4 heads of D = E/4
// as ONE launch per (head, sample) plus the K-split combine pass, instead of the five launches of the
// generic plan (GroupNorm finalise, in-projection conv, attention core, out-projection conv + the
// finalise in front of it).  At 54-216 tokens per sample these launches are latency chains: the whole
// block is 0.8 % of the UNet's FLOPs.
//
// One 768-thread workgroup = (head h, sample b) -- 12 waves, because LDS admits one workgroup per CU and every
// phase is a latency chain at one wave per SIMD:
//   1. x[b] (S x E, channels-last) -> LDS; GroupNorm statistics of all 8 groups straight from that copy
//      (two passes, fixed reduction order), normalised in place with the affine folded in;
//      the head's weight slices (3D rows of W_in, D columns of W_out) ride in the same single round of global
//      loads -- the kernel never waits on global memory again;
//   2. [q|k|v]_h = xn * W_in[h-slices]^T + b_in on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32):
//      wave w owns the 16-row blocks w, w+4, ...; both operands come from LDS rows in the REFERENCE weight
//      layout (a lane reads 16 contiguous bytes of one row: with the k values of an MFMA chosen as
//      {16j + 4*kq + jj} both fragments are plain float4 reads, no packing); q is pre-scaled by 1/sqrt(D);
//   3. softmax(q k^T) v per query row on the vector ALUs (online softmax, K / V rows broadcast from LDS),
//      written over q;
//   4. partial out-projection  y_h = o_h * W_out[:, h*D:(h+1)*D]^T  (S x E) on the matrix cores into
//      part[h][b]; ksplit_combine_kernel then sums the heads in a fixed order, adds the out-proj bias
//      and the residual x and emits the GroupNorm statistics of the result (cm_misc.hip).
// Nothing is reduced with atomics; sample b never touches another sample's data, so a chain's result
// does not depend on the batch it runs in.
#include "cm_kernels.h"

#include <algorithm>

namespace cm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ATT_NT = 768;                  // 12 waves, 3 per SIMD: one workgroup per CU (LDS), so latency hiding comes from here
constexpr int ATT_KS = 3;                    // key slices of the softmax phase (each query row: PARTS x KS lanes)

template <int D>
__global__ __launch_bounds__(ATT_NT) void attn_head_kernel(const AttnBlockArgs a) {
  constexpr int NT = ATT_NT, NW = NT / 64;
  constexpr int KD = D < 16 ? 16 : D;        // head dim padded to one 16-wide k chunk (zeros beyond D)
  constexpr int DS = KD + 4;                 // LDS row stride of q / k / v and of the out-projection slice
  constexpr int NBQ = (D + 15) / 16;         // 16-column blocks per q / k / v
  constexpr int NBT = 3 * NBQ;
  constexpr int PARTS = D / 8;               // lanes per (query row, key slice) in the softmax phase (8 dims each)
  constexpr int E = 4 * D;                   // nn.MultiheadAttention(E, 4 heads), layers.py:10 (checked on the host)
  constexpr int XS = E + 8;                  // LDS row stride of x / W_in rows: conflict-free ds_read_b128 fragments
  constexpr int Q4 = E >> 2;                 // channel quads (= D); NT % Q4 == 0
  constexpr int RL = NT / Q4;                // row lanes
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int S = a.S;
  float *Xs = sm;                            // [S][XS]
  float *Wi = Xs + (size_t)S * XS;           // [NBT * 16][XS]  in-projection rows of this head (q | k | v)
  float *Wo = Wi + (size_t)NBT * 16 * XS;    // [E][DS]         out-projection columns of this head
  float *Qs = Wo + (size_t)E * DS;           // [S][DS]   (later: the attention output o_h)
  float *Ks = Qs + (size_t)S * DS;
  float *Vs = Ks + (size_t)S * DS;
  float *bi = Vs + (size_t)S * DS;           // [NBT * 16] in-projection bias of this head
  float *red = bi + NBT * 16;                // [RL * 8] row-lane partials + [8] group totals
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;

  // ---- 0. ONE round of global loads: x[b], the head's weight slices, the affine -----------
  // A dependent global load costs 1.5-2 us under load on this part (profiles/round1_notes.md), so every load of
  // the kernel is issued here, back to back into registers (branch-free: clamped addresses), before the first
  // LDS store waits for any of them; nothing below this block touches global memory until the final stores.
  const int q4 = tid % Q4, rl = tid / Q4;
  const float *xb = a.x + (size_t)b * S * E;
  constexpr int UX = 3;                                    // x rows per thread and batch
  constexpr int NWI = (NBT * 16 + RL - 1) / RL;            // W_in rows per thread
  constexpr int NWO = (E * (KD / 4) + NT - 1) / NT;        // W_out float4 per thread
  f32x4 vx[UX], vwi[NWI], vwo[NWO];
#pragma unroll
  for (int u = 0; u < UX; ++u) vx[u] = *reinterpret_cast<const f32x4 *>(xb + (size_t)min(rl + u * RL, S - 1) * E + 4 * q4);
#pragma unroll
  for (int u = 0; u < NWI; ++u) {
    const int wr = min(rl + u * RL, NBT * 16 - 1);
    const int which = wr / (NBQ * 16), c = min(wr % (NBQ * 16), D - 1);   // rows beyond D of a padded block: any valid row
    vwi[u] = *reinterpret_cast<const f32x4 *>(a.w_in + (size_t)(which * E + h * D + c) * E + 4 * q4);
  }
#pragma unroll
  for (int u = 0; u < NWO; ++u) {
    const int i = min(tid + u * NT, E * (KD / 4) - 1);
    const int n = i / (KD / 4), c4 = min(i % (KD / 4), D / 4 - 1);
    vwo[u] = *reinterpret_cast<const f32x4 *>(a.w_out + (size_t)n * E + h * D + 4 * c4);
  }
  const f32x4 ga = *reinterpret_cast<const f32x4 *>(a.gamma + 4 * q4);
  const f32x4 be = *reinterpret_cast<const f32x4 *>(a.beta + 4 * q4);
  float bval = 0.f;
  {
    const int tb = min(tid, NBT * 16 - 1);
    const int which = tb / (NBQ * 16), c = min(tb % (NBQ * 16), D - 1);
    bval = a.b_in[which * E + h * D + c];
  }
  float s1 = 0.f;
#pragma unroll
  for (int u = 0; u < UX; ++u) {
    const int row = rl + u * RL;
    if (row < S) {
      *reinterpret_cast<f32x4 *>(Xs + row * XS + 4 * q4) = vx[u];
      s1 += (vx[u][0] + vx[u][1]) + (vx[u][2] + vx[u][3]);
    }
  }
#pragma unroll
  for (int u = 0; u < NWI; ++u) {
    const int wr = rl + u * RL;
    if (wr < NBT * 16) *reinterpret_cast<f32x4 *>(Wi + wr * XS + 4 * q4) = vwi[u];
  }
#pragma unroll
  for (int u = 0; u < NWO; ++u) {
    const int i = tid + u * NT;
    if (i < E * (KD / 4)) {
      const int n = i / (KD / 4), c4 = i % (KD / 4);
      *reinterpret_cast<f32x4 *>(Wo + n * DS + 4 * c4) = (4 * c4 < D) ? vwo[u] : f32x4{0.f, 0.f, 0.f, 0.f};   // zero beyond D
    }
  }
  if (tid < NBT * 16) bi[tid] = bval;
  for (int row0 = UX * RL; row0 < S; row0 += UX * RL) {   // samples with more than UX * RL tokens: further batches
#pragma unroll
    for (int u = 0; u < UX; ++u) vx[u] = *reinterpret_cast<const f32x4 *>(xb + (size_t)min(row0 + rl + u * RL, S - 1) * E + 4 * q4);
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int row = row0 + rl + u * RL;
      if (row < S) {
        *reinterpret_cast<f32x4 *>(Xs + row * XS + 4 * q4) = vx[u];
        s1 += (vx[u][0] + vx[u][1]) + (vx[u][2] + vx[u][3]);
      }
    }
  }

  // ---- 1. GroupNorm of x[b] from the LDS copy (two passes, fixed reduction order) -------------
  constexpr int QPG = Q4 / 8;                // quads per group (GroupNorm(8, E): E/8 channels = D/8 quads... x4 heads)
  const int grp = q4 / QPG;
  auto group_sum = [&](float v) -> float {   // returns the group total to every thread
    // the QPG quads of a group sit in adjacent lanes of one wave: fold them first, then the RL row lanes
#pragma unroll
    for (int mk = 1; mk < QPG; mk <<= 1) v += __shfl_xor(v, mk);
    __syncthreads();
    if ((q4 & (QPG - 1)) == 0) red[rl * 8 + grp] = v;
    __syncthreads();
    if (tid < 8) {
      float t = 0.f;
      for (int r = 0; r < RL; ++r) t += red[r * 8 + tid];
      red[RL * 8 + tid] = t;
    }
    __syncthreads();
    return red[RL * 8 + grp];
  };
  const float cnt = (float)(E / 8) * (float)S;
  const float mean = group_sum(s1) / cnt;
  float s2 = 0.f;
  for (int row = rl; row < S; row += RL) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(Xs + row * XS + 4 * q4);
    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  const float rstd = rsqrtf(group_sum(s2) / cnt + a.eps);   // biased variance, layers.py:9 (nn.GroupNorm)
  {
    const f32x4 sc = ga * rstd;
    const f32x4 sh = be - sc * mean;
    for (int row = rl; row < S; row += RL) {
      f32x4 *p = reinterpret_cast<f32x4 *>(Xs + row * XS + 4 * q4);
      *p = *p * sc + sh;
    }
  }
  __syncthreads();

  // ---- 2. q, k, v of this head: one 16 x 16 output block per wave and round -------------------
  const int nmb = (S + 15) >> 4;
  constexpr int nj = E >> 4;                 // 16-wide k chunks of the in-projection
  const float qscale = rsqrtf((float)D);
  for (int item = wave; item < nmb * NBT; item += NW) {
    const int mb = item / NBT, nb = item - mb * NBT;
    const int arow = min(mb * 16 + r16, S - 1);
    const float *ap = Xs + arow * XS + 4 * kq;
    const float *bp = Wi + (nb * 16 + r16) * XS + 4 * kq;
    f32x4 af[nj], bf[nj];
#pragma unroll
    for (int j = 0; j < nj; ++j) {
      af[j] = *reinterpret_cast<const f32x4 *>(ap + 16 * j);
      bf[j] = *reinterpret_cast<const f32x4 *>(bp + 16 * j);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < nj; ++j)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][jj], bf[j][jj], acc, 0, 0, 0);
    const int which = nb / NBQ, c = (nb % NBQ) * 16 + r16;
    float *dst = which == 0 ? Qs : (which == 1 ? Ks : Vs);
    const float bias = bi[nb * 16 + r16];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = mb * 16 + 4 * kq + reg;
      float v = acc[reg] + bias;
      if (which == 0) v *= qscale;
      if (c >= D) v = 0.f;                    // padding columns of the k chunk (D < 16)
      if (row < S) dst[row * DS + c] = v;
    }
  }
  __syncthreads();

  // ---- 3. softmax(q k^T) v: a query row = PARTS lanes x ATT_KS key slices, merged in slice order ------
  {
    constexpr int LPR = PARTS * ATT_KS;       // lanes per query row
    constexpr int ROWS = NT / LPR;
    float *mrg = Xs;                          // x and W_in are dead: [min(S, ROWS)][KS][PARTS][12] merge scratch (m, l, -, -, o[8])
    const int part = tid % PARTS, ksl = (tid / PARTS) % ATT_KS, prow = tid / LPR;
    const int kper = (S + ATT_KS - 1) / ATT_KS;
    const int k0 = ksl * kper, k1 = min(S, k0 + kper);
    for (int row0 = 0; row0 < S; row0 += ROWS) {
      const int row = row0 + prow;
      const int rowc = row < S ? row : S - 1;  // surplus lanes shadow the last row (shuffles stay convergent)
      const f32x4 q0 = *reinterpret_cast<const f32x4 *>(Qs + rowc * DS + 8 * part);
      const f32x4 q1 = *reinterpret_cast<const f32x4 *>(Qs + rowc * DS + 8 * part + 4);
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
      float mx = -3.0e38f, l = 0.f;
      for (int j = k0; j < k1; ++j) {
        const f32x4 kv0 = *reinterpret_cast<const f32x4 *>(Ks + j * DS + 8 * part);
        const f32x4 kv1 = *reinterpret_cast<const f32x4 *>(Ks + j * DS + 8 * part + 4);
        // two independent half-sums: half the dependent-FMA chain of a single accumulator
        float sa = q0[0] * kv0[0], sb = q1[0] * kv1[0];
        sa = fmaf(q0[1], kv0[1], sa); sb = fmaf(q1[1], kv1[1], sb);
        sa = fmaf(q0[2], kv0[2], sa); sb = fmaf(q1[2], kv1[2], sb);
        sa = fmaf(q0[3], kv0[3], sa); sb = fmaf(q1[3], kv1[3], sb);
        float sc = sa + sb;
#pragma unroll
        for (int m = 1; m < PARTS; m <<= 1) sc += __shfl_xor(sc, m);
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(Vs + j * DS + 8 * part);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(Vs + j * DS + 8 * part + 4);
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
      // merge the key slices of a row in slice order (every q row of this pass has been read by now)
      __syncthreads();
      if (row < S) {
        float *mp = mrg + ((prow * ATT_KS + ksl) * PARTS + part) * 12;
        mp[0] = mx; mp[1] = l;
        *reinterpret_cast<f32x4 *>(mp + 4) = o0;
        *reinterpret_cast<f32x4 *>(mp + 8) = o1;
      }
      __syncthreads();
      if (ksl == 0 && row < S) {
        float M = -3.0e38f;
#pragma unroll
        for (int s2i = 0; s2i < ATT_KS; ++s2i) M = fmaxf(M, mrg[((prow * ATT_KS + s2i) * PARTS + part) * 12]);
        float L = 0.f;
        f32x4 O0 = {0.f, 0.f, 0.f, 0.f}, O1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2i = 0; s2i < ATT_KS; ++s2i) {
          const float *mp = mrg + ((prow * ATT_KS + s2i) * PARTS + part) * 12;
          const float w = (mp[1] > 0.f) ? __expf(mp[0] - M) : 0.f;   // an empty slice (l = 0) contributes nothing
          L += mp[1] * w;
          O0 += *reinterpret_cast<const f32x4 *>(mp + 4) * w;
          O1 += *reinterpret_cast<const f32x4 *>(mp + 8) * w;
        }
        const float inv = 1.0f / L;
        *reinterpret_cast<f32x4 *>(Qs + row * DS + 8 * part) = O0 * inv;
        *reinterpret_cast<f32x4 *>(Qs + row * DS + 8 * part + 4) = O1 * inv;
      }
      __syncthreads();
    }
  }

  // ---- 4. partial out-projection of this head: one 16 x 16 block per wave and round -------------------
  {
    constexpr int NJ = KD / 16;
    constexpr int nbe = E >> 4;              // 16-column blocks of the output
    float *pout = a.part + ((size_t)h * a.B + b) * (size_t)S * E;
    for (int item = wave; item < nmb * nbe; item += NW) {
      const int mb = item / nbe, nb = item - mb * nbe;
      const int arow = min(mb * 16 + r16, S - 1);
      f32x4 af[NJ], bf[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        af[j] = *reinterpret_cast<const f32x4 *>(Qs + arow * DS + 16 * j + 4 * kq);
        bf[j] = *reinterpret_cast<const f32x4 *>(Wo + (nb * 16 + r16) * DS + 16 * j + 4 * kq);
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][jj], bf[j][jj], acc, 0, 0, 0);
      const int n = nb * 16 + r16;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = mb * 16 + 4 * kq + reg;
        if (row < S) pout[(size_t)row * E + n] = acc[reg];
      }
    }
  }
}

size_t attn_block_lds_bytes(int S, int E, int heads) {
  const int D = E / heads, KD = D < 16 ? 16 : D, NBT = 3 * ((D + 15) / 16);
  // (the softmax merge scratch, S * (D/8) * ATT_KS * 12 floats, overlays x and W_in, which are dead by then)
  return ((size_t)S * (E + 8) + (size_t)NBT * 16 * (E + 8) + (size_t)E * (KD + 4) + (size_t)3 * S * (KD + 4) + (size_t)NBT * 16 +
          (size_t)(ATT_NT / (E / 4)) * 8 + 8) * sizeof(float);
}

bool attn_block_ok(int S, int E, int heads, int groups) {
  if (heads != 4 || E % heads) return false;                      // the kernel's E = 4 * D is a compile-time constant
  const int D = E / heads;
  if (D != 8 && D != 16 && D != 32) return false;
  if (groups != 8 || E % (4 * groups)) return false;              // group = whole channel quads, 8 statistics lanes
  return S >= 1 && attn_block_lds_bytes(S, E, heads) <= 160 * 1024;
}

hipError_t launch_attn_block(const AttnBlockArgs &a, hipStream_t st) {
  if (!attn_block_ok(a.S, a.E, a.heads, a.groups)) return hipErrorInvalidValue;
  const int D = a.E / a.heads;
  const size_t smem = attn_block_lds_bytes(a.S, a.E, a.heads);
#define CM_ATTNB(DD)                                                                                        \
  if (D == DD) {                                                                                            \
    static bool attr_set[64] = {false};                                                                     \
    int dev = 0;                                                                                            \
    (void)hipGetDevice(&dev);                                                                               \
    if (!attr_set[dev & 63]) {                                                                              \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_head_kernel<DD>),              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);           \
      if (e != hipSuccess) return e;                                                                        \
      attr_set[dev & 63] = true;                                                                            \
    }                                                                                                       \
    hipLaunchKernelGGL((attn_head_kernel<DD>), dim3(a.heads, a.B), dim3(ATT_NT), smem, st, a);                 \
    return hipGetLastError();                                                                               \
  }
  CM_ATTNB(8) CM_ATTNB(16) CM_ATTNB(32)
#undef CM_ATTNB
  return hipErrorInvalidValue;
}

}  // namespace cm
