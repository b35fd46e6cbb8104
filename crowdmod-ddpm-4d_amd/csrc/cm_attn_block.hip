// Fused AttentionBlock (reference: /root/reference/models/backbones/layers.py:5-18):
//     x + MHA(GroupNorm(8, E)(x))          over S = Z*Y*X tokens, E channels, 4 heads of D = E/4
// as ONE launch per (head, sample) plus the K-split combine pass, instead of the five launches of the
// generic plan (GroupNorm finalise, in-projection conv, attention core, out-projection conv + the
// finalise in front of it).  At 54-216 tokens per sample these launches are latency chains: the whole
// block is 0.8 % of the UNet's FLOPs.
//
// One 256-thread workgroup = (head h, sample b):
//   1. x[b] (S x E, channels-last) -> LDS; GroupNorm statistics of all 8 groups straight from that copy
//      (two passes, fixed reduction order), normalised in place with the affine folded in;
//   2. [q|k|v]_h = xn * W_in[h-slices]^T + b_in on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32):
//      wave w owns the 16-row blocks w, w+4, ...; A fragments come from LDS, B fragments straight from the
//      REFERENCE weight layout [3E][E] (a lane reads 16 contiguous bytes of one weight row: with the k
//      values of an MFMA chosen as {16j + 4*kq + jj} both operands are plain float4 loads, no packing);
//      q is pre-scaled by 1/sqrt(D);
//   3. softmax(q k^T) v per query row on the vector ALUs (online softmax, K / V rows broadcast from LDS),
//      written over q;
//   4. partial out-projection  y_h = o_h * W_out[:, h*D:(h+1)*D]^T  (S x E) on the matrix cores into
//      part[h][b]; ksplit_combine_kernel then sums the heads in a fixed order, adds the out-proj bias
//      and the residual x and emits the GroupNorm statistics of the result (cm_misc.hip).
// Nothing is reduced with atomics; sample b never touches another sample's data, so a chain's result
// does not depend on the batch it runs in.
#include "cm_kernels.h"

namespace cm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(256) void attn_head_kernel(const AttnBlockArgs a) {
  constexpr int KD = D < 16 ? 16 : D;        // head dim padded to one 16-wide k chunk (zeros beyond D)
  constexpr int DS = KD + 4;                 // LDS row stride of q / k / v
  constexpr int NBQ = (D + 15) / 16;         // 16-column blocks per q / k / v
  constexpr int NBT = 3 * NBQ;
  constexpr int PARTS = D >= 8 ? D / 8 : 1;  // lanes per query row in the softmax phase (8 dims each)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int S = a.S, E = a.E;
  const int XS = E + 8;                      // LDS row stride of x: conflict-free ds_read_b128 of the A fragments
  float *Xs = sm;                            // [S][XS]
  float *Qs = Xs + (size_t)S * XS;           // [S][DS]   (later: the attention output o_h)
  float *Ks = Qs + (size_t)S * DS;
  float *Vs = Ks + (size_t)S * DS;
  float *red = Vs + (size_t)S * DS;          // [256] + [16] reduction scratch
  float *gst = red + 256;                    // [8] group mean, [8] group rstd
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;

  // ---- 1. stage x[b] and normalise it ------------------------------------------------
  const int Q4 = E >> 2;                     // channel quads; 256 % Q4 == 0 (checked on the host)
  const int RL = 256 / Q4;                   // row lanes
  const int q4 = tid % Q4, rl = tid / Q4;
  const float *xb = a.x + (size_t)b * S * E;
  float s1 = 0.f;
  for (int row = rl; row < S; row += RL) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + (size_t)row * E + 4 * q4);
    *reinterpret_cast<f32x4 *>(Xs + row * XS + 4 * q4) = v;
    s1 += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const int cg = E / a.groups;               // channels per group (multiple of 4)
  const int qpg = cg >> 2;                   // quads per group
  const int grp = q4 / qpg;
  // entries of group g in `red`: threads (rl, q) with q in [g*qpg, (g+1)*qpg): RL * qpg = 32 of them
  auto group_sum = [&](float v) -> float {   // returns the group total to every thread (fixed order)
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    if (tid < a.groups) {
      float t = 0.f;
      for (int r = 0; r < RL; ++r)
        for (int q = 0; q < qpg; ++q) t += red[r * Q4 + tid * qpg + q];
      red[256 + tid] = t;
    }
    __syncthreads();
    return red[256 + grp];
  };
  const float cnt = (float)cg * (float)S;
  const float mean = group_sum(s1) / cnt;
  float s2 = 0.f;
  for (int row = rl; row < S; row += RL) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(Xs + row * XS + 4 * q4);
    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  const float rstd = rsqrtf(group_sum(s2) / cnt + a.eps);   // biased variance, layers.py:9 (nn.GroupNorm)
  {
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(a.gamma + 4 * q4);
    const f32x4 be = *reinterpret_cast<const f32x4 *>(a.beta + 4 * q4);
    const f32x4 sc = ga * rstd;
    const f32x4 sh = be - sc * mean;
    for (int row = rl; row < S; row += RL) {
      f32x4 *p = reinterpret_cast<f32x4 *>(Xs + row * XS + 4 * q4);
      *p = *p * sc + sh;
    }
  }
  __syncthreads();

  // ---- 2. q, k, v of this head ---------------------------------------------------------
  const int nmb = (S + 15) >> 4;
  const int nj = E >> 4;                     // 16-wide k chunks of the in-projection
  const float qscale = rsqrtf((float)D);
  for (int mb = wave; mb < nmb; mb += 4) {
    const int arow = min(mb * 16 + r16, S - 1);
    const float *ap = Xs + arow * XS + 4 * kq;
    const float *bp[NBT];
    int ncol[NBT];
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb) {
      const int which = nb / NBQ, c = (nb % NBQ) * 16 + r16;
      const int n = min(which * E + h * D + c, 3 * E - 1);
      ncol[nb] = n;
      bp[nb] = a.w_in + (size_t)n * E + 4 * kq;
    }
    f32x4 acc[NBT];
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bf[NBT], bn[NBT];
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb) bf[nb] = *reinterpret_cast<const f32x4 *>(bp[nb]);
    for (int j = 0; j < nj; ++j) {
      const f32x4 af = *reinterpret_cast<const f32x4 *>(ap + 16 * j);
      const int jn = (j + 1 < nj) ? j + 1 : j;
#pragma unroll
      for (int nb = 0; nb < NBT; ++nb) bn[nb] = *reinterpret_cast<const f32x4 *>(bp[nb] + 16 * jn);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int nb = 0; nb < NBT; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[jj], bf[nb][jj], acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NBT; ++nb) bf[nb] = bn[nb];
    }
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb) {
      const int which = nb / NBQ, c = (nb % NBQ) * 16 + r16;
      float *dst = which == 0 ? Qs : (which == 1 ? Ks : Vs);
      const float bias = a.b_in[ncol[nb]];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = mb * 16 + 4 * kq + reg;
        float v = acc[nb][reg] + bias;
        if (which == 0) v *= qscale;
        if (c >= D) v = 0.f;                  // padding columns of the k chunk (D < 16)
        if (row < S && c < KD) dst[row * DS + c] = v;
      }
    }
  }
  __syncthreads();

  // ---- 3. softmax(q k^T) v, one query row per PARTS lanes ----------------------------------
  {
    constexpr int ROWS = 256 / PARTS;
    const int part = tid % PARTS, prow = tid / PARTS;
    for (int row0 = 0; row0 < S; row0 += ROWS) {
      const int row = row0 + prow;
      const int rowc = row < S ? row : S - 1;  // surplus lanes shadow the last row (shuffles stay convergent)
      f32x4 q0, q1;
      if constexpr (D >= 8) {
        q0 = *reinterpret_cast<const f32x4 *>(Qs + rowc * DS + 8 * part);
        q1 = *reinterpret_cast<const f32x4 *>(Qs + rowc * DS + 8 * part + 4);
      } else {
        q0 = *reinterpret_cast<const f32x4 *>(Qs + rowc * DS);
        q1 = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
      float mx = -3.0e38f, l = 0.f;
      for (int j = 0; j < S; ++j) {
        const f32x4 k0 = *reinterpret_cast<const f32x4 *>(Ks + j * DS + 8 * part);
        const f32x4 k1 = *reinterpret_cast<const f32x4 *>(Ks + j * DS + 8 * part + 4);
        float sc = q0[0] * k0[0];
        sc = fmaf(q0[1], k0[1], sc); sc = fmaf(q0[2], k0[2], sc); sc = fmaf(q0[3], k0[3], sc);
        sc = fmaf(q1[0], k1[0], sc); sc = fmaf(q1[1], k1[1], sc); sc = fmaf(q1[2], k1[2], sc); sc = fmaf(q1[3], k1[3], sc);
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
      __syncthreads();                        // every q row of this pass has been read: o may overwrite q
      if (row < S) {
        const float inv = 1.0f / l;
        *reinterpret_cast<f32x4 *>(Qs + row * DS + 8 * part) = o0 * inv;
        if constexpr (D >= 8) *reinterpret_cast<f32x4 *>(Qs + row * DS + 8 * part + 4) = o1 * inv;
      }
    }
  }
  __syncthreads();

  // ---- 4. partial out-projection of this head ----------------------------------------------
  {
    constexpr int NJ = KD / 16;
    const int nbe = E >> 4;                  // 16-column blocks of the output (<= 16, checked on the host)
    float *pout = a.part + ((size_t)h * a.B + b) * (size_t)S * E;
    for (int mb = wave; mb < nmb; mb += 4) {
      const int arow = min(mb * 16 + r16, S - 1);
      f32x4 af[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) af[j] = *reinterpret_cast<const f32x4 *>(Qs + arow * DS + 16 * j + 4 * kq);
      for (int nb0 = 0; nb0 < nbe; nb0 += 4) {
        f32x4 acc[4];
        f32x4 bf[4][NJ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int n = min((nb0 + u) * 16 + r16, E - 1);
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            // k columns beyond this head's D hold zeros in o: any in-range weight address will do there
            const int col = min(h * D + 16 * j + 4 * kq, E - 4);
            bf[u][j] = *reinterpret_cast<const f32x4 *>(a.w_out + (size_t)n * E + col);
          }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int u = 0; u < 4; ++u)
              acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][jj], bf[u][j][jj], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = (nb0 + u) * 16 + r16;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = mb * 16 + 4 * kq + reg;
            if (row < S && nb0 + u < nbe) pout[(size_t)row * E + n] = acc[u][reg];
          }
        }
      }
    }
  }
}

size_t attn_block_lds_bytes(int S, int E, int heads) {
  const int D = E / heads, KD = D < 16 ? 16 : D;
  return ((size_t)S * (E + 8) + (size_t)3 * S * (KD + 4) + 256 + 16 + 16) * sizeof(float);
}

bool attn_block_ok(int S, int E, int heads, int groups) {
  if (heads < 1 || E % heads) return false;
  const int D = E / heads;
  if (D != 8 && D != 16 && D != 32 && D != 64) return false;
  if (E % 16 || E > 256 || 256 % (E / 4)) return false;          // quads tile the workgroup; combine handles C <= 256
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
    hipLaunchKernelGGL((attn_head_kernel<DD>), dim3(a.heads, a.B), dim3(256), smem, st, a);                 \
    return hipGetLastError();                                                                               \
  }
  CM_ATTNB(8) CM_ATTNB(16) CM_ATTNB(32) CM_ATTNB(64)
#undef CM_ATTNB
  return hipErrorInvalidValue;
}

}  // namespace cm
