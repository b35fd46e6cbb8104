// nn.Upsample(x2, nearest) + 3x3x3 conv (UpSample.forward, /root/reference/models/backbones/layers.py:93-96) in its parity
// form -- output voxel u = 2 i + p reads the 2x2x2 source voxels i + e + p - 1 with host-summed weights (cm_model.cpp:
// parity_weights) -- with the source tile staged ONCE for four parity classes.
//
// The generic kernel (cm_conv.hip, FAST = 8) gives every parity class its own workgroup: eight workgroups stage the same
// low-resolution halo box, and its four waves split K, so each adds a cross-wave reduction (round 2 profile: 4.9 vector
// instructions per matrix instruction, matrix pipe 0.57 busy on the 64 -> 64 layer).  Here
//   * workgroup = (sample, source tile, p_z, 32 NB output channels); its four waves ARE the four (p_y, p_x) classes: they
//     read the same staged box at lane offsets (p_y, p_x) and own disjoint outputs -- no reduction, no shared accumulators;
//   * per 32-channel chunk the box of (TZ + 1) x (TY + 2) x (TX + 2) source voxels is copied to LDS once (the input of an
//     upsample conv is a raw block output: no normalisation on load), then every wave runs 8 taps x 4 channel groups of
//     MBW x NB x 4 exact-fp32 matrix instructions with its class's weights (register ring, refilled after use);
//   * `planes` tiles (TY TX <= 32, row block = source plane): a (row block, z tap) pair that reads the zero padding plane
//     (first plane with e_z = 0 for p_z = 0, last plane with e_z = 1 for p_z = 1) is never issued -- the products are exact
//     zeros, so the result is bit-identical to the full form;
//   * epilogue per wave: bias, channels-last store of its class's voxels, GroupNorm statistics per (row block, channel) in
//     the slot format of gn_finalize.
// Accumulation order per output: chunks ascending, taps (e_z, e_y, e_x) ascending, channels ascending -- the generic kernel
// sums its four waves' channel groups at the end instead, so the two agree to fp32 rounding, not bit for bit.
#include "cm_kernels.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

namespace cm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// tabH[(p * 2 + pz)][HV]: in-sample source voxel of halo voxel h of tile position p for class p_z, or -1 (zero padding);
// tabM[p][32 MBW][2]: row m -> halo index of its source voxel at (e = 0, p_y = p_x = 0) | packed global source coordinates
// (Z << 20 | Y << 10 | X) or -1 (padding row).
template <int MBW, int NB, int OCC, int PREC = 0>
__global__ __launch_bounds__(256, OCC) void conv_ups_kernel(const ConvArgs a, const int *__restrict__ tabH, const int *__restrict__ tabM,
                                                         int HV, int ntp, int HX, int HYX, int planes, int NBP) {
  // F16 (reduced-precision plan, cm_model_set_precision): the staged box is rounded to f16 (row = 32 halves + 8 pad halves), the
  // weights arrive as f16 fragments of 8 halves (pack_ups_f16: one column block per n tile), and a step is ONE
  // v_mfma_f32_32x32x16_f16 per (row block, column block) over 16 channels -- 16 steps per chunk; fp32 accumulation
  // PREC = 2 (B6): fp32 products formed on the bf16 matrix instruction from exact three-way splits x = hi + mid + lo
  // (8 + 8 + 8 mantissa bits: the split loses nothing).  Six cross terms per product -- hi*lo, lo*hi, mid*mid, hi*mid, mid*hi,
  // hi*hi; the three dropped ones are <= 2^-24 of the product -- accumulate in fp32: the error against an fp64 reference is
  // fp32-level (tools/sim_six_term.py, K = 512, one rounding per product as the worst case: rms 7.9e-7 of the result's rms against
  // 4.1e-7 of the k-ordered fp32 chain, three terms 4.4e-6; measured: whole-forward error against the reference 2.1e-6, exact-fp32 path 2.3e-6),
  // and 16 channels take 6 x 32 cycles instead of 8 x 64.  The staged box is split once per voxel at the LDS write (three
  // bf16 planes per row), the weights arrive pre-split (pack_ups_b6).
  // PREC = 4 (h2, default plan): f16 two-way splits, three cross terms (cm_kernels.h: cm_split2_f16).  The source of this conv is RAW
  // (the block output, not a GroupNorm output), so its range is taken from the data: the producer's slot statistics bound every element
  // of the sample (cm_h2_sample_scale), the staged values are multiplied by that power of two before the split and the accumulators by
  // its inverse (and by the weight scale's) in the epilogue -- a per-sample block exponent, exact.
  constexpr bool F16 = PREC == 1, B6 = PREC >= 2, H2 = PREC == 4;   // 3: relaxed plan -- three cross terms on the six-term fragments
  constexpr int U0 = PREC >= 3 ? 3 : 0, NTW = PREC >= 3 ? 2 : 3;
  constexpr int NTM = B6 ? 3 : 1;                // operand terms
  constexpr int S = B6 ? 52 : (F16 ? 20 : 36);   // LDS row stride in dwords: 32 channels (x 3 bf16 planes) + pad
  constexpr int NST = PREC ? 16 : 32;            // steps per 32-channel chunk
  constexpr int NLD = 10;                        // halo items (voxel, channel quad) per thread and chunk: 8 HV / 256 <= 10 (HV <= 320)
  constexpr int RD = B6 ? 2 : 4;                 // weight ring depth in (tap, channel group) steps; NST % RD == 0
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *A = lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  int tile = blockIdx.x;
  if (!(gridDim.x & 7) && !(a.dbg & 4096)) tile = (tile & 7) * (int)(gridDim.x >> 3) + (tile >> 3);   // XCD-aware order (cm_conv.hip)
  const int b = tile / ntp, p = tile - b * ntp;
  const int nt = blockIdx.y, pz = blockIdx.z;
  const int py = wave >> 1, px = wave & 1;
  const int par = (pz * 2 + py) * 2 + px;
  const unsigned Vs = (unsigned)(a.Zs * a.Ys * a.Xs), Vo = (unsigned)(a.Zo * a.Yo * a.Xo);

  // ---- geometry from the host tables -------------------------------------------------------------------------------
  const int nit = (8 * HV + 255) >> 8;
  const int q = tid & 7;
  int hoff[NLD];
  unsigned hok = 0;
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int h = (tid >> 3) + 32 * k;
    const int o = (k < nit && h < HV) ? tabH[(size_t)(p * 2 + pz) * HV + h] : -1;
    hok |= (o >= 0 ? 1u : 0u) << k;
    hoff[k] = o >= 0 ? o : 0;
  }
  int abase[MBW], ovox[MBW];
#pragma unroll
  for (int j = 0; j < MBW; ++j) {
    const int m = j * 32 + r;
    const int t0 = tabM[((size_t)p * 32 * MBW + m) * 2], t1 = tabM[((size_t)p * 32 * MBW + m) * 2 + 1];
    abase[j] = (t0 + py * HX + px) * S + 4 * hh;
    const int Z = t1 >> 20, Y = (t1 >> 10) & 1023, X = t1 & 1023;
    ovox[j] = t1 >= 0 ? ((2 * Z + pz) * a.Yo + 2 * Y + py) * a.Xo + 2 * X + px : -1;
  }
  // planes tiles whose box spans the whole Z extent: (row block j, z tap ez) reads source plane j + ez + pz - 1; the pair that
  // lands outside the grid -- (0, 0) for p_z = 0, (MBW - 1, 1) for p_z = 1 -- multiplies zeros and is not issued.  The three
  // cases are three instantiations of the loop body (a run-time test around matrix instructions makes the compiler copy the
  // accumulators at every join: 2 000 register moves per chunk in the first version of this kernel)
  const int mode = (planes && a.bz == a.Zs) ? 1 + pz : 0;
  const int nch = a.C0 >> 5;
  float sa = 1.f;                                  // h2: this sample's activation scale (a power of two)
  if constexpr (H2) sa = cm_h2_sample_scale(a.gp0, a.gc0, a.gns0, a.C0, b, A, tid, 256);

  auto body = [&](auto mode_c) {
  constexpr int MODE = decltype(mode_c)::value;
  f32x16 acc[MBW][NB];
#pragma unroll
  for (int j = 0; j < MBW; ++j)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][nb][e] = 0.f;

  // weights of this wave's class: pack_conv_weights order [n tile][chunk][step = tap * 4 + k8][NBP][lane] 16 B
  const int c32 = nt * NB;                        // first 32-channel output block of this workgroup
  const int nsteps = nch * NST;
  // fp32: column block nb of a packed n tile sits 64 fragments after nb - 1, a step NBP * 64 after the previous one;
  // f16 / B6:  [column block][step][term][lane]
  const int wstep = PREC ? NTM * 64 : NBP * 64, wnb = PREC ? nsteps * NTM * 64 : 64;
  const f32x4 *wbase = reinterpret_cast<const f32x4 *>(a.wfrag + (size_t)par * a.wpar_stride) +
                       (PREC ? (size_t)c32 * nsteps * NTM * 64 : ((size_t)(c32 / NBP) * nch * 32 * NBP + (c32 % NBP)) * 64) + lane;
  f32x4 bw[RD][NTM][NB];
#pragma unroll
  for (int s = 0; s < RD; ++s)
#pragma unroll
    for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bw[s][tm][nb] = wbase[(size_t)s * wstep + (size_t)tm * 64 + (size_t)nb * wnb];
  const f32x4 *wrun = wbase + (size_t)RD * wstep;      // next refill; advanced one step at a time (an address per step, hoisted,
                                                       //  is 64 registers)
  const int n = nt * 32 * NB + r;                 // (+ 32 nb)
  float bias_pre[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bias_pre[nb] = a.bias[n + 32 * nb < a.Co ? n + 32 * nb : 0];

  const float *const srcb = a.src0 + (size_t)b * Vs * a.C0 + 4 * q;
  const unsigned cb = (unsigned)a.C0 * 4u;
  for (int ch = 0; ch < nch; ++ch) {
    if (ch) __syncthreads();                      // every wave has read the previous chunk
    // ---- stage: raw copy of the box's 32 channels ----------------------------------------------------------------------
#pragma unroll
    for (int half = 0; half < 2; ++half) {          // two batches of <= 5 loads: the ten together are the register peak
      f32x4 ld[NLD / 2];
#pragma unroll
      for (int kk = 0; kk < NLD / 2; ++kk) {
        const int k = half * (NLD / 2) + kk;
        if (k < nit) {
          if constexpr (F16) {
            if (a.h16 & 1) {                        // f16 source tensor: 4 halves per item
              const char *sh = reinterpret_cast<const char *>(a.src0) + ((size_t)b * Vs * a.C0 + 4 * q + ch * 32) * 2;
              const cm_f32x2_t two = *reinterpret_cast<const cm_f32x2_t *>(sh + __umul24((unsigned)hoff[k], cb >> 1));
              const f16x4 hv4 = __builtin_bit_cast(f16x4, two);
              ld[kk] = f32x4{(float)hv4[0], (float)hv4[1], (float)hv4[2], (float)hv4[3]};
              continue;
            }
          }
          ld[kk] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(srcb + ch * 32) + __umul24((unsigned)hoff[k], cb));
        }
      }
#pragma unroll
      for (int kk = 0; kk < NLD / 2; ++kk) {
        const int k = half * (NLD / 2) + kk;
        if (k < nit) {
          const int h = (tid >> 3) + 32 * k;
          const f32x4 w = ((hok >> k) & 1u) ? ld[kk] : f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (B6) {
            cm_u32x2_t t3[3];
            if constexpr (H2) cm_split2_f16(w * sa, t3);   // f16 hi / mid of x 2^k
            else cm_split3_bf16<NTW>(w, t3);        // hi / mid / lo planes, exact remainders
            if (h < HV) {
#pragma unroll
              for (int tm = 0; tm < NTW; ++tm) *reinterpret_cast<cm_u32x2_t *>(A + (size_t)h * S + 16 * tm + 2 * q) = t3[tm];
            }
          } else if constexpr (F16) {
            const f16x4 hv = {(_Float16)w[0], (_Float16)w[1], (_Float16)w[2], (_Float16)w[3]};
            if (h < HV) *reinterpret_cast<f16x4 *>(A + (size_t)h * S + 2 * q) = hv;
          } else {
            if (h < HV) *reinterpret_cast<f32x4 *>(A + (size_t)h * S + 4 * q) = w;
          }
        }
      }
      asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (a.dbg & 2) continue;
    // ---- matrix phase: 8 taps x 4 channel groups ----------------------------------------------------------------------
    f32x4 afr[2][NTM][MBW];
#pragma unroll
    for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
      for (int j = 0; j < MBW; ++j) afr[0][tm][j] = *reinterpret_cast<const f32x4 *>(A + abase[j] + 16 * tm);
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      // fp32: step = (tap, 8-channel group k8); f16 / B6: step = (tap, 16-channel group)
      const int t = PREC ? s >> 1 : s >> 2, ez = t >> 2;
      if (s + 1 < NST) {
        const int s1 = s + 1, t1 = PREC ? s1 >> 1 : s1 >> 2, kg1 = PREC ? s1 & 1 : s1 & 3;
        int toff = (((t1 >> 2) * HYX) + ((t1 >> 1) & 1) * HX + (t1 & 1)) * S + 8 * kg1;
        asm volatile("" : "+s"(toff));           // one address add per read, HERE (hoisted, the 32 x MBW sums cost 100+ registers)
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
          for (int j = 0; j < MBW; ++j) afr[s1 & 1][tm][j] = *reinterpret_cast<const f32x4 *>(A + abase[j] + toff + 16 * tm);
      }
#pragma unroll
      for (int j = 0; j < MBW; ++j) {
        if ((MODE == 1 && j == 0 && ez == 0) || (MODE == 2 && j == MBW - 1 && ez == 1)) continue;   // (compile-time)
        if constexpr (B6) {
          // (A term, B term), small products first: hi = 0, mid = 1, lo = 2
          constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int u = U0; u < 6; ++u) {
              if constexpr (H2)
                acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[s & 1][TA[u]][j]),
                                                                    __builtin_bit_cast(f16x8, bw[s % RD][TB[u]][nb]), acc[j][nb], 0, 0, 0);
              else
                acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[s & 1][TA[u]][j]),
                                                                     __builtin_bit_cast(bf16x8, bw[s % RD][TB[u]][nb]), acc[j][nb], 0, 0, 0);
            }
        } else if constexpr (F16) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[s & 1][0][j]), __builtin_bit_cast(f16x8, bw[s % RD][0][nb]),
                                                                acc[j][nb], 0, 0, 0);
        } else {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[s & 1][0][j][jj], bw[s % RD][0][nb][jj], acc[j][nb], 0, 0, 0);
        }
      }
      // refill this ring slot AFTER the matrix instructions that read it; the fence keeps the request here (hipcc would
      // sink it to its first use, and every step would pay an L2 round trip)
      {
        const int g = ch * NST + s + RD;
        if (g < nsteps) {
#pragma unroll
          for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bw[s % RD][tm][nb] = wrun[(size_t)tm * 64 + (size_t)nb * wnb];
        }
        wrun += wstep;
        asm volatile("" : "+v"(wrun));
      }
      asm volatile("" ::: "memory");
    }
  }

  // ---- epilogue: lane (r, hh) of block (j, nb) holds rows (e & 3) + 8 (e >> 2) + 4 hh, channel n + 32 nb ---------------------
  if (a.dbg & 4) {
    if (acc[0][0][0] == 123.456f) a.out[0] = 1.f;
    return;
  }
  float *const outb = a.out + (size_t)b * Vo * a.out_cs;
  const int ns = ntp * 8 * MBW;
  const float osc = H2 ? a.h2_oscale * (1.0f / sa) : 1.f;    // (both factors powers of two: exact)
#pragma unroll
  for (int j = 0; j < MBW; ++j) {
    int orow[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) orow[e] = __shfl(ovox[j], (e & 3) + 8 * (e >> 2) + 4 * hh);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int nn = n + 32 * nb;
      const bool nok = nn < a.Co;
      float rs[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) rs[e] = (H2 ? acc[j][nb][e] * osc : acc[j][nb][e]) + bias_pre[nb];
      if (F16 && (a.h16 & 4)) {
        _Float16 *oh = reinterpret_cast<_Float16 *>(a.out) + (size_t)b * Vo * a.out_cs;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (nok && orow[e] >= 0) oh[(size_t)orow[e] * a.out_cs + nn] = (_Float16)rs[e];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (nok && orow[e] >= 0) outb[(size_t)orow[e] * a.out_cs + nn] = rs[e];
      }
      if (a.stat_part || a.astat) {
        float s1 = 0.f, cnt = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (orow[e] >= 0) { s1 += rs[e]; cnt += 1.f; }
        s1 += __shfl_xor(s1, 32);
        cnt += __shfl_xor(cnt, 32);
        const float mean = cnt > 0.f ? s1 / cnt : 0.f;
        float q2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (orow[e] >= 0) { const float dd = rs[e] - mean; q2 += dd * dd; }
        q2 += __shfl_xor(q2, 32);
        const int slot = ((p * 2 + pz) * 4 + wave) * MBW + j;
        if (a.astat) {
          if (hh == 0 && nok && cnt > 0.f) cm_stat_atomic(a.astat + ((size_t)b * a.astat_C + nn) * 3, s1, mean, q2);
        } else {
          if (hh == 0 && nok) {
            float *sp2 = a.stat_part + (((size_t)b * ns + slot) * a.stat_C + nn) * 2;
            sp2[0] = mean;
            sp2[1] = q2;
          }
          if (lane == 0 && nn == 0) a.stat_cnt[(size_t)b * ns + slot] = cnt;
        }
      }
    }
  }
  };
  if (mode == 0) body(std::integral_constant<int, 0>{});
  else if (mode == 1) body(std::integral_constant<int, 1>{});
  else body(std::integral_constant<int, 2>{});
}

// After an optimizer step the fp32 parity fragments (pack_conv_weights order, re-derived on the device by gather_pack) are the
// source of truth: split them again into the hi / mid / lo planes of pack_ups_b6.  One thread per (class, co, ci, tap).
__global__ __launch_bounds__(256) void ups_b6_repack_kernel(const float *__restrict__ wf, long long wpar_stride, unsigned short *__restrict__ w6,
                                                            long long w6_stride_halves, int Co, int Ci, int NBP) {
  const long long n = 8LL * Co * Ci * 8, i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i & 7);
  long long q = i >> 3;
  const int ci = (int)(q % Ci); q /= Ci;
  const int co = (int)(q % Co);
  const int par = (int)(q / Co);
  const int nch = Ci >> 5, ch = ci >> 5, cl = ci & 31;
  // source: [n tile][chunk][step = tap * 4 + k8][NBP][lane][4], ci = chunk * 32 + 8 k8 + 4 hh + jj, co = (n tile * NBP + nb) * 32 + r
  const int c32 = co >> 5, r = co & 31, k8 = cl >> 3, hs = (cl >> 2) & 1, jj = cl & 3;
  const float w = wf[(size_t)par * wpar_stride +
                     ((((size_t)(c32 / NBP) * nch + ch) * 32 + t * 4 + k8) * NBP + (c32 % NBP)) * 256 + (32 * hs + r) * 4 + jj];
  // destination: [column block][chunk][tap][m][term][lane][8], ci = chunk * 32 + 16 m + 8 hh + e
  const int mg = cl >> 4, hd = (cl >> 3) & 1, e = cl & 7;
  unsigned short *dst = w6 + (size_t)par * w6_stride_halves + ((((((size_t)c32 * nch + ch) * 8 + t) * 2 + mg) * 3) * 64 + 32 * hd + r) * 8 + e;
  float rem = w;
#pragma unroll
  for (int tm = 0; tm < 3; ++tm) {
    const __bf16 hb = (__bf16)rem;
    dst[(size_t)tm * 64 * 8] = __builtin_bit_cast(unsigned short, hb);
    rem -= (float)hb;
  }
}

hipError_t launch_ups_b6_repack(const float *wfrag, long long wpar_stride, float *w6, long long w6_stride, int Co, int Ci, int NBP, hipStream_t st) {
  if (Co % 32 || Ci % 32 || (NBP != 1 && NBP != 2)) return hipErrorInvalidValue;
  const long long n = 8LL * Co * Ci * 8;
  hipLaunchKernelGGL(ups_b6_repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wfrag, wpar_stride,
                     reinterpret_cast<unsigned short *>(w6), w6_stride * 2, Co, Ci, NBP);
  return hipGetLastError();
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// Source tile (tz, ty, tx) for a source grid (Z, Y, X): planes tiles (ty tx <= 32, tz <= 4 row blocks, one per plane, padding
// taps skipped when the tile spans the whole Z extent) or linear tiles (tz ty tx rows in <= 5 blocks); the staged box
// (tz + 1)(ty + 2)(tx + 2) must fit 320 voxels.  Score = matrix instructions per useful row (lower is better).
bool conv_ups_pick(int Z, int Y, int X, int *tz, int *ty, int *tx, int *mbw, int *planes) {
  if (const char *e = diag_env("CM_UPS_TILE")) {
    int z = 0, y = 0, x = 0, pl = 0;
    if (sscanf(e, "%d,%d,%d,%d", &z, &y, &x, &pl) == 4 && z > 0 && y > 0 && x > 0 && Z % z == 0 && Y % y == 0 && X % x == 0 &&
        (z + 1) * (y + 2) * (x + 2) <= 320) {
      const int w = pl ? z : (z * y * x + 31) / 32;
      if (w <= 5 && (!pl || y * x <= 32)) { *tz = z; *ty = y; *tx = x; *mbw = w; *planes = pl; return true; }
    }
  }
  double best = 1e30;
  for (int z = 1; z <= Z; ++z)
    for (int y = 1; y <= Y; ++y)
      for (int x = 1; x <= X; ++x) {
        if (Z % z || Y % y || X % x) continue;
        if ((z + 1) * (y + 2) * (x + 2) > 320) continue;
        const int rows = z * y * x;
        for (int pl = 0; pl < 2; ++pl) {
          int w;
          double issued;                             // row blocks x z taps issued per tile and (e_y, e_x, channel group)
          if (pl) {
            if (y * x > 32 || z > 4) continue;
            w = z;
            issued = 2.0 * w - (z == Z ? 1.0 : 0.0);   // one (block, tap) pair is padding for either p_z when the tile spans Z
          } else {
            w = (rows + 31) / 32;
            if (w > 5) continue;
            issued = 2.0 * w;
          }
          if (w < 2 && rows < 48) continue;           // too little matrix work per staged box
          const double cost = issued / rows * (1.0 + 0.02 * (5 - w));   // (slight preference for more work per workgroup)
          if (cost < best) { best = cost; *tz = z; *ty = y; *tx = x; *mbw = w; *planes = pl; }
        }
      }
  return best < 1e29;
}

struct UpsTabs { int *tH = nullptr, *tM = nullptr; int HV = 0, ntp = 0; };
static hipError_t ups_tabs_get(const ConvArgs &a, int mbw, int planes, UpsTabs *out) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int, int, int, int, int, int>, UpsTabs> cache;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const auto key = std::make_tuple(dev, a.Zs, a.Ys, a.Xs, a.bz, a.by, a.bx, mbw, planes);
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it == cache.end()) {
    const int HZ = a.bz + 1, HY = a.by + 2, HX = a.bx + 2, HV = HZ * HY * HX;
    const int ntz = a.Zs / a.bz, nty = a.Ys / a.by, ntx = a.Xs / a.bx, ntp = ntz * nty * ntx;
    const int MR = 32 * mbw, rpp = a.by * a.bx;
    std::vector<int> tH((size_t)ntp * 2 * HV, -1), tM((size_t)ntp * MR * 2, 0);
    for (int tz = 0; tz < ntz; ++tz)
      for (int ty = 0; ty < nty; ++ty)
        for (int tx = 0; tx < ntx; ++tx) {
          const int p = (tz * nty + ty) * ntx + tx;
          const int z0 = tz * a.bz, y0 = ty * a.by, x0 = tx * a.bx;
          for (int pz = 0; pz < 2; ++pz)
            for (int h = 0; h < HV; ++h) {
              const int hz = h / (HY * HX), rem = h % (HY * HX), hy = rem / HX, hx = rem % HX;
              const int cz = z0 + pz - 1 + hz, cy = y0 - 1 + hy, cx = x0 - 1 + hx;     // source voxel i + e + p - 1, e = p = 0 at h = 0
              if (cz >= 0 && cz < a.Zs && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs)
                tH[((size_t)p * 2 + pz) * HV + h] = (cz * a.Ys + cy) * a.Xs + cx;
            }
          for (int m = 0; m < MR; ++m) {
            int z, y, x;
            bool valid;
            if (planes) { z = m / 32; const int i = m % 32; valid = i < rpp; y = valid ? i / a.bx : 0; x = valid ? i % a.bx : 0; }
            else { valid = m < a.bz * rpp; z = valid ? m / rpp : 0; const int i = valid ? m % rpp : 0; y = i / a.bx; x = i % a.bx; }
            tM[((size_t)p * MR + m) * 2] = (z * HY + y) * HX + x;
            tM[((size_t)p * MR + m) * 2 + 1] = valid ? ((z0 + z) << 20) | ((y0 + y) << 10) | (x0 + x) : -1;
          }
        }
    UpsTabs t;
    t.HV = HV; t.ntp = ntp;
    hipError_t e = hipMalloc((void **)&t.tH, tH.size() * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&t.tM, tM.size() * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(t.tH, tH.data(), tH.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t.tM, tM.data(), tM.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    it = cache.emplace(key, t).first;
  }
  *out = it->second;
  return hipSuccess;
}

// `a` describes the parity-form conv as cm_model.cpp builds it (par = 1, ntaps = 8, CK = 32, weights packed with NB = nbp);
// a.bz / by / bx here = the SOURCE tile of conv_ups_pick.  a.f16 = 1: wfrag holds pack_ups_f16 fragments, 2: pack_ups_b6 (nbp ignored)
// 32-channel column blocks per workgroup: one (more, smaller workgroups hide each other's staging and epilogue)
static int ups_nb(const ConvArgs &a, int mbw, int nbp) {
  if (const char *e = diag_env("CM_UPS_NB")) {
    const int v = atoi(e);
    if (v == 2 && mbw <= 2 && (nbp == 2 || a.f16) && a.Co % 64 == 0) return 2;
  }
  return 1;
}
// requested waves per SIMD (= workgroups per CU).  Two: the 4-row-block variant needs 188 registers, and the ATC launches are
// 1024 / 512 workgroups = whole rounds of two per CU (three or four per CU measured the same or slower: 144.6 / 166.9 us vs 144.4)
static int ups_occ(int) { return 2; }

bool conv_ups_ok(const ConvArgs &a, int mbw, int planes, int nbp) {
  return a.par == 1 && a.ntaps == 8 && a.td == 2 && a.CK == 32 && a.C1 == 0 && a.C0 % 32 == 0 && a.Co % 32 == 0 && !a.gn && !a.pm && !a.temb &&
         !a.resid && !a.s2w && a.ks <= 1 && a.Zo == 2 * a.Zs && a.Yo == 2 * a.Ys && a.Xo == 2 * a.Xs && mbw >= 1 && mbw <= 5 &&
         a.bz > 0 && a.by > 0 && a.bx > 0 && a.Zs % a.bz == 0 && a.Ys % a.by == 0 && a.Xs % a.bx == 0 &&
         (a.bz + 1) * (a.by + 2) * (a.bx + 2) <= 320 && (planes ? (a.by * a.bx <= 32 && a.bz == mbw) : a.bz * a.by * a.bx <= 32 * mbw) &&
         (nbp == 1 || nbp == 2) &&
         a.Zo < 512 && a.Yo < 1024 && a.Xo < 1024 && a.ntz == a.Zs / a.bz && a.nty == a.Ys / a.by && a.ntx == a.Xs / a.bx;
}

int conv_ups_slots(const ConvArgs &a, int mbw) { return (a.Zs / a.bz) * (a.Ys / a.by) * (a.Xs / a.bx) * 8 * mbw; }

hipError_t launch_conv_ups(const ConvArgs &a_in, int mbw, int planes, int nbp, hipStream_t st) {
  ConvArgs a = a_in;
  a.dbg = conv_dbg_flags();
  if (!conv_ups_ok(a, mbw, planes, nbp)) return hipErrorInvalidValue;
  UpsTabs tb;
  hipError_t et = ups_tabs_get(a, mbw, planes, &tb);
  if (et != hipSuccess) return et;
  const int nb = ups_nb(a, mbw, nbp), occ = ups_occ(mbw);   // (a workgroup's NB column blocks sit in one packed n tile)
  const int prec = a.f16;                          // 0: fp32 matrix instruction, 1: f16 operands, 2: bf16 x 6 split products, 3: three of the six, 4: h2
  if (prec == 4 && !(a.gp0 && a.gc0 && a.gns0 > 0)) return hipErrorInvalidValue;   // h2 takes the source's range from its slot statistics
  const size_t lds = (size_t)tb.HV * (prec >= 2 ? 52 : (prec == 1 ? 20 : 36)) * sizeof(float);
  const dim3 grid((unsigned)(a.B * tb.ntp), (unsigned)(a.Co / (32 * nb)), 2);
  const int HX = a.bx + 2, HYX = (a.by + 2) * HX;
#define CM_UPS_GO(M, N, O, H)                                                                       \
  if (mbw == M && nb == N && occ == O && prec == H) {                                               \
    static bool attr_set[64] = {false};                                                             \
    int dev = 0;                                                                                    \
    (void)hipGetDevice(&dev);                                                                       \
    if (!attr_set[dev & 63]) {                                                                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_ups_kernel<M, N, O, H>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return e;                                                                \
      attr_set[dev & 63] = true;                                                                    \
    }                                                                                               \
    hipLaunchKernelGGL((conv_ups_kernel<M, N, O, H>), grid, dim3(256), lds, st, a, tb.tH, tb.tM, tb.HV, tb.ntp, HX, HYX, planes, nbp); \
    return hipGetLastError();                                                                       \
  }
#define CM_UPS_OCCS(M, N) CM_UPS_GO(M, N, 2, 0) CM_UPS_GO(M, N, 2, 1) CM_UPS_GO(M, N, 2, 2) CM_UPS_GO(M, N, 2, 3) CM_UPS_GO(M, N, 2, 4)
  CM_UPS_OCCS(1, 1) CM_UPS_OCCS(2, 1) CM_UPS_OCCS(3, 1) CM_UPS_OCCS(4, 1) CM_UPS_OCCS(5, 1) CM_UPS_OCCS(2, 2)
#undef CM_UPS_OCCS
#undef CM_UPS_GO
  return hipErrorInvalidValue;
}

}  // namespace cm
