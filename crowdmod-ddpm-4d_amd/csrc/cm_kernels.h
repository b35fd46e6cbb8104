// Internal launch interface between the host orchestration (cm_model.cpp) and
// the gfx950 kernels (cm_conv.hip, cm_misc.hip).  Not part of the public ABI.
//
// Internal activation layout: channels-last  [B][Z][Y][X][C]  fp32 with
//   Z = frames (reference L), Y = rows (reference H), X = cols (reference W)
// and C a multiple of 8.  A 3-D convolution is invariant under a consistent
// permutation of the spatial axes of input and kernel, so tap (dz,dy,dx) of the
// internal kernel is element [kH=dy][kW=dx][kL=dz] of the reference weight.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

namespace cm {

// Diagnostic environment switches (kernel ablations, tile experiments: DESIGN.md section 6) are read ONLY when
// CM_DIAG=1 is set as well -- a stray CM_* variable in a production environment cannot change which kernels run or
// what they compute.  Product switches read with plain getenv: CM_LANES, CM_USE_GRAPH, CM_LIB_PATH (Python side).
inline const char *diag_env(const char *name) {
  static const bool on = [] { const char *d = getenv("CM_DIAG"); return d && d[0] == '1' && d[1] == 0; }();
  return on ? getenv(name) : nullptr;
}

// One implicit-GEMM convolution launch:  out[m][n] = sum_k A[m][k] * W[k][n]
//   m = output voxel (b,z,y,x), n = output channel, k = (tap, input channel).
#ifdef __HIPCC__
// Exact three-way bf16 split of four fp32 values (the six-term kernels): term tm of the four values as two
// packed dwords (element e in half e & 1 of dword e >> 1 -- the memory order of a bf16x4), hi = RNE(x), mid = RNE(x - hi),
// lo = RNE(x - hi - mid); the remainders are exact.  Written on the packed conversion result (one v_cvt_pk_bf16_f32 per PAIR,
// the rounded values recovered by a shift / mask) because the compiler, given element-wise (__bf16) casts, converts every element
// twice -- once alone for the subtraction, once packed for the store: 34 vector instructions per split instead of 22.
typedef float cm_f32x2_t __attribute__((ext_vector_type(2)));
typedef float cm_f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 cm_bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned cm_u32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 cm_f16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned cm_cvt_pk_bf16(float a, float b) {
  const cm_f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cm_bf16x2_t));
}
// NT = 2: only hi and mid (the relaxed plan's two-way split: three cross terms hi x mid, mid x hi, hi x hi, ~16 mantissa bits per
// product); t[2] is left untouched.
template <int NT = 3>
__device__ __forceinline__ void cm_split3_bf16(cm_f32x4_t x, cm_u32x2_t (&t)[3]) {
#pragma unroll
  for (int tm = 0; tm < NT; ++tm) {
    const unsigned p01 = cm_cvt_pk_bf16(x[0], x[1]), p23 = cm_cvt_pk_bf16(x[2], x[3]);
    t[tm] = cm_u32x2_t{p01, p23};
    if (tm < NT - 1) {
      x[0] -= __uint_as_float(p01 << 16); x[1] -= __uint_as_float(p01 & 0xffff0000u);
      x[2] -= __uint_as_float(p23 << 16); x[3] -= __uint_as_float(p23 & 0xffff0000u);
    }
  }
}
// Two-way f16 split of four fp32 values (the default plan's "h2" arithmetic on layers whose input is GroupNorm + SiLU output, i.e.
// bounded): hi = RNE_f16(x), mid = RNE_f16(x - hi) (the remainder x - hi is exact in fp32).  hi + mid carries 22 of the 24 mantissa
// bits; three cross terms hi x mid, mid x hi, hi x hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation give the accuracy of the
// six-term bf16 form (tools/sim_six_term.py: 9.8e-7 rms against fp64 for both, 5.2e-7 for a plain fp32 chain) at half the matrix
// instructions and a 12- instead of 22-instruction split.  f16 has 5 exponent bits: callers bound their operands.
typedef _Float16 cm_f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cm_split2_f16(cm_f32x4_t x, cm_u32x2_t (&t)[3]) {
  const cm_f32x2_t a = {x[0], x[1]}, b = {x[2], x[3]};
  const cm_f16x2_t ha = __builtin_convertvector(a, cm_f16x2_t), hb = __builtin_convertvector(b, cm_f16x2_t);
  t[0] = cm_u32x2_t{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
  const cm_f32x2_t ra = a - __builtin_convertvector(ha, cm_f32x2_t), rb = b - __builtin_convertvector(hb, cm_f32x2_t);
  const cm_f16x2_t ma = __builtin_convertvector(ra, cm_f16x2_t), mb = __builtin_convertvector(rb, cm_f16x2_t);
  t[1] = cm_u32x2_t{__builtin_bit_cast(unsigned, ma), __builtin_bit_cast(unsigned, mb)};
}
#endif

struct ConvArgs {
  const float *src0;   // [B][Zs][Ys][Xs][C0]
  const float *src1;   // second source of a channel concat (torch.cat dim=1), or null
  int C0, C1;          // multiples of CK
  const float *gn;     // [B][2][C0+C1] per-(sample,channel) scale row, shift row; or null
  int silu;            // apply SiLU after the affine (GroupNorm->SiLU fused on load)
  const float *pm;     // optional per-(sample, input channel) multiplier applied after the activation:
  int pm_stride;       //   the Dropout3d keep-mask / (1-p) of training mode (layers.py:42,71); row stride
  const float *wfrag;  // weights in MFMA fragment order (see pack_conv_weights)
  const float *bias;   // [Co padded to TN]
  const float *temb;   // [rows][temb_stride] time-embedding projection table, or null
  int temb_stride;
  const long long *tidx;  // [B] timestep per sample (row of temb)
  const float *resid;  // residual, channels-last with channel stride res_cs, or null
  int res_cs;
  float *out;          // [B][Zo][Yo][Xo][out_cs]
  int out_cs;
  int Co;              // valid output channels
  int B;
  int Zs, Ys, Xs;      // source dims (before nearest-upsample)
  int Zo, Yo, Xo;      // output dims
  int ntaps;           // td^3: 27 (3x3x3, zero pad 1), 8 (parity sub-kernel of an upsampled conv) or 1
  int td;              // taps per dimension: 3, 2 or 1
  int par;             // 1: nearest-x2-upsample + 3x3x3 conv evaluated as 8 parity classes of 2x2x2
                       //    convs on the low-resolution source (gridDim.z = 8); output voxel = 2*i + parity
  long long wpar_stride;  // floats between the packed weight sets of consecutive parities
  int ks;              // > 1: K is split over gridDim.z workgroups (chunk ranges); each writes its RAW partial
                       //      sums to out + z * kpart (bias / temb / residual / statistics happen in ksplit_combine)
  long long kpart;     // floats between consecutive partial outputs
  int stride;          // 1 or 2
  int ups;             // 1: source is nearest-upsampled x2 on the fly
  int bs, bz, by, bx;  // output box of one workgroup: samples x z x y x x
  int nts, ntz, nty, ntx;  // number of boxes along each of those
  int CK;              // channel chunk staged per pass: 8, 16 or 32
  int nch0, nch1;      // chunks in src0 / src1
  const int *hvtab;    // [HV] packed halo-box coordinates  s<<26 | hz<<18 | hy<<9 | hx  (host-built)
  const int *mtab;     // [32*MB] packed output-box coordinates s<<26 | z<<18 | y<<9 | x, or -1 (padding row)
  // fused GroupNorm statistics of the OUTPUT (one sample per tile required): every 32-row
  // accumulator block writes (mean, M2) per channel to stat_part[b][slot][stat_C][2] and its
  // row count to stat_cnt[b][slot], slot = tile_in_sample * MB + mb, stat_ns slots per sample
  float *stat_part;
  float *stat_cnt;
  int stat_C, stat_ns;
  // fused 1x1x1 skip path (ResnetBlock.match_input, layers.py:46,74) of a stride-1 27-tap conv on
  // the register-ring path: extra K chunks of 32 channels read from the RAW block input (no
  // normalisation), centre tap only; s2w packed like wfrag with ntaps = 1, CK = 32
  const float *s2src0; const float *s2src1;
  int s2C0, s2C1;
  const float *s2w;
  int dbg;             // ablation switches for performance studies (0 in production)
  float *dbg_buf;      // cycle-stamp sink of the diagnostic build paths
  int stagger;         // start delay (in 64-cycle units) applied to every other first-wave workgroup
  int f16;             // 1: wfrag holds f16 fragments (4 halves per lane and step); specialised parity form only
  int zsplit;          // weight gradient of a 27-tap layer on a two-plane grid: mtab rows [0,32) are plane 0, [32,64) plane 1,
                       //   and a row block skips the z tap that multiplies its padding plane (18 of 27 taps, as the forward)
  // Round 4 -- f16 ACTIVATIONS in HBM (reduced-precision plan, BASELINE configs[4]; the reference's autocast makes every conv
  // output fp16, ddpm.py:116-120): bit mask of the tensors of this launch that are stored as _Float16 instead of float (same
  // channels-last layout, same element strides): 1 src0, 2 src1, 4 out, 8 resid, 16 s2src0, 32 s2src1.  Only kernels that
  // implement it ever receive a non-zero mask (cm_model.cpp: plan_h16): conv_f16d, conv_first (out), conv_ups (f16 form),
  // conv_smalln (src).  Accumulation, GroupNorm statistics (taken from the fp32 accumulators), SiLU and the sampler stay fp32.
  int h16;
  // Default plan, "h2" layers (a.f16 == 4): the weight fragments hold f16 hi / mid terms of w * 2^k (cm_split2_f16; k chosen per
  // layer at pack time so that the mid terms are normal numbers); the kernel multiplies its accumulators by h2_oscale = 2^-k.
  float h2_oscale;
  // Round 4 -- GroupNorm statistics WITHOUT the gn_finalize launch (inference plan).  Producer side: instead of slot partials,
  // every 32-row block ADDS its per-channel sums to astat[b][astat_C][3] (64-bit fixed point: sum x * 2^16 as two's complement,
  // floor(sum x^2 / 2^32), (sum x^2 mod 2^32) * 2^20): integer adds are exact and order-independent, so the totals are
  // bit-identical however the launch is tiled, sharded or scheduled -- cm_stat_atomic.  Consumer side: the workgroup finalises
  // the GroupNorm of its input itself from the accumulators of its (one or two) source tensors -- cm_gn_rows_from_sums.
  unsigned long long *astat;             // output accumulator rows of sample 0 of this launch, or null (then stat_part as before)
  int astat_C;                           //   accumulator channels per sample (row stride in channels)
  const unsigned long long *gs0, *gs1;   // consumer: accumulators of src0 / src1, [B][gs_C][3] from sample 0 of this launch (null: a.gn rows)
  int gs_C;
  const float *gs_gamma, *gs_beta;       //   affine of this layer's GroupNorm over the C0 + C1 input channels
  int gs_groups;
  float gs_eps;
  // Round 4 -- the same consumer-side finalisation from the producers' SLOT partials, when they are few (half / quarter
  // resolution: <= 32 slots per sample): part [B][gns][C][2] (mean, M2), cnt [B][gns], from sample 0 of this launch, as
  // gn_finalize reads them (conv_qr2 has done this since round 3).  Shares gs_gamma / gs_beta / gs_groups / gs_eps.
  const float *gp0, *gc0, *gp1, *gc1;
  int gns0, gns1;
};
#ifdef __HIPCC__
// ---- statistics accumulators (ConvArgs::astat) ----------------------------------------------------------------------------
constexpr unsigned long long CM_STAT_POISON = 1ull << 63;      // a non-finite or out-of-range contribution was added
// One 32-row block's contribution for one channel: s1 = sum of its rows, mean = s1 / cnt, m2 = sum (x - mean)^2 (all fp32, as
// the slot format holds them).  sum x^2 = m2 + s1 * mean is formed in fp64, so the fixed-point words lose nothing the fp32
// partials had.  Range: |sum x| < 2^46 and sum x^2 < 2^94 per sample and channel; beyond that (or NaN / Inf) the poison bit is set
// and the consumer's statistics come out NaN, as non-finite slot partials did.
__device__ __forceinline__ void cm_stat_atomic(unsigned long long *acc, float s1, float mean, float m2) {
  const double q = (double)m2 + (double)s1 * (double)mean;
  const double sx = (double)s1 * 65536.0;
  if (!(fabs(sx) < 4.0e18) || !(q < 1.9e28)) {                 // (also catches NaN)
    __hip_atomic_fetch_or(acc + 1, CM_STAT_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const double qh = floor(q * (1.0 / 4294967296.0));
  const unsigned long long QH = (unsigned long long)qh;
  const unsigned long long QL = (unsigned long long)((q - qh * 4294967296.0) * 1048576.0 + 0.5);
  const long long SX = __double2ll_rn(sx);
  __hip_atomic_fetch_add(acc, (unsigned long long)SX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(acc + 1, QH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(acc + 2, QL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (sum x, sum x^2) of one channel from its accumulator words; NaN when poisoned
__device__ __forceinline__ void cm_stat_read(const unsigned long long *acc, double *sx, double *sq) {
  const unsigned long long w0 = acc[0], w1 = acc[1], w2 = acc[2];
  if (w1 & CM_STAT_POISON) { *sx = __builtin_nan(""); *sq = __builtin_nan(""); return; }
  *sx = (double)(long long)w0 * (1.0 / 65536.0);
  *sq = (double)w1 * 4294967296.0 + (double)w2 * (1.0 / 1048576.0);
}
// Consumer prologue: scale / shift rows of sample `b` for the C0 + C1 input channels into rows[2 (C0 + C1)] (LDS or global):
// rows[c] = gamma[c] * rstd(group of c), rows[Ctot + c] = beta[c] - mean * rows[c]   (nn.GroupNorm, layers.py:30,41: biased
// variance, eps inside the root).  Called by ALL `nth` threads of the workgroup (it contains one barrier): pass 1 -- thread c
// reads its channel's three words (ONE round trip for the whole workgroup; a loop over the group's channels per thread paid one
// per channel, 6 us per workgroup) into `scratch` [Ctot][2] doubles in LDS; pass 2 -- every channel sums its group's entries
// (4 ... 24 LDS reads) and finishes in registers.  V = voxels per sample.  The caller synchronises before it reads `rows`.
// Chan et al. pairwise combination of (n, mean, M2) triples (cm_misc.hip: chan_combine; same arithmetic)
__device__ __forceinline__ void cm_chan_combine(float &n, float &mean, float &m2, float nb, float meanb, float m2b) {
  if (nb == 0.f) return;
  const float nt = n + nb;
  const float d = meanb - mean;
  const float f = nb * __builtin_amdgcn_rcpf(nt);
  mean += d * f;
  m2 += m2b + d * d * n * f;
  n = nt;
}
// Consumer prologue, slot form (ConvArgs::gp0): rows[2 (C0 + C1)] of sample `b` from the producers' slot partials.  Called by all
// `nth` threads (one barrier inside).  Pass 1: thread c merges the <= 32 slots of its channel in slot order (their loads issued
// together) into scratch[Ctot][2] (mean, M2 over the V voxels); pass 2: every channel merges its group's channels in channel order.
__device__ __forceinline__ void cm_gn_rows_from_slots(const ConvArgs &a, int b, int V, float *rows, float *scratch, int tid, int nth) {
  const int Ctot = a.C0 + a.C1, cg = Ctot / a.gs_groups;
  for (int c = tid; c < Ctot; c += nth) {
    const bool s0 = c < a.C0;
    const float *p = s0 ? a.gp0 : a.gp1, *nn = s0 ? a.gc0 : a.gc1;
    const int ns = s0 ? a.gns0 : a.gns1, Cx = s0 ? a.C0 : a.C1, cc = s0 ? c : c - a.C0;
    float N = 0.f, M = 0.f, S2 = 0.f;
    for (int s8 = 0; s8 < ns; s8 += 8) {            // eight slots per round, all their loads in flight together
      float cn[8];
      cm_f32x2_t q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sl = s8 + u < ns ? s8 + u : ns - 1;
        cn[u] = s8 + u < ns ? nn[(size_t)b * ns + sl] : 0.f;
        q[u] = *reinterpret_cast<const cm_f32x2_t *>(p + (((size_t)b * ns + sl) * Cx + cc) * 2);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) cm_chan_combine(N, M, S2, cn[u], q[u][0], q[u][1]);
    }
    scratch[2 * c] = M;
    scratch[2 * c + 1] = S2;
  }
  __syncthreads();
  for (int c = tid; c < Ctot; c += nth) {
    const int g0 = (c / cg) * cg;
    float N = 0.f, M = 0.f, S2 = 0.f;
    for (int k = g0; k < g0 + cg; ++k) cm_chan_combine(N, M, S2, (float)V, scratch[2 * k], scratch[2 * k + 1]);
    const float rstd = rsqrtf(S2 / N + a.gs_eps);
    const float sc = rstd * a.gs_gamma[c];
    rows[c] = sc;
    rows[Ctot + c] = a.gs_beta[c] - M * sc;
  }
}
// h2 on a RAW (not normalised) source tensor: a power of two `s` for sample `b` such that every element satisfies |x| s < 2^14, from
// the producer's slot statistics of that tensor -- a channel's merged (mean, M2) bounds each of its elements: |x - mean| <= sqrt(M2),
// so |x| <= max_c (|mean_c| + sqrt(M2_c)) =: m, and s = 2^(13 - floor(log2 m)) puts m s in [2^13, 2^14) (f16 max 65504: four times
// the headroom the fp32 rounding of the statistics could ever need).  All `nth` threads call it (two barriers inside); `scratch`:
// >= nth / 64 floats of LDS that nothing else uses until the call returns.  Deterministic per sample (fixed merge order).
__device__ __forceinline__ float cm_h2_sample_scale(const float *__restrict__ gp, const float *__restrict__ gc, int ns, int C, int b,
                                                    float *scratch, int tid, int nth) {
  float bm = 0.f;
  for (int c = tid; c < C; c += nth) {
    float N = 0.f, M = 0.f, S2 = 0.f;
    for (int s8 = 0; s8 < ns; s8 += 8) {
      float cn[8];
      cm_f32x2_t q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sl = s8 + u < ns ? s8 + u : ns - 1;
        cn[u] = s8 + u < ns ? gc[(size_t)b * ns + sl] : 0.f;
        q[u] = *reinterpret_cast<const cm_f32x2_t *>(gp + (((size_t)b * ns + sl) * C + c) * 2);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) cm_chan_combine(N, M, S2, cn[u], q[u][0], q[u][1]);
    }
    bm = fmaxf(bm, fabsf(M) + sqrtf(fmaxf(S2, 0.f)));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bm = fmaxf(bm, __shfl_xor(bm, off));
  if ((tid & 63) == 0) scratch[tid >> 6] = bm;
  __syncthreads();
  float m = scratch[0];
  for (int w = 1; w < (nth >> 6); ++w) m = fmaxf(m, scratch[w]);
  __syncthreads();
  const unsigned e = (__float_as_uint(m) >> 23) & 255u;       // m < 2^(e - 126)
  int se = 267 - (int)e;                                      // biased exponent of 2^(140 - e): m s < 2^14
  se = se < 32 ? 32 : (se > 222 ? 222 : se);
  if (!(m > 0.f) || e == 255u) se = 127;                       // all-zero / non-finite statistics: no scaling
  return __uint_as_float((unsigned)se << 23);
}
__device__ __forceinline__ void cm_gn_rows_from_sums(const ConvArgs &a, int b, int V, float *rows, double *scratch, int tid, int nth) {
  const int Ctot = a.C0 + a.C1, cg = Ctot / a.gs_groups;
  for (int c = tid; c < Ctot; c += nth) {
    const unsigned long long *p = c < a.C0 ? a.gs0 + ((size_t)b * a.gs_C + c) * 3 : a.gs1 + ((size_t)b * a.gs_C + (c - a.C0)) * 3;
    const unsigned long long w0 = p[0], w1 = p[1], w2 = p[2];
    const bool bad = (w1 & CM_STAT_POISON) != 0ull;
    const double sx = (double)(long long)w0 * (1.0 / 65536.0);
    const double sq = (double)w1 * 4294967296.0 + (double)w2 * (1.0 / 1048576.0);
    scratch[2 * c] = bad ? __builtin_nan("") : sx;
    scratch[2 * c + 1] = bad ? __builtin_nan("") : sq;
  }
  __syncthreads();
  const double inv_n = 1.0 / ((double)cg * (double)V);
  for (int c = tid; c < Ctot; c += nth) {
    const int g0 = (c / cg) * cg;
    double sx = 0.0, sq = 0.0;
    for (int k = g0; k < g0 + cg; ++k) { sx += scratch[2 * k]; sq += scratch[2 * k + 1]; }
    const double mean = sx * inv_n;
    double var = sq * inv_n - mean * mean;
    var = var < 0.0 ? 0.0 : var;                   // (NaN stays NaN)
    const float rstd = 1.0f / sqrtf((float)var + a.gs_eps);
    const float sc = a.gs_gamma[c] * rstd;
    rows[c] = sc;
    rows[Ctot + c] = a.gs_beta[c] - (float)mean * sc;
  }
}
#endif

bool conv_zsplit_variant(const ConvArgs &a, int MB, int NB);
// does launch_conv have an f16-operand instantiation for this parity-form tile?
bool conv_par_f16_variant(int MB, int NB, int bz, int by, int bx);

// Diagnostic switches of the conv kernels (ConvArgs::dbg): CM_CONV_DBG from the environment, or the value
// set through cm_debug_conv_flags (>= 0) -- tests flip the XCD tile remap inside one process with it.
extern int conv_dbg_override;
int conv_dbg_flags();
size_t conv_lds_bytes(const ConvArgs &a, int MB, int NB);
// Host-side builders of ConvArgs::hvtab / mtab for the box stored in `a`.
void conv_build_tables(const ConvArgs &a, int MB, int *hvtab /*[conv_halo_voxels]*/, int *mtab /*[32*MB]*/);
int conv_halo_voxels(const ConvArgs &a);
// MB x NB = number of 32x32 accumulator blocks per wave (workgroup tile 32MB x 32NB).
hipError_t launch_conv(const ConvArgs &a, int MB, int NB, hipStream_t st);
// 1x1x1 stride-1 conv without output statistics as a flat-row GEMM on f16 matrix-core operands (reduced-precision plan);
// a.wfrag = pack_1x1_f16 fragments [n tile][16-channel group][block][lane][8 halves]
bool conv1x1_f16_ok(const ConvArgs &a, int NB);
hipError_t launch_conv1x1_f16(const ConvArgs &a, int NB, hipStream_t st);
bool conv_variant_exists(int MB, int NB);
// Upsample conv, parity form with the source tile staged once for four parity classes (cm_conv_ups.hip); a.bz / by / bx =
// the SOURCE tile of conv_ups_pick, nbp = the NB the weights were packed with
bool conv_ups_pick(int Z, int Y, int X, int *tz, int *ty, int *tx, int *mbw, int *planes);
bool conv_ups_ok(const ConvArgs &a, int mbw, int planes, int nbp);
int conv_ups_slots(const ConvArgs &a, int mbw);
hipError_t launch_conv_ups(const ConvArgs &a, int mbw, int planes, int nbp, hipStream_t st);
// bf16 x 3 split fragments (pack_ups_b6 order) re-derived on the device from the fp32 parity fragments (after an optimizer step)
hipError_t launch_ups_b6_repack(const float *wfrag, long long wpar_stride, float *w6, long long w6_stride, int Co, int Ci, int NBP, hipStream_t st);
// 3x3x3 conv with <= 8 output channels on the vector ALUs (cm_conv_small.hip); weights packed as
// [chunk][tap][ci in chunk][NCO = 4 or 8] (internal tap order), zero beyond Co.
bool conv_smalln_ok(const ConvArgs &a, int MB);
size_t conv_smalln_lds(const ConvArgs &a, int MB);
hipError_t launch_conv_smalln(const ConvArgs &a, int MB, const float *wsm, hipStream_t st);

// The UNet's first conv (cm_conv_io.hip): C <= 8 input channels read from the [..][C0] input tensor, whole
// weight set of a 32-channel output tile in registers, waves split the voxels of a (bz x by x full-X) tile.
// wpk: [Co/32][27 * cin / 2][64 lanes]; a.bx == a.Xo, a.ntx == 1.
bool conv_first_ok(const ConvArgs &a, int cin);
int conv_first_blocks(const ConvArgs &a);    // 32-row accumulator blocks (= statistics slots) per tile
size_t conv_first_lds(const ConvArgs &a, int cin);
hipError_t launch_conv_first(const ConvArgs &a, int cin, const float *wpk, hipStream_t st);

// Winograd F(2x2, 3x3) over (Y, X), direct over Z (cm_conv_wino.hip): stride-1 3x3x3 convs whose output box
// tiles as bz x by x bx with 16 < bz * (by/2) * (bx/2) <= 32 (cm_conv_wino.hip lists the instantiated shapes; a last
// tile that would stick out is shifted back and owns only its own patches).  a.wfrag: [Co/32][Ci/16][4 = xi_y][24 = (dz*2 + k8)*4 + xi_x]
// [64 lanes][4] transformed weights; statistics slots per tile: 4 (the (a, b) sub-blocks).
bool conv_wino_tile_ok(int bz, int by, int bx);
bool conv_wino_ok(const ConvArgs &a);
size_t conv_wino_lds(int bz, int by, int bx, bool f16, int nbw);
int conv_wino_nbw(int bz, int Co);
int conv_wino_nbw_run(const ConvArgs &a, bool f16);   // ... of one launch (a.B, a.ntz / nty / ntx set): see cm_conv_wino.hip
bool conv_wino_two_step(int bz, int by, int bx, bool f16, int nbw);
// f16: a.wfrag holds the f16 packing (3 groups per chunk and wave, 8 halves per lane): fp32 accumulate, f16 operands
hipError_t launch_conv_wino(const ConvArgs &a, bool f16, hipStream_t st);
// six-term bf16 form (a.f16 = 2, wfrag = pack_wino_b6 fragments): two-tile table-driven kernel only
bool conv_wino_b6_ok(int bz, int by, int bx, int Co, int Zo);
// its split fragments from the fp32 Winograd fragments (n_floats of them), e.g. after an optimizer step
hipError_t launch_wino_b6_repack(const float *wwino, float *w6, long long n_floats, hipStream_t st);
// Weight gradient in the Winograd domain (same tiles): partials part[G][ncb][nkb][3][16][32 ci][32 co], then G^T dU G
hipError_t launch_wgrad_wino(const ConvArgs &a, const float *dy, int dy_cs, float *part, int G, int ncb, int nkb, hipStream_t st);
hipError_t launch_wgrad_wino_reduce(const float *part, int G, int ncb, int nkb, int Co, int Ci, float *dW, hipStream_t st);
// tile (bz, by, bx) the Winograd kernel would use for an output grid, or false if none of its shapes fits
bool conv_wino_pick(int Zo, int Yo, int Xo, int *bz, int *by, int *bx);

// Whole-sample 3x3x3 convolution of the lowest resolution (cm_conv_qr.hip): two z planes, <= 64 voxels per plane; one
// 512-thread workgroup = (sample, 32 output channels).  The GroupNorm of the input is finalised inside the kernel from the
// producers' per-slot (mean, M2) partials; the output's statistics are written in the same slot format (2 or 4 slots).
struct QrArgs {
  const float *src0, *src1;      // channels-last [B][2][Y][X][C0 / C1] (torch.cat dim=1 of the two)
  int C0, C1;
  const float *part0, *cnt0;     // slot partials of src0: part [B][ns0][C0][2] (mean, M2), cnt [B][ns0]
  int ns0;
  const float *part1, *cnt1;
  int ns1;
  const float *gamma, *beta;     // affine of this layer's GroupNorm over the C0 + C1 channels (null: no normalisation)
  int groups;
  float eps;
  int silu;
  float *gn_out;                 // optional [B][2][C0 + C1] copy of the folded scale / shift rows
  const float *wq6;              // optional: the same weights as exact bf16 x 3 splits (pack_qr_b6) -> six-term products in conv_qr2
  int three;                     // with wq6: 1 = only the three leading cross terms (relaxed plan, cm_model_set_precision);
                                 // 2 = wq6 holds h2 fragments (f16 hi / mid of w * 2^k, cm_split2_f16): three f16 cross terms
  float h2_oscale;               //     ... and the accumulators are multiplied by 2^-k
  const float *wq;               // [Co / 32][g = k8 * 9 + dy * 3 + dx][dz][64 lanes][4]: W[co = 32 nt + lane % 32][ci = 8 k8 + 4 (lane / 32) + jj][(dz, dy, dx)]
  const float *bias;
  const float *temb;
  int temb_stride;
  const long long *tidx;
  const float *resid;
  int res_cs;
  float *out;
  int out_cs, Co, B, Y, X;
  float *stat_part, *stat_cnt;   // output statistics: part [B][2 MBP][stat_C][2], cnt [B][2 MBP]; MBP = 1 (Y X <= 32) or 2
  int stat_C;
  const float *s2src0, *s2src1;  // fused 1x1x1 skip conv on the RAW block input (null s2w: none)
  int s2C0, s2C1;
  const float *s2w;              // [Co / 32][Cs / 8][64 lanes][4]
  int qshift;                    // (set by the launcher) log2 of the staging's channel-quad lane count
  const float *pm;               // optional per-(sample, input channel) multiplier after the activation: Dropout3d of the training forward (conv_qr2 only)
  int pm_stride;
  int raw;                       // 1 with gamma == null: no normalisation at all, conv_qr2 still applies (data gradients)
};
bool conv_qr_ok(const QrArgs &a);
bool conv_qr2_b6_ok(const QrArgs &a);
// split fragments of conv_qr2's six-term form from the fp32 fragments `wq` (n_floats of them; Ci input channels), e.g. after an optimizer step
hipError_t launch_qr_b6_repack(const float *wq, float *wq6, long long n_floats, int Ci, hipStream_t st);
hipError_t launch_conv_qr(const QrArgs &a, hipStream_t st);

// Direct 3x3x3 stride-1 conv with f16 operands (cm_conv_f16.hip; reduced-precision plan): a.bz/by/bx = output box (divides the
// grid), mbw = 32-row blocks per wave; a.wfrag: [Co/(32 NB)][Ci/16][27][NB][64 lanes][8 halves], a.s2w: [Co/(32 NB)][Cs/16][NB][64][8]
// with NB = 2 when Co % 64 == 0 else 1; statistics slots per sample: conv_f16d_slots.
bool conv_f16d_pick(int Z, int Y, int X, int *bz, int *by, int *bx, int *mbw);
bool conv_f16d_ok(const ConvArgs &a, int mbw);
int conv_f16d_slots(const ConvArgs &a, int mbw);
hipError_t launch_conv_f16d(const ConvArgs &a, int mbw, hipStream_t st);

// Direct 3x3x3 stride-1 conv in fp32 arithmetic from six bf16 cross terms (cm_conv_b6d.hip; fp32 plan, full-resolution layers):
// a.bz/by/bx = output box (divides the grid), nw = waves per workgroup (2 / 4), mbw = 32-row blocks per wave;
// a.wfrag: [Co/(32 NB)][Ci/16][27][NB][3 terms][64 lanes][8 bf16] (pack_b6d), a.s2w: [Co/(32 NB)][Cs/16][NB][3][64][8], NB =
// conv_b6d_nb(Co); statistics slots per sample: conv_b6d_slots.
bool conv_b6d_pick(int Z, int Y, int X, int *bz, int *by, int *bx, int *nw, int *mbw, int stride = 1, int Zs = 0, int Ys = 0, int Xs = 0);
bool conv_b6d_ok(const ConvArgs &a, int nw, int mbw);
int conv_b6d_slots(const ConvArgs &a, int nw, int mbw);
int conv_b6d_nb(int Co);
hipError_t launch_conv_b6d(const ConvArgs &a, int nw, int mbw, hipStream_t st);
// host-side geometry tables of one (grid, tile) pair (cm_conv_b6d.hip; also driven by the sanitizer self-test)
void conv_b6d_tables(int Z, int Y, int X, int bz, int by, int bx, int nw, int mbw, std::vector<int> &tS, std::vector<int> &tM,
                     int *NSP, int *HVP, int *PY, int *PZ, int *ntp, int *conflicts, int stride = 1, int Zs = 0, int Ys = 0, int Xs = 0);
int conv_b6d_nld(int stride);   // staging items per thread the kernel instance of this stride holds
// split fragments from the layer's fp32 weights in the REFERENCE layout [Co][Ci][kH][kW][kL] (or [Co][Ci], taps = 1), e.g. after an optimizer step
hipError_t launch_b6d_repack(const float *w, float *w6, int Co, int Ci, int taps, int NB, hipStream_t st);

// The UNet's last conv (32 -> C <= 4 channels) on the matrix core with the 27 taps packed into the columns (cm_conv_fin.hip):
// a.by / a.bx = in-plane tile of conv_fin_pick; wfin = launch_fin_pack fragments (three bf16 terms, or one f16 term with f16 = true)
bool conv_fin_pick(int Y, int X, int *by, int *bx);
bool conv_fin_ok(const ConvArgs &a);
hipError_t launch_conv_fin(const ConvArgs &a, const float *wfin, int mode /*0 six bf16 terms, 1 f16, 2 three bf16 terms, 3 three f16 terms (h2)*/, hipStream_t st);
hipError_t launch_fin_pack(const float *w_ref /*[Co][32][3][3][3] reference layout, device*/, float *wfin, int Co, int mode /*0 three bf16 terms, 1 one f16 term, 2 f16 hi / mid of w * wscale*/, hipStream_t st, float wscale = 1.f);
constexpr size_t CM_FIN_W_FLOATS = 4 * 2 * 3 * 64 * 4;   // fragment floats (six-term form; the f16 form uses a third)

// ---- small kernels --------------------------------------------------------
// Per-(sample, slice, channel) mean and M2 of a channels-last tensor.
//   part [B][nslice][C][2]
// Fall-backs of the accumulator scheme (round 4): gn rows [B][2][C0 + C1] from the accumulators (a consumer kernel that cannot
// finalise them itself), and accumulators from slot partials (a producer kernel that cannot add to them itself)
hipError_t launch_gn_from_sums(const ConvArgs &a, int V, float *gn_rows, hipStream_t st);
hipError_t launch_slots_to_sums(const float *part, const float *cnt, int nslots, int C, int B, unsigned long long *astat, int astat_C, hipStream_t st);
hipError_t launch_chan_stats(const float *x, int B, int V, int C, int nslice, float *part, float *cnt, hipStream_t st);
// Combine partial statistics (per slot: mean, M2 in part[b][slot][C][2], row count in
// cnt[b][slot]) of (up to) two concatenated tensors into GroupNorm scale/shift rows:
//   gn[b][0][c] = rstd_g*gamma_c ; gn[b][1][c] = beta_c - mean_g*rstd_g*gamma_c
hipError_t launch_gn_finalize(const float *part0, const float *cnt0, int ns0, int C0, const float *part1,
                              const float *cnt1, int ns1, int C1, int V, const float *gamma, const float *beta,
                              int groups, float eps, float *gn, float *mr, int B, hipStream_t st);
// Second pass of a K-split convolution: out = sum_s part[s] (fixed order) + bias + temb + residual,
// plus the GroupNorm statistics of out per 32-row slot.  part: [S][B][V][C], C <= 256.
struct CombineArgs {
  const float *part; int S; long long stride;
  const float *bias; const float *temb; int temb_stride; const long long *tidx;
  const float *resid; int res_cs;
  float *out; int C, V, B;
  float *stat_part; float *stat_cnt; int nslots;
  // optional fused GroupNorm finalisation of the CONSUMER (launch_combine_gn: one workgroup per sample, V <= 64):
  // statistics of this tensor (channels [0, C)) and of a second, finished tensor (channels [C, C + fin_C1), its
  // slot partials fin_p1 / fin_n1) -> scale / shift per (sample, channel) as launch_gn_finalize writes them
  const float *fin_gamma, *fin_beta; float *fin_gn, *fin_mr;
  const float *fin_p1, *fin_n1; int fin_ns1, fin_C1, fin_groups; float fin_eps;
};
hipError_t launch_ksplit_combine(const CombineArgs &a, hipStream_t st);
bool combine_gn_ok(const CombineArgs &a);
hipError_t launch_combine_gn(const CombineArgs &a, hipStream_t st);
// reference layout [B,C,H,W,P] + [B,C,H,W,F]  ->  channels-last [B][P+F][H][W][8]
hipError_t launch_assemble_input(const float *past, const float *future, float *x8, int B, int C, int H, int W,
                                 int P, int F, int which /*1 past,2 future,3 both*/, hipStream_t st);
// channels-last eps [B][L][H][W][cs] frames >= P  ->  reference layout [B,C,H,W,F]
hipError_t launch_extract_output(const float *eps_cl, int cs, float *out, int B, int C, int H, int W, int P, int F,
                                 hipStream_t st);
// Time-embedding MLP + all per-block dense_1 projections, one row per t value:
//   out[row][0..nproj) = Wd @ silu(W2 @ silu(W1 @ table[t_row] + b1) + b2) + bd
hipError_t launch_time_mlp(const float *table, const float *W1, const float *b1, const float *W2, const float *b2,
                           const float *Wd, const float *bd, int te, int tx, int nproj, int nrows, float *temb_raw,
                           float *out, const long long *rowidx, hipStream_t st);
// softmax(q k^T / sqrt(d)) v per (sample, head); qkv channels-last [B][S][3E]
hipError_t launch_attn_core(const float *qkv, float *out, int B, int S, int E, int heads, hipStream_t st);
// the same on f16 matrix-core operands (fp32 accumulate, fp32 softmax): reduced-precision plan, head dimension 32
hipError_t launch_attn_core_f16(const float *qkv, float *out, int B, int S, int E, int heads, hipStream_t st);

// Fused AttentionBlock (cm_attn_block.hip): GroupNorm + in-projection + softmax(q k^T) v + partial out-projection
// per (head, sample); the heads are summed by launch_ksplit_combine (S = heads, stride = B * S * E).
struct AttnBlockArgs {
  const float *x;                 // [B][S][E] block input (channels-last), also the residual
  const float *gamma, *beta;      // [E] attention.group_norm affine
  const float *w_in, *b_in;       // mhsa.in_proj_weight [3E][E] (reference layout), in_proj_bias [3E]
  const float *w_out;             // mhsa.out_proj.weight [E][E] (reference layout)
  float *part;                    // [heads][B][S][E] partial out-projections
  int B, S, E, heads, groups;
  float eps;
};
bool attn_block_ok(int S, int E, int heads, int groups);
size_t attn_block_lds_bytes(int S, int E, int heads);
hipError_t launch_attn_block(const AttnBlockArgs &a, hipStream_t st);

// Per-step scalars of the sampling loop as a device table, so that one captured graph of a step can be
// replayed for every step: the step kernels read row tab[*kctr]; step_begin advances the counter.
struct StepRow { int t; float c_x, c_eps, c_noise, guid; int draw; int step; int pad; };
struct StepArgs {
  float *x;             // [B,C,H,W,F] reference layout, updated in place
  const float *eps_cl;  // channels-last UNet output [B][L][H][W][cs]
  int cs;
  float *x8;            // UNet input tensor [B][L][H][W][8]: future frames rewritten with the new x
  const float *noise;   // [B,C,H,W,F] or null -> Philox
  float *hist;          // optional copy of the new x
  int B, C, H, W, P, F;
  float c_x, c_eps, c_noise;  // x' = c_x * x + c_eps * eps + c_noise * z
  float guid;                 // sparsity guidance: x'[:,0] -= guid * sign(x'[:,0])
  unsigned long long seed;
  long long sample_id_base;
  int step;             // Philox stream index of this step
  int draw;             // 0: z = 0
  // graph replay: when tab != null the scalars above come from tab[*kctr] and noise / hist are the BASE pointers
  // of the [nsteps(+1)][Bfull][per] arrays (row k resp. k + 1, sample offset boff within the row)
  const StepRow *tab; const int *kctr;
  long long row_stride;  // Bfull * per
  long long boff;        // b0 * per
  // plain loop: this step also writes the NEXT step's time index into the UNet's t buffer (saves a launch per step)
  long long *t_next; long long t_next_v;
  // round 4: this step's denoiser has consumed the GroupNorm accumulators (ConvArgs::astat); clear this lane's rows for the next step
  unsigned long long *zero_u64; long long zero_n;
};
hipError_t launch_sampler_step(const StepArgs &a, hipStream_t st);
hipError_t launch_q_sample(const float *x0, const long long *t, const float *eps, const float *sab, const float *s1m,
                           float *xt, int B, long long per, hipStream_t st);
hipError_t launch_fill_t(long long *t, int B, long long value, hipStream_t st);
// *count = number of NaN / Inf elements of x[0..n)  (one workgroup, no atomics)
hipError_t launch_count_nonfinite(const float *x, long long n, int *count, hipStream_t st);
// graph replay: ++*kctr, then t[0..B) = tab[*kctr].t  (one workgroup)
hipError_t launch_step_begin(long long *t, int B, const StepRow *tab, int *kctr, hipStream_t st);
// Weight gradient of a 3x3x3 stride-1 conv with FEW channels on one side (the UNet's first conv: 8-channel input;
// its last conv: 3-4 output channels).  The generic kernel spends a 32x32 MFMA block per tap on them with 4-8x
// padding; here the narrow side is packed with the taps into the N dimension:
//   dW[co][ci][tap] = sum_v dY[v][co] A[v + tap][ci]
//   first conv (mirror 0): rows = co from dY at v (wide, unshifted), columns (tap, ci) from A at v + tap - 1
//   last conv  (mirror 1): rows = ci from A at u (wide, GroupNorm + SiLU applied on load), columns (tap, co) from dY at u - tap + 1
// NBLK = ceil(27 * CN / 32) MFMA blocks per voxel pair instead of 27.  Partials part[G][NBLK][32][32] (the four waves of a workgroup are summed in wave order before the store).
struct WgradPackArgs {
  const float *wide; int wide_cs;       // [B][Z][Y][X][wide_cs], channels [0, 32)
  const float *gn; int silu;            // [B][2][32] scale / shift rows of the wide tensor, or null
  const float *narrow; int narrow_cs;   // [B][Z][Y][X][narrow_cs], channels [0, cn_valid)
  int cn_valid, mirror;
  int B, Z, Y, X;
  int bz, by, bx, ntz, nty, ntx;        // tile box (bz * by * bx <= 256 rows, even) and tiles per sample
  float *part;
};
hipError_t launch_wgrad_pack(const WgradPackArgs &a, int CN, int G, hipStream_t st);
// part -> dW[Co][Ci][27] (reference layout): mirror 0: Co = 32 rows, Ci = cn_valid; mirror 1: Ci = 32 rows, Co = cn_valid
hipError_t launch_wgrad_pack_reduce(const float *part, int G, int CN, int cn_valid, int mirror, int Co, int Ci, float *dW, hipStream_t st);

// Weight gradient of a 1x1x1 stride-1 conv (skip convs, attention projections): dW[co][ci] = sum_n dy[n][co] a[n][ci] over
// the flat row index n = (sample, voxel); a = the conv's actual input (concat, GroupNorm affine, SiLU, dropout multiplier
// recomputed on load as in the forward).  One workgroup = 32 output channels x NKB 32-channel input blocks and a
// strided set of 128-row chunks: every input row is read once per co block instead of once per (co, ci) block pair.
// Partials part[G][ncb][nkb][32 co][32 ci] (waves summed in order before the store) -> launch_wgrad_reduce(ntaps 1, nvs 1).
hipError_t launch_wgrad_1x1(const ConvArgs &a, const float *dy, int dy_cs, long long V, float *part, int G, int ncb, int nkb, int NKB, hipStream_t st);

// Weight gradient of an upsample conv in the forward's parity form (cm_conv.hip): class p = (pz,py,px) reads dY at the
// output voxels 2i + p and the LOW-resolution input at i + e + p - 1, e in {0,1}^3.  One workgroup = (tile group, class,
// 32 co, 32 ci); its four waves split the voxel pairs and each keeps all 8 tap blocks (one dY fragment feeds 8 MFMAs).
// Partials in launch_wgrad_reduce_par's layout [G * 8][ncb][nkb][8][32][32] (waves summed in order before the store).
// Tile box (a.bz, a.by, a.bx) in low-resolution voxels, a.ntz/nty/ntx tiles per sample; a.gn / a.pm must be null.
hipError_t launch_wgrad_par(const ConvArgs &a, const float *dy, int dy_cs, float *part, int G, int ncb, int nkb, hipStream_t st);

// Weight gradient of a 27-tap stride-1 conv on a grid with exactly two z planes (quarter resolution): dedicated form of
// wgrad_kernel<true>.  512 threads; one workgroup = (tile group, 32 co, TWO 32-channel input blocks); the halo holds the
// two real planes only (a row block never multiplies its padding plane), coordinates are arithmetic (no table loads in
// the tile loop), every load of a tile is issued in one batch; wave w owns in-plane tap q = w of every z tap, wave dz the
// ninth one (q = 8) of z tap dz.  Partials in wgrad_reduce's layout part[G][ncb][nkb][27][32 co][32 ci].
bool wgrad_zs_ok(const ConvArgs &a);
hipError_t launch_wgrad_zs(const ConvArgs &a, const float *dy, int dy_cs, float *part, int G, int ncb, int nkb, hipStream_t st);

// Deferred per-sample voxel sums of the backward pass (bias gradient and the broadcast time-embedding term of every
// conv): out[b][c] = sum_v x[b][v][c], optionally mirrored into out2 (the time-projection gradient row).  One launch for
// the whole job table; grid (B * ncb_max, njobs).
struct VsumJob { const float *x; int V, C, cs; float *out; int ostride; float *out2; int ostride2; };
hipError_t launch_voxel_sum_jobs(const VsumJob *jobs, int njobs, int B, int maxC, hipStream_t st);

// dst[i] = src[i] for a table of contiguous segments in ONE launch (the per-block slices of the time-projection
// gradients go to their parameters' slots: 22 device-to-device copies per step otherwise)
struct CopyJob { float *dst; const float *src; long long n; };
hipError_t launch_copy_jobs(const CopyJob *jobs, int njobs, long long max_n, hipStream_t st);

// Deferred per-parameter batch reductions of the backward pass (bias, GroupNorm gamma / beta gradients): one launch
// for the whole job table instead of one ~5 us launch each.  out[c] = sum_b in[b * stride + c], fixed order.
struct BsumJob { const float *in; float *out; int C, stride; };
hipError_t launch_batch_sum_jobs(const BsumJob *jobs, int njobs, int B, int maxC, hipStream_t st);
// mean((a-b)^2) over n elements -> *loss (single workgroup partials + deterministic final sum)
hipError_t launch_mse_loss(const float *a, const float *b, long long n, float *partial, float *loss, hipStream_t st);
// Dropout3d keep-mask / (1-p) per (sample, channel) from the device Philox stream
hipError_t launch_dropout_mask(float *mask, int B, int C, float p, unsigned long long seed, long long sample_id_base,
                               int step, hipStream_t st);
hipError_t launch_randn(float *x, int B, long long per, unsigned long long seed, long long sample_id_base, int step,
                        hipStream_t st);
// per-(sample, channel, frame) reductions of the sampling metrics (see cm_misc.hip)
hipError_t launch_frame_metrics(const float *pred, const float *gt, int N, int C, int H, int W, int F, double *out,
                                float *minmax, hipStream_t st);
// generic strided copy channels-last -> reference layout (debug hook)
hipError_t launch_cl_to_ref(const float *x_cl, int cs, float *out, int B, int C, int Z, int Y, int X, hipStream_t st);

// ---- backward pass (cm_train.hip) -----------------------------------------------------
hipError_t launch_wgrad(const ConvArgs &a, int MB, const float *dy, int dy_cs, float *part, int G, int ncb, int nkb,
                        hipStream_t st);
hipError_t launch_wgrad_reduce(const float *part, int G, int ncb, int nkb, int ntaps, int Co, int Ci, float *dW,
                               hipStream_t st, int force_nvs = 0 /* 1: one partial per group even for ntaps == 1 */);
// parity-form (upsample conv) partials [G * 8 parities][ncb][nkb][8][32][32] -> dW [Co][Ci][27] in reference layout
hipError_t launch_wgrad_reduce_par(const float *part, int G, int ncb, int nkb, int Co, int Ci, float *dW, hipStream_t st);
hipError_t launch_voxel_sum(const float *x, int B, int V, int C, int cs, float *out, int ostride,
                            float *scratch /* [16][B][ostride] */, hipStream_t st);
hipError_t launch_batch_sum(const float *in, int B, int C, int stride, float *out, int accumulate, hipStream_t st);
struct GnbArgs {
  const float *x0; const float *x1; int C0, C1;  // forward inputs (channels-last)
  const float *dA; int dA_cs;                    // gradient w.r.t. the activated tensor, [B][V][>=Ctot]
  const float *gn;                               // [B][2][Ctot] scale, shift of the forward
  const float *mr;                               // [B][2][Ctot] mean_g, rstd_g expanded per channel
  const float *gamma;
  const float *pm; int pm_stride;
  int silu;
  int V, B, groups;
  float *part;                                   // [B][nsl][Ctot][2]  (sum dy, sum dy*xh)
  int nsl;
  float *coef;                                   // [B][3][Ctot]  A_c = rstd*gamma, Bg = rstd*S1/N, Cg = rstd*S2/N
  float *dgb;                                    // [B][2][Ctot]  per-sample dgamma, dbeta
  float *g0; float *g1; int acc0, acc1;          // outputs: gradients of x0 / x1 (accumulate flags)
};
hipError_t launch_gn_backward(const GnbArgs &a, hipStream_t st);
hipError_t launch_add_into(float *dst, int dcs, const float *src, int scs, int C, long long rows, int accumulate,
                           hipStream_t st);
hipError_t launch_upsample2(const float *x, float *y, int B, int Z, int Y, int X, int C, hipStream_t st);
hipError_t launch_sumpool2(const float *y, float *x, int B, int Z, int Y, int X, int C, int accumulate, hipStream_t st);
hipError_t launch_zero_stuff2(const float *y, float *u, int B, int Z, int Y, int X, int C, hipStream_t st);
hipError_t launch_mse_grad(const float *pred, const float *target, float *g, int B, int C, int H, int W, int P, int F,
                           hipStream_t st);
// `big`: global scratch of attn_bwd_scratch_floats() floats when the S x S matrices exceed LDS (0 floats = not needed)
size_t attn_bwd_scratch_floats(int B, int S, int E, int heads);
hipError_t launch_attn_bwd(const float *qkv, const float *dO, float *dqkv, int B, int S, int E, int heads, float *big, hipStream_t st);
struct TimeBwdArgs {
  const float *table; const long long *t;
  const float *W1, *b1, *W2, *b2, *Wd, *bd;
  int te, tx, nproj, B;
  const float *dproj;     // [B][nproj]
  float *ws;              // workspace: e[B][te], h1[B][tx], z1[B][tx], tev[B][tx], s[B][tx], dte[B][tx], dz1[B][tx]
  float *dW1, *db1, *dW2, *db2, *dWd, *dbd;
};
hipError_t launch_time_bwd(const TimeBwdArgs &a, hipStream_t st);
hipError_t launch_adam(float *p, const float *g, float *m, float *v, long long n, float lr, float b1, float b2,
                       float eps, float wd, int step, hipStream_t st);
hipError_t launch_gather_pack(const float *W, const int *idx, int nk, float *packed, long long n, hipStream_t st);
// Winograd-transformed weights (cm_conv_wino.hip layout, pack_wino) re-derived from the master parameters after an optimizer
// step: element (n tile, chunk, xi_y, z tap, k half, xi_x, lane, jj) = sum_{dy,dx} G[xi_y][dy] G[xi_x][dx] W[idx27[(co, ci)][dz, dy, dx]].
// idx27: [Co][Ci][27] indices into the flat parameter buffer (internal tap order; the data gradient passes the flipped /
// transposed map).  One launch for all such tensors; 27 indices per (co, ci) pair instead of 9 (index, coefficient) terms per
// packed element (48 elements per pair and z tap triple): ~20x less traffic than the generic gather.
struct WinoPackJob { float *dst; const int *idx27; int Co, Ci, Ci_pad, pad; long long n; long long blk0; };
hipError_t launch_wino_pack_jobs(const float *W, const WinoPackJob *jobs, const int *blk2job, long long nblocks, hipStream_t st);

struct PackJob { float *dst; const int *idx; const float *coef; long long start; int nk; int pad; long long n; long long blk0; };   // coef: optional weights of the nk terms; n elements from workgroup blk0 on
// blk2job[workgroup] = job index (each job owns ceil(n / 256) consecutive workgroups: no per-element search)
hipError_t launch_gather_pack_jobs(const float *W, const PackJob *jobs, const int *blk2job, long long nblocks, hipStream_t st);

}  // namespace cm
