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

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace cm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division sequence
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// FAST = 0: generic step loop; 27 / 8: CK == 32 with that many taps (register weight ring);
// 127: the 27-tap ring path plus the fused 1x1x1 skip chunks (its own variant so that the plain
// kernels do not carry its registers)
// BZ x BY x BX != 0: the output box is a compile-time constant (stride 1, one sample per tile, no
// parity / upsampling): halo extents, LDS strides and every table entry fold into immediates, which
// removes the table loads and most of the integer / scalar-spill traffic of the prologue and staging.
// FAST = 8 specialisations are the parity (upsample) form, STR = 2 the strided 27-tap conv.
#ifndef CM_SPEC_OCC
#define CM_SPEC_OCC 2
#endif
// OCC != 0: requested waves per SIMD (= workgroups per CU); 3 for the MB = 3 full-resolution tiles, whose
// smaller accumulator set fits 168 VGPRs.
template <int MB, int NB, int FAST, int BZ = 0, int BY = 0, int BX = 0, int STR = 1, int OCC = 0>
__global__ __launch_bounds__(256, (OCC ? OCC : (BZ != 0 && MB * NB <= 4 ? CM_SPEC_OCC : 2))) void conv_mfma_kernel(const ConvArgs a) {
  constexpr bool SPEC = BZ != 0;
  constexpr bool SPAR = SPEC && (FAST % 100) == 8;   // specialised parity form: 2 taps per dimension
  // FAST >= 200 (reduced-precision plan, parity form only): the staged tile is rounded to f16 (row = 32 channels
  // = 64 B + 8 B pad), the weights arrive as f16, and a wave's 8-channel slice of a tap is ONE
  // v_mfma_f32_32x32x8_f16 (fp32 accumulate) instead of four fp32 instructions
  constexpr bool F16 = FAST >= 200 && FAST < 300;
  static_assert(!F16 || SPAR, "f16 operands are instantiated for the specialised parity form only");
  // FAST 327 / 427 (z-split form of a 27-tap layer whose grid has exactly TWO z planes -- the quarter resolution of
  // every reference grid): row block mb holds the voxels of plane z = mb, so the z tap that would read the padding
  // plane (dz = 0 for z = 0, dz = 2 for z = 1) is never issued and the padding planes are never staged: 18 instead
  // of 27 taps of MFMAs per block and half the halo box.  Bit-identical to the full form (the skipped products are
  // exact zeros).  427 = with the fused 1x1x1 skip convolution (as 127).
  constexpr bool ZS = SPEC && FAST >= 300;
  static_assert(!ZS || (MB == 2 && BZ == 2 && BY * BX <= 32 && STR == 1 && ((FAST % 100) == 27 || FAST == 308)), "z-split: one plane per row block");
  // FAST 308: the same for the parity form of an upsample conv whose SOURCE has two planes (decoder_blocks.2): row block mb
  // holds the source plane i = mb; class pz = 0 never issues (i = 0, e_z = 0), class pz = 1 never (i = 1, e_z = 1) -- 6 of 8
  // (row block, z tap) pairs, and two staged planes instead of three
  constexpr bool ZSP = ZS && (FAST % 100) == 8;
  constexpr bool SKIPC = FAST == 127 || FAST == 427;
  constexpr int STD = SPAR ? 2 : 3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TM = 32 * MB;
  constexpr int TN = 32 * NB;
  constexpr int NBLK = MB * NB;
  constexpr int RB = NBLK < 4 ? NBLK : 4;  // accumulator blocks reduced per round

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // diagnostic build path (dbg bit 3): wall-clock stamps (100 MHz) of this workgroup's phases
  unsigned long long tst[6] = {0, 0, 0, 0, 0, 0};
#define CM_RT(i) if (a.dbg & 8) tst[i] = __builtin_amdgcn_s_memrealtime();
  CM_RT(0)
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
  // consecutive block ids -- spatial neighbours that share halo voxels -- would land on 8 different L2s.
  // Block id i therefore takes logical tile (i % 8) * (n / 8) + i / 8: every XCD works through a contiguous
  // range of tiles (whole samples), and a neighbour's halo is an L2 hit instead of a fabric / HBM access.
  int tile = blockIdx.x;
  if (!(gridDim.x & 7) && !(a.dbg & 4096)) tile = (tile & 7) * (int)(gridDim.x >> 3) + (tile >> 3);
  const int tx = tile % a.ntx; tile /= a.ntx;
  const int ty = tile % a.nty; tile /= a.nty;
  const int tz = tile % a.ntz;
  const int ts = tile / a.ntz;
  const int nt = blockIdx.y;
  // De-phase the workgroups that share a CU.  They are identical programs started
  // together, so left alone they stage, compute and store in lock-step and the matrix
  // cores idle during everybody's load/store phases.  Delaying the second resident
  // set once is enough: every later workgroup starts when an earlier one retires.
  // (Placement is not architecturally defined -- a wrong guess only costs the delay.)
  if (a.stagger > 0) {
    const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x;
    const unsigned slot = lin >> 8;  // 256 CUs: blocks b and b+256 are expected to share a CU
    if (slot >= 1 && slot < 4 && (slot & 1)) {
      for (int i = 0; i < a.stagger; i += 64) __builtin_amdgcn_s_sleep(64);
    }
  }
  const int a_bs = SPEC ? 1 : a.bs, a_bz = SPEC ? BZ : a.bz, a_by = SPEC ? BY : a.by, a_bx = SPEC ? BX : a.bx;
  const int a_stride = SPEC ? STR : a.stride, a_par = SPEC ? (SPAR ? 1 : 0) : a.par, a_ups = SPEC ? 0 : a.ups,
            a_td = SPEC ? STD : a.td;
  const int b0 = ts * a_bs, z0 = tz * a_bz, y0 = ty * a_by, x0 = tx * a_bx;

  // parity class of an upsampled conv: out voxel u = 2i + p reads source voxels i + e + p - 1,
  // e in {0,1}, with the taps that fall on the same source voxel pre-summed on the host
  const int par = a_par ? (int)blockIdx.z : 0;
  const int pz = (par >> 2) & 1, py = (par >> 1) & 1, px = par & 1;
  const int os = a_par ? 2 : 1;
  const int pad = (a_td == 3) ? 1 : 0;
  const int td = a_td;
  const int HZ = ZS ? 2 : (a_bz - 1) * a_stride + td;
  const int HY = (a_by - 1) * a_stride + td;
  const int HX = (a_bx - 1) * a_stride + td;
  const int HV1 = HZ * HY * HX;
  const int HV = a_bs * HV1;
  const int HVp = (HV + 3) & ~3;
  const int S = F16 ? 18 : (SPEC ? 32 : a.CK) + 4;   // LDS row stride in dwords
  // packed row / halo coordinates: host tables in general, arithmetic (constant divisors) when specialised
  auto mtab_at = [&](int m) -> int {
    if constexpr (ZS) {
      const int rr = m & 31, y = rr / BX, x = rr - y * BX;
      return rr < BY * BX ? ((m >> 5) << 18) | (y << 9) | x : -1;
    } else if constexpr (SPEC) {
      if (m >= BZ * BY * BX) return -1;
      const int z = m / (BY * BX), rem = m - z * (BY * BX), y = rem / BX, x = rem - y * BX;
      return (z << 18) | (y << 9) | x;
    } else {
      return a.mtab[m];
    }
  };
  auto hvtab_at = [&](int hv) -> int {
    if constexpr (SPEC) {
      constexpr int cHY = (BY - 1) * STR + STD, cHX = (BX - 1) * STR + STD;
      const int hz = hv / (cHY * cHX), rem = hv - hz * (cHY * cHX), hy = rem / cHX, hx = rem - hy * cHX;
      return (hz << 18) | (hy << 9) | hx;
    } else {
      return a.hvtab[hv];
    }
  };

  int *outoff = reinterpret_cast<int *>(lds);  // [TM] output voxel index or -1
  int *outb = outoff + TM;                     // [TM] sample index of the row
  float *A = lds + 2 * TM;                     // [HV][S] staged halo tile / reduction scratch
  (void)HVp;

  // ---- output row table (coordinates come packed from the host: no divisions) ----
  // All table loads of the prologue (row table, LDS row bases, halo coordinates) are issued
  // unconditionally and back to back, and decoded branch-free: a load behind a divergent branch
  // costs its own memory round trip (~1 us under load), and there were six of them in a row.
  static_assert(TM <= 256, "one row-table pass");
  const int pk_row = mtab_at(tid < TM ? tid : TM - 1);
  int pk_ab[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) pk_ab[mb] = mtab_at(mb * 32 + r);
  {
    const int pk = pk_row;
    const bool v = pk >= 0;
    const int x = pk & 511, y = (pk >> 9) & 511, z = (pk >> 18) & 255, s = v ? (pk >> 26) : 0;
    const int b = b0 + s, oz = os * (z0 + z) + pz, oy = os * (y0 + y) + py, ox = os * (x0 + x) + px;
    const bool in = v && b < a.B && oz < a.Zo && oy < a.Yo && ox < a.Xo;
    const int off = in ? ((b * a.Zo + oz) * a.Yo + oy) * a.Xo + ox : -1;
    const int bb = (v && b < a.B) ? b : 0;
    if (tid < TM) { outoff[tid] = off; outb[tid] = bb; }
  }
  const int Zc = a.Zs << a_ups, Yc = a.Ys << a_ups, Xc = a.Xs << a_ups;
  const int cz0 = ZS ? 0 : z0 * a_stride + (a_par ? pz - 1 : -pad), cy0 = y0 * a_stride + (a_par ? py - 1 : -pad),
            cx0 = x0 * a_stride + (a_par ? px - 1 : -pad);

  // ---- per-lane LDS row base of each of this wave's MB row blocks -----------
  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int pk = pk_ab[mb] >= 0 ? pk_ab[mb] : 0;
    const int x = pk & 511, y = (pk >> 9) & 511, z = (pk >> 18) & 255, s = pk >> 26;
    const int hv = ((s * HZ + (ZS ? 0 : z) * a_stride) * HY + y * a_stride) * HX + x * a_stride;   // ZS: plane 0, the tap adds mb + dz - 1
    abase[mb] = hv * S + (F16 ? 2 : 4) * h;
  }

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.0f;

  const int a_CK = SPEC ? 32 : a.CK;
  const int K4 = a_CK >> 2, K8 = a_CK >> 3;
  const int nsteps = a.ntaps * K8;
  const int nchunks = a.nch0 + a.nch1;
  const int Ctot = a.C0 + a.C1;
  const f32x4 *wtile = reinterpret_cast<const f32x4 *>(a.wfrag + (size_t)par * a.wpar_stride) +
                       (size_t)nt * nchunks * nsteps * NB * 64 + lane;
  // K split over workgroups (tiny-spatial layers): this one owns chunks [ch0, ch1).
  // (Splitting the taps as well was tried and measured slower -- every extra workgroup repeats the
  // fixed setup / staging / epilogue phases that bound those layers -- and made the tap loop bounds
  // dynamic, which cost registers in every variant; see profiles/round1_notes.md.)
  const int kz = (a.ks > 1) ? (int)blockIdx.z : 0;
  const int ch0 = (a.ks > 1) ? kz * nchunks / a.ks : 0;
  const int ch1 = (a.ks > 1) ? (kz + 1) * nchunks / a.ks : nchunks;
  float *const outp = a.out + (size_t)kz * a.kpart;

  const int q4 = tid % K4, v0 = tid / K4, vstep = 256 / K4;

  constexpr bool fast = FAST != 0;           // host guarantees CK == 32 && ntaps == FAST
  constexpr int TAPS = FAST ? FAST % 100 : 27;
  constexpr int PD = (TAPS == 27) ? CM_PD27 : CM_PD8;   // weight prefetch depth of the fast path (taps), TAPS % PD == 0
  f32x4 bq[PD][NB];
  const f32x4 *wrun = wtile + ((size_t)(ch0 * TAPS + PD) * 4 + wave) * NB * 64;  // next ring refill
  int wleft = (ch1 - ch0) * TAPS - PD;                                            // refills still to issue
  // f16 operands: the same stream with 8-byte fragments (4 halves per lane and step)
  f32x2 bq2[PD][NB];
  const f32x2 *wtile2 = reinterpret_cast<const f32x2 *>(a.wfrag + (size_t)par * a.wpar_stride) +
                        (size_t)nt * nchunks * nsteps * NB * 64 + lane;
  const f32x2 *wrun2 = wtile2 + ((size_t)(ch0 * TAPS + PD) * 4 + wave) * NB * 64;
  if constexpr (fast && !F16) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const f32x4 *wp = wtile + ((size_t)ch0 * (TAPS * 4) + (wave + 4 * d)) * NB * 64;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bq[d][nb] = wp[nb * 64];
    }
  }
  if constexpr (F16) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const f32x2 *wp = wtile2 + ((size_t)ch0 * (TAPS * 4) + (wave + 4 * d)) * NB * 64;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bq2[d][nb] = wp[nb * 64];
    }
  }

  // Fast staging (32-channel chunks, one sample per tile): this thread's halo voxels and
  // their source offsets are resolved ONCE per workgroup; every chunk then issues all its
  // loads back to back (one memory latency per chunk instead of one per dependent step --
  // a global load costs 1.5-2 us under load on this part, profiles/round1_notes.md).
  // halo float4 per thread: 12 (HV <= 384) in general; a specialisation sizes it to its own box
  constexpr int cHZs = ZS ? 2 : (BZ - 1) * STR + STD, cHYs = (BY - 1) * STR + STD, cHXs = (BX - 1) * STR + STD;
  constexpr int cRJ = (cHZs * cHYs + 31) / 32;
  constexpr int NVM = !SPEC ? 12
                      : (cRJ <= 2 && cRJ * cHXs <= 16 && cRJ * cHXs > 12) ? 16                 // row mode, 13-16 slots
                      : (cHZs * cHYs * cHXs > 384 && cHZs * cHYs * cHXs <= 512) ? 16 : 12;      // voxel mode up to 512 voxels
  const bool fstage = fast && a_bs == 1 && (HV <= NVM * 32 || (SPEC && cRJ <= 2 && cRJ * cHXs <= NVM));
  int soff[NVM];
  unsigned okmask = 0;
  // Row mode: a thread owns one or two whole x-rows of the halo box (row = hz * HY + hy), so the
  // coordinate decode and the bounds of z / y happen once per row instead of once per voxel.
  const int HR = HZ * HY, RJ = (HR + 31) >> 5;
  const bool rowmode = fstage && a_ups == 0 && RJ <= 2 && RJ * HX <= NVM && (SPEC || !(a.dbg & 512));
  // slot k of a thread: (row lane j, x) in row mode, voxel v0 + 32 k otherwise
  auto slot_j = [&](int k) { return RJ == 2 ? k / (NVM / 2) : 0; };
  auto slot_x = [&](int k) { return RJ == 2 ? k % (NVM / 2) : k; };
  auto slot_used = [&](int k) { return rowmode ? (slot_x(k) < HX && slot_j(k) < RJ) : (k * 32 < HV); };
  auto slot_hv = [&](int k) { return rowmode ? (v0 + 32 * slot_j(k)) * HX + slot_x(k) : v0 + k * 32; };
  auto slot_mine = [&](int k) { return rowmode ? (v0 + 32 * slot_j(k) < HR) : (v0 + k * 32 < HV); };
  if (rowmode) {
    int rowbase[2];
    bool rowok[2];
    int pkr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) pkr[j] = hvtab_at(min(v0 + 32 * j, HR - 1) * HX);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = v0 + 32 * j;
      const int pk = pkr[j];
      const int cy = cy0 + ((pk >> 9) & 511), cz = cz0 + ((pk >> 18) & 255);
      rowok[j] = rr < HR && b0 < a.B && cz >= 0 && cz < Zc && cy >= 0 && cy < Yc;
      rowbase[j] = ((b0 * a.Zs + cz) * a.Ys + cy) * a.Xs;
    }
#pragma unroll
    for (int k = 0; k < NVM; ++k) {
      const int j = slot_j(k), cx = cx0 + slot_x(k);
      const bool ok = rowok[j] && slot_used(k) && cx >= 0 && cx < Xc;
      soff[k] = ok ? rowbase[j] + cx : 0;
      okmask |= (ok ? 1u : 0u) << k;
    }
  } else if (fstage) {
    int pkk[NVM];
#pragma unroll
    for (int k = 0; k < NVM; ++k) {
      const int hv = v0 + k * 32;
      pkk[k] = hvtab_at(hv < HV ? hv : HV - 1);
    }
#pragma unroll
    for (int k = 0; k < NVM; ++k) {
      const int hv = v0 + k * 32;
      const int cx = cx0 + (pkk[k] & 511), cy = cy0 + ((pkk[k] >> 9) & 511), cz = cz0 + ((pkk[k] >> 18) & 255);
      const bool ok = hv < HV && b0 < a.B && cz >= 0 && cz < Zc && cy >= 0 && cy < Yc && cx >= 0 && cx < Xc;
      soff[k] = ok ? ((b0 * a.Zs + (cz >> a_ups)) * a.Ys + (cy >> a_ups)) * a.Xs + (cx >> a_ups) : 0;
      okmask |= (ok ? 1u : 0u) << k;
    }
  }

  CM_RT(1)
  for (int ch = ch0; ch < ch1; ++ch) {
    const float *src;
    int Cs, c0, cg0;
    if (ch < a.nch0) { src = a.src0; Cs = a.C0; c0 = ch * a_CK; cg0 = c0; }
    else { src = a.src1; Cs = a.C1; c0 = (ch - a.nch0) * a_CK; cg0 = a.C0 + c0; }
    __syncthreads();  // tables ready (first pass) / previous chunk fully consumed
    // ---- stage the halo tile of this channel chunk ---------------------------
    // (loads are issued in batches of SU so that their latencies overlap)
    constexpr int SU = 6;   // loads in flight per thread per batch (their latencies overlap)
    const float *srcq = src + c0 + 4 * q4;
    // Branch-free: loads inside divergent branches make hipcc fall back to vmcnt(0) waits.
    // Out-of-range halo voxels read voxel 0 (a cached line) and are zeroed before the
    // LDS write; the hvtab entries beyond HV are clamped to the last valid one.
    // GroupNorm scale / shift of this thread's channel quad: one pair per chunk when the tile
    // holds a single sample (the common case), looked up per voxel otherwise
    f32x4 sc1 = {1.f, 1.f, 1.f, 1.f}, sh1 = {0.f, 0.f, 0.f, 0.f}, pm1 = {1.f, 1.f, 1.f, 1.f};
    if (a.pm && a_bs == 1)
      pm1 = *reinterpret_cast<const f32x4 *>(a.pm + (size_t)(b0 < a.B ? b0 : 0) * a.pm_stride + cg0 + 4 * q4);
    if (a.gn && a_bs == 1) {
      const float *g = a.gn + (size_t)(b0 < a.B ? b0 : 0) * 2 * Ctot + cg0 + 4 * q4;
      sc1 = *reinterpret_cast<const f32x4 *>(g);
      sh1 = *reinterpret_cast<const f32x4 *>(g + Ctot);
    }
    if (fstage && !(a.dbg & 1)) {
      f32x4 v[NVM];
#pragma unroll
      for (int k = 0; k < NVM; ++k)
        if (slot_used(k)) v[k] = *reinterpret_cast<const f32x4 *>(srcq + (size_t)soff[k] * Cs);
#pragma unroll
      for (int k = 0; k < NVM; ++k) {
        const int hv = slot_hv(k);
        if (slot_used(k)) {
          f32x4 w = v[k];
          if (a.gn && !(a.dbg & 128)) {
            w = w * sc1 + sh1;
            if (a.silu) { w[0] = silu_f(w[0]); w[1] = silu_f(w[1]); w[2] = silu_f(w[2]); w[3] = silu_f(w[3]); }
          }
          if (a.pm) w = w * pm1;
          if (!((okmask >> k) & 1u)) w = f32x4{0.f, 0.f, 0.f, 0.f};
          if (slot_mine(k)) {
            if constexpr (F16) {
              const f16x4 hw = {(_Float16)w[0], (_Float16)w[1], (_Float16)w[2], (_Float16)w[3]};
              *reinterpret_cast<f16x4 *>(&A[hv * S + 2 * q4]) = hw;
            } else {
              *reinterpret_cast<f32x4 *>(&A[hv * S + 4 * q4]) = w;
            }
          }
        }
      }
    }
    // (dead code in a specialisation whose halo box fits the fast staging)
    constexpr bool SPEC_FSTAGE = SPEC && (cHZs * cHYs * cHXs <= NVM * 32 || (cRJ <= 2 && cRJ * cHXs <= NVM));
    for (int hv0 = v0; !SPEC_FSTAGE && hv0 < HV && !(a.dbg & 1) && !fstage; hv0 += vstep * SU) {
      int pk[SU];
      f32x4 v[SU];
      int bbv[SU];
      bool ok[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int hv = hv0 + u * vstep;
        pk[u] = a.hvtab[hv < HV ? hv : HV - 1];
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int hv = hv0 + u * vstep;
        const int cx = cx0 + (pk[u] & 511), cy = cy0 + ((pk[u] >> 9) & 511), cz = cz0 + ((pk[u] >> 18) & 255);
        const int b = b0 + (pk[u] >> 26);
        ok[u] = hv < HV && b < a.B && cz >= 0 && cz < Zc && cy >= 0 && cy < Yc && cx >= 0 && cx < Xc;
        const int off = ok[u] ? ((b * a.Zs + (cz >> a.ups)) * a.Ys + (cy >> a.ups)) * a.Xs + (cx >> a.ups) : 0;
        bbv[u] = b < a.B ? b : 0;
        v[u] = *reinterpret_cast<const f32x4 *>(srcq + (size_t)off * Cs);
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int hv = hv0 + u * vstep;
        f32x4 w = v[u];
        if (a.gn && !(a.dbg & 128)) {
          f32x4 sc = sc1, sh = sh1;
          if (a_bs != 1) {
            const float *g = a.gn + (size_t)bbv[u] * 2 * Ctot + cg0 + 4 * q4;
            sc = *reinterpret_cast<const f32x4 *>(g);
            sh = *reinterpret_cast<const f32x4 *>(g + Ctot);
          }
          w = w * sc + sh;
          if (a.silu) { w[0] = silu_f(w[0]); w[1] = silu_f(w[1]); w[2] = silu_f(w[2]); w[3] = silu_f(w[3]); }
        }
        if (a.pm) {
          f32x4 pmv = pm1;
          if (a_bs != 1) pmv = *reinterpret_cast<const f32x4 *>(a.pm + (size_t)bbv[u] * a.pm_stride + cg0 + 4 * q4);
          w = w * pmv;
        }
        if (!ok[u]) w = f32x4{0.f, 0.f, 0.f, 0.f};
        if (hv < HV) *reinterpret_cast<f32x4 *>(&A[hv * S + 4 * q4]) = w;
      }
    }
    __syncthreads();
    if (ch == ch0) { CM_RT(2) }
    // ---- this wave's share of the (tap, 8-channel) steps ---------------------
    if (a.dbg & 2) continue;
    if constexpr (ZSP) {
      auto body = [&](auto PZ) {
        constexpr int pzc = decltype(PZ)::value;
        auto on = [](int t, int mb) { const int ez = t >> 2; return mb == 0 ? !(ez == 0 && pzc == 0) : !(ez == 1 && pzc == 1); };
        auto off = [&](int t, int mb) { const int ez = t >> 2, ey = (t >> 1) & 1, ex = t & 1; return (((mb + ez + pzc - 1) * cHYs + ey) * cHXs + ex) * 36; };
        const float *Aw = A + wave * 8;
        f32x4 afn[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) afn[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          if (on(0, mb)) afn[mb] = *reinterpret_cast<const f32x4 *>(&Aw[abase[mb] + off(0, mb)]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int d = t % PD;
          f32x4 af[MB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) af[mb] = afn[mb];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            if (t + 1 < 8 && on(t + 1, mb)) afn[mb] = *reinterpret_cast<const f32x4 *>(&Aw[abase[mb] + off(t + 1, mb)]);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
              if (on(t, mb)) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                  acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mb][jj], bq[d][nb][jj], acc[mb][nb], 0, 0, 0);
              }
          if (wleft > 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bq[d][nb] = wrun[nb * 64];
            wrun += 4 * NB * 64;
            --wleft;
          }
        }
      };
      if (pz == 0) body(std::integral_constant<int, 0>{}); else body(std::integral_constant<int, 1>{});
    } else if constexpr (ZS) {
      // all 27 taps unrolled: LDS offsets are immediates, and a row block skips the z tap that reads its padding plane
      auto zs_on = [](int t, int mb) { const int dz = t / 9; return mb == 0 ? dz >= 1 : dz <= 1; };
      auto zs_off = [&](int t, int mb) { const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3; return (((mb + dz - 1) * cHYs + dy) * cHXs + dx) * 36; };
      const float *Aw = A + wave * 8;
      f32x4 afn[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) afn[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        if (zs_on(0, mb)) afn[mb] = *reinterpret_cast<const f32x4 *>(&Aw[abase[mb] + zs_off(0, mb)]);
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const int d = t % PD;
        f32x4 af[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) af[mb] = afn[mb];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          if (t + 1 < 27 && zs_on(t + 1, mb)) afn[mb] = *reinterpret_cast<const f32x4 *>(&Aw[abase[mb] + zs_off(t + 1, mb)]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            if (zs_on(t, mb)) {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mb][jj], bq[d][nb][jj], acc[mb][nb], 0, 0, 0);
            }
        if (wleft > 0) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bq[d][nb] = wrun[nb * 64];
          wrun += 4 * NB * 64;
          --wleft;
        }
      }
    } else if constexpr (fast) {
      // 27 taps x 4 k8-steps = 108 steps per chunk, 27 per wave: wave w owns channels
      // [8w, 8w+8) of the chunk for every tap.  Weight fragments run PD steps ahead of
      // the MFMAs in a register ring (static indices: 27 = 9 x PD) and the stream
      // continues across chunk boundaries, so neither the L2 latency nor the staging
      // barrier exposes a weight load.
      // A fragments are read one tap ahead of their MFMAs (register double buffer), so the
      // LDS latency (bank conflicts included) hides behind 4*MB*NB MFMAs as well.
      // The matrix cores and the scalar / vector issue of a SIMD do not overlap for fp32 MFMAs
      // (tools/ubench/mfma_valu_overlap.hip: MFMA wave + VALU wave on one SIMD take the SUM of their
      // times), so every instruction in this loop costs its ~4 issue cycles on top of the MFMAs.
      // Tap offsets and the weight pointer therefore advance incrementally: the PD = TD unrolled taps
      // of one iteration are the dx steps of one (dz, dy) row of the kernel.
      constexpr int TD = (TAPS == 27) ? 3 : 2;
      static_assert(PD == TD, "the unrolled taps must be the dx steps of one kernel row");
      const int step_x = S, step_y = (HX - (TD - 1)) * S, step_z = ((HY - (TD - 1)) * HX - (TD - 1)) * S;
      int aoff = wave * (F16 ? 4 : 8), dyc = 0;
      f32x4 afn[MB];
      f32x2 afn2[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        if constexpr (F16) afn2[mb] = *reinterpret_cast<const f32x2 *>(&A[abase[mb] + aoff]);
        else afn[mb] = *reinterpret_cast<const f32x4 *>(&A[abase[mb] + aoff]);
      }
#pragma unroll 1
      for (int i0 = 0; i0 < TAPS; i0 += PD) {
        const int step_row = (i0 + PD >= TAPS) ? 0 : (dyc == TD - 1 ? step_z : step_y);  // (the last tap re-reads itself)
        dyc = (dyc == TD - 1) ? 0 : dyc + 1;
#pragma unroll
        for (int d = 0; d < PD; ++d) {
          f32x4 af[MB];
          f32x2 af2[MB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) { af[mb] = afn[mb]; af2[mb] = afn2[mb]; }
          aoff += (d < PD - 1) ? step_x : step_row;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            if constexpr (F16) afn2[mb] = *reinterpret_cast<const f32x2 *>(&A[abase[mb] + aoff]);
            else afn[mb] = *reinterpret_cast<const f32x4 *>(&A[abase[mb] + aoff]);
          }
          if constexpr (F16) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(f16x4, af2[mb]), __builtin_bit_cast(f16x4, bq2[d][nb]),
                                                                   acc[mb][nb], 0, 0, 0);
          } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
              for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                  acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mb][jj], bq[d][nb][jj], acc[mb][nb], 0, 0, 0);
          }
          // refill this ring slot with the fragments PD taps ahead (possibly next chunk): the stream
          // of one wave is linear in (chunk, tap), so a running pointer and a countdown suffice
          if (wleft > 0) {
            if constexpr (F16) {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) bq2[d][nb] = wrun2[nb * 64];
              wrun2 += 4 * NB * 64;
            } else {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) bq[d][nb] = wrun[nb * 64];
              wrun += 4 * NB * 64;
            }
            --wleft;
          }
        }
      }
    } else {
      const f32x4 *wch = wtile + (size_t)ch * nsteps * NB * 64;
      f32x4 bf[NB], bfn[NB];
      if (wave < nsteps) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bf[nb] = wch[(size_t)(wave * NB + nb) * 64];
      }
      for (int s = wave; s < nsteps; s += 4) {
        const int t = s / K8, j = s - t * K8;
        int tapoff = 0;
        if (td > 1) {
          const int dz = t / (td * td), rem = t - dz * td * td, dy = rem / td, dx = rem - dy * td;
          tapoff = (dz * HY + dy) * HX + dx;
        }
        const int aoff = tapoff * S + j * 8;
        const int sn = (s + 4 < nsteps) ? s + 4 : s;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bfn[nb] = wch[(size_t)(sn * NB + nb) * 64];
        f32x4 af[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) af[mb] = *reinterpret_cast<const f32x4 *>(&A[abase[mb] + aoff]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mb][jj], bf[nb][jj], acc[mb][nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bf[nb] = bfn[nb];
      }
    }
  }

  if constexpr (SKIPC) {
    if (a.s2w && kz == 0) {
      // ---- fused 1x1x1 skip convolution: K chunks over the raw block input, centre tap ----
      const int n2a = a.s2C0 >> 5, n2 = (a.s2C0 + a.s2C1) >> 5;
      const f32x4 *w2 = reinterpret_cast<const f32x4 *>(a.s2w) + (size_t)nt * n2 * 4 * NB * 64 + lane;
      const int q8 = tid & 7;
      for (int c2 = 0; c2 < n2; ++c2) {
        const float *src2;
        int Cs2, c02;
        if (c2 < n2a) { src2 = a.s2src0; Cs2 = a.s2C0; c02 = c2 * 32; }
        else { src2 = a.s2src1; Cs2 = a.s2C1; c02 = (c2 - n2a) * 32; }
        f32x4 wq[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wq[nb] = w2[((size_t)c2 * 4 + wave) * NB * 64 + nb * 64];
        __syncthreads();  // previous consumer of the A region is done
        f32x4 sv[MB];
#pragma unroll
        for (int k = 0; k < MB; ++k) {
          const int off = outoff[(tid >> 3) + 32 * k];
          sv[k] = *reinterpret_cast<const f32x4 *>(src2 + (size_t)(off >= 0 ? off : 0) * Cs2 + c02 + 4 * q8);
          if (off < 0) sv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < MB; ++k) *reinterpret_cast<f32x4 *>(&A[((tid >> 3) + 32 * k) * S + 4 * q8]) = sv[k];
        __syncthreads();
        f32x4 af2[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) af2[mb] = *reinterpret_cast<const f32x4 *>(&A[(mb * 32 + r) * S + wave * 8 + 4 * h]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af2[mb][jj], wq[nb][jj], acc[mb][nb], 0, 0, 0);
      }
    }
  }

  CM_RT(3)
  // ---- cross-wave reduction (rounds of RB blocks) + epilogue -------------------
  if (a.dbg & 4) {
    if (acc[0][0][0] == 123.456f) a.out[0] = 1.f;  // keep the accumulators live
    return;
  }
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
        // sum the four partials in wave order 0..3 whatever the owner: the result
        // must not depend on the tile geometry (batch-shard bit-exactness)
        const int owner = blk & 3;
        f32x16 v;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) v[reg] = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          if (w == owner) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) v[reg] += acc[mb][nb][reg];
          } else {
            const int slot = w - (w > owner ? 1 : 0);
            const float *sp = A + ((i * 3 + slot) * 16) * 64 + lane;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) v[reg] += sp[reg * 64];
          }
        }
        const int n = nt * TN + nb * 32 + r;
        {
          // branch-free gathers, then the stores (see the note on wait counts above)
          const bool nok = n < a.Co;
          const int nc = nok ? n : 0;
          const float bias = a.bias[nc];
          int offs[16];
          float rs[16];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) offs[reg] = outoff[mb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) rs[reg] = v[reg] + bias;
          if (a.temb) {  // wave-uniform
            if (a_bs == 1) {
              // one sample per tile: one time-embedding row for the whole block
              const float tv = a.temb[(size_t)a.tidx[b0 < a.B ? b0 : 0] * a.temb_stride + nc];
#pragma unroll
              for (int reg = 0; reg < 16; ++reg) rs[reg] += tv;
            } else {
#pragma unroll
              for (int reg = 0; reg < 16; ++reg) {
                const long long t = a.tidx[outb[mb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h]];
                rs[reg] += a.temb[(size_t)t * a.temb_stride + nc];
              }
            }
          }
          if (a.resid) {  // wave-uniform
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int oc = offs[reg] >= 0 ? offs[reg] : 0;
              rs[reg] += a.resid[(size_t)oc * a.res_cs + nc];
            }
          }
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            if (nok && offs[reg] >= 0) outp[(size_t)offs[reg] * a.out_cs + n] = rs[reg];
          if ((a.stat_part || a.astat) && !(a.dbg & 256)) {
            // fused GroupNorm statistics of this 32-row block (two-pass on registers, the
            // two lane halves merged with one cross-lane exchange): layers.py:30,41 read them
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
            const int slot = ((par * a.ntz + tz) * a.nty + ty) * a.ntx * MB + tx * MB + mb;
            if (a.astat) {                          // accumulator statistics (no gn_finalize launch): exact integer adds
              if (h == 0 && nok && b0 < a.B && cnt > 0.f) cm_stat_atomic(a.astat + ((size_t)b0 * a.astat_C + n) * 3, s1, mean, q);
            } else {
              if (h == 0 && nok && b0 < a.B) {
                float *sp = a.stat_part + (((size_t)b0 * a.stat_ns + slot) * a.stat_C + n) * 2;
                sp[0] = mean;
                sp[1] = q;
              }
              if (lane == 0 && n == 0 && b0 < a.B) a.stat_cnt[(size_t)b0 * a.stat_ns + slot] = cnt;
            }
          }
        }
      }
    }
  }
  if ((a.dbg & 8) && tid == 0 && a.dbg_buf) {
    tst[4] = __builtin_amdgcn_s_memrealtime();
    const size_t w = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (w < 8192) {
      unsigned long long *d = reinterpret_cast<unsigned long long *>(a.dbg_buf) + w * 8;
      for (int i = 0; i < 5; ++i) d[i] = tst[i];
    }
  }
}

// --------------------------------------------------------------------------------------------------------------
// 1x1x1 stride-1 convolution without output statistics (skip convs, attention in-projection, their data gradients;
// layers.py:14,46,74) as a flat-row GEMM: no LDS, no barriers.  A wave owns 32 consecutive rows (sample, voxel) and
// NB 32-channel output blocks; per 32-channel chunk it loads its A fragments straight from global memory (16 bytes
// per lane and k8 step, GroupNorm affine / SiLU / dropout multiplier applied in registers), the packed weight
// fragments of the chunk, and issues 16 NB MFMAs.  The generic kernel stages a per-sample halo tile through LDS with
// two barriers per chunk for the same 16 MFMAs per block: these launches were latency-bound at 25-55 us.
// --------------------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void conv1x1_flat_kernel(const ConvArgs a, long long N, int V) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const long long row0 = ((long long)blockIdx.x * 4 + wave) * 32;
  if (row0 >= N) return;                          // (wave-uniform; no barriers in this kernel)
  const int nt = blockIdx.y;
  const long long n = row0 + r, nn = n < N ? n : N - 1;
  const int b = (int)(nn / V);
  const int Ctot = a.C0 + a.C1;
  const int n0 = a.C0 >> 5, nchunks = n0 + (a.C1 >> 5);
  const f32x4 *wt = reinterpret_cast<const f32x4 *>(a.wfrag) + (size_t)nt * nchunks * 4 * NB * 64 + lane;
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
  for (int ch = 0; ch < nchunks; ++ch) {
    const float *src;
    int Cs, c0, cg0;
    if (ch < n0) { src = a.src0; Cs = a.C0; c0 = ch * 32; cg0 = c0; }
    else { src = a.src1; Cs = a.C1; c0 = (ch - n0) * 32; cg0 = a.C0 + c0; }
    const float *ap = src + (size_t)nn * Cs + c0 + 4 * h;
    f32x4 av[4], wv[4][NB];
#pragma unroll
    for (int k8 = 0; k8 < 4; ++k8) av[k8] = *reinterpret_cast<const f32x4 *>(ap + 8 * k8);
#pragma unroll
    for (int k8 = 0; k8 < 4; ++k8)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wv[k8][nb] = wt[((size_t)(ch * 4 + k8) * NB + nb) * 64];
    if (a.gn) {
      const float *g = a.gn + (size_t)b * 2 * Ctot + cg0 + 4 * h;
#pragma unroll
      for (int k8 = 0; k8 < 4; ++k8) {
        av[k8] = av[k8] * *reinterpret_cast<const f32x4 *>(g + 8 * k8) + *reinterpret_cast<const f32x4 *>(g + Ctot + 8 * k8);
        if (a.silu) { av[k8][0] = silu_f(av[k8][0]); av[k8][1] = silu_f(av[k8][1]); av[k8][2] = silu_f(av[k8][2]); av[k8][3] = silu_f(av[k8][3]); }
      }
    }
    if (a.pm) {
      const float *pmp = a.pm + (size_t)b * a.pm_stride + cg0 + 4 * h;
#pragma unroll
      for (int k8 = 0; k8 < 4; ++k8) av[k8] = av[k8] * *reinterpret_cast<const f32x4 *>(pmp + 8 * k8);
    }
#pragma unroll
    for (int k8 = 0; k8 < 4; ++k8)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k8][jj], wv[k8][nb][jj], acc[nb], 0, 0, 0);
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int co = (nt * NB + nb) * 32 + r;
    if (co >= a.Co) continue;
    const float bias = a.bias[co];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const long long orow = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (orow < N) {
        float v = acc[nb][reg] + bias;
        if (a.resid) v += a.resid[(size_t)orow * a.res_cs + co];
        a.out[(size_t)orow * a.out_cs + co] = v;
      }
    }
  }
}

// The same flat-row GEMM on f16 matrix-core operands (reduced-precision plan, BASELINE configs[4] "fp16 with MFMA 1x1 convs"; the
// reference's autocast covers the attention projections, ddpm.py:116-120 / layers.py:14): a wave owns 32 rows and NB 32-channel
// output blocks, a step is 16 input channels -- lane (row r, half h) loads channels 16 g + 8 h .. + 7 (two 16-byte loads), applies
// the GroupNorm affine / SiLU / dropout multiplier in fp32, rounds to f16, and issues ONE v_mfma_f32_32x32x16_f16 per output block
// against the packed f16 fragments (pack_1x1_f16: [n tile][16-channel group][block][lane][8 halves]); fp32 accumulation, bias and
// residual in fp32.  No statistics (conv1x1_f16_ok): the attention in-projection and unfused skip convs.  (A variant whose
// 32-row blocks were cut per sample and wrote statistics slots -- the out-projection -- measured 1 % SLOWER per step than the
// generic fp32 kernel on the 24x72 grid and is not kept: that launch is a latency chain, not matrix work.)
typedef _Float16 f16x8c __attribute__((ext_vector_type(8)));
template <int NB>
__global__ __launch_bounds__(256) void conv1x1_f16_kernel(const ConvArgs a, long long N, int V) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const long long row0 = ((long long)blockIdx.x * 4 + wave) * 32;
  if (row0 >= N) return;                          // (wave-uniform; no barriers in this kernel)
  const int nt = blockIdx.y;
  const long long n = row0 + r, nn = n < N ? n : N - 1;
  const int b = (int)(nn / V);
  const int Ctot = a.C0 + a.C1;
  const int g0n = a.C0 >> 4, ngr = g0n + (a.C1 >> 4);
  const f32x4 *wt = reinterpret_cast<const f32x4 *>(a.wfrag) + (size_t)nt * ngr * NB * 64 + lane;
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
  // four 16-channel groups per round: all their loads (rows, affine rows, weight fragments) are issued before the first use --
  // one memory round trip per 64 channels (a load -> use loop pays one per group: 8 dependent round trips for 128 channels)
  constexpr int GB = 4;
  for (int g0 = 0; g0 < ngr; g0 += GB) {
    f32x4 a0[GB], a1[GB], s0[GB], s1[GB], h0[GB], h1[GB], p0[GB], p1[GB], wv[GB][NB];
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      const int g = g0 + u < ngr ? g0 + u : ngr - 1;
      const float *src;
      int Cs, c0, cg0;
      if (g < g0n) { src = a.src0; Cs = a.C0; c0 = g * 16; cg0 = c0; }
      else { src = a.src1; Cs = a.C1; c0 = (g - g0n) * 16; cg0 = a.C0 + c0; }
      const float *ap = src + (size_t)nn * Cs + c0 + 8 * h;
      a0[u] = *reinterpret_cast<const f32x4 *>(ap);
      a1[u] = *reinterpret_cast<const f32x4 *>(ap + 4);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wv[u][nb] = wt[((size_t)g * NB + nb) * 64];
      if (a.gn) {
        const float *gp = a.gn + (size_t)b * 2 * Ctot + cg0 + 8 * h;
        s0[u] = *reinterpret_cast<const f32x4 *>(gp); s1[u] = *reinterpret_cast<const f32x4 *>(gp + 4);
        h0[u] = *reinterpret_cast<const f32x4 *>(gp + Ctot); h1[u] = *reinterpret_cast<const f32x4 *>(gp + Ctot + 4);
      }
      if (a.pm) {
        const float *pmp = a.pm + (size_t)b * a.pm_stride + cg0 + 8 * h;
        p0[u] = *reinterpret_cast<const f32x4 *>(pmp); p1[u] = *reinterpret_cast<const f32x4 *>(pmp + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      if (g0 + u >= ngr) break;
      f32x4 x0 = a0[u], x1 = a1[u];
      if (a.gn) {
        x0 = x0 * s0[u] + h0[u];
        x1 = x1 * s1[u] + h1[u];
        if (a.silu) {
          x0[0] = silu_f(x0[0]); x0[1] = silu_f(x0[1]); x0[2] = silu_f(x0[2]); x0[3] = silu_f(x0[3]);
          x1[0] = silu_f(x1[0]); x1[1] = silu_f(x1[1]); x1[2] = silu_f(x1[2]); x1[3] = silu_f(x1[3]);
        }
      }
      if (a.pm) { x0 = x0 * p0[u]; x1 = x1 * p1[u]; }
      const f16x8c af = {(_Float16)x0[0], (_Float16)x0[1], (_Float16)x0[2], (_Float16)x0[3],
                         (_Float16)x1[0], (_Float16)x1[1], (_Float16)x1[2], (_Float16)x1[3]};
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(f16x8c, wv[u][nb]), acc[nb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int co = (nt * NB + nb) * 32 + r;
    if (co >= a.Co) continue;
    const float bias = a.bias[co];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const long long orow = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (orow < N) {
        float v = acc[nb][reg] + bias;
        if (a.resid) v += a.resid[(size_t)orow * a.res_cs + co];
        a.out[(size_t)orow * a.out_cs + co] = v;
      }
    }
  }
}

bool conv1x1_f16_ok(const ConvArgs &a, int NB) {
  return a.ntaps == 1 && a.td == 1 && a.stride == 1 && !a.par && !a.ups && !(a.C0 & 15) && !(a.C1 & 15) && !a.stat_part && !a.astat && a.ks <= 1 &&
         !a.temb && !a.s2w && (NB == 1 || NB == 2) && a.Zs == a.Zo && a.Ys == a.Yo && a.Xs == a.Xo;
}

hipError_t launch_conv1x1_f16(const ConvArgs &a, int NB, hipStream_t st) {
  if (!conv1x1_f16_ok(a, NB)) return hipErrorInvalidValue;
  const int V = a.Zo * a.Yo * a.Xo;
  const long long N = (long long)a.B * V;
  const dim3 grid((unsigned)((N + 127) / 128), (unsigned)((a.Co + 32 * NB - 1) / (32 * NB)));
  if (NB == 1) hipLaunchKernelGGL(conv1x1_f16_kernel<1>, grid, dim3(256), 0, st, a, N, V);
  else hipLaunchKernelGGL(conv1x1_f16_kernel<2>, grid, dim3(256), 0, st, a, N, V);
  return hipGetLastError();
}

bool conv1x1_flat_ok(const ConvArgs &a, int NB) {
  return a.ntaps == 1 && a.td == 1 && a.stride == 1 && !a.par && !a.ups && a.CK == 32 && !(a.C0 & 31) && !(a.C1 & 31) && !a.stat_part && !a.astat &&
         a.ks <= 1 && !a.temb && !a.s2w && !a.f16 && (NB == 1 || NB == 2) && a.Zs == a.Zo && a.Ys == a.Yo && a.Xs == a.Xo &&
         !(conv_dbg_flags() & 65536);
}

size_t conv_lds_bytes(const ConvArgs &a, int MB, int NB) {
  const int HZ = (a.bz - 1) * a.stride + a.td;
  const int HY = (a.by - 1) * a.stride + a.td;
  const int HX = (a.bx - 1) * a.stride + a.td;
  const int HV = a.bs * HZ * HY * HX;
  const int HVp = (HV + 3) & ~3;
  const int S = a.CK + 4;
  const int nblk = MB * NB;
  const int rb = nblk < 4 ? nblk : 4;
  size_t tile = (size_t)HV * S;
  size_t red = (size_t)rb * 3 * 16 * 64;
  (void)HVp;
  size_t words = (size_t)64 * MB + (tile > red ? tile : red);
  return words * 4;
}

int conv_halo_voxels(const ConvArgs &a) {
  const int HZ = (a.bz - 1) * a.stride + a.td;
  const int HY = (a.by - 1) * a.stride + a.td;
  const int HX = (a.bx - 1) * a.stride + a.td;
  return a.bs * HZ * HY * HX;
}

void conv_build_tables(const ConvArgs &a, int MB, int *hvtab, int *mtab) {
  const int HZ = (a.bz - 1) * a.stride + a.td;
  const int HY = (a.by - 1) * a.stride + a.td;
  const int HX = (a.bx - 1) * a.stride + a.td;
  int i = 0;
  for (int s = 0; s < a.bs; ++s)
    for (int hz = 0; hz < HZ; ++hz)
      for (int hy = 0; hy < HY; ++hy)
        for (int hx = 0; hx < HX; ++hx) hvtab[i++] = (s << 26) | (hz << 18) | (hy << 9) | hx;
  const int nbox = a.bs * a.bz * a.by * a.bx;
  for (int m = 0; m < 32 * MB; ++m) {
    if (m >= nbox) { mtab[m] = -1; continue; }
    const int x = m % a.bx;
    int q = m / a.bx;
    const int y = q % a.by; q /= a.by;
    const int z = q % a.bz;
    const int s = q / a.bz;
    mtab[m] = (s << 26) | (z << 18) | (y << 9) | x;
  }
}

#define CM_CONV_VARIANTS(X) \
  X(1, 1) X(2, 1) X(3, 1) X(4, 1) X(5, 1) X(6, 1) X(7, 1) X(8, 1) \
  X(1, 2) X(2, 2) X(3, 2) X(4, 2) \
  X(1, 4) X(2, 4)

bool conv_par_f16_variant(int MB, int NB, int bz, int by, int bx) {
  return (MB == 5 && NB == 1 && bz == 4 && by == 6 && bx == 6) || (MB == 2 && NB == 2 && bz == 2 && by == 3 && bx == 9) ||
         (MB == 3 && NB == 1 && bz == 2 && by == 7 && bx == 6) || (MB == 2 && NB == 2 && bz == 4 && by == 4 && bx == 4);
}

// does launch_conv run this (stride-1, one sample per tile) op in a z-split form?  (executed-FLOP accounting: 18 of 27
// taps resp. 6 of 8 parity taps are issued)
bool conv_zsplit_variant(const ConvArgs &a, int MB, int NB) {
  if (conv_dbg_flags() & (2048 | 16384)) return false;
  if (!(MB == 2 && NB == 2 && a.bz == 2 && a.ntz == 1 && a.bs == 1 && a.stride == 1 && !a.ups && a.CK == 32 && a.Zs == 2)) return false;
  if (a.par) return a.ntaps == 8 && a.td == 2 && !a.f16 && a.by == 3 && a.bx == 9;
  return a.ntaps == 27 && a.td == 3 && a.Zo == 2 && ((a.by == 3 && a.bx == 9) || (a.by == 6 && a.bx == 5));
}

bool conv_variant_exists(int MB, int NB) {
#define X(m, n) if (MB == m && NB == n) return true;
  CM_CONV_VARIANTS(X)
#undef X
  return false;
}

static void conv_dbg_report(const ConvArgs &a, int MB, int NB, int F, dim3 grid, unsigned long long *buf, hipStream_t st) {
  static int shown = 0;
  (void)hipStreamSynchronize(st);
  if (shown++ >= 60) return;
  const size_t nwg = std::min<size_t>((size_t)grid.x * grid.y * grid.z, 8192);
  std::vector<unsigned long long> h(nwg * 8);
  (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  double ph[4] = {0, 0, 0, 0};
  for (size_t w = 0; w < nwg; ++w) {
    if (!h[w * 8]) continue;
    t0 = std::min(t0, h[w * 8]); t1 = std::max(t1, h[w * 8 + 4]);
    for (int i = 0; i < 4; ++i) ph[i] += (double)(h[w * 8 + i + 1] - h[w * 8 + i]) / nwg;
  }
  fprintf(stderr, "conv<%d,%d,%d> Ci=%d Co=%d taps=%d ks=%d grid=%ux%ux%u: span %.1f us | per WG (us): setup %.2f first-stage %.2f chunks(mfma+stage) %.2f reduce+epilogue %.2f\n",
          MB, NB, F, a.C0 + a.C1, a.Co, a.ntaps, a.ks, grid.x, grid.y, grid.z, (t1 - t0) / 100.0, ph[0] / 100, ph[1] / 100, ph[2] / 100, ph[3] / 100);
}

int conv_dbg_override = -1;
int conv_dbg_flags() {
  static const int env = cm::diag_env("CM_CONV_DBG") ? atoi(cm::diag_env("CM_CONV_DBG")) : 0;  // ablation switches (perf studies only)
  return conv_dbg_override >= 0 ? conv_dbg_override : env;
}

hipError_t launch_conv(const ConvArgs &a_in, int MB, int NB, hipStream_t st) {
  const int dbg = conv_dbg_flags();
  if (conv1x1_flat_ok(a_in, NB)) {
    const int V = a_in.Zo * a_in.Yo * a_in.Xo;
    const long long N = (long long)a_in.B * V;
    const dim3 grid((unsigned)((N + 127) / 128), (unsigned)((a_in.Co + 32 * NB - 1) / (32 * NB)));
    if (NB == 1) hipLaunchKernelGGL(conv1x1_flat_kernel<1>, grid, dim3(256), 0, st, a_in, N, V);
    else hipLaunchKernelGGL(conv1x1_flat_kernel<2>, grid, dim3(256), 0, st, a_in, N, V);
    return hipGetLastError();
  }
  static const int stg = cm::diag_env("CM_CONV_STAGGER") ? atoi(cm::diag_env("CM_CONV_STAGGER")) : -1;
  ConvArgs a = a_in;
  a.dbg = dbg;
  if (stg >= 0) a.stagger = stg;
  static unsigned long long *dbgbuf = nullptr;
  if (dbg & 8) {
    if (!dbgbuf) (void)hipMalloc((void **)&dbgbuf, 8192 * 8 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbgbuf, 0, 8192 * 8 * sizeof(unsigned long long), st);
    a.dbg_buf = reinterpret_cast<float *>(dbgbuf);
  }
  const size_t lds = conv_lds_bytes(a, MB, NB);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int TN = 32 * NB;
  if (a.par && a.ks > 1) return hipErrorInvalidValue;
  dim3 grid((unsigned)(a.nts * a.ntz * a.nty * a.ntx), (unsigned)((a.Co + TN - 1) / TN),
            a.par ? 8u : (a.ks > 1 ? (unsigned)a.ks : 1u));
  const int fastk = (a.CK == 32 && (a.ntaps == 27 || a.ntaps == 8)) ? a.ntaps : 0;
  if (a.f16 && !(fastk == 8 && a.bs == 1 && a.stride == 1 && a.par && !a.ups && a.td == 2)) return hipErrorInvalidValue;
#define CM_LAUNCH_T(KERNEL, m, n, f)                                                             \
  {                                                                                              \
    static bool attr_set[64] = {false};                                                          \
    int dev = 0;                                                                                 \
    (void)hipGetDevice(&dev);                                                                    \
    if (!attr_set[dev & 63]) {                                                                   \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL),                 \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return e;                                                             \
      attr_set[dev & 63] = true;                                                                 \
    }                                                                                            \
    hipLaunchKernelGGL(KERNEL, grid, dim3(256), lds, st, a);                                     \
    if (dbg & 8) conv_dbg_report(a, m, n, f, grid, dbgbuf, st);                                  \
    return hipGetLastError();                                                                    \
  }
#define CM_LAUNCH(m, n, f) CM_LAUNCH_T((conv_mfma_kernel<m, n, f>), m, n, f)
  // shape-specialised instantiations for the hot tile shapes of the reference grids
  const bool specok = fastk == 27 && a.bs == 1 && a.stride == 1 && !a.par && !a.ups && a.td == 3 && !(dbg & 2048);
#define CM_SPEC(m, n, z, y, x)                                                     \
  if (specok && MB == m && NB == n && a.bz == z && a.by == y && a.bx == x) {       \
    if (a.s2w) CM_LAUNCH_T((conv_mfma_kernel<m, n, 127, z, y, x>), m, n, 127)      \
    CM_LAUNCH_T((conv_mfma_kernel<m, n, 27, z, y, x>), m, n, 27)                   \
  }
#define CM_SPEC3(m, n, z, y, x)                                                    \
  if (specok && MB == m && NB == n && a.bz == z && a.by == y && a.bx == x) {       \
    if (a.s2w) CM_LAUNCH_T((conv_mfma_kernel<m, n, 127, z, y, x, 1, 3>), m, n, 127) \
    CM_LAUNCH_T((conv_mfma_kernel<m, n, 27, z, y, x, 1, 3>), m, n, 27)             \
  }
  // z-split form: grids with exactly two z planes (quarter resolution), one plane per row block
  const bool zsok = specok && a.Zo == 2 && a.Zs == 2 && a.ntz == 1 && MB == 2 && NB == 2 && a.bz == 2 && !(dbg & 16384);
#define CM_SPEC_ZS(y, x)                                                           \
  if (zsok && a.by == y && a.bx == x) {                                            \
    if (a.s2w) CM_LAUNCH_T((conv_mfma_kernel<2, 2, 427, 2, y, x>), 2, 2, 427)      \
    CM_LAUNCH_T((conv_mfma_kernel<2, 2, 327, 2, y, x>), 2, 2, 327)                 \
  }
  CM_SPEC_ZS(3, 9)   // ATC 3 x 9 x 2
  CM_SPEC_ZS(6, 5)   // 2x grid 6 x 18 x 2
  // (HERMES-CR-120's 7 x 6 x 2 quarter resolution keeps its tuned MB3 2x7x6 tile: 84 of 96 rows; a 7 x 4 z-split tile was not tuned)
#undef CM_SPEC_ZS
  CM_SPEC3(3, 1, 8, 6, 2)  // full resolution, three workgroups per CU (128 VGPRs): -7 % vs MB4 8x4x4 at 2 per CU
  CM_SPEC3(3, 1, 8, 4, 3)  //   (8x3x4, 4x4x6, 4x6x4, 8x2x6 measured slower: 118 / 136 / 111 / 134 us vs 108 on the 32->32 layer)
#undef CM_SPEC3
  CM_SPEC(4, 1, 8, 4, 4)   // ATC / 2x grid full resolution
  CM_SPEC(2, 2, 2, 3, 9)   // ATC half and quarter resolution
  CM_SPEC(2, 2, 2, 6, 5)
  CM_SPEC(1, 2, 2, 3, 5)
  CM_SPEC(4, 1, 4, 4, 8)   // HERMES-CR-120 / 2x grid
  CM_SPEC(3, 1, 2, 7, 6)
  CM_SPEC(2, 2, 2, 7, 4)
  CM_SPEC(2, 2, 4, 3, 5)
  CM_SPEC(2, 2, 4, 4, 4)
#undef CM_SPEC
  // parity (upsample) form and stride-2 convs
  const bool specpar = fastk == 8 && a.bs == 1 && a.stride == 1 && a.par && !a.ups && a.td == 2 && !(dbg & 2048);
  const bool specs2 = fastk == 27 && a.bs == 1 && a.stride == 2 && !a.par && !a.ups && a.td == 3 && !a.s2w && !(dbg & 2048);
#define CM_SPEC_PAR(m, n, z, y, x)                                                 \
  if (specpar && MB == m && NB == n && a.bz == z && a.by == y && a.bx == x) {      \
    if (a.f16) CM_LAUNCH_T((conv_mfma_kernel<m, n, 208, z, y, x>), m, n, 208)      \
    if constexpr (m == 2 && n == 2 && z == 2 && y * x <= 32) {                     \
      if (a.Zs == 2 && a.ntz == 1 && !(dbg & 16384)) CM_LAUNCH_T((conv_mfma_kernel<2, 2, 308, 2, y, x>), 2, 2, 308) \
    }                                                                              \
    CM_LAUNCH_T((conv_mfma_kernel<m, n, 8, z, y, x>), m, n, 8)                     \
  }
#define CM_SPEC_S2(m, n, z, y, x)                                                  \
  if (specs2 && MB == m && NB == n && a.bz == z && a.by == y && a.bx == x)         \
    CM_LAUNCH_T((conv_mfma_kernel<m, n, 27, z, y, x, 2>), m, n, 27)
  CM_SPEC_PAR(5, 1, 4, 6, 6)
  CM_SPEC_PAR(2, 2, 2, 3, 9)
  CM_SPEC_PAR(3, 1, 2, 7, 6)
  CM_SPEC_PAR(2, 2, 4, 4, 4)
  CM_SPEC_S2(1, 1, 4, 2, 4)
  CM_SPEC_S2(1, 2, 1, 3, 9)
  CM_SPEC_S2(1, 1, 2, 4, 4)
  CM_SPEC_S2(1, 1, 2, 7, 2)
  CM_SPEC_S2(1, 2, 2, 3, 5)
  CM_SPEC_S2(1, 2, 2, 7, 2)
#undef CM_SPEC_PAR
#undef CM_SPEC_S2
#define X(m, n)                                    \
  if (MB == m && NB == n) {                        \
    if (fastk == 27 && a.s2w) CM_LAUNCH(m, n, 127) \
    if (fastk == 27) CM_LAUNCH(m, n, 27)           \
    if (fastk == 8) CM_LAUNCH(m, n, 8)             \
    CM_LAUNCH(m, n, 0)                             \
  }
  CM_CONV_VARIANTS(X)
#undef X
  return hipErrorInvalidValue;
}

}  // namespace cm
