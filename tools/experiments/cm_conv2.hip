// 3x3x3 implicit-GEMM convolution, persistent + software-pipelined variant.
//
// Used for the layers with many output voxels per sample (full / half resolution,
// ~85 % of the UNet's FLOPs; /root/reference/models/backbones/layers.py:32,43,94).
// Same math and the same K-split MFMA inner loop as cm_conv.hip (every wave owns the
// whole 32*MB x 32*NB output tile and one 8-channel slice of each tap, so a weight
// byte is fetched once per workgroup and used for 32*MB rows), but the workgroup is
// persistent and pipelines ACROSS work units:
//
//   unit = (tile, 32-channel chunk);  tiles come from an atomic counter.
//   while the matrix cores work on the LDS image of unit u, the halo of unit u+1 is
//   already in flight global -> VGPR (its addresses need no table and no division:
//   the per-thread halo coordinates are loaded once per kernel).  After the MFMA
//   phase it is normalised (GroupNorm affine + SiLU) and written to LDS.  The weight
//   ring keeps streaming across unit boundaries.
//
// Measured motivation (profiles/round1_notes.md): on the non-persistent kernel the
// load -> LDS -> MFMA -> reduce -> store phases of a workgroup are chains of dependent
// global-memory latencies at 2 waves/SIMD; they do not overlap with anything and cost
// as much as the MFMA phase itself.
#include "cm_kernels.h"

#include <cstdio>
#include <cstdlib>

namespace cm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division sequence
__device__ __forceinline__ float silu2_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global load/store of the wave (vmcnt(0)), which would drain the halo
// prefetch, the weight ring and the epilogue stores at each of the 2-4 barriers per unit.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// MB x NB accumulator blocks per wave; NV = halo float4 per thread (256 threads).
template <int MB, int NB, int NV>
__global__ __launch_bounds__(256, 2) void conv3_persist_kernel(const ConvArgs a, int *__restrict__ ctr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TM = 32 * MB;
  constexpr int NBLK = MB * NB;
  constexpr int RB = NBLK < 4 ? NBLK : 4;
  constexpr int PD = 3;   // weight ring depth in taps (27 = 9 x PD)
  constexpr int S = 36;   // LDS row stride (dwords) of a 32-channel chunk
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nt = blockIdx.y;
  if (!(a.dbg & 32)) __builtin_amdgcn_s_setprio(3);

  const int HZ = a.bz + 2, HY = a.by + 2, HX = a.bx + 2;  // stride 1, pad 1, one sample per tile
  const int HV = HZ * HY * HX;
  const int ntiles = a.nts * a.ntz * a.nty * a.ntx;
  const int nchunks = a.nch0 + a.nch1;
  const int Ctot = a.C0 + a.C1;
  const int Zc = a.Zs << a.ups, Yc = a.Ys << a.ups, Xc = a.Xs << a.ups;

  int *ctrl = reinterpret_cast<int *>(lds);  // [4]
  int *outoff = ctrl + 4;                    // [TM] output voxel index or -1
  float *A = lds + 4 + TM;                   // [HV][S] halo image / reduction scratch

  // ---- static per-thread staging geometry ---------------------------------------
  const int q4 = tid & 7, vl = tid >> 3;     // channel quad, voxel lane (32 lanes)
  int hvpk[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int hv = vl + k * 32;
    hvpk[k] = hv < HV ? a.hvtab[hv] : -1;
  }
  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int pk = a.mtab[mb * 32 + r];
    int hv = 0;
    if (pk >= 0) hv = (((pk >> 18) & 255) * HY + ((pk >> 9) & 511)) * HX + (pk & 511);
    abase[mb] = hv * S + 4 * h + 8 * wave;   // this wave's 8-channel slice
  }

  // ---- work distribution: tiles from an atomic counter, two in flight ---------------
  if (tid == 0) {
    ctrl[0] = atomicAdd(&ctr[nt * 2], 1);
    ctrl[1] = atomicAdd(&ctr[nt * 2], 1);
  }
  __syncthreads();
  int cur = ctrl[0], nxt = ctrl[1], nn = 0;
  __syncthreads();

  f32x4 pre[NV];   // halo of the next unit, in flight
  f32x4 gsc, gsh;  // its GroupNorm scale / shift for this thread's channel quad
  unsigned pvalid = 0;

  auto tile_origin = [&](int tile, int &b0, int &z0, int &y0, int &x0) {
    const int tx = tile % a.ntx; tile /= a.ntx;
    const int ty = tile % a.nty; tile /= a.nty;
    const int tz = tile % a.ntz;
    b0 = tile / a.ntz; z0 = tz * a.bz; y0 = ty * a.by; x0 = tx * a.bx;
  };

  auto issue_loads = [&](int tile, int uch) {
    int b0, z0, y0, x0;
    tile_origin(tile, b0, z0, y0, x0);
    const float *src;
    int Cs, c0, cg0;
    if (uch < a.nch0) { src = a.src0; Cs = a.C0; c0 = uch * 32; cg0 = c0; }
    else { src = a.src1; Cs = a.C1; c0 = (uch - a.nch0) * 32; cg0 = a.C0 + c0; }
    const float *srcq = src + ((size_t)b0 * a.Zs * a.Ys * a.Xs) * Cs + c0 + 4 * q4;
    pvalid = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      // branch-free: out-of-range halo voxels load voxel 0 of the sample (a cached line)
      // and are zeroed when the image is written (pvalid), so the wait counts stay exact
      const int pk = hvpk[k];
      const int cx = x0 - 1 + (pk & 511), cy = y0 - 1 + ((pk >> 9) & 511), cz = z0 - 1 + ((pk >> 18) & 255);
      const bool ok = pk >= 0 && cz >= 0 && cz < Zc && cy >= 0 && cy < Yc && cx >= 0 && cx < Xc;
      const int off = ok ? (((cz >> a.ups)) * a.Ys + (cy >> a.ups)) * a.Xs + (cx >> a.ups) : 0;
      pre[k] = *reinterpret_cast<const f32x4 *>(srcq + (size_t)off * Cs);
      pvalid |= (ok ? 1u : 0u) << k;
    }
    if (a.gn) {
      const float *g = a.gn + (size_t)b0 * 2 * Ctot + cg0 + 4 * q4;
      gsc = *reinterpret_cast<const f32x4 *>(g);
      gsh = *reinterpret_cast<const f32x4 *>(g + Ctot);
    }
  };

  auto write_lds = [&]() {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (hvpk[k] >= 0) {
        f32x4 v = pre[k];
        if (a.gn) {
          v = v * gsc + gsh;
          if (a.silu) { v[0] = silu2_f(v[0]); v[1] = silu2_f(v[1]); v[2] = silu2_f(v[2]); v[3] = silu2_f(v[3]); }
        }
        if (!((pvalid >> k) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};  // zero padding (after the activation)
        *reinterpret_cast<f32x4 *>(&A[(vl + k * 32) * S + 4 * q4]) = v;
      }
    }
  };

  auto build_rows = [&](int tile) {
    int b0, z0, y0, x0;
    tile_origin(tile, b0, z0, y0, x0);
    for (int m = tid; m < TM; m += 256) {
      const int pk = a.mtab[m];
      int off = -1;
      if (pk >= 0) {
        const int oz = z0 + ((pk >> 18) & 255), oy = y0 + ((pk >> 9) & 511), ox = x0 + (pk & 511);
        if (oz < a.Zo && oy < a.Yo && ox < a.Xo) off = ((b0 * a.Zo + oz) * a.Yo + oy) * a.Xo + ox;
      }
      outoff[m] = off;
    }
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  // weights: [ntile][chunk][step = tap*4 + k8][nb][lane][4]; this wave uses k8 == wave
  const f32x4 *wtile = reinterpret_cast<const f32x4 *>(a.wfrag) + ((size_t)nt * nchunks * 108 + wave) * NB * 64 + lane;
  f32x4 bq[PD][NB];
  float eadd[NB];

  if (cur < ntiles) {
    issue_loads(cur, 0);
    write_lds();
    build_rows(cur);
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bq[d][nb] = wtile[(size_t)(4 * d * NB + nb) * 64];
  }
  __syncthreads();

  // optional per-phase cycle accounting (dbg bit 3): wave 0's view, written to stat_part
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = (a.dbg & 8) ? clock64() : 0;
#define CM_STAMP(i)                                   \
  if (a.dbg & 8) {                                    \
    const long long now_ = clock64();                 \
    tph[i] += now_ - tlast;                           \
    tlast = now_;                                     \
  }
  int ch = 0;
  while (cur < ntiles) {
    // ---- next unit ---------------------------------------------------------------
    int ntile = cur, nch = ch + 1;
    bool has_next = true;
    if (nch >= nchunks) {
      nch = 0;
      ntile = nxt;
      has_next = nxt < ntiles;
    }
    if (ch == 0 && tid == 0) ctrl[2] = atomicAdd(&ctr[nt * 2], 1);  // tile index two ahead
    if (ch == 0) {  // epilogue constants of this tile: bias + time-embedding row of its sample
      // (uniform values go through the scalar unit; issued BEFORE the halo prefetch because
      //  vector-memory waits are in order: a dependent load behind the prefetch would wait for it)
      const int b0 = __builtin_amdgcn_readfirstlane(cur) / (a.ntz * a.nty * a.ntx);
      const float *trow = a.temb ? a.temb + (size_t)a.tidx[b0] * a.temb_stride : nullptr;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = (nt * NB + nb) * 32 + r;
        float e = 0.f;
        if (n < a.Co) {
          e = a.bias[n];
          if (trow) e += trow[n];
        }
        eadd[nb] = e;
      }
    }
    if (has_next && !(a.dbg & 1)) issue_loads(ntile, nch);
    CM_STAMP(0)

    // ---- MFMA phase over the LDS image of (cur, ch): 27 taps, this wave's k8 slice ----
    // The MFMA stream needs one issue slot per 64 cycles; everything else in this kernel is
    // short vector/LDS/memory code that shares the SIMD with the co-resident workgroup's
    // MFMA stream.  Run the MFMA loop at low priority and the rest at high priority, so the
    // neighbour's load/store phases are not starved of issue slots by our matrix work.
    if (!(a.dbg & 32)) __builtin_amdgcn_s_setprio(0);
    if (!(a.dbg & 2)) {
      const int ch_ring = has_next ? nch : ch;  // past the last unit the ring re-reads valid addresses
#pragma unroll 1
      for (int t0 = 0; t0 < 27; t0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
          const int t = t0 + d;
          const int dz = t / 9, rem = t - dz * 9, dy = rem / 3, dx = rem - dy * 3;
          const int aoff = ((dz * HY + dy) * HX + dx) * S;
          f32x4 af[MB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) af[mb] = *reinterpret_cast<const f32x4 *>(&A[abase[mb] + aoff]);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mb][jj], bq[d][nb][jj], acc[mb][nb], 0, 0, 0);
          int tn = t + PD, chn = ch;
          if (tn >= 27) { tn -= 27; chn = ch_ring; }
          const f32x4 *wp = wtile + ((size_t)chn * 108 + 4 * tn) * NB * 64;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bq[d][nb] = wp[nb * 64];
        }
      }
    }

    if (!(a.dbg & 32)) __builtin_amdgcn_s_setprio(3);
    CM_STAMP(1)
    const bool last_chunk = (ch == nchunks - 1);
    lds_barrier();  // (A) every wave is done reading the LDS image
    CM_STAMP(2)
    if (last_chunk && !(a.dbg & 4)) {
      // ---- cross-wave reduction through LDS (scratch aliases the image) + epilogue ----
#pragma unroll
      for (int g0 = 0; g0 < NBLK; g0 += RB) {
        if (g0 > 0) lds_barrier();
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
        CM_STAMP(6)
        lds_barrier();
        CM_STAMP(7)
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int blk = g0 + i;
          if (blk < NBLK && wave == (blk & 3)) {
            const int mb = blk / NB, nb = blk % NB;
            const int owner = blk & 3;
            f32x16 v;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) v[reg] = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {  // fixed wave order: result independent of geometry
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
            if (a.dbg & 8) {
              float keep = 0.f;
              for (int reg = 0; reg < 16; ++reg) keep += v[reg];
              asm volatile("" ::"v"(keep));
              const long long now_ = clock64();
              tph[6] += 0;
              tph[7] += now_ - tlast;  // reduce reads (accounted into slot 7 together with the barrier)
              tlast = now_;
            }
            const int n = (nt * NB + nb) * 32 + r;
            {
              // Branch-free gather of the residual, then the stores.  Loads inside divergent
              // branches make hipcc fall back to s_waitcnt vmcnt(0) before EVERY store (each
              // store then waits for the previous store's write-ack); and out / resid may alias
              // as far as the compiler knows, so all loads come first.
              const bool nok = n < a.Co;
              const int nc = nok ? n : 0;
              const float add = eadd[nb];
              int offs[16];
              float rs[16];
#pragma unroll
              for (int reg = 0; reg < 16; ++reg) offs[reg] = outoff[mb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h];
              if (a.resid) {  // wave-uniform
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                  const int oc = offs[reg] >= 0 ? offs[reg] : 0;
                  rs[reg] = a.resid[(size_t)oc * a.res_cs + nc];
                }
              } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) rs[reg] = 0.f;
              }
#pragma unroll
              for (int reg = 0; reg < 16; ++reg) rs[reg] += v[reg] + add;
#pragma unroll
              for (int reg = 0; reg < 16; ++reg)
                if (nok && offs[reg] >= 0) a.out[(size_t)offs[reg] * a.out_cs + n] = rs[reg];
              if (a.stat_part) {
                // fused GroupNorm statistics of this 32-row block (see cm_conv.hip)
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
                const int tps = a.ntz * a.nty * a.ntx;
                const int sb = cur / tps, slot = (cur - sb * tps) * MB + mb;
                if (h == 0 && nok) {
                  float *sp = a.stat_part + (((size_t)sb * a.stat_ns + slot) * a.stat_C + n) * 2;
                  sp[0] = mean;
                  sp[1] = q;
                }
                if (lane == 0 && n == 0) a.stat_cnt[(size_t)sb * a.stat_ns + slot] = cnt;
              }
            }
          }
        }
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
      CM_STAMP(6)
      lds_barrier();  // scratch and row table free again
    }
    CM_STAMP(3)
    if (has_next) {
      write_lds();
      if (nch == 0) build_rows(ntile);
    }
    if (ch == 0) nn = ctrl[2];
    CM_STAMP(4)
    lds_barrier();  // (B) next image complete
    CM_STAMP(5)
    if (ch + 1 < nchunks) ++ch;
    else { cur = nxt; nxt = nn; ch = 0; }
  }

  if ((a.dbg & 8) && tid == 0 && a.dbg_buf) {
    for (int i = 0; i < 8; ++i) a.dbg_buf[(blockIdx.x + blockIdx.y * gridDim.x) * 8 + i] = (float)tph[i];
  }
  // ---- the last workgroup to finish re-arms the counter for the next launch --------
  if (tid == 0) {
    const int done = atomicAdd(&ctr[nt * 2 + 1], 1);
    if (done == (int)gridDim.x - 1) {
      ctr[nt * 2] = 0;
      ctr[nt * 2 + 1] = 0;
    }
  }
}

size_t conv2_lds_bytes(const ConvArgs &a, int MB, int NB) {
  const int HV = (a.bz + 2) * (a.by + 2) * (a.bx + 2);
  const int nblk = MB * NB;
  const int rb = nblk < 4 ? nblk : 4;
  const size_t tile = (size_t)HV * 36, red = (size_t)rb * 3 * 16 * 64;
  return ((size_t)4 + 32 * MB + (tile > red ? tile : red)) * 4;
}

// halo float4 per thread (256 threads, 8 channel quads -> 32 voxel lanes)
int conv2_nv(const ConvArgs &a) {
  const int HV = (a.bz + 2) * (a.by + 2) * (a.bx + 2);
  const int per = (HV + 31) / 32;
  return per <= 8 ? 8 : (per <= 12 ? 12 : (per <= 16 ? 16 : 0));
}

// only the instantiations that fit 256 VGPRs (2 waves/SIMD) without spilling
#define CM_CONV2_VARIANTS(X)                                           \
  X(2, 1, 8) X(2, 1, 12) X(3, 1, 8) X(3, 1, 12) X(4, 1, 8) X(4, 1, 12) \
  X(1, 2, 8) X(2, 2, 8) X(2, 2, 12)

bool conv2_variant_exists(int MB, int NB, int NV) {
#define X(m, n, v) if (MB == m && NB == n && NV == v) return true;
  CM_CONV2_VARIANTS(X)
#undef X
  return false;
}

hipError_t launch_conv2(const ConvArgs &a_in, int MB, int NB, int grid_x, int *ctr, hipStream_t st) {
  static const int dbg = getenv("CM_CONV_DBG") ? atoi(getenv("CM_CONV_DBG")) : 0;
  ConvArgs a = a_in;
  a.dbg = dbg;
  static float *dbgbuf = nullptr;
  if (dbg & 8) {
    if (!dbgbuf) (void)hipMalloc((void **)&dbgbuf, 4096 * 8 * sizeof(float));
    a.dbg_buf = dbgbuf;
  }
  if (a.ntaps != 27 || a.CK != 32 || a.stride != 1 || a.bs != 1) return hipErrorInvalidValue;
  const int NV = conv2_nv(a);
  const size_t lds = conv2_lds_bytes(a, MB, NB);
  if (!NV || lds > 160 * 1024) return hipErrorInvalidValue;
  const int ntn = (a.Co + 32 * NB - 1) / (32 * NB);
  dim3 grid((unsigned)grid_x, (unsigned)ntn);
#define X(m, n, v)                                                                                      \
  if (MB == m && NB == n && NV == v) {                                                                  \
    static bool attr_set[64] = {false};                                                                 \
    int dev = 0;                                                                                        \
    (void)hipGetDevice(&dev);                                                                           \
    if (!attr_set[dev & 63]) {                                                                          \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3_persist_kernel<m, n, v>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
      if (e != hipSuccess) return e;                                                                    \
      attr_set[dev & 63] = true;                                                                        \
    }                                                                                                   \
    hipLaunchKernelGGL((conv3_persist_kernel<m, n, v>), grid, dim3(256), lds, st, a, ctr);              \
    if (dbg & 8) {                                                                                      \
      (void)hipStreamSynchronize(st);                                                                   \
      static int shown = 0;                                                                             \
      if (shown < 40) {                                                                                 \
        ++shown;                                                                                        \
        float hb[64 * 8];                                                                               \
        (void)hipMemcpy(hb, dbgbuf, sizeof(hb), hipMemcpyDeviceToHost);                                 \
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                                       \
        for (int w = 0; w < 64; ++w) for (int i = 0; i < 8; ++i) sum[i] += hb[w * 8 + i] / 64.0;        \
        fprintf(stderr, "conv2<%d,%d,%d> Ci=%d Co=%d grid=%d: issue %.0f mfma %.0f barA %.0f epi-stores+bar %.0f wlds %.0f barB %.0f | partial-writes %.0f bar+reduce-reads %.0f cycles/WG\n", \
                m, n, v, a.C0 + a.C1, a.Co, grid_x, sum[0], sum[1], sum[2], sum[3], sum[4], sum[5], sum[6], sum[7]);   \
      }                                                                                                 \
    }                                                                                                   \
    return hipGetLastError();                                                                           \
  }
  CM_CONV2_VARIANTS(X)
#undef X
  return hipErrorInvalidValue;
}

}  // namespace cm
