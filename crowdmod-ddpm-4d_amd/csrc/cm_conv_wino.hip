// 3x3x3 stride-1 convolution with a Winograd F(2x2, 3x3) transform over the two in-plane axes (Y, X) and
// direct summation over the frame axis Z -- the full-resolution nn.Conv3d layers of the reference UNet
// (/root/reference/models/backbones/layers.py:32,43; the layers that are 46 % of a denoise step).
//
// Why: on gfx950 the exact-fp32 matrix instruction runs at the fp32 VECTOR rate and does not overlap other
// vector work on its SIMD (profiles/round1_notes.md), so a direct implicit GEMM cannot beat
// 27 * Ci * Co MACs per voxel plus its staging / epilogue instructions.  F(2x2, 3x3) needs 16 multiplies
// per 2x2 outputs and z tap instead of 36: 12 MACs per voxel and (ci, co) pair instead of 27 -- 2.25x fewer
// matrix instructions for the same fp32 result up to rounding (measured error vs the reference: see the
// parity tests; the transforms only add / subtract and scale weights by 1/2, 1/4).
//
// One 256-thread workgroup = one sample's BZ x BY x BX output box (BZ * (BY/2) * (BX/2) <= 32 patch rows):
//   * per 16-channel chunk every thread item (input plane zi, patch, channel quad) loads its 4x4 input
//     patch straight from global memory (GroupNorm affine + SiLU + Dropout3d multiplier applied on the fly,
//     zero padding as a mask), applies B^T d B in registers and writes the 16 frequency components to LDS as
//     U[xi][zi * NP + patch][ci]: the 32 rows of a component are CONSECUTIVE, so the A fragments are
//     conflict-free ds_read_b128 and the z tap is a plain row offset;
//   * wave w owns the frequency row xi_y = w (4 components xi_x): 4 accumulator blocks 32 rows x 32 channels,
//     K = 3 z taps x 16 channels per chunk; weights (G g G^T, packed on the host per wave in consumption
//     order) stream global -> VGPR through a two-group register ring;  no cross-wave K reduction;
//   * output transform: A^T along x in registers (4 blocks -> 2), exchange through LDS, A^T along y: wave w
//     ends with the output sub-block (a, b) = (w >> 1, w & 1) of every patch -- 32 voxels x 32 channels -- and
//     runs the same epilogue as the direct kernel (bias, time-embedding row, residual, channels-last store,
//     GroupNorm statistics of the block in the slot format of gn_finalize).
#include "cm_kernels.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace cm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_w(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// a - b on four floats as two v_pk_add_f32 with negated second operand: the compiler scalarises a vector fsub into
// four v_sub_f32 (there is no packed subtract), and three quarters of the Winograd transforms are subtractions
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub2(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x4 pk_sub(f32x4 a, f32x4 b) {
  const f32x2 lo = pk_sub2(a.xy, b.xy), hi = pk_sub2(a.zw, b.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}

// Epilogue of one 32-voxel x 32-channel output sub-block (same arithmetic as the direct kernel's): bias, time-embedding
// row, residual, channels-last store, GroupNorm statistics of the block in the slot format of gn_finalize.
// `bias`, `tv`: bias and time-embedding row value of this lane's output channel, loaded in the kernel's PROLOGUE -- the
// row sits behind two dependent loads (t index, table) that would otherwise be exposed at the very end of the workgroup.
__device__ __forceinline__ void wino_epilogue(const ConvArgs &a, const f32x16 &v, const int *outoff, int nt, int wave, int lane,
                                              int b0, int bs, int slot, float bias, float tv) {
  const int r = lane & 31, hh = lane >> 5;
  const int n = nt * 32 + r;
  const bool nok = n < a.Co;
  const int nc = nok ? n : 0;
  int offs[16];
  float rs[16];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) offs[reg] = outoff[wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) rs[reg] = v[reg] + bias;
  if (a.temb) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) rs[reg] += tv;
  }
  if (a.resid) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int oc = offs[reg] >= 0 ? offs[reg] : 0;
      rs[reg] += a.resid[(size_t)oc * a.res_cs + nc];
    }
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg)
    if (nok && offs[reg] >= 0) a.out[(size_t)offs[reg] * a.out_cs + n] = rs[reg];
  if (a.stat_part || a.astat) {
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
      if (offs[reg] >= 0) { const float dd = rs[reg] - mean; q += dd * dd; }
    q += __shfl_xor(q, 32);
    if (a.astat) {
      if (hh == 0 && nok && b0 < a.B && cnt > 0.f) cm_stat_atomic(a.astat + ((size_t)b0 * a.astat_C + n) * 3, s1, mean, q);
    } else {
      if (hh == 0 && nok && b0 < a.B) {
        float *sp2 = a.stat_part + (((size_t)b0 * a.stat_ns + slot) * a.stat_C + n) * 2;
        sp2[0] = mean;
        sp2[1] = q;
      }
      if (lane == 0 && n == 0 && b0 < a.B) a.stat_cnt[(size_t)b0 * a.stat_ns + slot] = cnt;
    }
  }
}

// Fused 1x1x1 skip convolution (ResnetBlock.match_input, layers.py:46,74; inference plan) on a finished sub-block:
// operand loads of up to SKB 32-channel chunks starting at chunk c0s (A: the RAW block input at this wave's 32 output
// voxels, B: the packed 1x1 weights), and the MFMAs over them.
constexpr int WINO_SKB = 3;                  // 32-channel chunks per batch (full-resolution skips have 2 or 3)
__device__ __forceinline__ void wino_skip_load(const ConvArgs &a, int nt, int n2a, int n2, int c0s, int svox, int lane,
                                               f32x4 (&sa)[WINO_SKB][4], f32x4 (&sw)[WINO_SKB][4]) {
  const int hh = lane >> 5;
  const f32x4 *w2 = reinterpret_cast<const f32x4 *>(a.s2w) + (size_t)nt * n2 * 4 * 64 + lane;
#pragma unroll
  for (int u = 0; u < WINO_SKB; ++u) {
    const int c2 = c0s + u < n2 ? c0s + u : n2 - 1;
    const float *s2 = c2 < n2a ? a.s2src0 + (size_t)svox * a.s2C0 + c2 * 32 : a.s2src1 + (size_t)svox * a.s2C1 + (c2 - n2a) * 32;
#pragma unroll
    for (int k8 = 0; k8 < 4; ++k8) {
      sa[u][k8] = *reinterpret_cast<const f32x4 *>(s2 + 8 * k8 + 4 * hh);
      sw[u][k8] = w2[(size_t)(c2 * 4 + k8) * 64];
    }
  }
}
__device__ __forceinline__ void wino_skip_mfma(int n2, int c0s, const f32x4 (&sa)[WINO_SKB][4], const f32x4 (&sw)[WINO_SKB][4], f32x16 &v) {
#pragma unroll
  for (int u = 0; u < WINO_SKB; ++u)
    if (c0s + u < n2) {
#pragma unroll
      for (int k8 = 0; k8 < 4; ++k8)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) v = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[u][k8][jj], sw[u][k8][jj], v, 0, 0, 0);
    }
}

// one 32-channel chunk at a time (persistent kernel: its epilogue holds the NEXT tile's prefetched halo loads as well, so
// the three-chunk batch above would not fit the register file)
__device__ __forceinline__ void wino_skip_load1(const ConvArgs &a, int nt, int n2a, int n2, int c2, int svox, int lane,
                                                f32x4 (&sa)[4], f32x4 (&sw)[4]) {
  const int hh = lane >> 5;
  const f32x4 *w2 = reinterpret_cast<const f32x4 *>(a.s2w) + (size_t)nt * n2 * 4 * 64 + lane;
  const float *s2 = c2 < n2a ? a.s2src0 + (size_t)svox * a.s2C0 + c2 * 32 : a.s2src1 + (size_t)svox * a.s2C1 + (c2 - n2a) * 32;
#pragma unroll
  for (int k8 = 0; k8 < 4; ++k8) {
    sa[k8] = *reinterpret_cast<const f32x4 *>(s2 + 8 * k8 + 4 * hh);
    sw[k8] = w2[(size_t)(c2 * 4 + k8) * 64];
  }
}
__device__ __forceinline__ void wino_skip_mfma1(const f32x4 (&sa)[4], const f32x4 (&sw)[4], f32x16 &v) {
#pragma unroll
  for (int k8 = 0; k8 < 4; ++k8)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) v = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[k8][jj], sw[k8][jj], v, 0, 0, 0);
}

// Tile = BZ planes x PY x PX patches (2x2 outputs each), ROWS = BZ * PY * PX <= 32 rows of the accumulator block
// (rows beyond ROWS are padding).  Where the patch grid is not a multiple of PY / PX the LAST tile of a row is
// shifted back inside the grid; it recomputes a few patches of its neighbour but owns (stores, counts in the
// statistics) only its own ones -- a.ntx / a.nty tiles of step PX / PY patches.
// F16 (reduced-precision plan, BASELINE configs[4]; the reference trains under fp16 autocast, ddpm.py:116-120): the
// transformed inputs are rounded to f16 when they are written to LDS (row = 16 channels = 32 B + 16 B pad), the
// transformed weights arrive as f16, and ONE v_mfma_f32_32x32x16_f16 per (z tap, component) contracts the whole
// 16-channel chunk with fp32 accumulation -- 1/16 of the matrix-pipe time of the 8 fp32 instructions it replaces.
// GroupNorm / SiLU, the transforms, the epilogue and the GroupNorm statistics stay fp32.
// TWO (the 8 x 2 x 2 full-resolution tile, where it fits next to U at two workgroups per CU): staging in two steps --
// (A) every halo voxel is normalised / activated ONCE into an LDS image R, its global loads issued before the
// previous chunk's matrix phase so that they complete under it; (B) the items transform their 4x4 patches out of R.
// The one-step form evaluates GroupNorm + SiLU per patch, i.e. 1.78x per voxel (patches overlap), and the kernel is
// vector-issue bound (10 vector instructions per MFMA, profiles/round2_pmc_summary.csv).  With TWO the fp32 U image is
// unpadded with an XOR swizzle of the 16-byte column by (row >> 2) instead of 4 pad dwords per row.
// NBW = 2 (layers with >= 64 output channels): a 512-thread workgroup computes TWO 32-channel output tiles from one
// staged U image -- waves 0-3 the first, waves 4-7 the second -- so the staging / transform work per output halves.
// In the one-step form the 8 waves then share the items as HALF items: lane pair (2k, 2k+1) takes patch rows
// {0,1} / {2,3} of item k, transforms along x locally, swaps ONE transformed row with its partner (DPP quad_perm) and
// finishes two of the four y components each -- 94 % of the lanes busy instead of 47 %.
template <int BZ, int PY, int PX, int OCC, bool F16, bool TWO, int NBW = 1>
__global__ __launch_bounds__(256 * NBW, OCC) void conv_wino_kernel(const ConvArgs a) {
  constexpr int NT = 256 * NBW;
  constexpr bool HALF = NBW == 2 && !TWO;       // half items in the one-step staging
  constexpr int NP = PY * PX, ROWS = BZ * NP;
  static_assert(ROWS <= 32 && ROWS > 16, "one (partly filled) 32-row accumulator block per frequency component");
  constexpr int HZ = BZ + 2, UR = HZ * NP;      // input planes, rows per component in LDS
  constexpr bool SWZ = TWO && !F16 && PY * PX == 4;   // unpadded fp32 rows, 16-byte column XOR-swizzled by (row >> 2) & 3
  constexpr int CS = 16, S = F16 ? 12 : (SWZ ? CS : CS + 4);  // channel chunk, LDS row stride in dwords (conflict-free b128 for consecutive rows)
  constexpr int RYH = 2 * PY + 2, RXH = 2 * PX + 2, RV = HZ * RYH * RXH, RS_ = BZ == 8 ? 24 : 20;   // activated halo image R[RV][24] (TWO)
  constexpr int RK = (RV * 4 + NT - 1) / NT;     // step-A rounds per thread
  constexpr int NITEMS = HZ * NP * (CS / 4);    // staging items: (plane, patch, channel quad)
  static_assert(NITEMS <= 256, "one staging item per thread (or lane pair)");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int *outoff = reinterpret_cast<int *>(lds);   // [4 sub-blocks (a, b)][32 rows] output voxel index or -1
  float *U = lds + 128;                         // [16][UR][S]; later the exchange buffer [4 waves][2][16][64]
  float *R = U + 16 * UR * S;                   // TWO: [RV][24] normalised + activated halo of the current chunk

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wv8 & 3, nbw = wv8 >> 2;     // frequency row xi_y of this wave, its output tile within the workgroup
  const int r = lane & 31, hh = lane >> 5;
  int tile = blockIdx.x;
  if (!(gridDim.x & 7) && !(a.dbg & 4096)) tile = (tile & 7) * (int)(gridDim.x >> 3) + (tile >> 3);  // XCD-aware order (cm_conv.hip)
  const int tx = tile % a.ntx; tile /= a.ntx;
  const int ty = tile % a.nty; tile /= a.nty;
  const int tz = tile % a.ntz;
  const int b0 = tile / a.ntz;
  const int nt = blockIdx.y * NBW + nbw;
  // epilogue operands of this lane's output channel, requested now (see wino_epilogue)
  const int nc_epi = nt * 32 + r < a.Co ? nt * 32 + r : 0;
  const float bias_pre = a.bias[nc_epi];
  // patch origin of the tile (shifted back inside the grid if it would stick out) and the first patch it OWNS
  const int pyt = a.Yo >> 1, pxt = a.Xo >> 1;
  const int py0 = min(ty * PY, pyt - PY), px0 = min(tx * PX, pxt - PX);
  const int z0 = tz * BZ, y0 = 2 * py0, x0 = 2 * px0;
  const int bs = b0 < a.B ? b0 : 0;             // (grid is exact; kept for safety)
  const float tv_pre = a.temb ? a.temb[(size_t)a.tidx[bs] * a.temb_stride + nc_epi] : 0.f;

  if (tid < 128) {
    const int ab = tid >> 5, row = tid & 31;
    const int zr = row / NP, pr = row % NP, py = pr / PX, px = pr % PX;
    const int oz = z0 + zr, oy = y0 + 2 * py + (ab >> 1), ox = x0 + 2 * px + (ab & 1);
    const bool own = row < ROWS && py0 + py >= ty * PY && px0 + px >= tx * PX;
    outoff[tid] = (own && b0 < a.B && oz < a.Zo && oy < a.Yo && ox < a.Xo) ? ((b0 * a.Zo + oz) * a.Yo + oy) * a.Xo + ox : -1;
  }
  // ---- this thread's staging item: source voxel offsets of its 4x4 patch, resolved once ---------------
  // TWO: halo planes outside the grid (z0 - 1 < 0, z0 + BZ >= Zs) are zero padding for EVERY chunk: their rows of R and U are
  // zeroed once below and neither staged nor transformed -- at full resolution (BZ = Zo) that is 2 of the 10 planes, on the
  // two-tile half-resolution launches 1 of 4: a fifth / quarter of the activation and transform instructions of a chunk
  const int pad_lo = (TWO && z0 == 0) ? 1 : 0, pad_hi = (TWO && z0 + BZ >= a.Zs) ? 1 : 0;
  const int nreal = HZ - pad_lo - pad_hi;          // (workgroup-uniform)
  const int itid = HALF ? tid >> 1 : tid, hf = HALF ? tid & 1 : 0;   // HALF: lane pair = item, hf = its patch rows {2hf, 2hf+1}
  const bool stager = itid < (TWO ? nreal * NP * (CS / 4) : NITEMS);
  const int it = stager ? itid : 0;
  const int quad = it & 3, patch = (it >> 2) % NP, zi = it / (4 * NP) + pad_lo;
  constexpr int NR = HALF ? 2 : 4;              // patch rows this thread loads
  int soff[NR * 4];
  unsigned okmask = 0;
  {
    const int py = patch / PX, px = patch % PX;
    const int cz = z0 - 1 + zi;
    const bool zok = stager && b0 < a.B && cz >= 0 && cz < a.Zs;
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cy = y0 + 2 * py - 1 + i + 2 * hf, cx = x0 + 2 * px - 1 + j;
        const bool ok = zok && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs;
        soff[i * 4 + j] = ok ? ((bs * a.Zs + cz) * a.Ys + cy) * a.Xs + cx : 0;
        okmask |= (ok ? 1u : 0u) << (i * 4 + j);
      }
  }
  const int urow = zi * NP + patch;
  float *const uw = U + (size_t)urow * S + (F16 ? 2 : 4) * (SWZ ? (quad ^ ((urow >> 2) & 3)) : quad);   // + xi * UR * S per component
  // TWO, step A: voxel (tid >> 2) + 64 k of the halo box, channel quad tid & 3
  int asoff[RK];
  unsigned aok = 0;
  const int RVR = (pad_lo + nreal) * (RYH * RXH);   // end of the real planes in the halo-voxel numbering
  const int rkr = TWO ? (nreal * (RYH * RXH) * 4 + NT - 1) / NT : 0;   // step-A rounds that hold real voxels (<= RK)
  if constexpr (TWO) {
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      const int v = (tid >> 2) + (NT / 4) * k + pad_lo * (RYH * RXH);     // voxel of the real planes only
      const int vz = v / (RYH * RXH), rem = v - vz * (RYH * RXH), vy = rem / RXH, vx = rem - vy * RXH;
      const int cz = z0 - 1 + vz, cy = y0 - 1 + vy, cx = x0 - 1 + vx;
      const bool ok = v < RVR && b0 < a.B && cz >= 0 && cz < a.Zs && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs;
      asoff[k] = ok ? ((bs * a.Zs + cz) * a.Ys + cy) * a.Xs + cx : 0;
      aok |= (ok ? 1u : 0u) << k;
    }
  }
  const int aq = tid & 3;
  // TWO, step B: R offset of the item's patch origin (plane zi, rows 2py.., cols 2px..)
  const int rbase = ((zi * RYH + 2 * (patch / PX)) * RXH + 2 * (patch % PX)) * RS_ + 4 * quad;

  f32x16 acc[4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[x][i] = 0.f;

  const int n0 = a.C0 >> 4, nchunks = n0 + (a.C1 >> 4);
  const int Ctot = a.C0 + a.C1;
  // packed weights: [n tile][chunk][wave = xi_y][group g][xi_x][lane] 16 bytes per lane:
  //   fp32: g = dz * 2 + k8 (6 groups), 4 floats  ci = 8*k8 + 4*hh + jj;   f16: g = dz (3 groups), 8 halves  ci = 8*hh + j
  constexpr int NG = F16 ? 3 : 6;
  const f32x4 *wbase = reinterpret_cast<const f32x4 *>(a.wfrag) + (size_t)nt * nchunks * (4 * NG * 4 * 64) + wave * (NG * 4 * 64) + lane;
  // register ring of weight groups: slot of group g is g % RS, the next group (possibly the next chunk's first) is
  // prefetched into slot (g + 1) % RS -- RS must divide NG so that a chunk always starts on slot 0
  constexpr int RS = F16 ? 3 : 2;
  static_assert(NG % RS == 0, "ring slots must line up at chunk boundaries");
  f32x4 bq[RS][4];
#pragma unroll
  for (int x = 0; x < 4; ++x) bq[0][x] = wbase[x * 64];

  const int ar = min(r, ROWS - 1);
  const float *arow = U + (size_t)(wave * 4) * UR * S + (size_t)ar * S + (SWZ ? 0 : 4 * hh);   // component xi_y = wave, xi_x = 0, tap 0
  f32x4 ald[RK];                                 // TWO: step-A loads of the NEXT chunk, in flight during the matrix phase
  if constexpr (TWO) {
    const float *sp0 = a.src0 + 4 * aq;          // chunk 0 always comes from src0
#pragma unroll
    for (int k = 0; k < RK; ++k)
      if (k < rkr) ald[k] = *reinterpret_cast<const f32x4 *>(sp0 + (size_t)asoff[k] * a.C0);
    // zero the padding planes of R and U once (no chunk ever writes them)
    for (int pz = 0; pz < 2; ++pz) {
      if (!(pz ? pad_hi : pad_lo)) continue;
      const int plane = pz ? HZ - 1 : 0;
      for (int i = tid; i < RYH * RXH * (RS_ / 4); i += NT)
        *reinterpret_cast<f32x4 *>(R + (size_t)plane * (RYH * RXH) * RS_ + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int i = tid; i < 16 * NP * (S / 4); i += NT) {
        const int xi = i / (NP * (S / 4)), rem = i - xi * (NP * (S / 4));
        *reinterpret_cast<f32x4 *>(U + ((size_t)xi * UR + plane * NP) * S + 4 * rem) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }

  f32x4 dn[TWO ? 1 : NR * 4];                    // one-step form: raw patch loads of the NEXT chunk, in flight during the matrix phase
  constexpr bool PRE = HALF;                     // (the full-item one-step form would spill with 16 more loads held)
  if constexpr (PRE) {
    const float *sp0 = a.src0 + 4 * quad;
#pragma unroll
    for (int k = 0; k < NR * 4; ++k) dn[k] = *reinterpret_cast<const f32x4 *>(sp0 + (size_t)soff[k] * a.C0);
  }

  for (int ch = 0; ch < nchunks; ++ch) {
    int cg0;
    if (ch < n0) cg0 = ch * CS; else cg0 = a.C0 + (ch - n0) * CS;
    // ---- stage + input transform ---------------------------------------------------------------------
    f32x4 d[16];
    f32x4 sc1 = {1.f, 1.f, 1.f, 1.f}, sh1 = {0.f, 0.f, 0.f, 0.f}, pm1 = {1.f, 1.f, 1.f, 1.f};
    const int gq = TWO ? aq : quad;               // channel quad this thread normalises
    if (a.gn) {
      const float *g = a.gn + (size_t)bs * 2 * Ctot + cg0 + 4 * gq;
      sc1 = *reinterpret_cast<const f32x4 *>(g);
      sh1 = *reinterpret_cast<const f32x4 *>(g + Ctot);
    }
    if (a.pm) pm1 = *reinterpret_cast<const f32x4 *>(a.pm + (size_t)bs * a.pm_stride + cg0 + 4 * gq);
    auto activate = [&](f32x4 w, bool ok) -> f32x4 {
      if (a.gn) {
        w = w * sc1 + sh1;
        if (a.silu) { w[0] = silu_w(w[0]); w[1] = silu_w(w[1]); w[2] = silu_w(w[2]); w[3] = silu_w(w[3]); }
      }
      if (a.pm) w = w * pm1;
      return ok ? w : f32x4{0.f, 0.f, 0.f, 0.f};   // zero padding of the ACTIVATED tensor
    };
    if constexpr (TWO) {
      // step A: this chunk's halo voxels (loaded during the previous matrix phase) -> activated image R
      if (a.dbg & 128) {                        // diagnostic: plain copy instead of GroupNorm + SiLU
#pragma unroll
        for (int k = 0; k < RK; ++k) {
          const int v = (tid >> 2) + (NT / 4) * k + pad_lo * (RYH * RXH);
          if (k < rkr && v < RVR) *reinterpret_cast<f32x4 *>(R + v * RS_ + 4 * aq) = ald[k];
        }
      } else if (a.gn && a.silu && !a.pm) {     // (workgroup-uniform: the plain ResnetBlock conv; no selects on the flags)
#pragma unroll
        for (int k = 0; k < RK; ++k)
          if (k < rkr) {
            const int v = (tid >> 2) + (NT / 4) * k + pad_lo * (RYH * RXH);
            f32x4 w = ald[k] * sc1 + sh1;
            w[0] = silu_w(w[0]); w[1] = silu_w(w[1]); w[2] = silu_w(w[2]); w[3] = silu_w(w[3]);
            if (!((aok >> k) & 1u)) w = f32x4{0.f, 0.f, 0.f, 0.f};
            if (v < RVR) *reinterpret_cast<f32x4 *>(R + v * RS_ + 4 * aq) = w;
          }
      } else {
#pragma unroll
        for (int k = 0; k < RK; ++k) {
          const int v = (tid >> 2) + (NT / 4) * k + pad_lo * (RYH * RXH);
          if (k < rkr && v < RVR) *reinterpret_cast<f32x4 *>(R + v * RS_ + 4 * aq) = activate(ald[k], (aok >> k) & 1u);
        }
      }
      __syncthreads();                        // R complete; every wave is past the previous chunk's matrix phase
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i * 4 + j] = *reinterpret_cast<const f32x4 *>(R + rbase + (i * RXH + j) * RS_);
    } else {
      if constexpr (!PRE) {
        const float *sp = (ch < n0 ? a.src0 + ch * CS : a.src1 + (ch - n0) * CS) + 4 * quad;
        const int Cs = ch < n0 ? a.C0 : a.C1;
#pragma unroll
        for (int k = 0; k < NR * 4; ++k) dn[k] = *reinterpret_cast<const f32x4 *>(sp + (size_t)soff[k] * Cs);
      }
#pragma unroll
      for (int k = 0; k < NR * 4; ++k) d[k] = activate(dn[k], (okmask >> k) & 1u);
    }
    // B^T d B: rows of B^T = (1,0,-1,0), (0,1,1,0), (0,-1,1,0), (0,1,0,-1); first along x (index j), then y (i)
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const f32x4 e0 = d[i * 4 + 0], e1 = d[i * 4 + 1], e2 = d[i * 4 + 2], e3 = d[i * 4 + 3];
      d[i * 4 + 0] = pk_sub(e0, e2); d[i * 4 + 1] = e1 + e2; d[i * 4 + 2] = pk_sub(e2, e1); d[i * 4 + 3] = pk_sub(e1, e3);
    }
    if constexpr (HALF) {
      // lane hf = 0 holds the x-transformed patch rows 0, 1; hf = 1 rows 2, 3.  y components: V0 = t0 - t2, V1 = t1 + t2
      // (lane 0), V2 = t2 - t1, V3 = t1 - t3 (lane 1): each lane needs its partner's INNER row (t2 resp. t1).
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 inner = hf ? d[j] : d[4 + j], outer = hf ? d[4 + j] : d[j];
        f32x4 c;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          c[e] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(inner[e]), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]; (__builtin_bit_cast of a vector ELEMENT reads element 0)
        d[j] = hf ? inner - c : outer - c;          // V0 = t0 - t2   |  V2 = t2 - t1
        d[4 + j] = hf ? c - outer : inner + c;      // V1 = t1 + t2   |  V3 = t1 - t3
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 e0 = d[0 * 4 + j], e1 = d[1 * 4 + j], e2 = d[2 * 4 + j], e3 = d[3 * 4 + j];
        d[0 * 4 + j] = pk_sub(e0, e2); d[1 * 4 + j] = e1 + e2; d[2 * 4 + j] = pk_sub(e2, e1); d[3 * 4 + j] = pk_sub(e1, e3);
      }
    }
    if constexpr (!TWO) __syncthreads();    // previous chunk's fragments have been read
    if (stager && !(a.dbg & 1)) {
#pragma unroll
      for (int k = 0; k < NR * 4; ++k) {                               // component xi = (2 hf + i) * 4 + j, k = i * 4 + j
        const int xi = HALF ? 8 * hf + k : k;
        if constexpr (F16) {
          const f16x4 hv = {(_Float16)d[k][0], (_Float16)d[k][1], (_Float16)d[k][2], (_Float16)d[k][3]};
          *reinterpret_cast<f16x4 *>(uw + (size_t)xi * UR * S) = hv;
        } else {
          *reinterpret_cast<f32x4 *>(uw + (size_t)xi * UR * S) = d[k];
        }
      }
    }
    __syncthreads();
    if (ch + 1 < nchunks) {                   // next chunk's loads: in flight under this chunk's matrix phase
      const int cn = ch + 1;
      const float *spn = (cn < n0 ? a.src0 + cn * CS : a.src1 + (cn - n0) * CS) + 4 * (TWO ? aq : quad);
      const int Cn = cn < n0 ? a.C0 : a.C1;
      if constexpr (TWO) {
#pragma unroll
        for (int k = 0; k < RK; ++k)
          if (k < rkr) ald[k] = *reinterpret_cast<const f32x4 *>(spn + (size_t)asoff[k] * Cn);
      } else if constexpr (PRE) {
#pragma unroll
        for (int k = 0; k < NR * 4; ++k) dn[k] = *reinterpret_cast<const f32x4 *>(spn + (size_t)soff[k] * Cn);
      }
    }
    // ---- matrix phase: NG groups (z tap [, 8-channel half]) x 4 components ---------------------------------
    if (a.dbg & 2) continue;
    auto aread = [&](int g, f32x4 (&af)[4]) {
      const int dz = F16 ? g : g >> 1, k8 = F16 ? 0 : g & 1;
      const int acol = SWZ ? 4 * ((2 * k8 + hh) ^ (((ar >> 2) + dz) & 3)) : 8 * k8;   // swizzled 16-byte column of row ar + dz * NP
#pragma unroll
      for (int x = 0; x < 4; ++x) af[x] = *reinterpret_cast<const f32x4 *>(arow + (size_t)x * UR * S + (size_t)dz * NP * S + acol);
    };
    f32x4 afr[2][4];                         // A fragments of the current / next group (static parity: no register copies)
    aread(0, afr[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // refill the other ring slot with the next group (possibly the next chunk's first)
      {
        const bool more = g < NG - 1 || ch + 1 < nchunks;
        const f32x4 *wn = wbase + (size_t)(g < NG - 1 ? ch : ch + 1) * (4 * NG * 4 * 64) + (size_t)(g < NG - 1 ? g + 1 : 0) * (4 * 64);
        if (more) {
#pragma unroll
          for (int x = 0; x < 4; ++x) bq[(g + 1) % RS][x] = wn[x * 64];
        }
      }
      if (g + 1 < NG) aread(g + 1, afr[(g + 1) & 1]);
      asm volatile("" ::: "memory");          // next group's weights and A fragments are REQUESTED here, not at their first use
      if constexpr (F16) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[g & 1][x]), __builtin_bit_cast(f16x8, bq[g % RS][x]), acc[x], 0, 0, 0);
      } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int x = 0; x < 4; ++x)
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[g & 1][x][jj], bq[g % RS][x][jj], acc[x], 0, 0, 0);
      }
    }
  }

  // ---- fused 1x1x1 skip convolution (ResnetBlock.match_input, layers.py:46,74; inference plan): issue the loads
  // of its operands now -- the RAW block input at this wave's 32 output voxels and the packed 1x1 weights -- so
  // that their latency hides behind the output transform; the MFMAs run on the finished sub-block below.
  if (a.dbg & 4) {
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.456f) a.out[0] = 1.f;   // keep the accumulators live
    return;
  }
  const int n2a = a.s2C0 >> 5, n2 = a.s2w ? (a.s2C0 + a.s2C1) >> 5 : 0;
  f32x4 sa[WINO_SKB][4], sw[WINO_SKB][4];
  int svox = 0;
  if (n2 > 0) {                              // wave-uniform
    const int o = outoff[wave * 32 + r];     // A operand: lane = (row r, k half hh)
    svox = o >= 0 ? o : 0;
    wino_skip_load(a, nt, n2a, n2, 0, svox, lane, sa, sw);
  }
  // ---- output transform: A^T = (1,1,1,0), (0,1,-1,-1) along x in registers, along y through LDS -----------
  f32x16 t0 = acc[0] + acc[1] + acc[2];
  f32x16 t1 = acc[1] - acc[2] - acc[3];
  __syncthreads();                          // U is dead: reuse as the exchange buffer
  float *const XC = U + (size_t)nbw * (4 * 2 * 16 * 64);   // this output tile's exchange buffer
  float *xb = XC + (size_t)(wave * 2) * 16 * 64 + lane;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) { xb[reg * 64] = t0[reg]; xb[(16 + reg) * 64] = t1[reg]; }
  __syncthreads();
  const int oa = wave >> 1, ob = wave & 1;   // this wave's output sub-block
  f32x16 v;
  {
    const float *p0 = XC + (size_t)((0 * 2 + ob) * 16) * 64 + lane, *p1 = XC + (size_t)((1 * 2 + ob) * 16) * 64 + lane;
    const float *p2 = XC + (size_t)((2 * 2 + ob) * 16) * 64 + lane, *p3 = XC + (size_t)((3 * 2 + ob) * 16) * 64 + lane;
    if (oa == 0) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) v[reg] = (p0[reg * 64] + p1[reg * 64]) + p2[reg * 64];
    } else {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) v[reg] = (p1[reg * 64] - p2[reg * 64]) - p3[reg * 64];
    }
  }
  if (n2 > 0) {
    for (int c0s = 0; c0s < n2; c0s += WINO_SKB) {
      if (c0s > 0) wino_skip_load(a, nt, n2a, n2, c0s, svox, lane, sa, sw);   // further batches: latency exposed
      wino_skip_mfma(n2, c0s, sa, sw, v);
    }
  }
  wino_epilogue(a, v, outoff, nt, wave, lane, b0, bs, (((tz * a.nty + ty) * a.ntx) + tx) * 4 + wave, bias_pre, tv_pre);
}

// ================================================================================================================
// Persistent two-step form (round 3).  Ablation of the kernel above on the full-resolution 32 -> 32 layer (78.7 us):
// matrix phase 44.5 us, everything else 34.2 us -- and the two ADD (no overlap): all workgroups have the same duration, so
// the whole chip marches through load burst -> staging -> matrix phase -> store burst in lock-step rounds, the
// latency-bound parts (first-chunk loads, GroupNorm rows, index tables, output stores) never hide under another
// workgroup's matrix phase, and every workgroup repeats ~1000 vector / ~670 scalar instructions of prologue and epilogue
// for 192 matrix instructions.  Here a workgroup is (tile position p, sample lane g) and loops over the samples
// b = g, g + G, ...:
//   * the tile geometry (source offsets of the halo box, output offsets, ownership) comes from two host-built tables and
//     is resolved ONCE per workgroup; per sample only scalar base pointers move (32-bit in-sample byte offsets);
//   * the halo loads AND the GroupNorm scale / shift rows of the next chunk -- which may be the first chunk of the NEXT
//     sample -- are issued before the current matrix phase: no cold start after the first tile, output stores of tile i
//     drain under the staging of tile i + 1;
//   * output exchange through LDS as 16-byte accesses (8 + 12 instead of 32 + 48 LDS instructions), store addresses as
//     one 24-bit multiply-add each, a mask-free store / statistics path when the wave owns all of its 32 rows.
// Same arithmetic in the same order as conv_wino_kernel<..., TWO = true>: results are bit-identical to it.
// SKIP: the block's 1x1x1 skip convolution is contracted onto the finished sub-block (its own instantiation, so that the
// plain layers do not carry its registers: with it in one body the kernel spilled ~100 registers).
// B6: fp32 products from exact three-way bf16 splits (cm_conv_ups.hip explains the arithmetic): the transformed input is split
// into hi / mid / lo planes at the U write, the transformed weights arrive pre-split (pack_wino_b6), and 16 channels of a
// (z tap, component) are six v_mfma_f32_32x32x16_bf16 (192 cycles) instead of eight fp32 instructions (512 cycles).
#ifndef CM_WINO_ABL
#define CM_WINO_ABL 0        // compile-time ablations of the six-term chunk loop (experiments only; results are wrong)
#endif
template <int BZ, int PY, int PX, bool F16, int NBW, bool SKIP, int B6 = 0>   // B6: 0 off, 1 six bf16 cross terms, 2 three (relaxed plan), 3 three f16 cross terms (h2)
__global__ __launch_bounds__(256 * NBW, 2) void conv_wino_p_kernel(const ConvArgs a, const int *__restrict__ tabA,
                                                                  const int *__restrict__ tabO, int G) {
  constexpr int NT = 256 * NBW;
  constexpr int NP = PY * PX, ROWS = BZ * NP;
  static_assert(ROWS <= 32 && ROWS > 16, "one (partly filled) 32-row accumulator block per frequency component");
  constexpr int HZ = BZ + 2, UR = HZ * NP;
  static_assert(!(F16 && B6), "one operand format");
  constexpr bool SWZ = !F16 && !B6 && PY * PX == 4;
  // CMP (B6 on the full-resolution tile, BZ = Zs = 8): the two zero-padding planes of the halo get NO rows -- U keeps one zero
  // block (rows [0, NP): it is halo plane 0, and halo plane 9 reads it too), R keeps the eight real planes only, rows are
  // unpadded -- so that the three bf16 planes of U (55 KB) and R (23 KB) still fit two workgroups per CU
  constexpr bool CMP = B6 && BZ == 8;
  constexpr int URC = CMP ? (HZ - 1) * NP : UR;          // U rows per component
  constexpr int CS = 16, S = CMP ? 24 : (B6 ? 28 : (F16 ? 12 : (SWZ ? CS : CS + 4)));   // B6: three bf16 planes of 8 dwords (+ 4 pad)
  constexpr int RYH = 2 * PY + 2, RXH = 2 * PX + 2, RV = HZ * RYH * RXH, RS_ = CMP ? 20 : (BZ == 8 ? 24 : 20);
  constexpr int PLV = RYH * RXH, RVC = CMP ? BZ * PLV : RV, RV0 = CMP ? PLV : 0;   // R voxels kept, first kept halo voxel
  constexpr int RK = ((CMP ? BZ * RYH * RXH : RV) * 4 + NT - 1) / NT;   // step-A rounds per thread (CMP: the real planes only)
  constexpr int NITEMS = HZ * NP * (CS / 4);
  static_assert(NITEMS <= 256, "one staging item per thread");
  constexpr int USZ = 16 * URC * S, XSZ = NBW * 4 * 2 * 16 * 64;
  constexpr int UX = USZ > XSZ ? USZ : XSZ;          // U, later the exchange buffer: R must NOT overlap it (next tile's step A)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int *outoff = reinterpret_cast<int *>(lds);       // [4 sub-blocks (a, b)][32 rows] in-sample output voxel index or -1
  float *U = lds + 128;
  float *R = U + UX;
  float *GNL = R + RVC * RS_;                       // accumulator statistics (a.gs0): scale / shift rows [2][C0 + C1] of the current sample

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wv8 & 3, nbw = wv8 >> 2;
  const int r = lane & 31, hh = lane >> 5;
  const int g0 = blockIdx.x, p = blockIdx.y;
  const int nt = blockIdx.z * NBW + nbw;
  if (g0 >= a.B) return;
  const int nc_epi = nt * 32 + r < a.Co ? nt * 32 + r : 0;
  const float bias_pre = a.bias[nc_epi];

  // ---- tile geometry from the tables (cm_conv_wino.hip: wino_tables) --------------------------------------------
  if (tid < 128) outoff[tid] = tabO[p * 128 + tid];
  int asoff[RK];
  unsigned aok = 0;
#pragma unroll
  for (int k = 0; k < RK; ++k) {
    // (the table is laid out for the full halo box, RKT rounds of NT / 4 voxels; CMP starts at the first real plane)
    constexpr int RKT = (RV * 4 + NT - 1) / NT;
    const int v = (tid >> 2) + (NT / 4) * k + RV0;
    asoff[k] = v < RKT * (NT / 4) ? tabA[p * RKT * (NT / 4) + v] : -1;
  }
#pragma unroll
  for (int k = 0; k < RK; ++k) {
    aok |= (asoff[k] >= 0 ? 1u : 0u) << k;
    asoff[k] = asoff[k] >= 0 ? asoff[k] : 0;
  }
  const int aq = tid & 3;
  // transform items (plane, patch, channel quad).  When the workgroup has twice as many threads as items (two-tile form),
  // both halves of the workgroup take every item: waves 0-3 compute the frequency columns xi_x in {0, 1}, waves 4-7 {2, 3}
  // (eight of the sixteen components each; the half is wave-uniform, so nothing is computed twice) -- the transform runs on
  // all eight waves instead of four, with half the arithmetic per thread
  // (CMP: the 128 items of the real planes, waves 0-1 / 2-3)
  constexpr bool HX2 = (NT == 512 && NITEMS <= 256) || CMP;
  constexpr int HXM = CMP ? 127 : 255;
  const int itid = HX2 ? (tid & HXM) + (CMP ? 4 * NP : 0) : tid, hf = HX2 ? (CMP ? wv8 >> 1 : wv8 >> 2) : 0;
  const int it = itid < NITEMS ? itid : 0;
  const int quad = it & 3, patch = (it >> 2) % NP, zi = it / (4 * NP);
  const bool stager = itid < NITEMS && !(CMP && (zi == 0 || zi == HZ - 1));   // CMP: padding planes are never transformed
  const int urow = zi * NP + patch;
  // CMP (compact six-term image): a component is [k half hh][row][term][4 dwords] -- row stride 12 dwords, the three terms of a
  // lane's fragment at immediate offsets 0 / 16 / 32 bytes: the 16 rows of a ds_read_b128 service group sit on 16 different
  // 16-byte bank groups (12 r mod 64 = 4 (3 r mod 16)); the term-major row of 24 dwords it replaces put rows r and r + 8 on the
  // same banks (2-way conflicts on every A fragment: the 0.45 conflict share of round 3)
  float *const uw = CMP ? U + (size_t)((quad >> 1) * URC + urow) * 12 + 2 * (quad & 1)
                        : U + (size_t)urow * S + ((F16 || B6) ? 2 : 4) * (SWZ ? (quad ^ ((urow >> 2) & 3)) : quad);
  constexpr int TST = CMP ? 4 : 8;                 // dwords between the terms of a row
  const int rbase = (((zi - (CMP ? 1 : 0)) * RYH + 2 * (patch / PX)) * RXH + 2 * (patch % PX)) * RS_ + 4 * quad;

  const int n0 = a.C0 >> 4, nchunks = n0 + (a.C1 >> 4);
  const int Ctot = a.C0 + a.C1;
  const unsigned Vs = (unsigned)(a.Zs * a.Ys * a.Xs), Vo = (unsigned)(a.Zo * a.Yo * a.Xo);
  constexpr int NG = (F16 || B6) ? 3 : 6;
  constexpr int NTM = B6 ? 3 : 1;                   // operand terms per fragment
  const f32x4 *wbase = reinterpret_cast<const f32x4 *>(a.wfrag) + (size_t)nt * nchunks * (4 * NG * 4 * NTM * 64) + wave * (NG * 4 * NTM * 64) + lane;
  constexpr int RS = F16 ? 3 : 2;
  static_assert(B6 || NG % RS == 0, "ring slots must line up at chunk boundaries");
  f32x4 bq[B6 ? 1 : RS][4];
  // B6: ring over the 12 (z tap, component) steps of a chunk, three term fragments per step, RD6 steps ahead
  constexpr int RD6 = 3;
  f32x4 b6[B6 ? RD6 : 1][3];
  if constexpr (B6) {
#pragma unroll
    for (int sx = 0; sx < RD6; ++sx)
#pragma unroll
      for (int tm = 0; tm < 3; ++tm) b6[sx][tm] = wbase[(sx * 3 + tm) * 64];
  } else {
#pragma unroll
    for (int x = 0; x < 4; ++x) bq[0][x] = wbase[x * 64];
  }

  const int ar = min(r, ROWS - 1);
  const float *arow = CMP ? U + (size_t)(wave * 4) * URC * S + (size_t)(hh * URC + ar) * 12
                          : U + (size_t)(wave * 4) * URC * S + (size_t)ar * S + (SWZ ? 0 : 4 * hh);
  // CMP: row block of halo plane z + dz; plane HZ - 1 (top padding) is the zero block 0
  constexpr int RST = CMP ? 12 : S;                // dwords between consecutive rows of a lane's k half
  int rowoff[3] = {0, NP * RST, 2 * NP * RST};
  if constexpr (CMP) {
    if (ar / NP == BZ - 1) rowoff[2] = -(ar / NP) * NP * RST;   // z = 7, dz = 2: block 0 instead of block 9
  }
  // this lane's output rows of the epilogue: reg -> row (reg & 3) + 8 (reg >> 2) + 4 hh of sub-block `wave`
  // (read once per workgroup after the table has landed, below)

  // ---- loads of (sample b, chunk ch): halo voxels of the thread's step-A slots + the GroupNorm rows of its channel quad
  f32x4 ald[RK], scn = {1.f, 1.f, 1.f, 1.f}, shn = {0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int b, int ch) {
    const bool s0 = ch < n0;
    const int Cn = s0 ? a.C0 : a.C1;
    const float *base = (s0 ? a.src0 + ch * CS : a.src1 + (ch - n0) * CS) + (size_t)b * Vs * Cn;      // wave-uniform
    const unsigned cb = (unsigned)Cn * 4u, aq16 = 16u * (unsigned)aq;
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      unsigned vo = (unsigned)asoff[k];
      asm volatile("" : "+v"(vo));               // keep ONE copy of the offsets live (no hoisted per-source products)
      ald[k] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(base) + (__umul24(vo, cb) + aq16));
    }
    if (a.gn) {
      const float *gp = a.gn + (size_t)b * 2 * Ctot + (s0 ? ch * CS : a.C0 + (ch - n0) * CS);              // wave-uniform
      scn = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(gp) + aq16);
      shn = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(gp + Ctot) + aq16);
    }
  };
  issue(g0, 0);
  __syncthreads();                               // outoff visible
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  // this lane's output rows of the epilogue: reg -> row (reg & 3) + 8 (reg >> 2) + 4 hh of sub-block `wave` (re-read per tile:
  // four 16-byte broadcast reads are cheaper than 16 registers held across the matrix phase)
  const i32x4 *const op4 = reinterpret_cast<const i32x4 *>(outoff + wave * 32 + 4 * hh);
  bool allv = true;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const i32x4 v4 = op4[2 * q];
    allv = allv && v4[0] >= 0 && v4[1] >= 0 && v4[2] >= 0 && v4[3] >= 0;
  }
  const bool wave_all = __all(allv) && nt * 32 + 31 < a.Co;       // the wave owns all 32 rows x 32 channels: mask-free path
  const int n = nt * 32 + r;
  const bool nok = n < a.Co;
  const int n2a = SKIP ? a.s2C0 >> 5 : 0, n2 = SKIP ? (a.s2C0 + a.s2C1) >> 5 : 0;
  const int slot = p * 4 + wave;

  const bool own_gn = a.gs0 != nullptr || a.gp0 != nullptr;  // the GroupNorm of the input is finalised in this workgroup
  const bool norm = a.gn != nullptr || own_gn;               // GroupNorm (+ SiLU) on load
  for (int b = g0; b < a.B; b += G) {
    const bool more_b = b + G < a.B;
    const float tv_pre = a.temb ? a.temb[(size_t)a.tidx[b] * a.temb_stride + nc_epi] : 0.f;
    if (own_gn) {
      // the GroupNorm of this sample's input finalised HERE (no gn_finalize launch) from the producers' slot partials (few slots:
      // half / quarter resolution) or accumulator rows: the halo loads of the first chunk are already in flight, the rows are
      // read from LDS in step A
      __syncthreads();                            // (the previous sample's last step A has read its rows, its epilogue the exchange buffer)
      if (a.gp0) cm_gn_rows_from_slots(a, b, (int)Vs, GNL, U, tid, NT);                     // (U is free here: scratch)
      else cm_gn_rows_from_sums(a, b, (int)Vs, GNL, reinterpret_cast<double *>(U), tid, NT);
      __syncthreads();
    }
    f32x16 acc[4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[x][i] = 0.f;

    for (int ch = 0; ch < nchunks; ++ch) {
      // ---- step A: this chunk's halo voxels -> activated image R ------------------------------------------
      f32x4 sc1 = scn, sh1 = shn;
      if (own_gn) {
        const int cb0 = (ch < n0 ? ch * CS : a.C0 + (ch - n0) * CS) + 4 * aq;
        sc1 = *reinterpret_cast<const f32x4 *>(GNL + cb0);
        sh1 = *reinterpret_cast<const f32x4 *>(GNL + Ctot + cb0);
      }
      f32x4 pm1 = {1.f, 1.f, 1.f, 1.f};
      if (!((CM_WINO_ABL & 32) && ch > 0)) {
      if (a.pm) pm1 = *reinterpret_cast<const f32x4 *>(a.pm + (size_t)b * a.pm_stride + (ch < n0 ? ch * CS : a.C0 + (ch - n0) * CS) + 4 * aq);
      if (norm && a.silu && !a.pm) {
#pragma unroll
        for (int k = 0; k < RK; ++k) {
          const int v = (tid >> 2) + (NT / 4) * k + RV0;
          f32x4 w = ald[k] * sc1 + sh1;
          w[0] = silu_w(w[0]); w[1] = silu_w(w[1]); w[2] = silu_w(w[2]); w[3] = silu_w(w[3]);
          if (!((aok >> k) & 1u)) w = f32x4{0.f, 0.f, 0.f, 0.f};
          if (v >= RV0 && v < RV0 + RVC) *reinterpret_cast<f32x4 *>(R + (v - RV0) * RS_ + 4 * aq) = w;
        }
      } else {
#pragma unroll
        for (int k = 0; k < RK; ++k) {
          const int v = (tid >> 2) + (NT / 4) * k + RV0;
          f32x4 w = ald[k];
          if (norm) {
            w = w * sc1 + sh1;
            if (a.silu) { w[0] = silu_w(w[0]); w[1] = silu_w(w[1]); w[2] = silu_w(w[2]); w[3] = silu_w(w[3]); }
          }
          if (a.pm) w = w * pm1;
          if (!((aok >> k) & 1u)) w = f32x4{0.f, 0.f, 0.f, 0.f};
          if (v >= RV0 && v < RV0 + RVC) *reinterpret_cast<f32x4 *>(R + (v - RV0) * RS_ + 4 * aq) = w;
        }
      }
      }
      __syncthreads();                          // R complete; every wave is past the previous matrix phase / exchange reads
      if constexpr (CMP) {
        if (ch == 0)                              // the zero block (the previous tile's exchange overwrote it): rows [0, NP) of both k halves
          for (int i = tid; i < 16 * 2 * NP * 3; i += NT) {
            const int xi = i / (2 * NP * 3), rem = i - xi * (2 * NP * 3), hf2 = rem / (NP * 3), r4 = rem - hf2 * (NP * 3);
            *reinterpret_cast<f32x4 *>(U + (size_t)xi * URC * S + (size_t)hf2 * URC * 12 + 4 * r4) = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
      // ---- step B: B^T d B of the item's 4x4 patch out of R ------------------------------------------------
      if (stager && !((CM_WINO_ABL & 4) && ch > 0)) {
        f32x4 d[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) d[i * 4 + j] = *reinterpret_cast<const f32x4 *>(R + rbase + (i * RXH + j) * RS_);
        auto put = [&](int k, const f32x4 v) {     // component k = xi_y * 4 + xi_x of this item
          if constexpr (B6) {
            constexpr int NTW = B6 >= 2 ? 2 : 3;    // (relaxed plan / h2: the lo plane is neither formed nor written nor read)
            cm_u32x2_t t3[3];
            if constexpr (B6 == 3) cm_split2_f16(v, t3);   // h2: f16 hi / mid of the transformed GroupNorm + SiLU output (bounded)
            else cm_split3_bf16<NTW>(v, t3);        // hi / mid / lo planes, exact remainders
#pragma unroll
            for (int tm = 0; tm < NTW; ++tm) *reinterpret_cast<cm_u32x2_t *>(uw + (size_t)k * URC * S + TST * tm) = t3[tm];
          } else if constexpr (F16) {
            const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *reinterpret_cast<f16x4 *>(uw + (size_t)k * URC * S) = hv;
          } else {
            *reinterpret_cast<f32x4 *>(uw + (size_t)k * URC * S) = v;
          }
        };
        if constexpr (HX2) {
          // x pass: only this lane's two frequency columns; y pass on them; same operations per component as the full form
          f32x4 c[4][2];
          if (hf == 0) {                           // (wave-uniform)
#pragma unroll
            for (int i = 0; i < 4; ++i) { c[i][0] = pk_sub(d[i * 4 + 0], d[i * 4 + 2]); c[i][1] = d[i * 4 + 1] + d[i * 4 + 2]; }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { c[i][0] = pk_sub(d[i * 4 + 2], d[i * 4 + 1]); c[i][1] = pk_sub(d[i * 4 + 1], d[i * 4 + 3]); }
          }
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const f32x4 e0 = c[0][jj], e1 = c[1][jj], e2 = c[2][jj], e3 = c[3][jj];
            const int j = 2 * hf + jj;
            put(0 * 4 + j, pk_sub(e0, e2)); put(1 * 4 + j, e1 + e2); put(2 * 4 + j, pk_sub(e2, e1)); put(3 * 4 + j, pk_sub(e1, e3));
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 e0 = d[i * 4 + 0], e1 = d[i * 4 + 1], e2 = d[i * 4 + 2], e3 = d[i * 4 + 3];
            d[i * 4 + 0] = pk_sub(e0, e2); d[i * 4 + 1] = e1 + e2; d[i * 4 + 2] = pk_sub(e2, e1); d[i * 4 + 3] = pk_sub(e1, e3);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 e0 = d[0 * 4 + j], e1 = d[1 * 4 + j], e2 = d[2 * 4 + j], e3 = d[3 * 4 + j];
            d[0 * 4 + j] = pk_sub(e0, e2); d[1 * 4 + j] = e1 + e2; d[2 * 4 + j] = pk_sub(e2, e1); d[3 * 4 + j] = pk_sub(e1, e3);
          }
#pragma unroll
          for (int k = 0; k < 16; ++k) put(k, d[k]);
        }
      }
      __syncthreads();
      // ---- next chunk's loads (possibly the next sample's first chunk): in flight under this matrix phase -----
#if !(CM_WINO_ABL & 2)
      if (ch + 1 < nchunks) issue(b, ch + 1);
      else if (more_b) issue(b + G, 0);
#endif
      // ---- matrix phase: NG groups (z tap [, 8-channel half]) x 4 components, A fragments one group ahead -------
      auto aread = [&](int g, f32x4 (&af)[4]) {
        const int dz = F16 ? g : g >> 1, k8 = F16 ? 0 : g & 1;
        const int acol = SWZ ? 4 * ((2 * k8 + hh) ^ (((ar >> 2) + dz) & 3)) : 8 * k8;
#pragma unroll
        for (int x = 0; x < 4; ++x) af[x] = *reinterpret_cast<const f32x4 *>(arow + (size_t)x * URC * S + (size_t)dz * NP * S + acol);
      };
      if constexpr (B6) {
        const float *ab = arow;                    // (F16-style fragment: 8 halves per lane at 4 hh dwords, planes 8 dwords apart)
        f32x4 af6[2][3];
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) af6[0][tm] = *reinterpret_cast<const f32x4 *>(ab + rowoff[0] + TST * tm);
#pragma unroll
        for (int sx = 0; sx < 12; ++sx) {          // step = (z tap g, component x)
          const int g = sx >> 2, x = sx & 3;
          (void)g; (void)x;
          if (sx + 1 < 12) {
            const int g1 = (sx + 1) >> 2, x1 = (sx + 1) & 3;
#pragma unroll
            for (int tm = 0; tm < 3; ++tm)
              af6[(sx + 1) & 1][tm] = *reinterpret_cast<const f32x4 *>(ab + (size_t)x1 * URC * S + rowoff[g1] + TST * tm);
          }
          // (A term, B term), small products first: hi = 0, mid = 1, lo = 2
          constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
          for (int u = (B6 >= 2 ? 3 : 0); u < ((CM_WINO_ABL & 8) ? (B6 >= 2 ? 4 : 1) : 6); ++u) {
            if constexpr (B6 == 3)
              acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af6[sx & 1][TA[u]]),
                                                              __builtin_bit_cast(f16x8, b6[sx % RD6][TB[u]]), acc[x], 0, 0, 0);
            else
              acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af6[sx & 1][TA[u]]),
                                                               __builtin_bit_cast(bf16x8, b6[sx % RD6][TB[u]]), acc[x], 0, 0, 0);
          }
          // refill the slot just read with step sx + RD6 (of this chunk, the next one, or the next sample's first)
          {
            constexpr int CHS = 4 * NG * 4 * 3 * 64;
            const int sn = sx + RD6;
            const bool wraps = sn >= 12;
            const bool more = !wraps || ch + 1 < nchunks || more_b;
            const int cn = !wraps ? ch : (ch + 1 < nchunks ? ch + 1 : 0);
            const f32x4 *wn = wbase + (size_t)cn * CHS + (size_t)((wraps ? sn - 12 : sn) * 3) * 64;
#if !(CM_WINO_ABL & 1)
            if (more) {
#pragma unroll
              for (int tm = 0; tm < 3; ++tm) b6[sx % RD6][tm] = wn[tm * 64];
            }
#else
            (void)wn; (void)more;
#endif
          }
          asm volatile("" ::: "memory");
        }
      } else {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        {
          const bool more = g < NG - 1 || ch + 1 < nchunks || more_b;
          const int cn = g < NG - 1 ? ch : (ch + 1 < nchunks ? ch + 1 : 0);
          const f32x4 *wn = wbase + (size_t)cn * (4 * NG * 4 * 64) + (size_t)(g < NG - 1 ? g + 1 : 0) * (4 * 64);
          if (more) {
#pragma unroll
            for (int x = 0; x < 4; ++x) bq[(g + 1) % RS][x] = wn[x * 64];
          }
        }
        f32x4 af[4];
        aread(g, af);
        if constexpr (F16) {
#pragma unroll
          for (int x = 0; x < 4; ++x)
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[x]), __builtin_bit_cast(f16x8, bq[g % RS][x]), acc[x], 0, 0, 0);
        } else {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int x = 0; x < 4; ++x)
              acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[x][jj], bq[g % RS][x][jj], acc[x], 0, 0, 0);
        }
      }
      }
    }

    // ---- tile epilogue ---------------------------------------------------------------------------------------
    // output transform: A^T = (1,1,1,0), (0,1,-1,-1) along x in registers, along y through LDS (16-byte accesses)
    f32x16 t0 = acc[0] + acc[1] + acc[2];
    f32x16 t1 = acc[1] - acc[2] - acc[3];
#ifndef CM_WINO_SKIP_EARLY
#define CM_WINO_SKIP_EARLY 0   // (measured: no gain -- 31.9 vs 32.8, 70.4 vs 69.7 us per layer, step 1.169 vs 1.166 ms)
#endif
    constexpr bool SKIP_EARLY = SKIP && B6 == 3 && CM_WINO_SKIP_EARLY;
    f32x4 ska[4], skw[4];
    int svox = 0;
    if constexpr (SKIP) {
      const int o = outoff[wave * 32 + r];
      svox = (o >= 0 ? o : 0) + (int)(b * Vo);
      // h2 (24 registers lighter than the six-term form): the skip conv's first operand chunk travels under the output transform
      if constexpr (SKIP_EARLY) wino_skip_load1(a, nt, n2a, n2, 0, svox, lane, ska, skw);
    }
    if constexpr ((CM_WINO_ABL & 16) != 0) {      // (ablation: no epilogue; one store keeps the accumulators alive)
      if (t0[0] + t1[0] == 12345.f) a.out[0] = t0[1];
      __syncthreads();
      continue;
    }
    __syncthreads();                            // U is dead: reuse as the exchange buffer
    f32x4 *const XC = reinterpret_cast<f32x4 *>(U) + (size_t)nbw * (4 * 2 * 4 * 64);   // [wave][blk 2][q 4][lane 64] float4
    {
      f32x4 *xb = XC + (size_t)(wave * 2) * 4 * 64 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xb[q * 64] = f32x4{t0[4 * q], t0[4 * q + 1], t0[4 * q + 2], t0[4 * q + 3]};
        xb[(4 + q) * 64] = f32x4{t1[4 * q], t1[4 * q + 1], t1[4 * q + 2], t1[4 * q + 3]};
      }
    }
    __syncthreads();
    const int oa = wave >> 1, ob = wave & 1;
    f32x16 v;
    {
      const f32x4 *p0 = XC + (size_t)(((oa + 0) * 2 + ob) * 4) * 64 + lane;       // oa = 0: waves 0,1,2 (+,+,+); oa = 1: waves 1,2,3 (+,-,-)
      const f32x4 *p1 = XC + (size_t)(((oa + 1) * 2 + ob) * 4) * 64 + lane;
      const f32x4 *p2 = XC + (size_t)(((oa + 2) * 2 + ob) * 4) * 64 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 u0 = p0[q * 64], u1 = p1[q * 64], u2 = p2[q * 64];
        const f32x4 w4 = oa == 0 ? (u0 + u1) + u2 : (u0 - u1) - u2;
        v[4 * q] = w4[0]; v[4 * q + 1] = w4[1]; v[4 * q + 2] = w4[2]; v[4 * q + 3] = w4[3];
      }
    }
    if constexpr (B6 == 3) v = v * a.h2_oscale;   // (h2 fragments hold w * 2^k; the skip conv below adds to the unscaled sums)
    if constexpr (SKIP) {
      // operands one 32-channel chunk at a time (double buffering them was the register peak of the kernel and spilled into the
      // chunk loop: 37 registers even in the h2 form); h2: the FIRST chunk was requested above the output transform
      for (int c2 = 0; c2 < n2; ++c2) {
        if (!(SKIP_EARLY && c2 == 0)) wino_skip_load1(a, nt, n2a, n2, c2, svox, lane, ska, skw);
        wino_skip_mfma1(ska, skw, v);
      }
    }
    // bias, time-embedding row, residual, channels-last store, GroupNorm statistics (slot format of gn_finalize)
    {
      float *const outb = a.out + (size_t)b * Vo * a.out_cs;
      const unsigned ocs4 = (unsigned)a.out_cs * 4u, n4 = (unsigned)(nok ? n : 0) * 4u;
      int orow[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const i32x4 v4 = op4[2 * q];
        orow[4 * q] = v4[0]; orow[4 * q + 1] = v4[1]; orow[4 * q + 2] = v4[2]; orow[4 * q + 3] = v4[3];
      }
      float rs[16];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) rs[reg] = v[reg] + bias_pre;
      if (a.temb) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) rs[reg] += tv_pre;
      }
      if (a.resid) {
        const float *const resb = a.resid + (size_t)b * Vo * a.res_cs;
        const unsigned rcs4 = (unsigned)a.res_cs * 4u;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const unsigned oc = (unsigned)(orow[reg] >= 0 ? orow[reg] : 0);
          rs[reg] += *reinterpret_cast<const float *>(reinterpret_cast<const char *>(resb) + (__umul24(oc, rcs4) + n4));
        }
      }
      if (wave_all) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          *reinterpret_cast<float *>(reinterpret_cast<char *>(outb) + (__umul24((unsigned)orow[reg], ocs4) + n4)) = rs[reg];
      } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if (nok && orow[reg] >= 0)
            *reinterpret_cast<float *>(reinterpret_cast<char *>(outb) + (__umul24((unsigned)orow[reg], ocs4) + n4)) = rs[reg];
      }
      if (a.stat_part || a.astat) {
        float s1 = 0.f, cnt = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if (orow[reg] >= 0) { s1 += rs[reg]; cnt += 1.f; }
        s1 += __shfl_xor(s1, 32);
        cnt += __shfl_xor(cnt, 32);
        const float mean = cnt > 0.f ? s1 / cnt : 0.f;
        float q = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if (orow[reg] >= 0) { const float dd = rs[reg] - mean; q += dd * dd; }
        q += __shfl_xor(q, 32);
        if (a.astat) {
          if (hh == 0 && nok && cnt > 0.f) cm_stat_atomic(a.astat + ((size_t)b * a.astat_C + n) * 3, s1, mean, q);
        } else {
          if (hh == 0 && nok) {
            float *sp2 = a.stat_part + (((size_t)b * a.stat_ns + slot) * a.stat_C + n) * 2;
            sp2[0] = mean;
            sp2[1] = q;
          }
          if (lane == 0 && n == 0) a.stat_cnt[(size_t)b * a.stat_ns + slot] = cnt;
        }
      }
    }
  }
}

// Geometry tables of the persistent kernel for one (grid, tile) pair, `nth` = threads per workgroup:
//   tabA[p][k][nth / 4]: in-sample SOURCE voxel index of halo voxel v = j + (nth / 4) k of tile position p, or -1
//                        (zero padding, or v beyond the halo box);   tabO[p][4 (a, b)][32 rows]: in-sample OUTPUT voxel
//   index of the row's (a, b) output, or -1 (row beyond the tile, or a patch a shifted last tile does not own).
// p = (tz * nty + ty) * ntx + tx, the statistics-slot order of conv_wino_kernel.
static void wino_tables(const ConvArgs &a, int nth, std::vector<int> &tA, std::vector<int> &tO) {
  const int PY = a.by / 2, PX = a.bx / 2, BZ = a.bz, NP = PY * PX, ROWS = BZ * NP;
  const int HZ = BZ + 2, RYH = 2 * PY + 2, RXH = 2 * PX + 2, RV = HZ * RYH * RXH;
  const int RK = (RV * 4 + nth - 1) / nth, q4 = nth / 4;
  const int ntp = a.ntz * a.nty * a.ntx;
  const int pyt = a.Yo >> 1, pxt = a.Xo >> 1;
  tA.assign((size_t)ntp * RK * q4, -1);
  tO.assign((size_t)ntp * 128, -1);
  for (int tz = 0; tz < a.ntz; ++tz)
    for (int ty = 0; ty < a.nty; ++ty)
      for (int tx = 0; tx < a.ntx; ++tx) {
        const int p = (tz * a.nty + ty) * a.ntx + tx;
        const int py0 = std::min(ty * PY, pyt - PY), px0 = std::min(tx * PX, pxt - PX);
        const int z0 = tz * BZ, y0 = 2 * py0, x0 = 2 * px0;
        for (int k = 0; k < RK; ++k)
          for (int j = 0; j < q4; ++j) {
            const int v = j + q4 * k;
            if (v >= RV) continue;
            const int vz = v / (RYH * RXH), rem = v - vz * (RYH * RXH), vy = rem / RXH, vx = rem - vy * RXH;
            const int cz = z0 - 1 + vz, cy = y0 - 1 + vy, cx = x0 - 1 + vx;
            if (cz >= 0 && cz < a.Zs && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs)
              tA[((size_t)p * RK + k) * q4 + j] = (cz * a.Ys + cy) * a.Xs + cx;
          }
        for (int ab = 0; ab < 4; ++ab)
          for (int row = 0; row < 32; ++row) {
            const int zr = row / NP, pr = row % NP, py = pr / PX, px = pr % PX;
            const int oz = z0 + zr, oy = y0 + 2 * py + (ab >> 1), ox = x0 + 2 * px + (ab & 1);
            const bool own = row < ROWS && py0 + py >= ty * PY && px0 + px >= tx * PX;
            if (own && oz < a.Zo && oy < a.Yo && ox < a.Xo) tO[(size_t)p * 128 + ab * 32 + row] = (oz * a.Yo + oy) * a.Xo + ox;
          }
      }
}

// instantiated tiles (bz planes, by / 2 x bx / 2 patches): full resolution 8 x 2 x 2 (32 rows), half resolution of the
// ATC / 2x grids 2 x 3 x 5 (30 rows), half resolution of the CR-120 grid 2 x 7 x 2 (28 rows)
#define CM_WINO_TILES(X) X(8, 2, 2) X(2, 3, 5) X(2, 7, 2)

bool conv_wino_tile_ok(int bz, int by, int bx) {
#define X(z, py, px) if (bz == z && by == 2 * py && bx == 2 * px) return true;
  CM_WINO_TILES(X)
#undef X
  return false;
}

bool conv_wino_pick(int Zo, int Yo, int Xo, int *bz, int *by, int *bx) {
  if (Yo % 2 || Xo % 2) return false;
  double best = 0;
#define X(z, py, px)                                                                                          \
  if (Zo % z == 0 && 2 * py <= Yo && 2 * px <= Xo) {                                                          \
    const int ny = (Yo / 2 + py - 1) / py, nx = (Xo / 2 + px - 1) / px;                                       \
    const double eff = (double)(Yo / 2) * (Xo / 2) / ((double)ny * py * nx * px) * (z * py * px) / 32.0;      \
    if (eff > best) { best = eff; *bz = z; *by = 2 * py; *bx = 2 * px; }                                      \
  }
  CM_WINO_TILES(X)
#undef X
  return best >= 0.6;
}

// two-step staging where the activated halo image fits next to U at two workgroups per CU: the full-resolution tile,
// and every tile of the f16 plan (its U image is 40 % smaller)
bool conv_wino_two_step(int bz, int by, int bx, bool f16, int nbw) { return (bz == 8 && by == 4 && bx == 4) || f16 || nbw == 2; }

// output tiles per workgroup: two (512 threads) for layers with a multiple of 64 output channels, except on the 8x2x2 tile
int conv_wino_nbw(int bz, int Co) { return (bz != 8 && Co % 64 == 0) ? 2 : 1; }
static int wino_cu_count();
// ... of ONE launch.  f16 operands (reduced-precision plan): the matrix phase is a sixteenth of the fp32 one and the launch is a latency
// chain, so two output tiles per workgroup only pay when the two-tile grid still fills the chip -- the quarter resolution of the
// 24 x 72 grid at B = 32 is 128 two-tile workgroups on 256 CUs: one tile per workgroup, 1.756 -> 1.689 ms per step of that plan
int conv_wino_nbw_run(const ConvArgs &a, bool f16) {
  const int nbw = conv_wino_nbw(a.bz, a.Co);
  static const bool keep = cm::diag_env("CM_WINO_F16_NBW2") != nullptr;
  if (f16 && nbw == 2 && !keep && (long long)a.B * a.ntz * a.nty * a.ntx * (a.Co / 64) < wino_cu_count()) return 1;
  return nbw;
}

size_t conv_wino_lds(int bz, int by, int bx, bool f16, int nbw) {
  const size_t ur = (size_t)(bz + 2) * (by / 2) * (bx / 2);
  const bool two = conv_wino_two_step(bz, by, bx, f16, nbw);
  const size_t u = 16 * ur * (f16 ? 12 : ((two && by * bx == 16) ? 16 : 20));   // (swizzled unpadded rows on the 2 x 2 patch tile)
  const size_t rimg = two ? (size_t)(bz + 2) * (by + 2) * (bx + 2) * (bz == 8 ? 24 : 20) : 0;
  const size_t x = (size_t)nbw * 4 * 2 * 16 * 64;  // exchange buffers of the output transform (overlay U and R)
  return (128 + (u + rimg > x ? u + rimg : x)) * sizeof(float);
}

bool conv_wino_ok(const ConvArgs &a) {
  return a.ntaps == 27 && a.td == 3 && a.stride == 1 && !a.par && !a.ups && a.ks <= 1 && a.bs == 1 && (!a.s2w || (a.s2C0 % 32 == 0 && a.s2C1 % 32 == 0)) &&
         a.C0 % 16 == 0 && a.C1 % 16 == 0 && conv_wino_tile_ok(a.bz, a.by, a.bx) && a.Zo % a.bz == 0 && a.Yo % 2 == 0 && a.Xo % 2 == 0 &&
         a.by <= a.Yo && a.bx <= a.Xo && a.nty == (a.Yo + a.by - 1) / a.by && a.ntx == (a.Xo + a.bx - 1) / a.bx && a.ntz == a.Zo / a.bz &&
         a.Zs == a.Zo && a.Ys == a.Yo && a.Xs == a.Xo && conv_wino_lds(a.bz, a.by, a.bx, false, conv_wino_nbw(a.bz, a.Co)) <= (conv_wino_nbw(a.bz, a.Co) == 2 ? 160 : 80) * 1024;
}

size_t conv_wino_p_lds(int bz, int by, int bx, bool f16, int nbw, bool b6) {
  const bool cmp = b6 && bz == 8;                   // compact form: no rows for the two padding planes (conv_wino_p_kernel)
  const size_t ur = (size_t)(bz + (cmp ? 1 : 2)) * (by / 2) * (bx / 2);
  const size_t u = 16 * ur * (cmp ? 24 : (b6 ? 28 : (f16 ? 12 : (by * bx == 16 ? 16 : 20))));
  const size_t x = (size_t)nbw * 4 * 2 * 16 * 64;
  const size_t rimg = (size_t)(bz + (cmp ? 0 : 2)) * (by + 2) * (bx + 2) * (cmp ? 20 : (bz == 8 ? 24 : 20));
  return (128 + std::max(u, x) + rimg) * sizeof(float);       // R behind max(U, exchange): see conv_wino_p_kernel
}

// device copies of the geometry tables, one per (device, grid, tile, threads): tiny (tens of KB), built on first use
struct WinoTabs { int *tA = nullptr, *tO = nullptr; };
static hipError_t wino_tabs_get(const ConvArgs &a, int nth, WinoTabs *out) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int, int, int, int, int>, WinoTabs> cache;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const auto key = std::make_tuple(dev, a.Zo, a.Yo, a.Xo, a.bz, a.by, a.bx, nth);
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it == cache.end()) {
    std::vector<int> tA, tO;
    wino_tables(a, nth, tA, tO);
    WinoTabs t;
    hipError_t e = hipMalloc((void **)&t.tA, tA.size() * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&t.tO, tO.size() * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(t.tA, tA.data(), tA.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t.tO, tO.data(), tO.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    it = cache.emplace(key, t).first;
  }
  *out = it->second;
  return hipSuccess;
}

static int wino_cu_count() {
  static int cus[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int &c = cus[dev & 63];
  if (!c) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    c = v;
  }
  return c;
}

// Split fragments of the six-term form, re-derived on the device from the fp32 Winograd fragments (pack_wino order
// [n tile][chunk][xi_y][step = (dz * 2 + k8) * 4 + xi_x][lane][4], ci = chunk * 16 + 8 k8 + 4 hh + jj) into pack_wino_b6 order
// [n tile][chunk][xi_y][dz][xi_x][term][lane][8 bf16], ci = chunk * 16 + 8 hh + j.  One thread per transformed weight.
__global__ __launch_bounds__(256) void wino_b6_repack_kernel(const float *__restrict__ wf, unsigned short *__restrict__ w6, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // i enumerates the SOURCE order: (((((nt * nch + chunk) * 4 + xy) * 24 + step) * 64) + lane) * 4 + jj
  const int jj = (int)(i & 3), lane = (int)((i >> 2) & 63);
  long long q = i >> 8;
  const int step = (int)(q % 24); q /= 24;
  const int xy = (int)(q & 3);
  const long long tc = q >> 2;                       // nt * nch + chunk
  const int xx = step & 3, k8 = (step >> 2) & 1, dz = step >> 3;
  const int r = lane & 31, hs = lane >> 5;
  const int cl = 8 * k8 + 4 * hs + jj;               // channel within the 16-channel chunk
  const int hd = cl >> 3, j = cl & 7;
  unsigned short *dst = w6 + ((((((size_t)tc * 4 + xy) * 3 + dz) * 4 + xx) * 3) * 64 + 32 * hd + r) * 8 + j;
  float rem = wf[i];
#pragma unroll
  for (int tm = 0; tm < 3; ++tm) {
    const __bf16 hb = (__bf16)rem;
    dst[(size_t)tm * 64 * 8] = __builtin_bit_cast(unsigned short, hb);
    rem -= (float)hb;
  }
}

hipError_t launch_wino_b6_repack(const float *wwino, float *w6, long long n_floats, hipStream_t st) {
  if (n_floats <= 0 || n_floats % (4 * 24 * 64 * 4)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(wino_b6_repack_kernel, dim3((unsigned)((n_floats + 255) / 256)), dim3(256), 0, st, wwino,
                     reinterpret_cast<unsigned short *>(w6), n_floats);
  return hipGetLastError();
}

// can this layer run the six-term bf16 form (wfrag = pack_wino_b6 fragments, a.f16 = 2)?  Two-tile table-driven kernel only.
bool conv_wino_b6_ok(int bz, int by, int bx, int Co, int Zo) {
  if (cm::diag_env("CM_NO_WINO_P") || !conv_wino_tile_ok(bz, by, bx)) return false;
  const int nbw = conv_wino_nbw(bz, Co);
  // two-tile layers (one workgroup per CU), or the full-resolution tile whose planes span the whole frame axis (compact LDS
  // form: the padding planes have no rows, two workgroups per CU)
  if (bz != 8) return nbw == 2 && conv_wino_p_lds(bz, by, bx, false, 2, true) <= 160 * 1024;
  return nbw == 1 && by == 4 && bx == 4 && Zo == 8 && 2 * conv_wino_p_lds(bz, by, bx, false, 1, true) <= 160 * 1024 && !cm::diag_env("CM_NO_WINO_B6_FULL");
}

hipError_t launch_conv_wino(const ConvArgs &a_in, bool f16, hipStream_t st) {
  ConvArgs a = a_in;
  a.dbg = conv_dbg_flags();
  if (!conv_wino_ok(a)) return hipErrorInvalidValue;
  const int nbw = conv_wino_nbw_run(a, f16);
  const bool b6 = a.f16 == 2 || a.f16 == 3 || a.f16 == 4;   // 3: the relaxed plan's three-term form on the same fragments;
  const bool t3 = a.f16 == 3, h2 = a.f16 == 4;               // 4: h2 fragments (f16 hi / mid of w * 2^k), three f16 cross terms
  if (b6 && (f16 || !conv_wino_b6_ok(a.bz, a.by, a.bx, a.Co, a.Zo))) return hipErrorInvalidValue;
  static const bool no_p = cm::diag_env("CM_NO_WINO_P") != nullptr;
#define CM_WINO_ATTR(KERNEL)                                                                        \
    static bool attr_set[64] = {false};                                                             \
    int dev = 0;                                                                                    \
    (void)hipGetDevice(&dev);                                                                       \
    if (!attr_set[dev & 63]) {                                                                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return e;                                                                \
      attr_set[dev & 63] = true;                                                                    \
    }
  // ---- persistent two-step form: workgroup = (tile position, sample lane), loops over its samples -----------------
  // Measured (round 3, ATC B = 64): on the one-round launches of the two-tile form (one workgroup per sample and tile) the
  // leaner prologue / epilogue is worth 1-2 %; on the full-resolution 8x2x2 tile (3.4 rounds) the persistent loop is no
  // faster per tile than the hardware's own workgroup scheduling -- the kernel is issue-bound, not latency-bound: SQ counters
  // show the SIMD 83 % busy (58 % matrix, 24 % other vector instructions) while two workgroups are resident -- and its static
  // sample lanes balance worse (85 vs 79 us).  CM_WINO_P=1 under CM_DIAG forces it everywhere for A/B runs.
  static const bool all_p = cm::diag_env("CM_WINO_P") != nullptr;
  if (conv_wino_two_step(a.bz, a.by, a.bx, f16, nbw) && !no_p && (nbw == 2 || all_p || b6)) {
    const size_t ldsp = conv_wino_p_lds(a.bz, a.by, a.bx, f16, nbw, b6) + ((a.gs0 || a.gp0) ? (size_t)2 * (a.C0 + a.C1) * sizeof(float) : 0);
    const int ntp = a.ntz * a.nty * a.ntx, nz = (a.Co + 31) / 32 / nbw;
    const int per_cu = (nbw == 1 && 2 * ldsp <= 160 * 1024) ? 2 : 1;
    const int slots = wino_cu_count() * per_cu;
    static const int g_force = cm::diag_env("CM_WINO_G") ? atoi(cm::diag_env("CM_WINO_G")) : 0;
    // (the full-resolution six-term form runs one tile per workgroup: the persistent loop's static sample lanes balance
    //  worse than the hardware's own workgroup order on its 3.4-round launches)
    const int G = g_force > 0 ? std::min(a.B, g_force) : (b6 && nbw == 1) ? a.B : std::max(1, std::min(a.B, slots / std::max(1, ntp * nz)));
    WinoTabs tb;
    hipError_t et = wino_tabs_get(a, 256 * nbw, &tb);
    if (et != hipSuccess) return et;
    const dim3 gridp((unsigned)G, (unsigned)ntp, (unsigned)nz);
#define CM_WINO_PGO1(KERNEL, THREADS)                                                               \
  {                                                                                                 \
    CM_WINO_ATTR(KERNEL)                                                                            \
    hipLaunchKernelGGL(KERNEL, gridp, dim3(THREADS), ldsp, st, a, tb.tA, tb.tO, G);                 \
    return hipGetLastError();                                                                       \
  }
#define CM_WINO_PGO(Z, PY_, PX_, F, NB, THREADS)                                                    \
  {                                                                                                 \
    if (a.s2w) CM_WINO_PGO1((conv_wino_p_kernel<Z, PY_, PX_, F, NB, true>), THREADS)                \
    CM_WINO_PGO1((conv_wino_p_kernel<Z, PY_, PX_, F, NB, false>), THREADS)                          \
  }
#define CM_WINO_PGO6N(Z, PY_, PX_, NB, THREADS)                                                     \
  {                                                                                                 \
    if (h2) {                                                                                       \
      if (a.s2w) CM_WINO_PGO1((conv_wino_p_kernel<Z, PY_, PX_, false, NB, true, 3>), THREADS)       \
      CM_WINO_PGO1((conv_wino_p_kernel<Z, PY_, PX_, false, NB, false, 3>), THREADS)                 \
    }                                                                                               \
    if (t3) {                                                                                       \
      if (a.s2w) CM_WINO_PGO1((conv_wino_p_kernel<Z, PY_, PX_, false, NB, true, 2>), THREADS)       \
      CM_WINO_PGO1((conv_wino_p_kernel<Z, PY_, PX_, false, NB, false, 2>), THREADS)                 \
    }                                                                                               \
    if (a.s2w) CM_WINO_PGO1((conv_wino_p_kernel<Z, PY_, PX_, false, NB, true, 1>), THREADS)         \
    CM_WINO_PGO1((conv_wino_p_kernel<Z, PY_, PX_, false, NB, false, 1>), THREADS)                   \
  }
#define CM_WINO_PGO61(Z, PY_, PX_, THREADS) CM_WINO_PGO6N(Z, PY_, PX_, 1, THREADS)
#define CM_WINO_PGO6(Z, PY_, PX_, THREADS) CM_WINO_PGO6N(Z, PY_, PX_, 2, THREADS)
#define X(z, py, px)                                                                                \
    if (a.bz == z && a.by == 2 * py && a.bx == 2 * px && ldsp <= 160 * 1024) {                      \
      if constexpr (z != 8) {                                                                       \
        if (nbw == 2 && b6) CM_WINO_PGO6(z, py, px, 512)                                            \
        if (nbw == 2 && f16) CM_WINO_PGO(z, py, px, true, 2, 512)                                   \
        if (nbw == 2) CM_WINO_PGO(z, py, px, false, 2, 512)                                         \
      }                                                                                             \
      if (f16) CM_WINO_PGO(z, py, px, true, 1, 256)                                                 \
      if constexpr (z == 8 && py == 2 && px == 2) {                                                 \
        if (b6) CM_WINO_PGO61(z, py, px, 256)                                                       \
        CM_WINO_PGO(z, py, px, false, 1, 256)                                                       \
      }                                                                                             \
    }
    CM_WINO_TILES(X)
#undef X
  }
  if (b6) return hipErrorInvalidValue;            // (only the table-driven kernel has the six-term form)
  const dim3 grid((unsigned)(a.B * a.ntz * a.nty * a.ntx), (unsigned)((a.Co + 31) / 32 / nbw));
  const size_t lds = conv_wino_lds(a.bz, a.by, a.bx, f16, nbw);
#define CM_WINO_GO(KERNEL, THREADS)                                                                 \
  {                                                                                                 \
    CM_WINO_ATTR(KERNEL)                                                                            \
    hipLaunchKernelGGL(KERNEL, grid, dim3(THREADS), lds, st, a);                                    \
    return hipGetLastError();                                                                       \
  }
#define X(z, py, px)                                                                                \
  if (a.bz == z && a.by == 2 * py && a.bx == 2 * px) {                                              \
    constexpr bool two = (z == 8 && py == 2 && px == 2);                                            \
    if constexpr (z != 8) {                                                                         \
      if (nbw == 2 && f16) CM_WINO_GO((conv_wino_kernel<z, py, px, 2, true, true, 2>), 512)         \
      if (nbw == 2) CM_WINO_GO((conv_wino_kernel<z, py, px, 2, false, true, 2>), 512)               \
    }                                                                                               \
    if (f16) CM_WINO_GO((conv_wino_kernel<z, py, px, 2, true, true, 1>), 256)                       \
    CM_WINO_GO((conv_wino_kernel<z, py, px, 2, false, two, 1>), 256)                                \
  }
  CM_WINO_TILES(X)
#undef X
  return hipErrorInvalidValue;
}

// ================================================================================================================
// Weight gradient of a Winograd layer in the Winograd domain (training step, ddpm.py:142-143 loss.backward()).
//   forward:  Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          =>   d(G g G^T)[xi] = sum over tiles, planes
//   of  V[xi](ci) * dM[xi](co)   with  V = B^T d B  (the forward's transformed input, activation recomputed on the
//   fly exactly as the forward stages it)  and  dM = A dY A^T  (the 2x2 output-gradient patch spread to 4x4);
//   finally  dg = G^T dU G  per (co, ci, z tap) in the reduce pass.  16 multiplies per 2x2 outputs and z tap instead
//   of 36 -- the same 2.25x saving as the forward, on the kernel that was 8.4 ms of the 22 ms training step.
// MFMA 32x32x2 with the PATCH ROW as the contraction index: A operand = V[xi][row][ci] (lane: ci, k = row parity),
// B operand = dM[xi][row][co], computed in registers from the raw 2x2 dY patch (one ds_read_b128 per row pair).
// Wave w owns frequency row xi_y = w: 4 components x 3 z taps = 12 accumulator blocks (32 ci x 32 co) that persist
// over all tiles of the workgroup; partials part[g][cb][kb][dz][xi][ci][co], summed and transformed by
// wgrad_wino_reduce_kernel in a fixed order.  One workgroup per CU (V image 80 KB + 192 accumulator registers).
template <int BZ, int PY, int PX>
__global__ __launch_bounds__(512, 2) void wgrad_wino_kernel(const ConvArgs a, const float *__restrict__ dy, int dy_cs,
                                                            float *__restrict__ part, int G) {
  constexpr int NP = PY * PX, ROWS = BZ * NP, HZ = BZ + 2, UR = HZ * NP;
  static_assert(ROWS <= 32 && ROWS > 16, "one (partly filled) 32-row block");
  constexpr int NIT = HZ * NP * 8;               // staging items (plane, patch, channel quad of 32 ci): one per thread
  static_assert(NIT <= 512, "one staging round of the 512 threads");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *V = lds;                                  // [16][UR][32]
  float *Yt = V + 16 * UR * 32;                    // [32 rows][32 co][4 = (a, b)]
  // TWO (full-resolution tile, round 4): two-step staging as in the forward kernel -- every halo voxel of the tile is loaded and
  // activated ONCE into R[voxel][32 ci (+ 4 pad)], the 4 x 4 patches are then read from LDS: a patch item used to load and activate its
  // 16 voxels itself, four times per voxel (GroupNorm + SiLU: two quarter-rate transcendentals per element).  The half-resolution
  // tile's V image (123 KB) leaves no room for R.
#ifndef CM_WGRAD_TWO
#define CM_WGRAD_TWO 1
#endif
  constexpr bool TWO = BZ == 8 && CM_WGRAD_TWO;
  constexpr int RYH = 2 * PY + 2, RXH = 2 * PX + 2, RVX = HZ * RYH * RXH, RSX = 36;
  float *R = Yt + 32 * 32 * 4;                     // [RVX][RSX] (TWO only)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = w8 & 3, xh = w8 >> 2;          // frequency row xi_y, and which two of its four xi_x components
  const int r = lane & 31, hh = lane >> 5;
  const int g = blockIdx.x, cb = blockIdx.y, kb = blockIdx.z;
  const int Ctot = a.C0 + a.C1;
  const int ci0 = kb * 32;
  const int pyt = a.Yo >> 1, pxt = a.Xo >> 1;
  const int ntile = a.B * a.ntz * a.nty * a.ntx;
  // A^T rows (1,1,1,0), (0,1,-1,-1)  =>  A = (1,0), (1,1), (1,-1), (0,-1): this wave's row of A along y
  const float cy0 = wave == 3 ? 0.f : 1.f, cy1 = wave == 0 ? 0.f : (wave == 1 ? 1.f : -1.f);

  f32x16 acc[3][2];
#pragma unroll
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[dz][x][i] = 0.f;

  for (int tile = g; tile < ntile; tile += G) {
    int tt = tile;
    const int tx = tt % a.ntx; tt /= a.ntx;
    const int ty = tt % a.nty; tt /= a.nty;
    const int tz = tt % a.ntz;
    const int b = tt / a.ntz;
    const int py0 = min(ty * PY, pyt - PY), px0 = min(tx * PX, pxt - PX);
    const int z0 = tz * BZ, y0 = 2 * py0, x0 = 2 * px0;
    __syncthreads();                               // previous tile's operands have been read
    if constexpr (TWO) {
      // ---- step A: every halo voxel of the tile, activated once, into R ---------------------------------------------
      constexpr int RK = (RVX * 8 + 511) / 512;
      const int quad = tid & 7;
      const int c = ci0 + 4 * quad;
      const bool cok = c < Ctot;
      const bool from0 = c < a.C0;
      const float *sp = from0 ? a.src0 + c : a.src1 + (c - a.C0);
      const int Cs = from0 ? a.C0 : a.C1;
      f32x4 sc1 = {1.f, 1.f, 1.f, 1.f}, sh1 = {0.f, 0.f, 0.f, 0.f}, pm1 = {1.f, 1.f, 1.f, 1.f};
      const int cc = cok ? c : 0;
      if (a.gn) {
        const float *gp = a.gn + (size_t)b * 2 * Ctot + cc;
        sc1 = *reinterpret_cast<const f32x4 *>(gp);
        sh1 = *reinterpret_cast<const f32x4 *>(gp + Ctot);
      }
      if (a.pm) pm1 = *reinterpret_cast<const f32x4 *>(a.pm + (size_t)b * a.pm_stride + cc);
      f32x4 ld[RK];
      unsigned okm = 0;
#pragma unroll
      for (int k = 0; k < RK; ++k) {
        const int v = (tid >> 3) + 64 * k;
        const int zi = v / (RYH * RXH), rem = v - zi * (RYH * RXH), ry = rem / RXH, rx = rem - ry * RXH;
        const int cz = z0 - 1 + zi, cyy = y0 - 1 + ry, cxx = x0 - 1 + rx;
        const bool ok = v < RVX && cok && cz >= 0 && cz < a.Zs && cyy >= 0 && cyy < a.Ys && cxx >= 0 && cxx < a.Xs;
        const int off = ok ? ((b * a.Zs + cz) * a.Ys + cyy) * a.Xs + cxx : 0;
        ld[k] = *reinterpret_cast<const f32x4 *>(sp + (size_t)off * Cs);
        okm |= (ok ? 1u : 0u) << k;
      }
#pragma unroll
      for (int k = 0; k < RK; ++k) {
        const int v = (tid >> 3) + 64 * k;
        f32x4 w = ld[k];
        if (a.gn) {
          w = w * sc1 + sh1;
          if (a.silu) { w[0] = silu_w(w[0]); w[1] = silu_w(w[1]); w[2] = silu_w(w[2]); w[3] = silu_w(w[3]); }
        }
        if (a.pm) w = w * pm1;
        if (!((okm >> k) & 1u)) w = f32x4{0.f, 0.f, 0.f, 0.f};
        if (v < RVX) *reinterpret_cast<f32x4 *>(R + (size_t)v * RSX + 4 * quad) = w;
      }
      __syncthreads();
    }
    // ---- V = B^T d B of the conv's actual input (GroupNorm + SiLU + Dropout3d multiplier as in the forward) ----
    {
      const int it = tid;
      const bool stager = it < NIT;
      const int itc = stager ? it : 0;
      const int quad = itc & 7, patch = (itc >> 3) % NP, zi = itc / (8 * NP);
      const int py = patch / PX, px = patch % PX;
      const int c = ci0 + 4 * quad;                // channel quad in the concatenated channel space
      f32x4 d[16];
      if constexpr (TWO) {
        const float *rb = R + (size_t)((zi * RYH + 2 * py) * RXH + 2 * px) * RSX + 4 * quad;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) d[i * 4 + j] = *reinterpret_cast<const f32x4 *>(rb + (size_t)(i * RXH + j) * RSX);
      } else {
      const bool from0 = c < a.C0;
      const float *sp = from0 ? a.src0 + c : a.src1 + (c - a.C0);
      const int Cs = from0 ? a.C0 : a.C1;
      const int cz = z0 - 1 + zi;
      const bool zok = stager && c < Ctot && cz >= 0 && cz < a.Zs;
      unsigned okmask = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cyy = y0 + 2 * py - 1 + i, cxx = x0 + 2 * px - 1 + j;
          const bool ok = zok && cyy >= 0 && cyy < a.Ys && cxx >= 0 && cxx < a.Xs;
          const int off = ok ? ((b * a.Zs + cz) * a.Ys + cyy) * a.Xs + cxx : 0;
          d[i * 4 + j] = *reinterpret_cast<const f32x4 *>(sp + (size_t)off * Cs);
          okmask |= (ok ? 1u : 0u) << (i * 4 + j);
        }
      f32x4 sc1 = {1.f, 1.f, 1.f, 1.f}, sh1 = {0.f, 0.f, 0.f, 0.f}, pm1 = {1.f, 1.f, 1.f, 1.f};
      const int cc = c < Ctot ? c : 0;
      if (a.gn) {
        const float *gp = a.gn + (size_t)b * 2 * Ctot + cc;
        sc1 = *reinterpret_cast<const f32x4 *>(gp);
        sh1 = *reinterpret_cast<const f32x4 *>(gp + Ctot);
      }
      if (a.pm) pm1 = *reinterpret_cast<const f32x4 *>(a.pm + (size_t)b * a.pm_stride + cc);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        f32x4 w = d[k];
        if (a.gn) {
          w = w * sc1 + sh1;
          if (a.silu) { w[0] = silu_w(w[0]); w[1] = silu_w(w[1]); w[2] = silu_w(w[2]); w[3] = silu_w(w[3]); }
        }
        if (a.pm) w = w * pm1;
        d[k] = ((okmask >> k) & 1u) ? w : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 e0 = d[i * 4 + 0], e1 = d[i * 4 + 1], e2 = d[i * 4 + 2], e3 = d[i * 4 + 3];
        d[i * 4 + 0] = pk_sub(e0, e2); d[i * 4 + 1] = e1 + e2; d[i * 4 + 2] = pk_sub(e2, e1); d[i * 4 + 3] = pk_sub(e1, e3);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 e0 = d[0 * 4 + j], e1 = d[1 * 4 + j], e2 = d[2 * 4 + j], e3 = d[3 * 4 + j];
        d[0 * 4 + j] = pk_sub(e0, e2); d[1 * 4 + j] = e1 + e2; d[2 * 4 + j] = pk_sub(e2, e1); d[3 * 4 + j] = pk_sub(e1, e3);
      }
      if (stager) {
        float *vw = V + (size_t)(zi * NP + patch) * 32 + 4 * quad;
#pragma unroll
        for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x4 *>(vw + (size_t)k * UR * 32) = d[k];
      }
    }
    // ---- raw dY patches of the 32 rows: Yt[row][co][(a, b)]; rows this tile does not own contribute nothing ----
    if (tid < 256) {
      const int row = tid >> 3, q = tid & 7;
      const int zr = row / NP, pr = row % NP, py = pr / PX, px = pr % PX;
      const bool own = row < ROWS && py0 + py >= ty * PY && px0 + px >= tx * PX;
      const int co = cb * 32 + 4 * q;
      f32x4 yv[4];
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        const int oz = z0 + zr, oy = y0 + 2 * py + (ab >> 1), ox = x0 + 2 * px + (ab & 1);
        const bool ok = own && oz < a.Zo && oy < a.Yo && ox < a.Xo && co < a.Co;
        const int off = ok ? ((b * a.Zo + oz) * a.Yo + oy) * a.Xo + ox : 0;
        yv[ab] = *reinterpret_cast<const f32x4 *>(dy + (size_t)off * dy_cs + (co < a.Co ? co : 0));
        if (!ok) yv[ab] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4)
        *reinterpret_cast<f32x4 *>(Yt + ((size_t)row * 32 + 4 * q + c4) * 4) = f32x4{yv[0][c4], yv[1][c4], yv[2][c4], yv[3][c4]};
    }
    __syncthreads();
    // ---- matrix phase: 16 row pairs x (4 components x 3 z taps) -------------------------------------------
    const float *vb = V + (size_t)(wave * 4) * UR * 32 + r;
#pragma unroll 2
    for (int p = 0; p < 16; ++p) {
      const int row = 2 * p + hh;
      const f32x4 yv = *reinterpret_cast<const f32x4 *>(Yt + ((size_t)row * 32 + r) * 4);
      const float t0 = cy0 * yv[0] + cy1 * yv[2], t1 = cy0 * yv[1] + cy1 * yv[3];      // A along y: rows a = 0, 1
      const float bm0 = xh ? t0 - t1 : t0, bm1 = xh ? -t1 : t0 + t1;                     // A along x: components 2 xh, 2 xh + 1
      const int vrow = min(row, ROWS - 1);
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        const float av0 = vb[((size_t)(2 * xh) * UR + vrow + dz * NP) * 32], av1 = vb[((size_t)(2 * xh + 1) * UR + vrow + dz * NP) * 32];
        acc[dz][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bm0, acc[dz][0], 0, 0, 0);
        acc[dz][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bm1, acc[dz][1], 0, 0, 0);
      }
    }
  }
  // ---- partials: part[g][cb][kb][dz][xi][ci][co] ----------------------------------------------------------
#pragma unroll
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      float *p = part + (((((size_t)g * gridDim.y + cb) * gridDim.z + kb) * 3 + dz) * 16 + wave * 4 + 2 * xh + x) * 1024;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int ci = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        p[ci * 32 + r] = acc[dz][x][reg];
      }
    }
}

// dW[co][ci][kH][kW][kL] (reference layout) = sum_g G^T dU G of the partials (fixed order)
__global__ __launch_bounds__(256) void wgrad_wino_reduce_kernel(const float *__restrict__ part, int G, int ncb, int nkb, int Co, int Ci,
                                                                float *__restrict__ dW) {
  // 32 outputs (consecutive co) x 8 group lanes per workgroup: lane l sums the groups l, l + 8, ... of all 16 components,
  // the lanes are merged in lane order through LDS (a fixed summation order), lane 0 transforms and stores
  __shared__ float sh[8][16][32];
  const long long n = (long long)ncb * nkb * 3 * 1024;
  const long long i = (long long)blockIdx.x * 32 + (threadIdx.x & 31);
  const int gl = threadIdx.x >> 5;
  const bool live = i < n;
  const long long ic = live ? i : 0;
  const int col = (int)(ic & 31), cil = (int)((ic >> 5) & 31);
  long long q = ic >> 10;
  const int dz = (int)(q % 3); q /= 3;
  const int kb = (int)(q % nkb);
  const int cb = (int)(q / nkb);
  const int co = cb * 32 + col, ci = kb * 32 + cil;
  const long long per_g = (long long)ncb * nkb * 3 * 16 * 1024;
  float u[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) u[k] = 0.f;
  const float *p = part + ((((size_t)cb * nkb + kb) * 3 + dz) * 16) * 1024 + cil * 32 + col;
  if (live)
    for (int g = gl; g < G; g += 8)
#pragma unroll
      for (int k = 0; k < 16; ++k) u[k] += p[(size_t)g * per_g + (size_t)k * 1024];
#pragma unroll
  for (int k = 0; k < 16; ++k) sh[gl][k][threadIdx.x & 31] = u[k];
  __syncthreads();
  if (gl != 0 || !live || co >= Co || ci >= Ci) return;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float t = sh[0][k][threadIdx.x];
#pragma unroll
    for (int l = 1; l < 8; ++l) t += sh[l][k][threadIdx.x];
    u[k] = t;
  }
  // G^T u G with G = (1,0,0), (1/2,1/2,1/2), (1/2,-1/2,1/2), (0,0,1):  G^T rows = (1, 1/2, 1/2, 0), (0, 1/2, -1/2, 0), (0, 1/2, 1/2, 1)
  float s[4][3];
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    const float u0 = u[y * 4 + 0], u1 = u[y * 4 + 1], u2 = u[y * 4 + 2], u3 = u[y * 4 + 3];
    s[y][0] = u0 + 0.5f * (u1 + u2);
    s[y][1] = 0.5f * (u1 - u2);
    s[y][2] = 0.5f * (u1 + u2) + u3;
  }
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const float o0 = s[0][dx] + 0.5f * (s[1][dx] + s[2][dx]);
    const float o1 = 0.5f * (s[1][dx] - s[2][dx]);
    const float o2 = 0.5f * (s[1][dx] + s[2][dx]) + s[3][dx];
    float *w = dW + ((size_t)co * Ci + ci) * 27;          // reference tap index (dy*3 + dx)*3 + dz
    w[(0 * 3 + dx) * 3 + dz] = o0;
    w[(1 * 3 + dx) * 3 + dz] = o1;
    w[(2 * 3 + dx) * 3 + dz] = o2;
  }
}

size_t wgrad_wino_lds(int bz, int by, int bx) {
  const size_t r = (bz == 8 && CM_WGRAD_TWO) ? (size_t)(bz + 2) * (by + 2) * (bx + 2) * 36 : 0;      // the activated halo image of the two-step form
  return ((size_t)16 * (bz + 2) * (by / 2) * (bx / 2) * 32 + 32 * 32 * 4 + r) * sizeof(float);
}

hipError_t launch_wgrad_wino(const ConvArgs &a, const float *dy, int dy_cs, float *part, int G, int ncb, int nkb, hipStream_t st) {
  if (!conv_wino_tile_ok(a.bz, a.by, a.bx) || a.stride != 1 || a.par || a.ups || (a.C0 % 4) || (a.C1 % 4)) return hipErrorInvalidValue;
  const size_t lds = wgrad_wino_lds(a.bz, a.by, a.bx);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
#define X(z, py, px)                                                                                        \
  if (a.bz == z && a.by == 2 * py && a.bx == 2 * px) {                                                      \
    static bool attr_set[64] = {false};                                                                     \
    int dev = 0;                                                                                            \
    (void)hipGetDevice(&dev);                                                                               \
    if (!attr_set[dev & 63]) {                                                                              \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_wino_kernel<z, py, px>),      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);           \
      if (e != hipSuccess) return e;                                                                        \
      attr_set[dev & 63] = true;                                                                            \
    }                                                                                                       \
    hipLaunchKernelGGL((wgrad_wino_kernel<z, py, px>), dim3(G, ncb, nkb), dim3(512), lds, st, a, dy, dy_cs, part, G); \
    return hipGetLastError();                                                                               \
  }
  CM_WINO_TILES(X)
#undef X
  return hipErrorInvalidValue;
}

hipError_t launch_wgrad_wino_reduce(const float *part, int G, int ncb, int nkb, int Co, int Ci, float *dW, hipStream_t st) {
  const long long n = (long long)ncb * nkb * 3 * 1024;
  hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, part, G, ncb, nkb, Co, Ci, dW);
  return hipGetLastError();
}

}  // namespace cm
