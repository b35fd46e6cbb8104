// Direct 3x3x3 stride-1 convolution with f16 matrix-core operands and fp32 accumulation -- the reduced-precision plan
// (BASELINE configs[4]; the reference's analogue is torch.amp.autocast around the denoiser,
// /root/reference/models/diffusion/ddpm.py:116-120) for the stride-1 nn.Conv3d of every full- and half-resolution
// ResnetBlock (/root/reference/models/backbones/layers.py:32,43).
//
// Why not Winograd here: v_mfma_f32_32x32x16_f16 runs at 16x the exact-fp32 matrix rate, so the 2.25x saving of
// F(2x2,3x3) is worth nothing next to the transform / exchange instructions it costs -- round 2's f16 Winograd layer
// (32 -> 32 on the 24x72 grid) took 106 us for 10 us of matrix instructions.  The direct form has no transforms:
//   * workgroup = 4 waves, one output box of <= 128 MBW voxels x 32 NB channels of ONE sample; wave w owns the row blocks
//     [w MBW, (w+1) MBW) and all NB column blocks (no cross-wave reduction);
//   * per 16-channel chunk the halo box is normalised (GroupNorm affine + SiLU, once per halo voxel), rounded to f16 and
//     written to LDS (row = 16 halves + 8 pad halves: conflict-free 16-byte fragment reads); the next chunk's loads are in
//     flight in registers under this chunk's matrix phase;
//   * per tap ONE matrix instruction per (row block, column block) contracts the 16 channels; weights stream global ->
//     VGPR in consumption order as f16 fragments (ring of 3 taps);
//   * the block's 1x1x1 skip convolution rides along as extra chunks (centre tap, raw input);
//   * epilogue: bias, time-embedding row, residual, fp32 channels-last store, GroupNorm statistics per (row block, channel)
//     in the slot format of gn_finalize.  GroupNorm statistics, SiLU and the accumulation stay fp32.
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
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_h(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// tabH[p][HVp]: in-sample source voxel of halo voxel h of tile position p, or -1 (zero padding / beyond the box);
// tabM[p][128 MBW]: row m -> (halo index of the row's voxel at tap (0,0,0)) | in-sample output voxel << 12 ... two ints per row:
//   tabM[2 m] = halo index (always valid, 0 for padding rows), tabM[2 m + 1] = in-sample output voxel or -1.
template <int MBW, int NB>
__global__ __launch_bounds__(256, 2) void conv_f16d_kernel(const ConvArgs a, const int *__restrict__ tabH, const int *__restrict__ tabM,
                                                          int HV, int ntp) {
  constexpr int MB = 4 * MBW;                    // row blocks per workgroup
  constexpr int AS = 12;                         // LDS row stride in dwords: 16 halves + 8 pad halves
  constexpr int NLD = 10;                        // halo items (voxel, channel quad) per thread and chunk: 4 HV / 256 <= 10 (HV <= 640)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *A = lds;                                // [HV][AS] f16 image of the current chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  int tile = blockIdx.x;
  if (!(gridDim.x & 7) && !(a.dbg & 4096)) tile = (tile & 7) * (int)(gridDim.x >> 3) + (tile >> 3);   // XCD-aware order (cm_conv.hip)
  const int b = tile / ntp, p = tile - b * ntp;
  const int nt = blockIdx.y;
  const int HXs = a.bx + 2, HYXs = (a.by + 2) * HXs;    // halo box strides (voxels)
  const unsigned Vs = (unsigned)(a.Zs * a.Ys * a.Xs);

  // ---- geometry from the host tables -------------------------------------------------------------------------------
  const int nit = (4 * HV + 255) >> 8;            // items per thread (quad q = tid & 3 is the same for all of them)
  const int q = tid & 3;
  int hoff[NLD];
  unsigned hok = 0;
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int h = (tid >> 2) + 64 * k;
    const int o = (k < nit && h < HV) ? tabH[(size_t)p * HV + h] : -1;
    hok |= (o >= 0 ? 1u : 0u) << k;
    hoff[k] = o >= 0 ? o : 0;
  }
  int abase[MBW], ovox[MBW];                       // per-lane: LDS dword offset of the row's voxel (tap 0,0,0), output voxel
#pragma unroll
  for (int j = 0; j < MBW; ++j) {
    const int m = (wave * MBW + j) * 32 + r;
    abase[j] = tabM[((size_t)p * 32 * MB + m) * 2] * AS + 4 * hh;
    ovox[j] = tabM[((size_t)p * 32 * MB + m) * 2 + 1];
  }
  const int n16 = (a.C0 + a.C1) >> 4, n0 = a.C0 >> 4;                 // 16-channel chunks of the main input
  const int ns16 = a.s2w ? (a.s2C0 + a.s2C1) >> 4 : 0, ns0 = a.s2C0 >> 4;   // ... of the fused skip input (raw, centre tap)
  const int nch = n16 + ns16;
  const int Ctot = a.C0 + a.C1;

  f32x16 acc[MBW][NB];
#pragma unroll
  for (int j = 0; j < MBW; ++j)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][nb][e] = 0.f;

  // weights: main  [nt][chunk][tap 27][nb][lane] 16 B, then skip [nt][chunk][nb][lane] 16 B
  const f32x4 *wmain = reinterpret_cast<const f32x4 *>(a.wfrag) + (size_t)nt * n16 * 27 * NB * 64 + lane;
  const f32x4 *wskip = a.s2w ? reinterpret_cast<const f32x4 *>(a.s2w) + (size_t)nt * ns16 * NB * 64 + lane : nullptr;

  // ---- loads of chunk c (main: GroupNorm rows too) -------------------------------------------------------------------
  f32x4 ld[NLD], scn = {1.f, 1.f, 1.f, 1.f}, shn = {0.f, 0.f, 0.f, 0.f};
  bool ldh = false;                                  // the loads in flight are 4 halves (8 bytes) per item: an f16 tensor (a.h16)
  auto issue = [&](int c) {
    const float *srcp;
    int Cn, cin;
    bool hsrc;
    if (c < n16) {
      const bool s0 = c < n0;
      Cn = s0 ? a.C0 : a.C1;
      srcp = s0 ? a.src0 : a.src1;
      cin = (s0 ? c : c - n0) * 16;
      hsrc = (a.h16 & (s0 ? 1 : 2)) != 0;
      if (a.gn) {
        const float *gp = a.gn + (size_t)b * 2 * Ctot + c * 16 + 4 * q;
        scn = *reinterpret_cast<const f32x4 *>(gp);
        shn = *reinterpret_cast<const f32x4 *>(gp + Ctot);
      }
    } else {
      const int cs = c - n16;
      const bool s0 = cs < ns0;
      Cn = s0 ? a.s2C0 : a.s2C1;
      srcp = s0 ? a.s2src0 : a.s2src1;
      cin = (s0 ? cs : cs - ns0) * 16;
      hsrc = (a.h16 & (s0 ? 16 : 32)) != 0;
    }
    ldh = hsrc;
    if (hsrc) {
      const char *base = reinterpret_cast<const char *>(srcp) + ((size_t)b * Vs * Cn + cin) * 2;
      const unsigned cb = (unsigned)Cn * 2u, q8 = 8u * (unsigned)q;
#pragma unroll
      for (int k = 0; k < NLD; ++k)
        if (k < nit) {
          const cm_f32x2_t two = *reinterpret_cast<const cm_f32x2_t *>(base + (__umul24((unsigned)hoff[k], cb) + q8));
          ld[k][0] = two[0]; ld[k][1] = two[1];
        }
    } else {
      const char *base = reinterpret_cast<const char *>(srcp) + ((size_t)b * Vs * Cn + cin) * 4;
      const unsigned cb = (unsigned)Cn * 4u, q16 = 16u * (unsigned)q;
#pragma unroll
      for (int k = 0; k < NLD; ++k)
        if (k < nit) ld[k] = *reinterpret_cast<const f32x4 *>(base + (__umul24((unsigned)hoff[k], cb) + q16));
    }
  };
  auto stage = [&](int c) {
    const bool main = c < n16;
    const f32x4 sc = scn, sh = shn;
    const bool hl = ldh;
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      if (k < nit) {
        const int h = (tid >> 2) + 64 * k;
        f32x4 w = ld[k];
        if (hl) {
          const f16x4 hv4 = __builtin_bit_cast(f16x4, cm_f32x2_t{ld[k][0], ld[k][1]});
          w = f32x4{(float)hv4[0], (float)hv4[1], (float)hv4[2], (float)hv4[3]};
        }
        if (main && a.gn && !(a.dbg & 128)) {
          w = w * sc + sh;
          if (a.silu) { w[0] = silu_h(w[0]); w[1] = silu_h(w[1]); w[2] = silu_h(w[2]); w[3] = silu_h(w[3]); }
        }
        if (!((hok >> k) & 1u)) w = f32x4{0.f, 0.f, 0.f, 0.f};
        const f16x4 hv = {(_Float16)w[0], (_Float16)w[1], (_Float16)w[2], (_Float16)w[3]};
        if (h < HV) *reinterpret_cast<f16x4 *>(A + (size_t)h * AS + 2 * q) = hv;
      }
  };
  issue(0);
  const int n = nt * 32 * NB + r;                 // (+ 32 nb)
  float bias_pre[NB], tv_pre[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int nn = n + 32 * nb < a.Co ? n + 32 * nb : 0;
    bias_pre[nb] = a.bias[nn];
    tv_pre[nb] = a.temb ? a.temb[(size_t)a.tidx[b] * a.temb_stride + nn] : 0.f;
  }
  // weight ring depth in taps: the fragments come from L2 (~700 cycles) and a tap is only 64-128 matrix cycles: with 2-3 taps
  // in flight the matrix phase ran at one tap per L2 round trip (dec3.conv_1 on 24x72: 100 us for 17 us of matrix work)
  constexpr int RD = NB == 2 ? 5 : 9;
  f32x4 bw[RD][NB];

  for (int c = 0; c < nch; ++c) {
    stage(c);
    __syncthreads();                              // image of chunk c complete
    if (c + 1 < nch) issue(c + 1);                // next chunk's loads: in flight under this matrix phase
    if (a.dbg & 2) { __syncthreads(); continue; }
    if (c < n16) {
      const f32x4 *wc = wmain + (size_t)c * 27 * NB * 64;
#pragma unroll
      for (int t = 0; t < RD; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bw[t][nb] = wc[(size_t)(t * NB + nb) * 64];
      f32x4 af[MBW], afn[MBW];
#pragma unroll
      for (int j = 0; j < MBW; ++j) af[j] = *reinterpret_cast<const f32x4 *>(A + abase[j]);
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        if (t + 1 < 27) {
          const int t1 = t + 1, dz = t1 / 9, dy = (t1 / 3) % 3, dx = t1 % 3;
          const int toff = (dz * HYXs + dy * HXs + dx) * AS;
#pragma unroll
          for (int j = 0; j < MBW; ++j) afn[j] = *reinterpret_cast<const f32x4 *>(A + abase[j] + toff);
        }
#pragma unroll
        for (int j = 0; j < MBW; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[j]), __builtin_bit_cast(f16x8, bw[t % RD][nb]), acc[j][nb], 0, 0, 0);
        // refill this ring slot AFTER the matrix instructions that read it (no register copies); the fence keeps the refill
        // here: without it the compiler sinks the load to just before its first use and every tap pays an L2 round trip
        if (t + RD < 27) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bw[t % RD][nb] = wc[(size_t)((t + RD) * NB + nb) * 64];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < MBW; ++j) af[j] = afn[j];
      }
    } else {
      // fused 1x1x1 skip conv (layers.py:46,74): centre tap of the raw block input
      const f32x4 *wc = wskip + (size_t)(c - n16) * NB * 64;
      const int toff = (HYXs + HXs + 1) * AS;
      f32x4 af[MBW], w4[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) w4[nb] = wc[(size_t)nb * 64];
#pragma unroll
      for (int j = 0; j < MBW; ++j) af[j] = *reinterpret_cast<const f32x4 *>(A + abase[j] + toff);
#pragma unroll
      for (int j = 0; j < MBW; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[j]), __builtin_bit_cast(f16x8, w4[nb]), acc[j][nb], 0, 0, 0);
    }
    __syncthreads();                              // every wave has read chunk c: the image may be overwritten
  }

  // ---- epilogue: lane (r, hh) of block (j, nb) holds rows (e & 3) + 8 (e >> 2) + 4 hh, channel n + 32 nb --------------------
  // the output voxel of a row lives in the lane with that r: fetch it with a wave shuffle (ds_bpermute)
  if (a.dbg & 4) {
    if (acc[0][0][0] == 123.456f) a.out[0] = 1.f;
    return;
  }
  float *const outb = a.out + (size_t)b * (size_t)(a.Zo * a.Yo * a.Xo) * a.out_cs;
  const float *const resb = a.resid ? a.resid + (size_t)b * (size_t)(a.Zo * a.Yo * a.Xo) * a.res_cs : nullptr;
  const int ns = ntp * MB;
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
      for (int e = 0; e < 16; ++e) rs[e] = acc[j][nb][e] + bias_pre[nb] + tv_pre[nb];
      if (resb) {
        if (a.h16 & 8) {
          const _Float16 *rh = reinterpret_cast<const _Float16 *>(a.resid) + (size_t)b * (size_t)(a.Zo * a.Yo * a.Xo) * a.res_cs;
#pragma unroll
          for (int e = 0; e < 16; ++e) rs[e] += (float)rh[(size_t)(orow[e] >= 0 ? orow[e] : 0) * a.res_cs + (nok ? nn : 0)];
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) rs[e] += resb[(size_t)(orow[e] >= 0 ? orow[e] : 0) * a.res_cs + (nok ? nn : 0)];
        }
      }
      if (a.h16 & 4) {
        _Float16 *oh = reinterpret_cast<_Float16 *>(a.out) + (size_t)b * (size_t)(a.Zo * a.Yo * a.Xo) * a.out_cs;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (nok && orow[e] >= 0) oh[(size_t)orow[e] * a.out_cs + nn] = (_Float16)rs[e];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (nok && orow[e] >= 0) outb[(size_t)orow[e] * a.out_cs + nn] = rs[e];
      }
      if (a.stat_part) {
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
        const int slot = p * MB + wave * MBW + j;
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

// ---- host side ---------------------------------------------------------------------------------------------------------
bool conv_f16d_pick(int Z, int Y, int X, int *bz, int *by, int *bx, int *mbw) {
  double best = 0;
  for (int z = 1; z <= Z; ++z)
    for (int y = 1; y <= Y; ++y)
      for (int x = 1; x <= X; ++x) {
        if (Z % z || Y % y || X % x) continue;
        const int rows = z * y * x;
        if (rows > 256 || rows < 64) continue;       // (<= 2 row blocks per wave: 3 or 4 spill registers into the chunk loop)
        const int w = (rows + 127) / 128;           // row blocks per wave
        const int hv = (z + 2) * (y + 2) * (x + 2);
        if (hv > 640) continue;
        const double eff = (double)rows / (128.0 * w), halo = (double)rows / hv;
        const double score = eff * (0.35 + 0.65 * halo) * (w >= 2 ? 1.0 : 0.8);
        if (score > best) { best = score; *bz = z; *by = y; *bx = x; *mbw = w; }
      }
  return best > 0;
}

struct F16dTabs { int *tH = nullptr, *tM = nullptr; int HV = 0, ntp = 0; };
static hipError_t f16d_tabs_get(const ConvArgs &a, int mbw, F16dTabs *out) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int, int, int, int, int>, F16dTabs> cache;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const auto key = std::make_tuple(dev, a.Zo, a.Yo, a.Xo, a.bz, a.by, a.bx, mbw);
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it == cache.end()) {
    const int HZ = a.bz + 2, HY = a.by + 2, HX = a.bx + 2, HV = HZ * HY * HX;
    const int ntz = a.Zo / a.bz, nty = a.Yo / a.by, ntx = a.Xo / a.bx, ntp = ntz * nty * ntx;
    const int rows = a.bz * a.by * a.bx, MR = 128 * mbw;
    std::vector<int> tH((size_t)ntp * HV, -1), tM((size_t)ntp * MR * 2, 0);
    for (int tz = 0; tz < ntz; ++tz)
      for (int ty = 0; ty < nty; ++ty)
        for (int tx = 0; tx < ntx; ++tx) {
          const int p = (tz * nty + ty) * ntx + tx;
          const int z0 = tz * a.bz, y0 = ty * a.by, x0 = tx * a.bx;
          for (int h = 0; h < HV; ++h) {
            const int hz = h / (HY * HX), rem = h % (HY * HX), hy = rem / HX, hx = rem % HX;
            const int cz = z0 - 1 + hz, cy = y0 - 1 + hy, cx = x0 - 1 + hx;
            if (cz >= 0 && cz < a.Zs && cy >= 0 && cy < a.Ys && cx >= 0 && cx < a.Xs) tH[(size_t)p * HV + h] = (cz * a.Ys + cy) * a.Xs + cx;
          }
          for (int m = 0; m < MR; ++m) {
            int hidx = 0, ov = -1;
            if (m < rows) {
              const int z = m / (a.by * a.bx), rem = m % (a.by * a.bx), y = rem / a.bx, x = rem % a.bx;
              hidx = (z * HY + y) * HX + x;          // halo voxel of tap (0, 0, 0) for this output voxel
              ov = ((z0 + z) * a.Yo + (y0 + y)) * a.Xo + (x0 + x);
            }
            tM[((size_t)p * MR + m) * 2] = hidx;
            tM[((size_t)p * MR + m) * 2 + 1] = ov;
          }
        }
    F16dTabs t;
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

bool conv_f16d_ok(const ConvArgs &a, int mbw) {
  return a.ntaps == 27 && a.stride == 1 && !a.par && !a.ups && a.C0 % 16 == 0 && a.C1 % 16 == 0 && a.Co % 32 == 0 && a.Zs == a.Zo && a.Ys == a.Yo &&
         a.Xs == a.Xo && mbw >= 1 && mbw <= 2 && a.Zo % a.bz == 0 && a.Yo % a.by == 0 && a.Xo % a.bx == 0 && a.bz * a.by * a.bx <= 128 * mbw &&
         (a.bz + 2) * (a.by + 2) * (a.bx + 2) <= 640 && (!a.s2w || (a.s2C0 % 16 == 0 && a.s2C1 % 16 == 0)) && !a.pm;
}

// statistics slots per sample of a launch with this geometry
int conv_f16d_slots(const ConvArgs &a, int mbw) { return (a.Zo / a.bz) * (a.Yo / a.by) * (a.Xo / a.bx) * 4 * mbw; }

hipError_t launch_conv_f16d(const ConvArgs &a_in, int mbw, hipStream_t st) {
  ConvArgs a = a_in;
  a.dbg = conv_dbg_flags();
  if (!conv_f16d_ok(a, mbw)) return hipErrorInvalidValue;
  F16dTabs tb;
  hipError_t et = f16d_tabs_get(a, mbw, &tb);
  if (et != hipSuccess) return et;
  const int nb = a.Co % 64 == 0 ? 2 : 1;
  const size_t lds = (size_t)tb.HV * 12 * sizeof(float);
  const dim3 grid((unsigned)(a.B * tb.ntp), (unsigned)(a.Co / (32 * nb)));
#define CM_F16D_GO(M, N)                                                                            \
  if (mbw == M && nb == N) {                                                                        \
    static bool attr_set[64] = {false};                                                             \
    int dev = 0;                                                                                    \
    (void)hipGetDevice(&dev);                                                                       \
    if (!attr_set[dev & 63]) {                                                                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_f16d_kernel<M, N>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return e;                                                                \
      attr_set[dev & 63] = true;                                                                    \
    }                                                                                               \
    hipLaunchKernelGGL((conv_f16d_kernel<M, N>), grid, dim3(256), lds, st, a, tb.tH, tb.tM, tb.HV, tb.ntp); \
    return hipGetLastError();                                                                       \
  }
  CM_F16D_GO(1, 1) CM_F16D_GO(2, 1) CM_F16D_GO(1, 2) CM_F16D_GO(2, 2)
#undef CM_F16D_GO
  return hipErrorInvalidValue;
}

}  // namespace cm
