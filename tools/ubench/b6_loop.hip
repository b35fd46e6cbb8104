// Microbenchmark for the direct six-term conv main loop: per tap a wave issues 3 weight-fragment loads (16 B per lane,
// global, every wave of the chip reads the same 81 KB per chunk -> L1 / L2 hits), MBW x 3 ds_read_b128 (the hi / mid / lo
// planes of its rows) and MBW x 6 v_mfma_f32_32x32x16_bf16.  Question: at which (MBW, waves per SIMD) is the matrix pipe
// the bound rather than the L1 / LDS paths?   usage: b6_loop [lds_kb_per_wg]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MBW, int RD, int GLD, bool LDS>
__global__ __launch_bounds__(256) void k(const f32x4 *__restrict__ w, int nchunk, float *out, int S) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  for (int i = tid; i < 2048; i += 256) lds[i] = (float)i * 1e-9f;
  __syncthreads();
  f32x16 acc[MBW];
#pragma unroll
  for (int j = 0; j < MBW; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const f32x4 *wl = w + lane;
  f32x4 bw[RD][3];
  const float *abase = lds + r * S + 4 * hh;
  for (int c = 0; c < nchunk; ++c) {
    const f32x4 *wc = wl + (size_t)(c & 1) * 27 * 3 * 64;
#pragma unroll
    for (int t = 0; t < RD; ++t)
#pragma unroll
      for (int tm = 0; tm < 3; ++tm) bw[t][tm] = GLD == 1 ? wc[(t * 3 + tm) * 64] : (GLD == 2 ? *reinterpret_cast<const f32x4 *>(lds + 4096 + ((t * 3 + tm) * 64 + lane) * 4) : f32x4{1.f, 2.f, 3.f, 4.f});
    f32x4 af[MBW][3], afn[MBW][3];
#pragma unroll
    for (int j = 0; j < MBW; ++j)
#pragma unroll
      for (int tm = 0; tm < 3; ++tm) af[j][tm] = *reinterpret_cast<const f32x4 *>(abase + j * 32 * S + 8 * tm);
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      if (t + 1 < 27) {
        const int toff = ((t + 1) % 9) * S * 2;
#pragma unroll
        for (int j = 0; j < MBW; ++j)
#pragma unroll
          for (int tm = 0; tm < 3; ++tm)
            afn[j][tm] = LDS ? *reinterpret_cast<const f32x4 *>(abase + j * 32 * S + toff + 8 * tm) : af[j][tm];
      }
      constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int j = 0; j < MBW; ++j)
#pragma unroll
        for (int u = 0; u < 6; ++u)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[j][TA[u]]), __builtin_bit_cast(bf16x8, bw[t % RD][TB[u]]), acc[j], 0, 0, 0);
      if (GLD && t + RD < 27) {
#pragma unroll
        for (int tm = 0; tm < 3; ++tm)
          bw[t % RD][tm] = GLD == 1 ? wc[((t + RD) * 3 + tm) * 64] : *reinterpret_cast<const f32x4 *>(lds + 4096 + (((t + RD) % 9 * 3 + tm) * 64 + lane) * 4);
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < MBW; ++j)
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) af[j][tm] = afn[j][tm];
    }
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < MBW; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[j][e];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MBW, int RD, int GLD, bool LDS>
static void run(const char *name, const f32x4 *w, float *out, int lds_kb, int S) {
  const int nchunk = 64, grid = 256 * 8;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<MBW, RD, GLD, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MBW, RD, GLD, LDS>), dim3(grid), dim3(256), lds_kb * 1024, 0, w, nchunk, out, S);
  hipEventRecord(e0);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MBW, RD, GLD, LDS>), dim3(grid), dim3(256), lds_kb * 1024, 0, w, nchunk, out, S);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  const double flops = (double)grid * 4 * nchunk * 27 * MBW * 6 * 32768.0;
  printf("%-28s lds %3d KB/WG S %2d: %8.3f ms  %7.1f TFLOP/s bf16  (%.2f of 2500)\n", name, lds_kb, S, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 2500.0);
}

int main(int argc, char **argv) {
  f32x4 *w;
  float *out;
  hipMalloc(&w, 2 * 27 * 3 * 64 * sizeof(f32x4));
  hipMemset(w, 0x3c, 2 * 27 * 3 * 64 * sizeof(f32x4));
  hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int lds_kb : {50, 76}) {        // 3, 2 workgroups per CU by LDS
    run<1, 4, 1, true>("MBW1 RD4 gld+lds", w, out, lds_kb, 28);
    run<1, 8, 1, true>("MBW1 RD8 gld+lds", w, out, lds_kb, 28);
    run<2, 4, 1, true>("MBW2 RD4 gld+lds", w, out, lds_kb, 28);
    run<2, 6, 1, true>("MBW2 RD6 gld+lds", w, out, lds_kb, 28);
    run<2, 8, 1, true>("MBW2 RD8 gld+lds", w, out, lds_kb, 28);
    run<3, 3, 1, true>("MBW3 RD3 gld+lds", w, out, lds_kb, 28);
    run<3, 5, 1, true>("MBW3 RD5 gld+lds", w, out, lds_kb, 28);
    run<4, 2, 1, true>("MBW4 RD2 gld+lds", w, out, lds_kb, 28);
    run<4, 3, 1, true>("MBW4 RD3 gld+lds", w, out, lds_kb, 28);
    run<1, 1, 2, true>("MBW1 RD1 B from LDS + A lds", w, out, lds_kb, 28);
    run<1, 2, 2, true>("MBW1 RD2 B from LDS + A lds", w, out, lds_kb, 28);
    run<2, 2, 2, true>("MBW2 RD2 B from LDS + A lds", w, out, lds_kb, 28);
  }
  return 0;
}
