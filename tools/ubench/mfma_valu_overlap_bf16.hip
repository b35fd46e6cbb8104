// Microbenchmark: can one wave's VALU work overlap another wave's MFMA stream on the same SIMD?
// grid = 256 CUs x k workgroups; workgroup = 8 waves: waves 0-3 (one per SIMD) run MFMAs, waves 4-7 run VALU FMAs
// (or transcendental / LDS / integer work).  Modes: 1 = MFMA waves only, 2 = VALU waves only, 3 = both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int mode, int nm, int nv, int kind, float *out, int prio) {
  const int wave = threadIdx.x >> 6;
  if (prio == 1) { if (wave >= 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
  if (prio == 2) { if (wave >= 4) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3); }
  __shared__ float sh[4096];
  sh[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 a0, a1, a2, a3;
    for (int i = 0; i < 16; ++i) { a0[i] = 0; a1[i] = 1; a2[i] = 2; a3[i] = 3; }
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8))); bf16x8 xb, yb; for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)(threadIdx.x * 1e-3f + i); yb[i] = (__bf16)(1.0f + i); }
    for (int i = 0; i < nm; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    if (!(mode & 2)) return;
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    int iv = threadIdx.x;
    if (kind == 0) {
      for (int i = 0; i < nv; ++i) {  // 4 independent FMA chains
        v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
      }
    } else if (kind == 1) {
      for (int i = 0; i < nv; ++i) {  // integer address-style math
        iv = iv * 36 + (iv >> 3); iv ^= i; v0 += (float)(iv & 7);
      }
    } else if (kind == 2) {
      for (int i = 0; i < nv; ++i) {  // LDS reads
        v0 += sh[(iv + i * 33) & 4095]; v1 += sh[(iv + i * 65) & 4095];
      }
    } else {
      for (int i = 0; i < nv; ++i) { v0 = __expf(v0) * 0.5f; v1 = __builtin_amdgcn_rcpf(v1 + 2.f); }
    }
    out[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3 + iv;
  }
}

int main(int argc, char **argv) {
  float *out;
  hipMalloc(&out, 4096 * 512 * 4);
  const int nm = 2000, grid = 256;
  for (int prio = 0; prio < 3; ++prio)
  for (int kind = 0; kind < 4; ++kind) {
    const int nv = kind == 0 ? 8000 : (kind == 1 ? 8000 : (kind == 2 ? 8000 : 2000));
    float t[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, mode, nm, nv, kind, out, prio);
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, mode, nm, nv, kind, out, prio);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&t[mode], e0, e1);
      t[mode] /= 5;
    }
    printf("prio %d (1: VALU waves high, 2: MFMA waves high) kind %d (0 fma, 1 int, 2 lds, 3 transcendental): MFMA only %.1f us | VALU only %.1f us | both %.1f us (sum %.1f, max %.1f)\n",
           prio, kind, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, (t[1] + t[2]) * 1e3, (t[1] > t[2] ? t[1] : t[2]) * 1e3);
  }
  printf("MFMA-only expectation: %d x 4 MFMAs x 32 cycles / 2.4 GHz = %.1f us\n", nm, nm * 4 * 32 / 2400.0);
  return 0;
}
