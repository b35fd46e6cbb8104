// Microbenchmark: independent VALU / LDS / global-load work interleaved with MFMAs in the SAME wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(int nm, int domfma, float *out, const float *in) {
  __shared__ float sh[4096];
  sh[threadIdx.x] = threadIdx.x;
  __syncthreads();
  f32x16 a0, a1, a2, a3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0; a1[i] = 1; a2[i] = 2; a3[i] = 3; }
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
  int iv = threadIdx.x;
  for (int i = 0; i < nm; ++i) {
    if (domfma) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      if (KIND == 0) v[u & 7] = fmaf(v[u & 7], 1.0001f, 0.5f);
      else if (KIND == 1) v[u & 7] += sh[(iv + u * 33 + i) & 4095];
      else if (KIND == 2) v[u & 7] = __expf(v[u & 7]) * 0.5f;
      else v[u & 7] += in[(size_t)((iv * 64 + u * 4099 + i * 131) & 0xFFFFF)];
    }
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + s;
}

template <int NV, int KIND>
void run(float *out, const float *in, const char *name) {
  const int nm = 2000;
  float t[2];
  for (int dm = 0; dm < 2; ++dm) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256), 0, 0, nm, dm, out, in);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256), 0, 0, nm, dm, out, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&t[dm], e0, e1);
    t[dm] /= 5;
  }
  printf("%-14s %2d ops per 4 MFMAs: ops alone %.1f us | with MFMAs %.1f us (MFMA alone 213 us)\n", name, NV, t[0] * 1e3, t[1] * 1e3);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&in, (1 << 20) * 4 + 4096);
  hipMemset(in, 0, (1 << 20) * 4 + 4096);
  run<0, 0>(out, in, "nothing");
  run<8, 0>(out, in, "fma");
  run<16, 0>(out, in, "fma");
  run<32, 0>(out, in, "fma");
  run<64, 0>(out, in, "fma");
  run<4, 1>(out, in, "lds read");
  run<8, 1>(out, in, "lds read");
  run<4, 2>(out, in, "exp");
  run<8, 2>(out, in, "exp");
  run<2, 3>(out, in, "global load");
  run<4, 3>(out, in, "global load");
  return 0;
}
