// Developer probe: issue rate of the float min / max flavours on gfx950 — v_add_f32, v_max_f32 (IEEE maxNum: ignores
// NaN), v_max3_f32, v_maximum_f32 / v_maximum3_f32 (IEEE-754-2019 maximum: propagates NaN; what the reducers use).
// 8 independent accumulators per lane, 4 waves per SIMD, 2 M ops per lane; reports cycles per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip && ./valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = seed + (float)(threadIdx.x + k);
  float x = seed * 0.5f, y = seed * 0.25f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (KIND == 0) a[k] = a[k] + x;
      else if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
      else if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
      else if (KIND == 3) asm volatile("v_maximum3_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(x));
      else if (KIND == 4) asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
      else if (KIND == 5) asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
      else if (KIND == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
      else if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
    }
    x += 1e-9f;
  }
  float s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, float* out) {
  const int iters = 1 << 16, blocks = 256 * 4;        // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, 16, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves x iters x 8 instructions
  const double instr_per_simd = 4.0 * iters * 8;
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);     // kHz
  printf("%-34s %8.3f ms  %.2f cycles per wave-instruction (at %d MHz)\n", name, ms, ms * 1e-3 * clk * 1e3 / instr_per_simd, clk / 1000);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
  run<0>("v_add_f32", out);
  run<1>("v_max_f32 (maxNum)", out);
  run<2>("v_max3_f32", out);
  run<3>("v_maximum3_f32 a, a, x, x", out);
  run<4>("v_maximum3_f32 a, a, x, y", out);
  run<5>("v_minimum3_f32 a, a, x, y", out);
  run<6>("v_exp_f32", out);
  run<7>("v_fma_f32", out);
  return 0;
}
