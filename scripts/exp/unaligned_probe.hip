#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) U16 { u32x4 v; };
// rows of RB bytes (multiple of 4), copy src row perm[r] -> dst row r
template <int MODE>
__global__ __launch_bounds__(256) void copy_rows(const char* __restrict__ src, char* __restrict__ dst, const int* __restrict__ perm,
                                                 long n_rows, int rb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * 16;
  for (int k = wave; k < 16; k += 4) {
    const long r = r0 + k;
    if (r >= n_rows) return;
    const char* s = src + (long)perm[r] * rb;
    char* d = dst + r * rb;
    if (MODE == 0) {            // 8-byte lanes
      for (int c = lane; c * 8 < rb; c += 64) *(u32x2*)(d + c * 8) = __builtin_nontemporal_load((const u32x2*)(s + c * 8));
    } else {                    // 16-byte lanes at 8-byte aligned addresses + 8-byte tail
      const int full = rb / 16;
      for (int c = lane; c < full; c += 64) {
        U16 v = *(const U16*)(s + c * 16);
        *(U16*)(d + c * 16) = v;
      }
      if (lane == 63 && (rb & 8)) *(u32x2*)(d + full * 16) = *(const u32x2*)(s + full * 16);
    }
  }
}
int main() {
  const int rb = 1000; const long n = 4 << 20;   // 4 GB
  char *src, *dst; int* perm;
  hipMalloc(&src, n * rb); hipMalloc(&dst, n * rb); hipMalloc(&perm, n * 4);
  std::vector<int> p(n); for (long i = 0; i < n; ++i) p[i] = (int)((i * 2654435761ull) % n);
  hipMemcpy(perm, p.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(src, 1, n * rb);
  std::vector<unsigned char> h(n * rb > (1 << 24) ? (1 << 24) : n * rb);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)(i * 7 + (i >> 8));
  hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(e0);
      if (mode == 0) copy_rows<0><<<(n + 15) / 16, 256>>>(src, dst, perm, n, rb);
      else copy_rows<1><<<(n + 15) / 16, 256>>>(src, dst, perm, n, rb);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d: %.3f ms  %.2f TB/s (err %d)\n", mode, ms, 2.0 * n * rb / ms / 1e9, (int)hipGetLastError());
    }
    // verify first rows
    std::vector<unsigned char> o(16 * rb); hipMemcpy(o.data(), dst, o.size(), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 16; ++r) {
      std::vector<unsigned char> s(rb); hipMemcpy(s.data(), src + (long)p[r] * rb, rb, hipMemcpyDeviceToHost);
      for (int i = 0; i < rb; ++i) bad += s[i] != o[r * rb + i];
    }
    printf("mode %d mismatches %d\n", mode, bad);
    hipMemset(dst, 0, n * rb);
  }
}
