#include <hip/hip_runtime.h>
__device__ __forceinline__ float xor16_sum(float x) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float x) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float ror_max(float v) {
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x121, 0xf, 0xf, false)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x122, 0xf, 0xf, false)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x124, 0xf, 0xf, false)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x128, 0xf, 0xf, false)));
  return v;
}
__global__ void k(float* p) {
  float x = p[threadIdx.x];
  p[threadIdx.x] = xor32_sum(xor16_sum(x));
  p[64 + threadIdx.x] = ror_max(x);
}
int main() {
  float h[128]; for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64);
  float* d; hipMalloc(&d, 512); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    float want = 0; for (int k = 0; k < 4; ++k) want += (float)((((i & 15) + 16 * k) * 37) % 64);
    float wm = 0; for (int c = 0; c < 16; ++c) wm = fmaxf(wm, (float)((((i & 48) + c) * 37) % 64));
    if (h[i] != want || h[64 + i] != wm) { ++bad; if (bad < 5) printf("lane %d: sum %g want %g, max %g want %g\n", i, h[i], want, h[64+i], wm); }
  }
  printf("bad=%d\n", bad); return bad != 0;
}
