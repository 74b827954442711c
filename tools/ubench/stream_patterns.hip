// Development micro-benchmark (not part of the library): HBM streaming rate on MI355X for the
// two per-instruction footprints the VI kernels can have on a [frames][64] fp32 matrix:
//   pattern 0: 1 KiB contiguous per wave-instruction (lane i -> 16 B at base + 16 i)
//   pattern 1: 16 rows x 64 B per wave-instruction (lane (j = i & 15, kg = i >> 4) -> row j,
//              bytes 16 kg of a 64-B piece) -- the MFMA 16x16x32 B-operand footprint
//   pattern 2: 8 rows x 128 B per wave-instruction (whole 128-B lines)
// modes: 0 read + reduce, 1 write, 2 copy (read A, write B: the E-step's traffic shape),
//        3 copy with nontemporal stores, 4 copy with nontemporal loads and stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE> __device__ __forceinline__ float4 ld(const float4* p) {
  if (MODE == 4) { f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
  return *p;
}
template <int MODE> __device__ __forceinline__ void st(float4* p, float4 v) {
  if (MODE >= 3) { f4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<f4*>(p)); }
  else *p = v;
}
template <int PAT, int MODE, int PF>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ in, float4* __restrict__ out, long ntiles, float* sink) {
  // a tile = 16 rows x 256 B = 4 KiB = 256 float4; 4 instructions per wave per tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long stride = (long)gridDim.x * 4;
  float acc = 0.f;
  long t = (long)blockIdx.x * 4 + wave;
  float4 buf[PF][4];
  auto idx = [&](long tile, int i) -> long {
    if (PAT == 0) return tile * 256 + i * 64 + lane;
    if (PAT == 2) {                              // 8 rows x 128 B per instruction (whole cache lines)
      const int j8 = lane & 7, h = lane >> 3;
      return tile * 256 + ((i & 1) * 8 + j8) * 16 + (i >> 1) * 8 + h;
    }
    const int j = lane & 15, kg = lane >> 4;
    return tile * 256 + j * 16 + i * 4 + kg;     // row j (16 float4 per row), piece i, 16 B kg
  };
  if (MODE != 1) {
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (t + p * stride < ntiles)
#pragma unroll
        for (int i = 0; i < 4; ++i) buf[p][i] = ld<MODE>(in + idx(t + p * stride, i));
  }
  for (; t < ntiles; t += PF * stride) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const long tt = t + p * stride;
      if (tt >= ntiles) break;
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MODE != 1) v[i] = buf[p][i]; else v[i] = make_float4(lane, i, tt, 1.f);
      }
      if (MODE != 1 && tt + PF * stride < ntiles) {
#pragma unroll
        for (int i = 0; i < 4; ++i) buf[p][i] = ld<MODE>(in + idx(tt + PF * stride, i));
      }
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) st<MODE>(out + idx(tt, i), v[i]);
      }
    }
  }
  if (MODE == 0 && acc == 12345.678f) *sink = acc;
}

// Two matrices read together (the M-step's shape: phi and Z, 4 KiB of each per 16 frames).
// ORDER 0: tiles dealt round-robin over all waves (globally sequential sweep);
// ORDER 1: every wave owns one contiguous range of tiles (as many concurrent streams as waves);
// ORDER 2: round-robin in blocks of 4 consecutive tiles.
template <int ORDER, int PF>
__global__ __launch_bounds__(256) void k2(const float4* __restrict__ a, const float4* __restrict__ b, long ntiles,
                                           float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nw = (long)gridDim.x * 4, w = (long)blockIdx.x * 4 + wave;
  const long per = (ntiles + nw - 1) / nw;                  // tiles per wave
  auto tile_of = [&](long j) -> long {                      // j-th tile of this wave, or -1
    if (j >= per) return -1;
    long t;
    if (ORDER == 0) t = j * nw + w;
    else if (ORDER == 1) t = w * per + j;
    else t = ((j >> 2) * nw + w) * 4 + (j & 3);
    return t < ntiles ? t : -1;
  };
  float acc = 0.f;
  float4 ba[PF][4], bb[PF][4];
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    const long t = tile_of(p);
    if (t >= 0)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ba[p][i] = a[t * 256 + i * 64 + lane];
        bb[p][i] = b[t * 256 + i * 64 + lane];
      }
  }
  for (long j = 0; j < per; j += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      if (tile_of(j + p) < 0) break;
      float4 va[4], vb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        va[i] = ba[p][i];
        vb[i] = bb[p][i];
      }
      const long tn = tile_of(j + p + PF);
      if (tn >= 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ba[p][i] = a[tn * 256 + i * 64 + lane];
          bb[p][i] = b[tn * 256 + i * 64 + lane];
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += va[i].x * vb[i].x + va[i].y * vb[i].y + va[i].z * vb[i].z + va[i].w * vb[i].w;
    }
  }
  if (acc == 12345.678f) *sink = acc;
}

template <int ORDER, int PF>
void run2(const char* name, float4* a, float4* b, long ntiles, float* sink, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k2<ORDER, PF>), dim3(blocks), dim3(256), 0, 0, a, b, ntiles, sink);
  CK(hipEventRecord(e0));
  const int R = 10;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL((k2<ORDER, PF>), dim3(blocks), dim3(256), 0, 0, a, b, ntiles, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= R;
  const double bytes = (double)ntiles * 4096 * 2;
  printf("%-28s blocks=%5d  %8.1f us  %7.2f TB/s\n", name, blocks, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

template <int PAT, int MODE, int PF>
void run(const char* name, float4* a, float4* b, long ntiles, float* sink, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PAT, MODE, PF>), dim3(blocks), dim3(256), 0, 0, a, b, ntiles, sink);
  CK(hipEventRecord(e0));
  const int R = 10;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL((k<PAT, MODE, PF>), dim3(blocks), dim3(256), 0, 0, a, b, ntiles, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= R;
  const double bytes = (double)ntiles * 4096 * (MODE >= 2 ? 2 : 1);
  printf("%-28s blocks=%5d  %8.1f us  %7.2f TB/s\n", name, blocks, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const long ntiles = argc > 1 ? atol(argv[1]) : 65536;          // 65536 tiles = 256 MiB per matrix
  float4 *a, *b; float* sink;
  CK(hipMalloc(&a, ntiles * 4096)); CK(hipMalloc(&b, ntiles * 4096)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, ntiles * 4096)); CK(hipMemset(b, 0, ntiles * 4096));
  for (int per_cu : {2, 4}) {
    const int blocks = 256 * per_cu;
    printf("--- two matrices, %d workgroups (x4 waves) per CU\n", per_cu);
    run2<0, 1>("read2 round-robin pf1", a, b, ntiles, sink, blocks);
    run2<1, 1>("read2 chunk/wave  pf1", a, b, ntiles, sink, blocks);
    run2<2, 1>("read2 rr blocks4  pf1", a, b, ntiles, sink, blocks);
    run2<0, 2>("read2 round-robin pf2", a, b, ntiles, sink, blocks);
    run2<1, 2>("read2 chunk/wave  pf2", a, b, ntiles, sink, blocks);
    run2<2, 2>("read2 rr blocks4  pf2", a, b, ntiles, sink, blocks);
  }
  if (argc > 2) return 0;
  for (int per_cu : {2, 4, 8}) {
    const int blocks = 256 * per_cu;
    printf("--- %d workgroups (x4 waves) per CU\n", per_cu);
    run<0, 0, 1>("read  contiguous pf1", a, b, ntiles, sink, blocks);
    run<1, 0, 1>("read  16x64B     pf1", a, b, ntiles, sink, blocks);
    run<0, 0, 2>("read  contiguous pf2", a, b, ntiles, sink, blocks);
    run<1, 0, 2>("read  16x64B     pf2", a, b, ntiles, sink, blocks);
    run<0, 1, 1>("write contiguous", a, b, ntiles, sink, blocks);
    run<1, 1, 1>("write 16x64B", a, b, ntiles, sink, blocks);
    run<2, 1, 1>("write 8x128B", a, b, ntiles, sink, blocks);
    run<0, 2, 1>("copy  contiguous pf1", a, b, ntiles, sink, blocks);
    run<1, 2, 1>("copy  16x64B     pf1", a, b, ntiles, sink, blocks);
    run<0, 3, 1>("copy  contiguous nt-st", a, b, ntiles, sink, blocks);
    run<1, 3, 1>("copy  16x64B     nt-st", a, b, ntiles, sink, blocks);
    run<2, 3, 1>("copy  8x128B     nt-st", a, b, ntiles, sink, blocks);
    run<2, 2, 1>("copy  8x128B     pf1", a, b, ntiles, sink, blocks);
    run<0, 4, 1>("copy  contiguous nt-ld+st", a, b, ntiles, sink, blocks);
    run<1, 4, 1>("copy  16x64B     nt-ld+st", a, b, ntiles, sink, blocks);
  }
  return 0;
}
