// Shared pieces of the M-step data-pass kernels (mstep.hip: register-streamed, mstep_dma.hip:
// LDS-DMA ring): the exact bf16x3 operand split and the six-product accumulate.
#pragma once

#include "common.h"

namespace bnpvi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kStatWaves = 4;          // waves per workgroup
constexpr int kMaxChunk = 8192;        // frames summed in fp32 before promotion to fp64
constexpr int kKStep = 16;             // frames per MFMA k-step

typedef _Float16 f16x8m __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2m __attribute__((ext_vector_type(2)));
typedef float f32x2m __attribute__((ext_vector_type(2)));

union Frag {
  uint32_t u[4];
  bf16x8 v;
  f16x8m vh;     // the same 16 bytes as eight f16 (two-piece split, mstep_wide.hip)
};

// ---- two-piece f16 split (mstep_wide.hip, F16 mode): x * scale = hi + lo + O(2^-22 |x * scale|)
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {            // {a (low half), b (high half)}, RNE
  const f16x2m h = __builtin_convertvector((f32x2m){a, b}, f16x2m);
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float f16_lo_as_float(uint32_t p) {
  return (float)__builtin_bit_cast(f16x2m, p)[0];
}
__device__ __forceinline__ float f16_hi_as_float(uint32_t p) {
  return (float)__builtin_bit_cast(f16x2m, p)[1];
}
__device__ __forceinline__ void split2h(const float (&x)[8], float scale, Frag& h, Frag& l) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float s0 = x[2 * q] * scale, s1 = x[2 * q + 1] * scale;
    const uint32_t hp = pack_f16(s0, s1);
    h.u[q] = hp;
    l.u[q] = pack_f16(s0 - f16_lo_as_float(hp), s1 - f16_hi_as_float(hp));
  }
}

// exact 3-way bf16 split of 8 fp32 values: x[e] = h[e] + m[e] + l[e]
__device__ __forceinline__ void split3(const float (&x)[8], Frag& h, Frag& m, Frag& l) {
#ifdef BNPVI_MABLATE_NOSPLIT
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h.u[q] = __float_as_uint(x[2 * q]);
    m.u[q] = __float_as_uint(x[2 * q + 1]);
    l.u[q] = __float_as_uint(x[q]);
  }
  return;
#endif
  uint32_t hb[8], mb[8], lb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint32_t u = __float_as_uint(x[e]);
    hb[e] = u;                                             // top 16 bits are the piece
    const float r1 = x[e] - __uint_as_float(u & 0xffff0000u);
    const uint32_t u1 = __float_as_uint(r1);
    mb[e] = u1;
    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
    lb[e] = __float_as_uint(r2);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // dword q = {elem 2q (low half), elem 2q+1 (high half)}, each its top 16 bits
    h.u[q] = __builtin_amdgcn_perm(hb[2 * q + 1], hb[2 * q], 0x07060302u);
    m.u[q] = __builtin_amdgcn_perm(mb[2 * q + 1], mb[2 * q], 0x07060302u);
    l.u[q] = __builtin_amdgcn_perm(lb[2 * q + 1], lb[2 * q], 0x07060302u);
  }
}

// the same split for one pair of elements -> dword q of the three fragments
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
#ifdef BNPVI_MABLATE_NOSPLIT
  h = __float_as_uint(x0);
  m = __float_as_uint(x1);
  l = __float_as_uint(x0);
  return;
#endif
  const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u);
  const float r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
  const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u);
  const float s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

template <int TL, int TD>
struct RawStep {
  float a[TL][8];     // phi[frame 8h+e][leaf column of tile a]
  float b[TD][8];     // Z  [frame 8h+e][dim  column of tile b]
};

// 6-product accumulate of one (A tile, B tile) pair
__device__ __forceinline__ void mac6(const Frag& ah, const Frag& am, const Frag& al, const Frag& bh,
                                     const Frag& bm, const Frag& bl, f32x16& acc) {
#ifdef BNPVI_MABLATE_NOMFMA
  acc[0] += __uint_as_float((ah.u[0] ^ bl.u[1] ^ am.u[2] ^ bm.u[3] ^ al.u[0] ^ bh.u[1] ^ ah.u[3] ^ bh.u[2] ^
                             am.u[1] ^ bm.u[0] ^ al.u[2] ^ bl.u[3]) & 0x3fffffffu);
  return;
#endif
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bh.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bl.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bm.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bh.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bm.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bh.v, acc, 0, 0, 0);
}

// product k (0..5) of mac6, same order
__device__ __forceinline__ void mac1(int k, const Frag& ah, const Frag& am, const Frag& al, const Frag& bh,
                                     const Frag& bm, const Frag& bl, f32x16& acc) {
#ifdef BNPVI_MABLATE_NOMFMA
  if (k == 0)
    acc[0] += __uint_as_float((ah.u[0] ^ bl.u[1] ^ am.u[2] ^ bm.u[3] ^ al.u[0] ^ bh.u[1] ^ ah.u[3] ^ bh.u[2] ^
                               am.u[1] ^ bm.u[0] ^ al.u[2] ^ bl.u[3]) & 0x3fffffffu);
  return;
#endif
  const Frag& a = (k == 0) ? al : (k == 2 || k == 3) ? am : ah;
  const Frag& b = (k == 0 || k == 3 || k == 5) ? bh : (k == 1) ? bl : bm;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}

// three products of the two-piece f16 split (lo*hi + hi*lo + hi*hi), same order in mac1h
__device__ __forceinline__ void mac3h(const Frag& ah, const Frag& al, const Frag& bh, const Frag& bl, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.vh, bh.vh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.vh, bl.vh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.vh, bh.vh, acc, 0, 0, 0);
}
__device__ __forceinline__ void mac1h(int k, const Frag& ah, const Frag& al, const Frag& bh, const Frag& bl,
                                      f32x16& acc) {
  const Frag& a = (k == 0) ? al : ah;
  const Frag& b = (k == 1) ? bl : bh;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.vh, b.vh, acc, 0, 0, 0);
}

// ---- S0 accumulation (mstep_dma.hip, mstep_wide.hip, mstep.hip)
// History.  Written as `s0 += t` the compiler sinks the update behind the divergent video-boundary
// path of the k-step and fuses the two leaf tiles into one packed add with swapped halves:
//     v_pk_add_f32 v[2:3], v[2:3], v[4:5]            ; last level of the 8-frame tree sum
//     s_cbranch_vccz .Ljoin                          ; TAKEN on the video-boundary path
//   .Ljoin:
//     v_pk_add_f32 v[14:15], v[14:15], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]   ; S0 += t
// In that form S0 of lanes 16-31 missed whole k-steps in about 1 of 1000 launches on MI355X (12 of
// 10000 against 0 of 17500 with the update kept in front of the branch); S1 -- same registers,
// through the MFMAs -- and the masses never did.  What the disassembly of the two builds (hipcc
// -S, ROCm 7.2, with and without -DBNPVI_NO_S0_PIN) rules out: (1) an LDS refill race -- every
// ds_read of a slot is followed by its s_waitcnt lgkmcnt(0) before the loop comes round to the
// global_load_lds that refills the slot, and the counted vmcnt((R-1)*P) covers the pieces in issue
// order, atomics of the boundary path included; (2) a wrong EXEC mask -- every s_and_saveexec of the
// boundary path is matched by its s_or_b64 exec before the join, and the join block runs under the
// loop's full mask; (3) a clobbered operand -- the boundary path writes v82-85, v89-95 only.  What
// differs is only the instruction quoted above: a packed fp32 add with op_sel whose source is the
// result of the packed add two instructions earlier, issued as the first VALU instruction after a
// taken branch.  HYPOTHESIS (round 2, never proven): a forwarding / dependency-check hazard of that
// instruction pair that the compiler does not pad.  Round 3 looked again and found NO support for it:
//   * LLVM's hazard tables for gfx940/950 know no such case (the dst-sel forwarding hazard concerns
//     producers that write a half register -- SDWA dst_sel / VOP3 op_sel[3] -- which v_pk_add_f32
//     does not), and no wait state is required between dependent VALU instructions otherwise;
//   * the present source built with -DBNPVI_NO_S0_PIN shows the very same pair at the join
//     (tools/isa_hazard_scan.py, profiles/r03_isa_hazard_scan.txt) and returned an exact S0 in
//     320 000 of 320 000 launches on two MI355X boxes (tools/dev/s0_ab.py: configs 1, 2, 3), where
//     round 2's rate would have given ~380 failures.
// So the cause of round 2's losses is UNKNOWN (that build differed in more than this pair, and it
// ran on other boxes).  The code below does not depend on the answer:
//   * S0 is updated by ONE scalar-width v_add_f32 per leaf tile, written as inline asm, so it can be
//     neither packed, nor given op_sel, nor moved across the branch;
//   * it sits in straight-line code in front of the boundary path, behind the (also pinned) tree sum.
// tests/test_gpu_round2.py::test_s0_is_exact_on_one_hot_responsibilities checks S0 bit for bit.
__device__ __forceinline__ void s0_accumulate(float& s0, float t) {
#ifdef BNPVI_NO_S0_PIN  // development only: the plain form, for tools/isa_hazard_scan.py to look at
  s0 += t;
#else
  asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(t));
#endif
}

// Debug builds only (-DBNPVI_S0_CROSSCHECK, `make variant NAME=s0chk DEFS=-DBNPVI_S0_CROSSCHECK SRCS="mstep
// mstep_dma mstep_wide"`): every lane carries a second S0 in fp64 -- double adds are never packed and take
// no op_sel -- and compares it with the fp32 sum when the kernel's frame loop ends.  fp32 accumulation
// differs from it by ~1e-7 * sqrt(k-steps) relative; a lost k-step by ~1 / k-steps.  A mismatch is
// printed (kernel, workgroup, lane, both sums), so ONE run of any test or benchmark against the debug
// library shows a regression of the S0 path whatever the data.
struct S0Check {
#ifdef BNPVI_S0_CROSSCHECK
  double sum = 0.0;
  __device__ __forceinline__ void add(float t) { sum += (double)t; }
  __device__ __forceinline__ void verify(float s0, const char* kernel, int a) const {
    const double tol = 1e-4 * (sum < 0 ? -sum : sum) + 1e-6;
    const double d = (double)s0 - sum;
    if (d > tol || -d > tol)
      printf("bnpvi S0 cross-check FAILED: %s wg (%d,%d) thread %d tile %d: fp32 %.9g fp64 %.9g\n", kernel,
             (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, a, (double)s0, sum);
  }
#else
  __device__ __forceinline__ void add(float) {}
  __device__ __forceinline__ void verify(float, const char*, int) const {}
#endif
};

// ---- LDS-DMA helpers (mstep_dma.hip, mstep_wide.hip)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void dma16(const void* g, float* l) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}
// the same with the non-temporal cache policy (aux bit 1): for bytes this kernel reads once and
// nothing reads again before they would be evicted anyway
__device__ __forceinline__ void dma16_nt(const void* g, float* l) {
#ifdef BNPVI_PHI_LOAD_PLAIN
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
#else
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 2);
#endif
}
__device__ __forceinline__ void dma4(const void* g, float* l) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 4, 0, 0);
}

// s_waitcnt vmcnt(n) only (expcnt / lgkmcnt left alone): gfx9 encoding
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// wide-tile variant for large trees (mstep_wide.hip); BNPVI_EUNSUPPORTED for shapes it does not cover
int launch_leaf_stats_wide(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                           const int32_t* vid, int64_t N, int D, int L, int V, double* stats, double* mass,
                           hipStream_t s);

// LDS-DMA variant (mstep_dma.hip); returns BNPVI_EUNSUPPORTED for shapes it does not take
int launch_leaf_stats_dma(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                          const int32_t* vid, int64_t N, int D, int L, int V, double* stats, double* mass,
                          hipStream_t s);

}  // namespace bnpvi
