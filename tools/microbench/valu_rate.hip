// valu_rate.hip -- what ONE wave64 VALU instruction costs a gfx950 SIMD, per instruction class and per resident-wave count.
//
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate && tools/microbench/valu_rate
//
// Why ("what's weak" 2): bench.py charged 4 clk per VALU instruction, MI355X_MICROARCH.md derives 2 clk from the f32
// peak, profiles/r03_microbench5.txt measured 3.09 (one configuration: dependent pairs, 4 waves per SIMD).  This file measures
// INDEPENDENT streams (8 accumulators per lane: distance 8 between dependent instructions) at 1, 2, 4 and 8 waves per SIMD and, for
// comparison, a fully dependent chain at 1 wave per SIMD (latency).
//
// Method: one block of 256 * W threads per CU-sized slot (W <= 4; W = 8: two blocks of 1024), 256 (512) blocks = one per CU;
// every wave brackets its loop with s_memtime (shader clocks, MI355X_MICROARCH.md: "tick = shader cycle") and reports the SIMD it
// ran on (HW_REG_HW_ID / XCC_ID), so the host can (a) check that every SIMD really held W waves and (b) compute
//     clk per wave-instruction per SIMD = mean elapsed clocks of the waves / (W * instructions per wave)
// with no assumption about the clock rate.  The wall-clock number (hipEvents) is printed next to it with the s_memtime /
// s_memrealtime ratio of the same launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 16;      // x 8 accumulators = 128 VALU instructions per loop iteration (+ 3 SALU)
constexpr int kAcc = 8;

struct WaveRec { unsigned long long clk, ref; unsigned hw, xcc; };

__device__ __forceinline__ unsigned hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v; }

// T(A): the text of one instruction on accumulator register A (read-modify-write); %8, %9 = operands b, c (never written).
// All 128 instructions of a loop iteration sit in ONE asm statement (between separate asm statements the compiler's hazard
// recogniser inserts an s_nop, which would cost an issue cycle per statement).
#define R16(X) X X X X X X X X X X X X X X X X
#define BODY8(T) asm volatile(R16(T("%0") T("%1") T("%2") T("%3") T("%4") T("%5") T("%6") T("%7"))                     \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)          \
                              : "v"(b), "v"(c) : "vcc", "s20", "s21");
#define BODY1(T) asm volatile(R16(T("%0") T("%0") T("%0") T("%0") T("%0") T("%0") T("%0") T("%0"))                     \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)          \
                              : "v"(b), "v"(c) : "vcc", "s20", "s21");

#define DEFINE_KERNEL2(NAME, TYPE, INIT, TYPEB, INITB, OP)                                                                             \
  template <bool DEP>                                                                                                    \
  __global__ __launch_bounds__(1024) void k_##NAME(WaveRec* __restrict__ out, int iters, float seed) {                   \
    TYPE a0 = INIT(0), a1 = INIT(1), a2 = INIT(2), a3 = INIT(3), a4 = INIT(4), a5 = INIT(5), a6 = INIT(6), a7 = INIT(7); \
    TYPEB b = INITB(9), c = INITB(11);                                                                                   \
    __syncthreads();                                                                                                     \
    asm volatile("s_mov_b64 vcc, 0x5555\ns_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");   /* (masks are initialised) */ \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                  \
    for (int it = 0; it < iters; ++it) {                                                                                 \
      if (DEP) { BODY1(OP) } else { BODY8(OP) }                                                                          \
    }                                                                                                                    \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                  \
    asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));                         \
    if ((threadIdx.x & 63) == 0) {                                                                                       \
      WaveRec r;                                                                                                         \
      r.clk = t1 - t0; r.ref = r1 - r0; r.hw = hw_id(); r.xcc = xcc_id();                                                \
      out[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;                                                \
    }                                                                                                                    \
  }

#define DEFINE_KERNEL(NAME, TYPE, INIT, OP) DEFINE_KERNEL2(NAME, TYPE, INIT, TYPE, INIT, OP)
#define INIT_F(i) (seed + 0.001f * (float)(threadIdx.x + 64 * (i)))
#define INIT_D(i) ((double)seed + 0.001 * (double)(threadIdx.x + 64 * (i)))
#define INIT_U(i) ((unsigned)(threadIdx.x * 2654435761u + (i)) | 1u)
typedef float v2f __attribute__((ext_vector_type(2)));
#define INIT_V2(i) (v2f{seed + 0.001f * (float)(threadIdx.x + (i)), seed + 0.5f})
#define INIT_U64(i) ((unsigned long long)(threadIdx.x * 2654435761u + (i)) | 1ull)

#define OP_FMA_F32(A) "v_fma_f32 " A ", " A ", %8, %9\n"
#define OP_MUL_F32(A) "v_mul_f32 " A ", " A ", %8\n"
#define OP_ADD_F32(A) "v_add_f32 " A ", " A ", %8\n"
#define OP_MAC_F32(A) "v_fmac_f32 " A ", %8, %9\n"
#define OP_PK_FMA_F32(A) "v_pk_fma_f32 " A ", " A ", %8, %9\n"
#define OP_PK_MUL_F32(A) "v_pk_mul_f32 " A ", " A ", %8\n"
#define OP_MUL_F64(A) "v_mul_f64 " A ", " A ", %8\n"
#define OP_FMA_F64(A) "v_fma_f64 " A ", " A ", %8, %9\n"
#define OP_ADD_F64(A) "v_add_f64 " A ", " A ", %8\n"
#define OP_EXP_F32(A) "v_exp_f32 " A ", " A "\n"
#define OP_LOG_F32(A) "v_log_f32 " A ", " A "\n"
#define OP_RCP_F32(A) "v_rcp_f32 " A ", " A "\n"
#define OP_SQRT_F32(A) "v_sqrt_f32 " A ", " A "\n"
#define OP_RCP_F64(A) "v_rcp_f64 " A ", " A "\n"
#define OP_MAD_U24(A) "v_mad_u32_u24 " A ", " A ", %8, %9\n"
#define OP_ADD_U32(A) "v_add_u32 " A ", " A ", %8\n"
#define OP_XOR_B32(A) "v_xor_b32 " A ", " A ", %8\n"
#define OP_LSHL_ADD(A) "v_lshl_add_u32 " A ", " A ", 1, %8\n"
#define OP_MUL_LO_U32(A) "v_mul_lo_u32 " A ", " A ", %8\n"
#define OP_MAD_U64_U32(A) "v_mad_u64_u32 " A ", vcc, %8, %9, " A "\n"
#define OP_CNDMASK(A) "v_cndmask_b32 " A ", " A ", %8, vcc\n"
#define OP_MOV_DPP(A) "v_mov_b32_dpp " A ", " A " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_ADD_DPP(A) "v_add_f32_dpp " A ", " A ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_FLOOR_F32(A) "v_floor_f32 " A ", " A "\n"
#define OP_CVT_I32_F32(A) "v_cvt_i32_f32 " A ", " A "\n"
#define OP_MAX_F32(A) "v_max_f32 " A ", " A ", %8\n"
#define OP_CMP_F32(A) "v_cmp_lt_f32 vcc, " A ", %8\n"
#define OP_READLANE(A) "v_readlane_b32 s20, " A ", 3\n"
#define OP_READFIRST(A) "v_readfirstlane_b32 s20, " A "\n"
#define OP_CVT_F64_F32_PAIR(A) "v_cvt_f64_f32 " A ", %8\n"     /* (64-bit accumulator written from an f32 source) */
#define OP_CVT_F32_F64_PAIR(A) "v_cvt_f32_f64 " A ", %8\n"     /* (32-bit accumulator written from an f64 source) */

DEFINE_KERNEL(fma_f32, float, INIT_F, OP_FMA_F32)
DEFINE_KERNEL(mul_f32, float, INIT_F, OP_MUL_F32)
DEFINE_KERNEL(add_f32, float, INIT_F, OP_ADD_F32)
DEFINE_KERNEL(fmac_f32, float, INIT_F, OP_MAC_F32)
DEFINE_KERNEL(pk_fma_f32, v2f, INIT_V2, OP_PK_FMA_F32)
DEFINE_KERNEL(pk_mul_f32, v2f, INIT_V2, OP_PK_MUL_F32)
DEFINE_KERNEL(mul_f64, double, INIT_D, OP_MUL_F64)
DEFINE_KERNEL(fma_f64, double, INIT_D, OP_FMA_F64)
DEFINE_KERNEL(add_f64, double, INIT_D, OP_ADD_F64)
DEFINE_KERNEL(exp_f32, float, INIT_F, OP_EXP_F32)
DEFINE_KERNEL(log_f32, float, INIT_F, OP_LOG_F32)
DEFINE_KERNEL(rcp_f32, float, INIT_F, OP_RCP_F32)
DEFINE_KERNEL(sqrt_f32, float, INIT_F, OP_SQRT_F32)
DEFINE_KERNEL(rcp_f64, double, INIT_D, OP_RCP_F64)
DEFINE_KERNEL(mad_u32_u24, unsigned, INIT_U, OP_MAD_U24)
DEFINE_KERNEL(add_u32, unsigned, INIT_U, OP_ADD_U32)
DEFINE_KERNEL(xor_b32, unsigned, INIT_U, OP_XOR_B32)
DEFINE_KERNEL(lshl_add_u32, unsigned, INIT_U, OP_LSHL_ADD)
DEFINE_KERNEL(mul_lo_u32, unsigned, INIT_U, OP_MUL_LO_U32)
DEFINE_KERNEL2(mad_u64_u32, unsigned long long, INIT_U64, unsigned, INIT_U, OP_MAD_U64_U32)
DEFINE_KERNEL(cndmask_b32, unsigned, INIT_U, OP_CNDMASK)
DEFINE_KERNEL(mov_dpp, unsigned, INIT_U, OP_MOV_DPP)
DEFINE_KERNEL(add_f32_dpp, float, INIT_F, OP_ADD_DPP)
DEFINE_KERNEL(floor_f32, float, INIT_F, OP_FLOOR_F32)
DEFINE_KERNEL(cvt_i32_f32, float, INIT_F, OP_CVT_I32_F32)
DEFINE_KERNEL(max_f32, float, INIT_F, OP_MAX_F32)
DEFINE_KERNEL(cmp_lt_f32, float, INIT_F, OP_CMP_F32)
DEFINE_KERNEL(readlane, unsigned, INIT_U, OP_READLANE)
DEFINE_KERNEL(readfirstlane, unsigned, INIT_U, OP_READFIRST)

// r06 second batch: which opcodes run at the double rate?  (v_cndmask_b32 with an UNINITIALISED vcc measured 22.8 clk in the first run:
// the variants below separate the mask source from the instruction)
#define OPN_and_b32(A) "v_and_b32 " A ", " A ", %8\n"
DEFINE_KERNEL(and_b32, unsigned, INIT_U, OPN_and_b32)
#define OPN_or_b32(A) "v_or_b32 " A ", " A ", %8\n"
DEFINE_KERNEL(or_b32, unsigned, INIT_U, OPN_or_b32)
#define OPN_lshlrev_b32(A) "v_lshlrev_b32 " A ", 1, " A "\n"
DEFINE_KERNEL(lshlrev_b32, unsigned, INIT_U, OPN_lshlrev_b32)
#define OPN_lshrrev_b32(A) "v_lshrrev_b32 " A ", 1, " A "\n"
DEFINE_KERNEL(lshrrev_b32, unsigned, INIT_U, OPN_lshrrev_b32)
#define OPN_ashrrev_i32(A) "v_ashrrev_i32 " A ", 1, " A "\n"
DEFINE_KERNEL(ashrrev_i32, unsigned, INIT_U, OPN_ashrrev_i32)
#define OPN_sub_u32(A) "v_sub_u32 " A ", " A ", %8\n"
DEFINE_KERNEL(sub_u32, unsigned, INIT_U, OPN_sub_u32)
#define OPN_min_i32(A) "v_min_i32 " A ", " A ", %8\n"
DEFINE_KERNEL(min_i32, unsigned, INIT_U, OPN_min_i32)
#define OPN_max_u32(A) "v_max_u32 " A ", " A ", %8\n"
DEFINE_KERNEL(max_u32, unsigned, INIT_U, OPN_max_u32)
#define OPN_add3_u32(A) "v_add3_u32 " A ", " A ", %8, %9\n"
DEFINE_KERNEL(add3_u32, unsigned, INIT_U, OPN_add3_u32)
#define OPN_and_or_b32(A) "v_and_or_b32 " A ", " A ", %8, %9\n"
DEFINE_KERNEL(and_or_b32, unsigned, INIT_U, OPN_and_or_b32)
#define OPN_or3_b32(A) "v_or3_b32 " A ", " A ", %8, %9\n"
DEFINE_KERNEL(or3_b32, unsigned, INIT_U, OPN_or3_b32)
#define OPN_xad_u32(A) "v_xad_u32 " A ", " A ", %8, %9\n"
DEFINE_KERNEL(xad_u32, unsigned, INIT_U, OPN_xad_u32)
#define OPN_lshl_or_b32(A) "v_lshl_or_b32 " A ", " A ", 1, %8\n"
DEFINE_KERNEL(lshl_or_b32, unsigned, INIT_U, OPN_lshl_or_b32)
#define OPN_bfe_u32(A) "v_bfe_u32 " A ", " A ", 3, 8\n"
DEFINE_KERNEL(bfe_u32, unsigned, INIT_U, OPN_bfe_u32)
#define OPN_mul_u32_u24(A) "v_mul_u32_u24 " A ", " A ", %8\n"
DEFINE_KERNEL(mul_u32_u24, unsigned, INIT_U, OPN_mul_u32_u24)
#define OPN_mov_b32(A) "v_mov_b32 " A ", %8\n"
DEFINE_KERNEL(mov_b32, unsigned, INIT_U, OPN_mov_b32)
#define OPN_perm_b32(A) "v_perm_b32 " A ", " A ", %8, %9\n"
DEFINE_KERNEL(perm_b32, unsigned, INIT_U, OPN_perm_b32)
#define OPN_alignbit_b32(A) "v_alignbit_b32 " A ", " A ", %8, 3\n"
DEFINE_KERNEL(alignbit_b32, unsigned, INIT_U, OPN_alignbit_b32)
#define OPN_sub_f32(A) "v_sub_f32 " A ", " A ", %8\n"
DEFINE_KERNEL(sub_f32, float, INIT_F, OPN_sub_f32)
#define OPN_min_f32(A) "v_min_f32 " A ", " A ", %8\n"
DEFINE_KERNEL(min_f32, float, INIT_F, OPN_min_f32)
#define OPN_mul_f32_inline2(A) "v_mul_f32 " A ", 2.0, " A "\n"
DEFINE_KERNEL(mul_f32_inline2, float, INIT_F, OPN_mul_f32_inline2)
#define OPN_fma_f32_sgpr(A) "v_fma_f32 " A ", " A ", s20, %9\n"
DEFINE_KERNEL(fma_f32_sgpr, float, INIT_F, OPN_fma_f32_sgpr)
#define OPN_fmaak_f32_literal(A) "v_fmaak_f32 " A ", " A ", %8, 0x3f8ccccd\n"
DEFINE_KERNEL(fmaak_f32_literal, float, INIT_F, OPN_fmaak_f32_literal)
#define OPN_mul_f32_e64_neg(A) "v_mul_f32_e64 " A ", -" A ", %8\n"
DEFINE_KERNEL(mul_f32_e64_neg, float, INIT_F, OPN_mul_f32_e64_neg)
#define OPN_fract_f32(A) "v_fract_f32 " A ", " A "\n"
DEFINE_KERNEL(fract_f32, float, INIT_F, OPN_fract_f32)
#define OPN_trunc_f32(A) "v_trunc_f32 " A ", " A "\n"
DEFINE_KERNEL(trunc_f32, float, INIT_F, OPN_trunc_f32)
#define OPN_rndne_f32(A) "v_rndne_f32 " A ", " A "\n"
DEFINE_KERNEL(rndne_f32, float, INIT_F, OPN_rndne_f32)
#define OPN_ldexp_f32(A) "v_ldexp_f32 " A ", " A ", 1\n"
DEFINE_KERNEL(ldexp_f32, float, INIT_F, OPN_ldexp_f32)
#define OPN_med3_f32(A) "v_med3_f32 " A ", " A ", %8, %9\n"
DEFINE_KERNEL(med3_f32, float, INIT_F, OPN_med3_f32)
#define OPN_cvt_f32_i32(A) "v_cvt_f32_i32 " A ", " A "\n"
DEFINE_KERNEL(cvt_f32_i32, float, INIT_F, OPN_cvt_f32_i32)
#define OPN_cvt_f32_u32(A) "v_cvt_f32_u32 " A ", " A "\n"
DEFINE_KERNEL(cvt_f32_u32, float, INIT_F, OPN_cvt_f32_u32)
#define OPN_cndmask_e64_sgpr(A) "v_cndmask_b32_e64 " A ", " A ", %8, s[20:21]\n"
DEFINE_KERNEL(cndmask_e64_sgpr, unsigned, INIT_U, OPN_cndmask_e64_sgpr)
#define OPN_cndmask_other_dst(A) "v_cndmask_b32 " A ", %8, %9, vcc\n"
DEFINE_KERNEL(cndmask_other_dst, unsigned, INIT_U, OPN_cndmask_other_dst)
#define OPN_cndmask_zero_vcc(A) "v_cndmask_b32 " A ", " A ", %8, vcc\n"
DEFINE_KERNEL(cndmask_zero_vcc, unsigned, INIT_U, OPN_cndmask_zero_vcc)
#define OPN_cmp_then_cndmask(A) "v_cmp_lt_f32 vcc, " A ", %8\nv_cndmask_b32 " A ", " A ", %9, vcc\n"
DEFINE_KERNEL(cmp_then_cndmask, float, INIT_F, OPN_cmp_then_cndmask)
#define OPN_cmp_e64_then_cndmask_e64(A) "v_cmp_lt_f32_e64 s[20:21], " A ", %8\nv_cndmask_b32_e64 " A ", " A ", %9, s[20:21]\n"
DEFINE_KERNEL(cmp_e64_then_cndmask_e64, float, INIT_F, OPN_cmp_e64_then_cndmask_e64)

// conversions: the destination is the accumulator, the source the never-written operand (a conversion cannot be chained without
// a second conversion; "dep" = 8 writes of the same register)
template <bool DEP>
__global__ __launch_bounds__(1024) void k_cvt_f64_f32(WaveRec* __restrict__ out, int iters, float seed) {
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  float b = INIT_F(9), c = INIT_F(11);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (DEP) { BODY1(OP_CVT_F64_F32_PAIR) } else { BODY8(OP_CVT_F64_F32_PAIR) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
  if ((threadIdx.x & 63) == 0) {
    WaveRec r;
    r.clk = t1 - t0; r.ref = r1 - r0; r.hw = hw_id(); r.xcc = xcc_id();
    out[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
  }
}
template <bool DEP>
__global__ __launch_bounds__(1024) void k_cvt_f32_f64(WaveRec* __restrict__ out, int iters, float seed) {
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  double b = INIT_D(9), c = INIT_D(11);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (DEP) { BODY1(OP_CVT_F32_F64_PAIR) } else { BODY8(OP_CVT_F32_F64_PAIR) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
  if ((threadIdx.x & 63) == 0) {
    WaveRec r;
    r.clk = t1 - t0; r.ref = r1 - r0; r.hw = hw_id(); r.xcc = xcc_id();
    out[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
  }
}

// ds_add_f64 / ds_read_b128 next to VALU work: does LDS issue share the VALU's slot?  (8 independent fma + 1 LDS op per 9 issues)
template <bool DEP>
__global__ __launch_bounds__(1024) void k_fma_plus_ds_add_f64(WaveRec* __restrict__ out, int iters, float seed) {
  __shared__ double win[1024 * 2];
  float a0 = INIT_F(0), a1 = INIT_F(1), a2 = INIT_F(2), a3 = INIT_F(3), a4 = INIT_F(4), a5 = INIT_F(5), a6 = INIT_F(6), a7 = INIT_F(7);
  float b = INIT_F(9), c = INIT_F(11);
  win[threadIdx.x] = 0.0; win[threadIdx.x + 1024] = 0.0;
  const unsigned addr = (unsigned)(threadIdx.x * 8);
  const double one = 1.0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (DEP) { BODY8(OP_FMA_F32) }
    else {
      asm volatile(R16(OP_FMA_F32("%0") OP_FMA_F32("%1") OP_FMA_F32("%2") OP_FMA_F32("%3") OP_FMA_F32("%4") OP_FMA_F32("%5")
                       OP_FMA_F32("%6") OP_FMA_F32("%7") "ds_add_f64 %10, %[one]\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(b), "v"(c), "v"(addr), [one] "v"(one) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
  if ((threadIdx.x & 63) == 0) {
    WaveRec r;
    r.clk = t1 - t0; r.ref = r1 - r0; r.hw = hw_id(); r.xcc = xcc_id();
    out[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
  }
}

typedef void (*Kern)(WaveRec*, int, float);
struct Op { const char* name; Kern indep, dep; int insts_per_body; };   // insts_per_body: VALU instructions per BODY8

static void run(const Op& op, int W, bool dep, WaveRec* dbuf, std::vector<WaveRec>& host, int ncu, int iters) {
  const int threads = W >= 4 ? 1024 : 256 * W;
  const int blocks = ncu * (W == 8 ? 2 : 1);
  const int waves = blocks * threads / 64;
  Kern k = dep ? op.dep : op.indep;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<<<blocks, threads>>>(dbuf, iters / 8, 1.0f);   // warm-up (code object load, clocks)
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<<<blocks, threads>>>(dbuf, iters, 1.0f);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(host.data(), dbuf, sizeof(WaveRec) * waves, hipMemcpyDeviceToHost));
  // waves per SIMD actually observed: key = (xcc, se, sh(not on gfx9: folded), cu, simd)
  std::map<unsigned, int> per_simd;
  double clk = 0.0, ref = 0.0;
  unsigned long long cmax = 0, cmin = ~0ull;
  for (int i = 0; i < waves; ++i) {
    const WaveRec& r = host[i];
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    const unsigned key = (r.xcc & 0xf) << 16 | (r.hw & 0xfff0 & ~0xc0u);
    per_simd[key]++;
    clk += (double)r.clk; ref += (double)r.ref;
    cmax = std::max(cmax, r.clk); cmin = std::min(cmin, r.clk);
  }
  int wmin = 1 << 30, wmax = 0;
  for (auto& kv : per_simd) { wmin = std::min(wmin, kv.second); wmax = std::max(wmax, kv.second); }
  const double insts = (double)iters * kUnroll * op.insts_per_body;   // per wave
  const double mean_clk = clk / waves;
  const double ghz = ref > 0 ? clk / ref * 0.1 : 0.0;                  // s_memrealtime: 100 MHz
  // W waves share the SIMD for the whole measured interval (they start together behind the barrier-less launch ramp; the spread
  // min..max is printed): SIMD time per wave-instruction = elapsed / (W * insts)
  const double eff_w = (double)waves / (double)per_simd.size();
  printf("%-18s %-5s W=%d  simds=%4zu waves/simd=[%d..%d]  %8.3f ms  clk/wave-instr/SIMD %6.3f  (per-wave latency %7.3f clk/instr; wave clk min/max %.3f; %.3f GHz; wall-derived %6.3f)\n",
         op.name, dep ? "dep" : "indep", W, per_simd.size(), wmin, wmax, ms, mean_clk / (eff_w * insts), mean_clk / insts,
         (double)cmin / (double)cmax, ghz, (double)ms * 1e-3 * ghz * 1e9 / (eff_w * insts));
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  int dev = 0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs, attribute clock %d kHz\n", prop.gcnArchName, ncu, prop.clockRate);
  printf("kernels: %d VALU instructions per loop iteration (8 independent accumulators x %d, or one accumulator), + 3 SALU; one block\n"
         "of 256 x W threads per CU (W = 8: two blocks of 1024); clk = s_memtime ticks (shader cycles) per wave\n", kAcc * kUnroll, kUnroll);
  WaveRec* dbuf;
  const size_t maxw = (size_t)ncu * 2 * 16;
  CHECK(hipMalloc(&dbuf, sizeof(WaveRec) * maxw));
  std::vector<WaveRec> host(maxw);
#define OPX(NAME) {#NAME, k_##NAME<false>, k_##NAME<true>, 8}
  const Op ops[] = {
      OPX(fma_f32), OPX(mul_f32), OPX(add_f32), OPX(fmac_f32), OPX(max_f32), OPX(floor_f32), OPX(cvt_i32_f32), OPX(cmp_lt_f32),
      OPX(pk_fma_f32), OPX(pk_mul_f32), OPX(mul_f64), OPX(fma_f64), OPX(add_f64),
      OPX(cvt_f64_f32), OPX(cvt_f32_f64),
      OPX(exp_f32), OPX(log_f32), OPX(rcp_f32), OPX(sqrt_f32), OPX(rcp_f64),
      OPX(mad_u32_u24), OPX(add_u32), OPX(xor_b32), OPX(lshl_add_u32), OPX(mul_lo_u32), OPX(mad_u64_u32), OPX(cndmask_b32),
      OPX(mov_dpp), OPX(add_f32_dpp), OPX(readlane), OPX(readfirstlane),
      OPX(and_b32), OPX(or_b32), OPX(lshlrev_b32), OPX(lshrrev_b32), OPX(ashrrev_i32), OPX(sub_u32), OPX(min_i32), OPX(max_u32), OPX(add3_u32), OPX(and_or_b32), OPX(or3_b32), OPX(xad_u32), OPX(lshl_or_b32), OPX(bfe_u32), OPX(mul_u32_u24), OPX(mov_b32), OPX(perm_b32), OPX(alignbit_b32), OPX(sub_f32), OPX(min_f32), OPX(mul_f32_inline2), OPX(fma_f32_sgpr), OPX(fmaak_f32_literal), OPX(mul_f32_e64_neg), OPX(fract_f32), OPX(trunc_f32), OPX(rndne_f32), OPX(ldexp_f32), OPX(med3_f32), OPX(cvt_f32_i32), OPX(cvt_f32_u32), OPX(cndmask_e64_sgpr), OPX(cndmask_other_dst), OPX(cndmask_zero_vcc),
      {"cmp_then_cndmask", k_cmp_then_cndmask<false>, k_cmp_then_cndmask<true>, 16}, {"cmp_e64_then_cndmask_e64", k_cmp_e64_then_cndmask_e64<false>, k_cmp_e64_then_cndmask_e64<true>, 16},
      {"8fma+1ds_add_f64", k_fma_plus_ds_add_f64<false>, k_fma_plus_ds_add_f64<true>, 8},
  };
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  const int iters = 2000;
  for (const Op& op : ops) {
    for (int W : {1, 2, 4, 8}) {
      if (quick && W != 1 && W != 4) continue;
      run(op, W, false, dbuf, host, ncu, iters);
    }
    run(op, 1, true, dbuf, host, ncu, iters);
    if (!strcmp(op.name, "8fma+1ds_add_f64")) run(op, 4, true, dbuf, host, ncu, iters);   // (dep = the same loop WITHOUT the LDS add)
  }
  CHECK(hipFree(dbuf));
  return 0;
}
