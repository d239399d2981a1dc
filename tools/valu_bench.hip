// valu_bench.hip -- issue cost of the VALU instructions the traversal kernels are made of, on gfx950.
// For each opcode: 8 independent dependency chains, 64 instructions per loop trip, W waves per SIMD (W = 1, 2, 4);
// prints shader cycles (s_memtime) per wave-instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  float a[8]; v2f p[8]; unsigned u[8];
  const float b = 1.0f + threadIdx.x * 1e-9f, c = 1e-9f; const v2f pb = {b, b}, pc = {c, c};
  const unsigned su0 = threadIdx.x * 7u + 0x01020304u, su1 = threadIdx.x * 13u + 0x00010203u;
  for (int i = 0; i < 8; i++) { a[i] = threadIdx.x + i; p[i] = (v2f){a[i], a[i] + 1.f}; u[i] = threadIdx.x * 2654435761u + i; }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP64(X)
#undef X
    } else if (OP == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      REP64(X)
#undef X
    } else if (OP == 2) {
#define X(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));
      REP64(X)
#undef X
    } else if (OP == 3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP64(X)
#undef X
    } else if (OP == 4) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : );
      REP64(X)
#undef X
    } else if (OP == 5) {
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(u[i]));
      REP64(X)
#undef X
    } else if (OP == 6) {
#define X(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
      REP64(X)
#undef X
    } else if (OP == 7) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP64(X)
#undef X
    } else if (OP == 8) {
#define X(i) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
      REP64(X)
#undef X
    } else if (OP == 9) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
      REP64(X)
#undef X
    }
#define OPCASE(N, STR, ...) else if (OP == N) { _Pragma("unroll") for (int r = 0; r < 8; r++) { REP8(OPX##N) } }
#define OPX10(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OPX11(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX12(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(su0));
#define OPX13(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(su0), "v"(su1));
#define OPX14(i) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX15(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define OPX16(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(su0), "v"(su1));
#define OPX17(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i]));
#define OPX18(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(su0));
#define OPX19(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(u[i]) : : "vcc");
#define OPX20(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define OPX21(i) asm volatile("v_cmp_le_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
#define OPX22(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(su0), "v"(su1));
#define OPX23(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(su0), "v"(su1));
#define OPX24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(su0));
#define OPX25(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define OPX26(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX27(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OPX28(i) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define OPX29(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX30(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX31(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(u[i]) : "v"(su0), "v"(su1));
#define OPX32(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u[i]) : "v"(su0));
#define OPX33(i) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(u[i]) : "v"(su0));
#define OPX34(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(su0));
#define OPX35(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(u[i]) : "v"(su0));
#define OPX36(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[i]) : "v"(su0));
#define OPX37(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(a[i]));
#define OPX38(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(su0));
#define OPX39(i) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(su0), "v"(su1));
// round 5: is the VOP2 form of v_cndmask (implicit VCC) really five times dearer than the VOP3 form with an SGPR pair (OPX12 vs OPX33)?  And what do the candidates for a shorter node step cost?
#define OPX40(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(su0));
#define OPX41(i) asm volatile("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(u[i]) : "v"(a[i]), "v"(b), "v"(su0) : "vcc");
#define OPX42(i) asm volatile("v_cmp_le_f32_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %3, s[20:21]" : "+v"(u[i]) : "v"(a[i]), "v"(b), "v"(su0) : "s20", "s21");
#define OPX43(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_1" : "+v"(u[i]) : "v"(su0));
#define OPX44(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8" : "+v"(u[i]) : "v"(su0), "v"(su1));
#define OPX45(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(su0));
#define OPX46(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(u[i]));
#define OPX47(i) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(su1), "v"(c));   /* first operand: an f16 DENORMAL in the low half */
#define OPX48(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(su0), "v"(su1));
#define OPX49(i) asm volatile("v_bfm_b32 %0, %0, %1" : "+v"(u[i]) : "v"(su0));
#define OPX50(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(u[i]));
#define OPX51(i) asm volatile("v_cvt_f32_ubyte0_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(a[i]) : "v"(u[i]));
#define OPX52(i) asm volatile("v_cmp_le_f32_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, 0, %3, s[20:21]" : "+v"(u[i]) : "v"(a[i]), "v"(b), "v"(su0) : "s20", "s21");
#define OPX53(i) asm volatile("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(u[i]) : "v"(a[i]), "v"(b) : "vcc");
#define OPX54(i) asm volatile("v_cmp_le_f32_e32 vcc, %1, %2\n\ts_nop 4\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(u[i]) : "v"(a[i]), "v"(b), "v"(su0) : "vcc");
#define OPX55(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
    OPCASE(10,) OPCASE(11,) OPCASE(12,) OPCASE(13,) OPCASE(14,) OPCASE(15,) OPCASE(16,) OPCASE(17,) OPCASE(18,) OPCASE(19,)
    OPCASE(20,) OPCASE(21,) OPCASE(22,) OPCASE(23,) OPCASE(24,) OPCASE(25,) OPCASE(26,) OPCASE(27,) OPCASE(28,) OPCASE(29,)
    OPCASE(30,) OPCASE(31,) OPCASE(32,) OPCASE(33,) OPCASE(34,) OPCASE(35,) OPCASE(36,) OPCASE(37,) OPCASE(38,) OPCASE(39,)
    OPCASE(40,) OPCASE(41,) OPCASE(42,) OPCASE(43,) OPCASE(44,) OPCASE(45,) OPCASE(46,) OPCASE(47,) OPCASE(48,) OPCASE(49,)
    OPCASE(50,) OPCASE(51,) OPCASE(52,) OPCASE(53,) OPCASE(54,) OPCASE(55,)
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0; unsigned su = 0;
  for (int i = 0; i < 8; i++) { s += a[i] + p[i].x + p[i].y; su += u[i]; }
  out[blockIdx.x * 256 + threadIdx.x] = s + su;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main(int argc, char** argv) {
  const int firstOp = argc > 1 ? atoi(argv[1]) : 0;
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_f32_ubyte1", "v_max3_f32", "v_cndmask_b32(dep)", "v_bfe_u32", "v_lshl_or_b32", "v_mul_f32", "v_cmp_le_f32", "v_and_b32",
                         "v_max_f32", "v_min3_f32", "v_cndmask_b32 vcc", "v_perm_b32", "v_fma_mix_f32", "v_cvt_f32_u32", "v_and_or_b32", "v_lshlrev_b32", "v_add_u32", "v_addc_co_u32",
                         "v_sub_f32", "v_cmp_le_f32 sgpr", "v_bfi_b32", "v_mad_u32_u24", "v_mul_u32_u24", "v_pk_mul_f32", "v_med3_f32", "v_min_f32", "v_cvt_f32_ubyte0", "v_mac_f32",
                         "v_fmac_f32", "v_pk_fma_f16", "v_pk_max_f16", "v_cndmask_b32 sgpr", "v_lshl_add_u32", "v_bcnt_u32_b32", "v_max_u32", "v_cvt_pk_u8_f32", "v_mov_b32", "v_sad_u8",
                         "v_cndmask_e64 vcc", "cmp_e32+cndmask_e32 vcc (2)", "cmp_e64+cndmask_e64 sgpr (2)", "v_lshlrev_b32_sdwa", "v_bitop3_b32", "v_alignbit_b32", "v_add_u32_dpp row_shr", "v_fma_mix f16 denormal",
                         "v_or3_b32", "v_bfm_b32", "v_ffbl_b32", "v_cvt_f32_ubyte0_sdwa", "cmp_e64+cndmask_e64 0,x (2)", "cmp_e32+addc vcc (2)", "cmp+s_nop4+cndmask vcc (2)", "v_pk_add_f32"};
  void (*fns[])(float*, int, long long*) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>, k<16>, k<17>, k<18>, k<19>,
                                            k<20>, k<21>, k<22>, k<23>, k<24>, k<25>, k<26>, k<27>, k<28>, k<29>, k<30>, k<31>, k<32>, k<33>, k<34>, k<35>, k<36>, k<37>, k<38>, k<39>,
                                            k<40>, k<41>, k<42>, k<43>, k<44>, k<45>, k<46>, k<47>, k<48>, k<49>, k<50>, k<51>, k<52>, k<53>, k<54>, k<55>};
  float* out; long long* cyc; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 256 * 8 * 4 * 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int op = firstOp; op < 56; op++)
    for (int w : {1, 4}) {
      const int blocks = 256 * w;
      hipLaunchKernelGGL(fns[op], dim3(blocks), dim3(256), 0, 0, out, 10, cyc);
      hipEventRecord(e0); hipLaunchKernelGGL(fns[op], dim3(blocks), dim3(256), 0, 0, out, iters, cyc); hipEventRecord(e1);
      hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(blocks * 4); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto v : h) avg += v; avg /= h.size();
      const double instr = (double)iters * 64;
      printf("VALU %-18s waves/SIMD %d: %.2f shader-clk per instr per wave, %.2f clk per instr per SIMD | wall %.3f ms -> %.2f clk/instr/SIMD at 2.4 GHz\n",
             names[op], w, avg / instr, avg / instr / w, ms, ms * 1e-3 * 2.4e9 / (instr * w));
    }
  return 0;
}
