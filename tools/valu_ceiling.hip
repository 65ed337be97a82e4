// Integer-VALU issue ceiling of gfx950, measured: wave-instructions per cycle per SIMD for the instructions the extractor
// kernels are made of (v_min3_u32, v_pk_max_u16, v_pk_add_u16 clamp, v_alignbyte_b32, v_mbcnt, v_dot4_u32_u8, v_perm_b32,
// v_and_or, v_cmp + v_cndmask, v_bcnt), as independent streams (8 chains per wave: no dependency stalls), for 1..8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_ceiling.hip -o valu_ceiling && ./valu_ceiling
// A wave64 instruction on a SIMD-32 takes 2 issue cycles => ceiling 0.5 wave-instructions / cycle / SIMD
// (MI355X_MICROARCH.md "Wave scheduling"); anything an opcode achieves below that is that opcode's own rate.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kIters = 2000;   // loop trips; each trip = 8 chains x 4 instructions = 32 VALU instructions

#define BODY8(OP)                                                                                                    \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)       \
               OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)       \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])      \
               : "v"(a), "v"(b))

#define OP_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_PKMAX(i) "v_pk_max_u16 %" #i ", %" #i ", %8\n"
#define OP_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8 clamp\n"
#define OP_ALIGN(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n"
#define OP_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define OP_DOT4(i) "v_dot4_u32_u8 %" #i ", %8, %9, %" #i "\n"
#define OP_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_BCNT(i) "v_bcnt_u32_b32 %" #i ", %8, %" #i "\n"
#define OP_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define OP_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define OP_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define OP_MAX3I16(i) "v_max3_i16 %" #i ", %" #i ", %8, %9\n"
#define OP_SAD(i) "v_sad_u8 %" #i ", %8, %9, %" #i "\n"
#define OP_ADD64(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n"
#define OP_MIN32(i) "v_min_u32_e32 %" #i ", %8, %" #i "\n"
#define OP_MIN64(i) "v_min_u32_e64 %" #i ", %8, %" #i "\n"
#define OP_XOR(i) "v_xor_b32_e32 %" #i ", %8, %" #i "\n"
#define OP_LSHL(i) "v_lshlrev_b32_e32 %" #i ", 1, %" #i "\n"
#define OP_MAXU16(i) "v_max_u16_e32 %" #i ", %8, %" #i "\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define OP_SDWA_MIN(i) "v_min_u32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define OP_SDWA_MAXU16(i) "v_max_u16_sdwa %" #i ", %8, %" #i " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:WORD_1\n"
#define OP_DPP_ADD(i) "v_add_u32_dpp %" #i ", %8, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_MOV(i) "v_mov_b32_e32 %" #i ", %8\n"
#define OP_SUBU16(i) "v_sub_u16_e32 %" #i ", %8, %" #i "\n"
#define OP_PKMINI16(i) "v_pk_min_i16 %" #i ", %" #i ", %8\n"
#define OP_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %10, %10\n"
#define OP_FMA32(i) "v_fmac_f32_e32 %" #i ", %8, %9\n"
#define OP_PKMAXF16(i) "v_pk_max_f16 %" #i ", %" #i ", %8\n"
#define OP_PKADDF16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
#define OP_PKFMAF16(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define OP_MAXF16(i) "v_max_f16_e32 %" #i ", %8, %" #i "\n"
#define OP_MINF32(i) "v_min_f32_e32 %" #i ", %8, %" #i "\n"
#define OP_MAX3F32(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_AND(i) "v_and_b32_e32 %" #i ", %8, %" #i "\n"
#define OP_OR(i) "v_or_b32_e32 %" #i ", %8, %" #i "\n"
#define OP_LSHR(i) "v_lshrrev_b32_e32 %" #i ", 1, %" #i "\n"
#define OP_MINU16(i) "v_min_u16_e32 %" #i ", %8, %" #i "\n"
#define OP_MINI16(i) "v_min_i16_e32 %" #i ", %8, %" #i "\n"
#define OP_ADDU16(i) "v_add_u16_e32 %" #i ", %8, %" #i "\n"
#define OP_MULU24(i) "v_mul_u32_u24_e32 %" #i ", %8, %" #i "\n"
#define OP_CVTUB(i) "v_cvt_f32_ubyte1_e32 %" #i ", %" #i "\n"
#define OP_CNDMASK(i) "v_cndmask_b32_e32 %" #i ", %8, %" #i ", vcc\n"
#define OP_CMPU16(i) "v_cmp_lt_u16_e32 vcc, %8, %" #i "\n"
#define OP_CMPF32(i) "v_cmp_lt_f32_e32 vcc, %8, %" #i "\n"
#define OP_PKADDU16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define OP_PKMULF32(i) "v_sub_u32_e32 %" #i ", %8, %" #i "\n"
#define OP_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_PKMINF16(i) "v_pk_min_f16 %" #i ", %" #i ", %8\n"
#define OP_MULF32(i) "v_mul_f32_e32 %" #i ", %8, %" #i "\n"
#define OP_ADDF32(i) "v_add_f32_e32 %" #i ", %8, %" #i "\n"
#define OP_MAXF32(i) "v_max_f32_e32 %" #i ", %8, %" #i "\n"
#define OP_CMPSEL(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"

template <int WHICH>
__global__ __launch_bounds__(512) void k_valu(unsigned* out, long long* cycles, unsigned a, unsigned b) {
  unsigned r[8];
  for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 2654435761u + i;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; it++) {
    if (WHICH == 0) BODY8(OP_MIN3);
    if (WHICH == 1) BODY8(OP_PKMAX);
    if (WHICH == 2) BODY8(OP_PKADD);
    if (WHICH == 3) BODY8(OP_ALIGN);
    if (WHICH == 4) BODY8(OP_MBCNT);
    if (WHICH == 5) BODY8(OP_DOT4);
    if (WHICH == 6) BODY8(OP_PERM);
    if (WHICH == 7) BODY8(OP_ANDOR);
    if (WHICH == 8) BODY8(OP_ADD);
    if (WHICH == 9) BODY8(OP_BCNT);
    if (WHICH == 10) BODY8(OP_MAD24);
    if (WHICH == 11) BODY8(OP_MULLO);
    if (WHICH == 12) BODY8(OP_LSHLADD);
    if (WHICH == 13) BODY8(OP_MAX3I16);
    if (WHICH == 14) BODY8(OP_SAD);
    if (WHICH == 15) BODY8(OP_CMPSEL);
    if (WHICH == 16) BODY8(OP_ADD64);
    if (WHICH == 17) BODY8(OP_MIN32);
    if (WHICH == 18) BODY8(OP_MIN64);
    if (WHICH == 19) BODY8(OP_XOR);
    if (WHICH == 20) BODY8(OP_LSHL);
    if (WHICH == 21) BODY8(OP_MAXU16);
    if (WHICH == 22) BODY8(OP_ADD3);
    if (WHICH == 23) BODY8(OP_BFE);
    if (WHICH == 24) BODY8(OP_SDWA_MIN);
    if (WHICH == 25) BODY8(OP_SDWA_MAXU16);
    if (WHICH == 26) BODY8(OP_DPP_ADD);
    if (WHICH == 27) BODY8(OP_MOV);
    if (WHICH == 28) BODY8(OP_SUBU16);
    if (WHICH == 29) BODY8(OP_PKMINI16);
    if (WHICH == 30) BODY8(OP_FMA32);
    if (WHICH == 31) BODY8(OP_PKMAXF16);
    if (WHICH == 32) BODY8(OP_PKADDF16);
    if (WHICH == 33) BODY8(OP_PKFMAF16);
    if (WHICH == 34) BODY8(OP_MAXF16);
    if (WHICH == 35) BODY8(OP_MINF32);
    if (WHICH == 36) BODY8(OP_MAX3F32);
    if (WHICH == 37) BODY8(OP_AND);
    if (WHICH == 38) BODY8(OP_OR);
    if (WHICH == 39) BODY8(OP_LSHR);
    if (WHICH == 40) BODY8(OP_MINU16);
    if (WHICH == 41) BODY8(OP_MINI16);
    if (WHICH == 42) BODY8(OP_ADDU16);
    if (WHICH == 43) BODY8(OP_MULU24);
    if (WHICH == 44) BODY8(OP_CVTUB);
    if (WHICH == 45) BODY8(OP_CNDMASK);
    if (WHICH == 46) BODY8(OP_CMPU16);
    if (WHICH == 47) BODY8(OP_CMPF32);
    if (WHICH == 48) BODY8(OP_PKADDU16);
    if (WHICH == 49) BODY8(OP_PKMULF32);
    if (WHICH == 50) BODY8(OP_MED3);
    if (WHICH == 51) BODY8(OP_PKMINF16);
    if (WHICH == 52) BODY8(OP_MULF32);
    if (WHICH == 53) BODY8(OP_ADDF32);
    if (WHICH == 54) BODY8(OP_MAXF32);
  }
  const long long t1 = __builtin_readcyclecounter();
  unsigned acc = 0;
  for (int i = 0; i < 8; i++) acc ^= r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

typedef void (*Kern)(unsigned*, long long*, unsigned, unsigned);

int main() {
  const char* names[] = {"v_min3_u32", "v_pk_max_u16", "v_pk_add_u16 clamp", "v_alignbyte_b32", "v_mbcnt_lo_u32_b32", "v_dot4_u32_u8", "v_perm_b32",
                         "v_and_or_b32", "v_add_u32", "v_bcnt_u32_b32", "v_mad_u32_u24", "v_mul_lo_u32", "v_lshl_add_u32", "v_max3_i16", "v_sad_u8",
                         "v_cmp_lt_u32 + v_cndmask_b32 (2 instr)", "v_add_u32_e64", "v_min_u32_e32", "v_min_u32_e64", "v_xor_b32_e32", "v_lshlrev_b32_e32",
                         "v_max_u16_e32", "v_add3_u32", "v_bfe_u32", "v_min_u32_sdwa (byte select)", "v_max_u16_sdwa (byte sel, word dst)",
                         "v_add_u32_dpp row_shr:1", "v_mov_b32_e32", "v_sub_u16_e32", "v_pk_min_i16", "v_fmac_f32_e32", "v_pk_max_f16", "v_pk_add_f16",
                         "v_pk_fma_f16", "v_max_f16_e32", "v_min_f32_e32", "v_max3_f32", "v_and_b32_e32", "v_or_b32_e32", "v_lshrrev_b32_e32",
                         "v_min_u16_e32", "v_min_i16_e32", "v_add_u16_e32", "v_mul_u32_u24_e32", "v_cvt_f32_ubyte1_e32", "v_cndmask_b32_e32",
                         "v_cmp_lt_u16_e32", "v_cmp_lt_f32_e32", "v_pk_add_u16 (no clamp)", "v_sub_u32_e32", "v_med3_f32", "v_pk_min_f16",
                         "v_mul_f32_e32", "v_add_f32_e32", "v_max_f32_e32"};
  Kern kerns[] = {k_valu<0>, k_valu<1>, k_valu<2>, k_valu<3>, k_valu<4>, k_valu<5>, k_valu<6>, k_valu<7>, k_valu<8>, k_valu<9>, k_valu<10>, k_valu<11>,
                  k_valu<12>, k_valu<13>, k_valu<14>, k_valu<15>, k_valu<16>, k_valu<17>, k_valu<18>, k_valu<19>, k_valu<20>, k_valu<21>, k_valu<22>,
                  k_valu<23>, k_valu<24>, k_valu<25>, k_valu<26>, k_valu<27>, k_valu<28>, k_valu<29>, k_valu<30>, k_valu<31>, k_valu<32>, k_valu<33>,
                  k_valu<34>, k_valu<35>, k_valu<36>, k_valu<37>, k_valu<38>, k_valu<39>, k_valu<40>, k_valu<41>, k_valu<42>, k_valu<43>, k_valu<44>,
                  k_valu<45>, k_valu<46>, k_valu<47>, k_valu<48>, k_valu<49>, k_valu<50>, k_valu<51>, k_valu<52>, k_valu<53>, k_valu<54>};
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  unsigned* out;
  long long* cyc;
  CHECK(hipMalloc(&out, sizeof(unsigned) * ncu * 2048));
  CHECK(hipMalloc(&cyc, sizeof(long long) * ncu * 4));
  std::printf("# %s, %d CUs, clock %d MHz; wave-instructions / cycle / SIMD (s_memtime cycles, 100 MHz-class counter scaled by wall clock below)\n", p.gcnArchName,
              ncu, p.clockRate / 1000);
  std::printf("%-42s %8s %8s %8s %8s   %s\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD", "Tinstr/s at 8 waves (whole chip)");
  for (int k = 0; k < 55; k++) {
    double ipc[4] = {0, 0, 0, 0}, tops = 0;
    const int wps[4] = {1, 2, 4, 8};
    for (int c = 0; c < 4; c++) {
      // waves per SIMD = wps: workgroups of 256 threads (one wave per SIMD), wps of them per CU
      const int blocks = ncu * wps[c];
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      hipLaunchKernelGGL(kerns[k], dim3(blocks), dim3(256), 0, 0, out, cyc, 0x01020304u, 0x7fu);   // warm-up
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(kerns[k], dim3(blocks), dim3(256), 0, 0, out, cyc, 0x01020304u, 0x7fu);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double instr_per_wave = (double)kIters * 32 * (k == 15 ? 2 : 1);
      const double total = instr_per_wave * 4.0 * blocks;                 // wave-instructions launched
      const double cycles = ms * 1e-3 * (double)p.clockRate * 1e3;        // shader cycles of wall time
      ipc[c] = total / ((double)ncu * 4.0) / cycles;                      // per SIMD per cycle
      if (c == 3) tops = total / (ms * 1e-3) / 1e12;
    }
    std::printf("%-42s %8.3f %8.3f %8.3f %8.3f   %.2f\n", names[k], ipc[0], ipc[1], ipc[2], ipc[3], tops);
  }
  return 0;
}
