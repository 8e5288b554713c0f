// Issue rate of wave64 instructions on gfx950 (one SIMD), measured: how many cycles a SIMD is occupied per wave
// instruction, for the instruction classes the integer raster kernels (k_pairs16, k_descent16, k_relax_bits) are made of.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
//
// Method: every wave executes ITER trips of a loop holding UNROLL copies of ONE instruction on NACC independent
// registers (no dependency closer than NACC instructions: issue-bound, not latency-bound).  The grid puts W waves on every
// SIMD of every CU (256 CUs x 4 SIMDs).  cycles per wave-instruction per SIMD = t * f / (W * ITER * UNROLL), f from
// s_memrealtime-free wall time and the clock the runtime reports; the RATIO between instruction classes does not depend
// on f.  Both 1 wave per SIMD (can one wave alone keep the SIMD issuing?) and 4 / 8 waves per SIMD are run.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ITER = 2000, UNROLL = 64, NACC = 8;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

template <int KIND>
__global__ __launch_bounds__(256) void k_probe(uint32_t *out, uint32_t seed) {
  uint32_t a[NACC];
  __shared__ uint32_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 4;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NACC; j++) a[j] = seed + threadIdx.x * 4 + j * 1024;
  uint32_t b = seed ^ 0x55u, c = seed | 3u;
  asm volatile("" : "+v"(b), "+v"(c));
  for (int it = 0; it < ITER; it++) {
    if (KIND == 0) {
#define OP(j) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
      BODY8(OP)
#undef OP
    } else if (KIND == 1) {
#define OP(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
      BODY8(OP)
#undef OP
    } else if (KIND == 2) {   // compare into VCC + select: two instructions per copy
#define OP(j) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[j]) : "v"(b), "v"(c) : "vcc");
      BODY8(OP)
#undef OP
    } else if (KIND == 3) {
#define OP(j) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[j]) : "v"(b));
      BODY8(OP)
#undef OP
    } else if (KIND == 4) {
#define OP(j) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
      BODY8(OP)
#undef OP
    } else if (KIND == 5) {   // packed FP32 (the rate the 157 TFLOP/s vector peak is quoted at)
      float2 *f = reinterpret_cast<float2 *>(a);
#define OP(j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f[(j) & 3]) : "v"(f[(j) & 3]));
      BODY8(OP)
#undef OP
    } else if (KIND == 6) {
#define OP(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
      BODY8(OP)
#undef OP
    } else if (KIND == 7) {   // DPP row shift
#define OP(j) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]));
      BODY8(OP)
#undef OP
    } else if (KIND == 8) {   // LDS: 16-bit reads at a lane-dependent address (the address feeds on the result: & keeps it in range)
#define OP(j) asm volatile("ds_read_u16 %0, %0\n\ts_waitcnt lgkmcnt(7)" : "+v"(a[j]));
#pragma unroll
      for (int j = 0; j < NACC; j++) a[j] &= 0x3FFCu;
      BODY8(OP)
      asm volatile("s_waitcnt lgkmcnt(0)");
#undef OP
    } else if (KIND == 9) {   // LDS: 32-bit reads
#define OP(j) asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(7)" : "+v"(a[j]));
#pragma unroll
      for (int j = 0; j < NACC; j++) a[j] &= 0x3FFCu;
      BODY8(OP)
      asm volatile("s_waitcnt lgkmcnt(0)");
#undef OP
    } else if (KIND == 10) {   // 64-bit ballot of a compare (SGPR pair destination)
      unsigned long long m;
#define OP(j) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(a[j]), "v"(b)); asm volatile("" ::"s"(m));
      BODY8(OP)
#undef OP
    } else if (KIND == 11) {   // v_perm_b32
#define OP(j) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
      BODY8(OP)
#undef OP
    } else if (KIND == 12) {   // 64-bit shift (funnel work of the bitmap search)
      unsigned long long *q = reinterpret_cast<unsigned long long *>(a);
#define OP(j) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[(j) & 3]));
      BODY8(OP)
#undef OP
    } else if (KIND == 13) {   // v_min3_u32
#define OP(j) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
      BODY8(OP)
#undef OP
    }
    // ---- second batch (r05): which of the everyday integer / float instructions run at the ~2.4-cycle rate of v_add_u32 / v_fma_f32
#define TWO(K, INS)                                                              \
    else if (KIND == K) {                                                        \
      _Pragma("unroll") for (int u = 0; u < UNROLL / NACC; u++)                  \
        _Pragma("unroll") for (int j = 0; j < NACC; j++) asm volatile(INS " %0, %0, %1" : "+v"(a[j]) : "v"(b)); \
    }
#define THREE(K, INS)                                                            \
    else if (KIND == K) {                                                        \
      _Pragma("unroll") for (int u = 0; u < UNROLL / NACC; u++)                  \
        _Pragma("unroll") for (int j = 0; j < NACC; j++) asm volatile(INS " %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); \
    }
    TWO(14, "v_and_b32") TWO(15, "v_or_b32") TWO(16, "v_xor_b32") TWO(17, "v_lshlrev_b32") TWO(20, "v_max_u32")
    TWO(21, "v_mul_u32_u24") TWO(25, "v_sub_u32") TWO(26, "v_min_f32") TWO(27, "v_max_f32") TWO(37, "v_add_f32") TWO(38, "v_mul_f32")
    TWO(41, "v_min_u16") TWO(42, "v_add_u16") TWO(44, "v_lshrrev_b32") TWO(46, "v_min_i32") TWO(47, "v_mul_lo_u32")
    THREE(22, "v_mad_u32_u24") THREE(23, "v_add3_u32") THREE(24, "v_lshl_add_u32") THREE(28, "v_min3_f32") THREE(30, "v_bfe_u32")
    THREE(32, "v_add_lshl_u32") THREE(33, "v_or3_b32") THREE(40, "v_alignbit_b32") THREE(48, "v_xad_u32") THREE(49, "v_med3_u32")
    THREE(50, "v_lshl_or_b32") THREE(51, "v_max3_u32")
    else if (KIND == 18) {   // compare into VCC alone
#pragma unroll
      for (int u = 0; u < UNROLL / NACC; u++)
#pragma unroll
        for (int j = 0; j < NACC; j++) asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(a[j]), "v"(b) : "vcc");
    } else if (KIND == 29) {
#pragma unroll
      for (int u = 0; u < UNROLL / NACC; u++)
#pragma unroll
        for (int j = 0; j < NACC; j++) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(a[j]), "v"(b) : "vcc");
    } else if (KIND == 19) {   // select alone (VCC as it is)
#pragma unroll
      for (int u = 0; u < UNROLL / NACC; u++)
#pragma unroll
        for (int j = 0; j < NACC; j++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b) : "vcc");
    } else if (KIND == 31) {
#pragma unroll
      for (int u = 0; u < UNROLL / NACC; u++)
#pragma unroll
        for (int j = 0; j < NACC; j++) asm volatile("v_mov_b32 %0, %1" : "=v"(a[j]) : "v"(b));
    } else if (KIND == 36) {
#pragma unroll
      for (int u = 0; u < UNROLL / NACC; u++)
#pragma unroll
        for (int j = 0; j < NACC; j++) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[j]) : "v"(b));
    } else if (KIND == 35) {   // LDS stores
#pragma unroll
      for (int j = 0; j < NACC; j++) a[j] &= 0x3FFCu;
#pragma unroll
      for (int u = 0; u < UNROLL / NACC; u++)
#pragma unroll
        for (int j = 0; j < NACC; j++) asm volatile("ds_write_b32 %0, %1" ::"v"(a[j]), "v"(b) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == 52 || KIND == 53 || KIND == 54 || KIND == 55) {
      // the SCALAR unit: 52 = s_add_u32 alone, 53 = s_and_b64 (lane-mask logic) alone, 54 = one s_add_u32 after every v_add_u32
      // (does scalar work ride along with vector work of the same wave?), 55 = one s_and_b64 after every v_min_u32
      uint32_t sa[NACC];
      unsigned long long sm[NACC];
#pragma unroll
      for (int j = 0; j < NACC; j++) { sa[j] = __builtin_amdgcn_readfirstlane(a[j]); sm[j] = __builtin_amdgcn_ballot_w64(a[j] > (uint32_t)j); asm volatile("" : "+s"(sa[j]), "+s"(sm[j])); }
#pragma unroll
      for (int u = 0; u < UNROLL / NACC; u++)
#pragma unroll
        for (int j = 0; j < NACC; j++) {
          if (KIND == 54) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
          if (KIND == 55) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
          if (KIND == 52 || KIND == 54) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sa[j])::"scc");
          else asm volatile("s_and_b64 %0, %0, exec" : "+s"(sm[j])::"scc");
        }
#pragma unroll
      for (int j = 0; j < NACC; j++) { a[j] ^= sa[j] ^ (uint32_t)sm[j]; }
    }
#undef TWO
#undef THREE
  }
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < NACC; j++) s ^= a[j];
  if (s == 0x12345678u) out[threadIdx.x] = s;   // (keeps the chain alive)
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int KIND>
static void run(const char *name, int per_copy, uint32_t *d, int cus, double ghz) {
  printf("%-44s", name);
  for (int wps : {1, 2, 4, 8}) {   // waves per SIMD
    // a block = 256 threads = one wave per SIMD of its CU; wps blocks per CU
    const int blocks = cus * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_probe<KIND><<<blocks, 256>>>(d, 7);   // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
      CK(hipEventRecord(e0));
      k_probe<KIND><<<blocks, 256>>>(d, 7);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    const double insts = (double)ITER * UNROLL * per_copy * wps;   // wave instructions one SIMD issued
    printf("  %dw/SIMD %6.2f cyc", wps, best * 1e-3 * ghz * 1e9 / insts);
  }
  printf("\n");
}

int main() {
  int dev = 0, cus = 0, khz = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev));
  const double ghz = khz * 1e-6;
  printf("gfx950 issue probe: %d CUs, runtime clock %.3f GHz; cycles a SIMD spends per wave64 instruction (at that clock)\n", cus, ghz);
  uint32_t *d; CK(hipMalloc(&d, 4096));
  run<0>("v_min_u32", 1, d, cus, ghz);
  run<1>("v_add_u32", 1, d, cus, ghz);
  run<2>("v_cmp_lt_u32 + v_cndmask_b32 (per instr)", 2, d, cus, ghz);
  run<13>("v_min3_u32", 1, d, cus, ghz);
  run<4>("v_and_or_b32", 1, d, cus, ghz);
  run<11>("v_perm_b32", 1, d, cus, ghz);
  run<3>("v_pk_min_u16", 1, d, cus, ghz);
  run<12>("v_lshlrev_b64", 1, d, cus, ghz);
  run<10>("v_cmp_lt_u32 -> SGPR pair (ballot)", 1, d, cus, ghz);
  run<7>("v_mov_b32_dpp row_shr:1", 1, d, cus, ghz);
  run<6>("v_fma_f32", 1, d, cus, ghz);
  run<5>("v_pk_add_f32 (2 flops per lane)", 1, d, cus, ghz);
  run<8>("ds_read_u16", 1, d, cus, ghz);
  run<9>("ds_read_b32", 1, d, cus, ghz);
  printf("-- second batch\n");
  run<31>("v_mov_b32", 1, d, cus, ghz);
  run<14>("v_and_b32", 1, d, cus, ghz);
  run<15>("v_or_b32", 1, d, cus, ghz);
  run<16>("v_xor_b32", 1, d, cus, ghz);
  run<33>("v_or3_b32", 1, d, cus, ghz);
  run<50>("v_lshl_or_b32", 1, d, cus, ghz);
  run<17>("v_lshlrev_b32", 1, d, cus, ghz);
  run<44>("v_lshrrev_b32", 1, d, cus, ghz);
  run<30>("v_bfe_u32", 1, d, cus, ghz);
  run<40>("v_alignbit_b32", 1, d, cus, ghz);
  run<25>("v_sub_u32", 1, d, cus, ghz);
  run<23>("v_add3_u32", 1, d, cus, ghz);
  run<24>("v_lshl_add_u32", 1, d, cus, ghz);
  run<32>("v_add_lshl_u32", 1, d, cus, ghz);
  run<48>("v_xad_u32", 1, d, cus, ghz);
  run<21>("v_mul_u32_u24", 1, d, cus, ghz);
  run<22>("v_mad_u32_u24", 1, d, cus, ghz);
  run<47>("v_mul_lo_u32", 1, d, cus, ghz);
  run<20>("v_max_u32", 1, d, cus, ghz);
  run<46>("v_min_i32", 1, d, cus, ghz);
  run<51>("v_max3_u32", 1, d, cus, ghz);
  run<49>("v_med3_u32", 1, d, cus, ghz);
  run<41>("v_min_u16", 1, d, cus, ghz);
  run<42>("v_add_u16", 1, d, cus, ghz);
  run<18>("v_cmp_lt_u32 -> vcc", 1, d, cus, ghz);
  run<19>("v_cndmask_b32 (vcc)", 1, d, cus, ghz);
  run<36>("v_mbcnt_lo_u32_b32", 1, d, cus, ghz);
  run<37>("v_add_f32", 1, d, cus, ghz);
  run<38>("v_mul_f32", 1, d, cus, ghz);
  run<26>("v_min_f32", 1, d, cus, ghz);
  run<27>("v_max_f32", 1, d, cus, ghz);
  run<28>("v_min3_f32", 1, d, cus, ghz);
  run<29>("v_cmp_lt_f32 -> vcc", 1, d, cus, ghz);
  run<35>("ds_write_b32", 1, d, cus, ghz);
  printf("-- the scalar unit (cycles per instruction COUNTED: all scalar ones, or the vector ones of a mixed stream)\n");
  run<52>("s_add_u32 alone", 1, d, cus, ghz);
  run<53>("s_and_b64 alone", 1, d, cus, ghz);
  run<54>("v_add_u32 + s_add_u32 pairs (per pair)", 1, d, cus, ghz);
  run<55>("v_min_u32 + s_and_b64 pairs (per pair)", 1, d, cus, ghz);
  return 0;
}
