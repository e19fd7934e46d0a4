// inst_cost.hip -- what does one wave64 instruction cost on gfx950, in SHADER CYCLES of the SIMD it runs on?
// Every test kernel runs 8 independent dependency chains per lane, `iters` times, W waves per SIMD (grid = 256 W
// workgroups of 4 waves, all resident at once).  Reported: HIP-event time per wave-instruction and SIMD (ns, best of 3,
// and relative to v_fma_f32 at the same occupancy), with the s_memtime ticks of the same run beside it.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/inst_cost.bin scripts/ubench/inst_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string>
#include <vector>

#define CHAINS 8
enum Op {
  FMA, MUL, ADD, FMA_SGPR, FMAC, PK_FMA, PK_MUL, PK_ADD, MIN_RAW, MAX_RAW, MED3, MIN3, FMINF_C, CMP_CND, CMP_ONLY, CND_ONLY,
  EXP, RCP, LOG, SQRT, RSQ, EXPF_C, DPP_QUAD, DPP_ROW_SHR1, DPP_ROW_ROR8, DPP_BCAST15, DPP_HALF_MIRROR, DPP_MOV, DPP_WAVE_SHR1,
  SWAP32, SWAP16, READLANE, READFIRST, READLANE_USE, IAND, IADD, LSHL, LSHL_ADD, MAD_U24, CVT_F2I, CVT_I2F, FLOOR,
  FMA_CLAMP, MUL_NEGABS, MOV, LDS_B128_BCAST, LDS_B32, LDS_B64_BCAST, BPERMUTE, SWIZZLE, MIX_EXP_3FMA, MIX_DPP_3FMA, MIX_SALU_FMA,
  MIX_CMP_3FMA, SUB_SGPR, FMA_2SGPR, MIX_EXP_7FMA, MIX_LDS_4FMA, ATOMIC_LDS_ADD, NOPS
};

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters, float b_in, float c_in) {
  __shared__ float4 lds[256];
  float a[CHAINS];
  float a2[CHAINS];
  for (int i = 0; i < CHAINS; i++) {
    a[i] = threadIdx.x * 0.001f + i;
    a2[i] = threadIdx.x * 0.002f + i;
  }
  lds[threadIdx.x] = make_float4(a[0], a[1], a[2], a[3]);
  __syncthreads();
  float b = b_in, c = c_in;
  float sb = __builtin_amdgcn_readfirstlane(b_in), sc = __builtin_amdgcn_readfirstlane(c_in);
  int si = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) {
      if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c));
      if (OP == FMA_2SGPR) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "s"(sb));
      if (OP == SUB_SGPR) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
      if (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == PK_FMA) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 x = {a[i], a2[i]}, bb = {b, b}, cc = {c, c};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(bb), "v"(cc));
        a[i] = x[0]; a2[i] = x[1];
      }
      if (OP == PK_MUL) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 x = {a[i], a2[i]}, bb = {b, b};
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(bb));
        a[i] = x[0]; a2[i] = x[1];
      }
      if (OP == PK_ADD) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 x = {a[i], a2[i]}, cc = {c, c};
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(cc));
        a[i] = x[0]; a2[i] = x[1];
      }
      if (OP == MIN_RAW) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == MAX_RAW) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == FMINF_C) a[i] = fminf(a[i], b);
      if (OP == CMP_CND) a[i] = a[i] > c ? b : a[i] + 0.f;
      if (OP == CMP_ONLY) {
        uint64_t m;
        asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(c));
        si += (int)m;
      }
      if (OP == CND_ONLY) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
      if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (OP == LOG) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
      if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
      if (OP == RSQ) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
      if (OP == EXPF_C) a[i] = __expf(a[i]);
      if (OP == DPP_QUAD) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (OP == DPP_ROW_SHR1) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (OP == DPP_ROW_ROR8) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (OP == DPP_BCAST15) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]));
      if (OP == DPP_HALF_MIRROR) asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (OP == DPP_MOV) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (OP == DPP_WAVE_SHR1) asm volatile("v_add_f32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (OP == SWAP32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a2[i]));
      if (OP == SWAP16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a2[i]));
      if (OP == READLANE) {
        int s;
        asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(a[i]));
        si += s;
      }
      if (OP == READFIRST) {
        int s;
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(a[i]));
        si += s;
      }
      if (OP == READLANE_USE) a[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[i]), it & 63)) + a[i];
      if (OP == IAND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
      if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
      if (OP == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == CVT_F2I) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
      if (OP == CVT_I2F) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
      if (OP == FLOOR) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
      if (OP == FMA_CLAMP) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == MUL_NEGABS) asm volatile("v_mul_f32 %0, -|%0|, %1" : "+v"(a[i]) : "v"(b));
      if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a2[i]));
      if (OP == LDS_B128_BCAST) {
        float4 r = lds[(it + i) & 255];
        a[i] += r.x + r.w;
      }
      if (OP == LDS_B64_BCAST) {
        float2 r = reinterpret_cast<float2 *>(lds)[(it + i) & 511];
        a[i] += r.x + r.y;
      }
      if (OP == LDS_B32) {
        float r = reinterpret_cast<float *>(lds)[((it + i) * 64 + threadIdx.x) & 1023];
        a[i] += r;
      }
      if (OP == BPERMUTE) a[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(((threadIdx.x + 1) & 63) << 2, __float_as_int(a[i])));
      if (OP == SWIZZLE) a[i] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a[i]), 0x041F));
      if (OP == MIX_EXP_3FMA) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a2[i]) : "v"(b), "v"(c));
      }
      if (OP == MIX_EXP_7FMA) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a2[i]) : "v"(b), "v"(c));
      }
      if (OP == MIX_DPP_3FMA) {
        asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a2[i]) : "v"(b), "v"(c));
      }
      if (OP == MIX_CMP_3FMA) {
        a[i] = a[i] > c ? b : a[i];
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a2[i]) : "v"(b), "v"(c));
      }
      if (OP == MIX_SALU_FMA) {
        asm volatile("v_fma_f32 %0, %0, %2, %3\n s_mul_i32 %1, %1, 3" : "+v"(a[i]), "+s"(si) : "v"(b), "v"(c));
      }
      if (OP == MIX_LDS_4FMA) {
        float4 r = lds[(it + i) & 255];
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a2[i]) : "v"(b), "v"(c));
        a[i] += r.x;
      }
      if (OP == ATOMIC_LDS_ADD) atomicAdd(reinterpret_cast<float *>(lds) + ((i * 64 + threadIdx.x) & 1023), a[i]);
      if (OP == NOPS) asm volatile("s_nop 0");
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = (float)si;
  for (int i = 0; i < CHAINS; i++) s += a[i] + a2[i];
  out[blockIdx.x * 64 + threadIdx.x] = s + lds[threadIdx.x].x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Row { std::string name; double ns[3]; double cyc[3]; };
std::vector<Row> rows;
double fma_ns[3] = {0, 0, 0};

// grid = 256 CUs x w workgroups of 4 waves: w waves on every SIMD, all resident at once.  Best of 3 runs of ~10^5 chain
// steps; the figure of merit is TIME per wave-instruction and SIMD (ns, and relative to v_fma_f32 at the same
// occupancy); the s_memtime cycles are printed beside it.
template <int OP> void run(const char *name, double ops_per_step) {
  Row r; r.name = name;
  int wi = 0;
  for (int w : {1, 4, 8}) {
    int blocks = 256 * w, iters = 20000;
    float *out; long long *cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, cyc, 2000, 1.0001f, 0.5f);
    double best = 1e30, bestc = 0;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, cyc, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(blocks);
      hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
      double sum = 0; for (auto v : h) sum += (double)v;
      if (ms < best) { best = ms; bestc = sum / blocks; }
    }
    double n = (double)iters * CHAINS * ops_per_step * w;  // wave-instructions per SIMD
    r.ns[wi] = best * 1e6 / n;
    r.cyc[wi] = bestc / n;
    wi++;
    hipFree(out); hipFree(cyc);
  }
  if (OP == FMA) for (int i = 0; i < 3; i++) fma_ns[i] = r.ns[i];
  rows.push_back(r);
  printf("%-30s ns/inst/SIMD @1,4,8 waves: %6.2f %6.2f %6.2f | x v_fma: %5.2f %5.2f %5.2f | s_memtime ticks: %6.2f %6.2f %6.2f\n", name,
         r.ns[0], r.ns[1], r.ns[2], r.ns[0] / fma_ns[0], r.ns[1] / fma_ns[1], r.ns[2] / fma_ns[2], r.cyc[0], r.cyc[1], r.cyc[2]);
  fflush(stdout);
}

int main() {
  run<FMA>("v_fma_f32", 1); run<MUL>("v_mul_f32", 1); run<ADD>("v_add_f32", 1); run<FMAC>("v_fmac_f32", 1);
  run<FMA_SGPR>("v_fma_f32 (1 sgpr)", 1); run<FMA_2SGPR>("v_fma_f32 (sgpr x2)", 1); run<SUB_SGPR>("v_sub_f32 (sgpr)", 1);
  run<FMA_CLAMP>("v_fma_f32 clamp", 1); run<MUL_NEGABS>("v_mul_f32 -|x|", 1); run<MOV>("v_mov_b32", 1);
  run<PK_FMA>("v_pk_fma_f32", 1); run<PK_MUL>("v_pk_mul_f32", 1); run<PK_ADD>("v_pk_add_f32", 1);
  run<MIN_RAW>("v_min_f32", 1); run<MAX_RAW>("v_max_f32", 1); run<MED3>("v_med3_f32", 1); run<MIN3>("v_min3_f32", 1);
  run<FMINF_C>("fminf() (C++: max+min)", 2); run<CMP_CND>("cmp+cndmask+add (C++)", 3); run<CMP_ONLY>("v_cmp -> sgpr (+s_add)", 1);
  run<CND_ONLY>("v_cndmask vcc", 1);
  run<EXP>("v_exp_f32", 1); run<RCP>("v_rcp_f32", 1); run<LOG>("v_log_f32", 1); run<SQRT>("v_sqrt_f32", 1); run<RSQ>("v_rsq_f32", 1);
  run<EXPF_C>("__expf (mul+exp)", 2);
  run<DPP_QUAD>("v_add dpp quad_perm", 1); run<DPP_ROW_SHR1>("v_add dpp row_shr:1", 1); run<DPP_ROW_ROR8>("v_add dpp row_ror:8", 1);
  run<DPP_BCAST15>("v_add dpp row_bcast15", 1); run<DPP_HALF_MIRROR>("v_add dpp half_mirror", 1); run<DPP_MOV>("v_mov dpp row_shr:1", 1);
  run<DPP_WAVE_SHR1>("v_add dpp wave_shr:1", 1);
  run<SWAP32>("v_permlane32_swap", 1); run<SWAP16>("v_permlane16_swap", 1);
  run<READLANE>("v_readlane (+s_add)", 1); run<READFIRST>("v_readfirstlane (+s_add)", 1); run<READLANE_USE>("readlane+v_add", 2);
  run<IAND>("v_and_b32", 1); run<IADD>("v_add_u32", 1); run<LSHL>("v_lshlrev_b32", 1); run<LSHL_ADD>("v_lshl_add_u32", 1);
  run<MAD_U24>("v_mad_u32_u24", 1); run<CVT_F2I>("v_cvt_i32_f32", 1); run<CVT_I2F>("v_cvt_f32_i32", 1); run<FLOOR>("v_floor_f32", 1);
  run<LDS_B128_BCAST>("ds_read_b128 bcast +2add (per 3)", 3); run<LDS_B64_BCAST>("ds_read_b64 bcast +2add (per 3)", 3); run<LDS_B32>("ds_read_b32 +add (per 2)", 2);
  run<BPERMUTE>("ds_bpermute_b32", 1); run<SWIZZLE>("ds_swizzle_b32", 1);
  run<MIX_EXP_3FMA>("1 exp + 3 fma (per 4)", 4); run<MIX_EXP_7FMA>("1 exp + 7 fma (per 8)", 8); run<MIX_DPP_3FMA>("1 dpp + 3 fma (per 4)", 4);
  run<MIX_CMP_3FMA>("cmp+cnd + 3 fma (per 5)", 5); run<MIX_SALU_FMA>("1 fma + 1 s_mul (per fma)", 1); run<MIX_LDS_4FMA>("b128 bcast + 4fma+add (per 6)", 6);
  run<NOPS>("s_nop 0", 1);
  return 0;
}
