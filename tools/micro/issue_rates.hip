// Issue-rate / latency probe for gfx950 written with inline asm so that the compiler cannot pack, reorder or drop anything:
// one wave per SIMD (625 single-wave blocks), each kernel runs ITERS x 64 copies of one instruction pattern.
// Prints ns per instruction and cycles at the 2.4 GHz peak clock.   hipcc --offload-arch=gfx950 -O3 issue_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define KERNEL(name, body, per_rep)                                                                              \
  __global__ __launch_bounds__(64) void name(float* out, int iters, float a) {                                    \
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    typedef float f2 __attribute__((ext_vector_type(2)));                                                        \
    typedef float f4 __attribute__((ext_vector_type(4))); f4 q0, q1; f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a};                                   \
    __shared__ float sm[1024];                                                                                    \
    sm[threadIdx.x] = x0;                                                                                        \
    unsigned addr = threadIdx.x * 4;                                                                             \
    for (int i = 0; i < iters; ++i) {                                                                            \
      REP16(body)                                                                                                \
    }                                                                                                            \
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y; \
  }                                                                                                              \
  static const int name##_per = 16 * (per_rep);

// 4 instructions per body unless noted
KERNEL(fma_dep, asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0" : "+v"(x0) : "v"(a));, 4)
KERNEL(fma_ind, asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));, 4)
KERNEL(pk_dep, asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p0) : "v"(pa));, 4)
KERNEL(pk_ind, asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));, 4)
KERNEL(pk_mul_ind, asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));, 4)
KERNEL(exp_dep, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0" : "+v"(x0));, 4)
KERNEL(exp_ind, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));, 4)
KERNEL(exp_fma_mix, asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));, 4)
KERNEL(rcp_ind, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));, 4)
KERNEL(dpp_dep, asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(x0));, 4)
KERNEL(dpp_ind, asm volatile("v_mov_b32_dpp %0, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 quad_perm:[2,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 quad_perm:[3,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[0,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));, 4)
KERNEL(fma_dpp_src, asm volatile("v_fmac_f32_dpp %0, %4, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %4, %5 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %4, %5 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %4, %5 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(x4));, 4)
KERNEL(dpp_then_fma, asm volatile("v_mov_b32_dpp %1, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_fma_f32 %0, %1, %2, %0\n v_mov_b32_dpp %1, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_fma_f32 %0, %1, %2, %0" : "+v"(x0), "+v"(x1) : "v"(a));, 4)
KERNEL(lds_rt, asm volatile("ds_write_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(x0) : "v"(addr) : "memory");, 1)
KERNEL(lds_read_dep, asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %2, %0" : "+v"(x0) : "v"(addr), "v"(a) : "memory");, 1)
KERNEL(pk_dep_nop, asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %0\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %0\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %0\n s_nop 0" : "+v"(p0) : "v"(pa));, 4)
KERNEL(fma_nop1, asm volatile("v_fma_f32 %0, %0, %1, %0\n s_nop 1\n v_fma_f32 %0, %0, %1, %0\n s_nop 1\n v_fma_f32 %0, %0, %1, %0\n s_nop 1\n v_fma_f32 %0, %0, %1, %0\n s_nop 1" : "+v"(x0) : "v"(a));, 4)
KERNEL(fma_salu, asm volatile("v_fma_f32 %0, %0, %1, %0\n s_add_u32 s20, s20, 1\n v_fma_f32 %0, %0, %1, %0\n s_add_u32 s20, s20, 1\n v_fma_f32 %0, %0, %1, %0\n s_add_u32 s20, s20, 1\n v_fma_f32 %0, %0, %1, %0\n s_add_u32 s20, s20, 1" : "+v"(x0) : "v"(a) : "s20");, 4)
KERNEL(fma_2salu, asm volatile("v_fma_f32 %0, %0, %1, %0\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n v_fma_f32 %0, %0, %1, %0\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n v_fma_f32 %0, %0, %1, %0\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n v_fma_f32 %0, %0, %1, %0\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1" : "+v"(x0) : "v"(a) : "s20", "s21");, 4)
KERNEL(fma_waitcnt, asm volatile("v_fma_f32 %0, %0, %1, %0\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %0\n s_waitcnt vmcnt(0)\n v_fma_f32 %0, %0, %1, %0\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %0\n s_waitcnt vmcnt(0)" : "+v"(x0) : "v"(a));, 4)
KERNEL(exp_nop_fma, asm volatile("v_exp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %0\n v_exp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %0" : "+v"(x0) : "v"(a));, 2)
KERNEL(exp2_pk, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n s_nop 0\n v_pk_add_f32 %2, %2, %3" : "+v"(x0), "+v"(x1), "+v"(p0) : "v"(pa));, 1)
KERNEL(dswrite_only, asm volatile("ds_write_b32 %1, %0\n ds_write_b32 %1, %0\n ds_write_b32 %1, %0\n ds_write_b32 %1, %0" : : "v"(x0), "v"(addr) : "memory");, 4)
KERNEL(dsread_nowait, asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n v_fma_f32 %3, %3, %4, %3\n v_fma_f32 %3, %3, %4, %3" : "=v"(q0), "=v"(q1) : "v"(addr), "v"(x0), "v"(a) : "memory");, 4)
KERNEL(barrier1, asm volatile("s_barrier" ::: "memory");, 1)

#define RUN(name) run(#name, (void*)name, name##_per)
static float* d;
static void run(const char* nm, void* fn, int per_iter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 1000;
  float a = 0.999f;
  void* args[] = {&d, (void*)&iters, &a};
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernel(fn, dim3(625), dim3(64), args, 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double n = (double)iters * per_iter;
  printf("%-14s %8.3f ms  %.3f ns/unit  %.2f cycles@2.4GHz\n", nm, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  hipMalloc(&d, 1 << 24);
  RUN(fma_dep); RUN(fma_ind); RUN(pk_dep); RUN(pk_ind); RUN(pk_mul_ind); RUN(exp_dep); RUN(exp_ind); RUN(exp_fma_mix);
  RUN(rcp_ind); RUN(dpp_dep); RUN(dpp_ind); RUN(fma_dpp_src); RUN(dpp_then_fma); RUN(lds_rt); RUN(lds_read_dep); RUN(pk_dep_nop); RUN(fma_nop1); RUN(fma_waitcnt); RUN(exp_nop_fma); RUN(exp2_pk); RUN(dswrite_only); RUN(dsread_nowait); RUN(barrier1);
  return 0;
}
