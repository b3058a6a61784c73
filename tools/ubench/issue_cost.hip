// Micro-benchmark: issue cost (cycles per wave instruction) of the instruction kinds the MPC kernels lean on, for ONE
// wavefront on an otherwise idle CU (the bench shape).  hipcc --offload-arch=gfx950 -O2 issue_cost.hip -o issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef double d2v __attribute__((ext_vector_type(2)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define KERNEL(name, setup, body)                                                          \
    __global__ void name(long long *out, double *sink, int iters)                           \
    {                                                                                       \
        __shared__ double lds[1024];                                                        \
        for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 1.0 + i;                      \
        __syncthreads();                                                                    \
        double a = 1.0 + threadIdx.x * 1e-3, b = 0.999, c = 0.5, d = 0.25, e = 2.0, f = 3.0, g = 4.0, h = 5.0; \
        int ia = threadIdx.x, ib = 3;                                                       \
        unsigned laddr = (threadIdx.x & 7) * 16;                                            \
        (void)laddr; (void)ia; (void)ib;                                                    \
        setup;                                                                              \
        long long t0 = __builtin_readcyclecounter();                                        \
        for (int it = 0; it < iters; ++it) { body; }                                        \
        long long t1 = __builtin_readcyclecounter();                                        \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                    \
        sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h + ia + ib;      \
    }

// 64 independent FMAs over 8 accumulators
KERNEL(k_fma_indep, , REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                                          : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(f));))
KERNEL(k_fma_dep, , REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(f));))
KERNEL(k_mul_f64, , REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                                        : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));))
KERNEL(k_add_f64, , REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                                        : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));))
KERNEL(k_fma_f32, float x0 = 1.f + threadIdx.x; float x1 = 2.f; float x2 = 3.f; float x3 = 4.f; float y = 0.999f;,
       REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y)););
       a += x0 + x1 + x2 + x3;)
KERNEL(k_mov_b32, , REP16(asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n v_mov_b32 %0, %3\n v_mov_b32 %1, %2" : "+v"(ia), "+v"(ib) : "v"(ia), "v"(ib));))
KERNEL(k_cndmask, , REP16(asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc"
                                        : "+v"(ia), "+v"(ib) : "v"(ia) : "vcc");))
KERNEL(k_readlane, int s0 = 0; int s1 = 0;, REP16(asm volatile("v_readlane_b32 %0, %2, 5\n v_readlane_b32 %1, %3, 7\n v_readlane_b32 %0, %3, 9\n v_readlane_b32 %1, %2, 11"
                                         : "=s"(s0), "=s"(s1) : "v"(ia), "v"(ib));); ia += s0 + s1;)
// chain: FMA result -> 2 readlane -> FMA (the back-substitution / Cholesky broadcast pattern), compiler-generated
__device__ __forceinline__ double rdl(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
#define RLCHAIN a = fma(rdl(a, 5), b, a);
KERNEL(k_readlane_chain, , REP16(RLCHAIN))
// same, 4 independent chains interleaved
#define RLCHAIN4 a = fma(rdl(a, 5), b, a); c = fma(rdl(c, 6), b, c); d = fma(rdl(d, 7), b, d); e = fma(rdl(e, 8), b, e);
KERNEL(k_readlane_chain4, , REP16(RLCHAIN4))
KERNEL(k_dpp_mov, , REP16(asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                                        "v_mov_b32_dpp %0, %3 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_mirror row_mask:0xf bank_mask:0xf"
                                        : "+v"(ia), "+v"(ib) : "v"(ia), "v"(ib));))
// DPP reduction step as compiled for f64: 2 dpp movs + dependent add
__device__ __forceinline__ double dppx(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
#define DPPSTEP a = a * b + dppx(a);
KERNEL(k_dpp_step, , REP16(DPPSTEP))
KERNEL(k_accvgpr, , REP16(asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_read_b32 %1, a1" : "+v"(ia), "+v"(ib) : : "a0", "a1");))
// LDS broadcast reads: 4 x b128, all lanes the same address, then wait
KERNEL(k_lds_b128_bcast, d2v r0_; d2v r1_; d2v r2_; d2v r3_; unsigned z_ = 0;,
       REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                          : "=v"(r0_), "=v"(r1_), "=v"(r2_), "=v"(r3_) : "v"(z_)); a += r0_.x + r1_.x + r2_.x + r3_.x;))
// same without the per-group wait: 16 reads in flight
KERNEL(k_lds_b128_burst, d2v r0_; d2v r1_; d2v r2_; d2v r3_; unsigned z_ = 0;,
       REP4(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n"
                         "ds_read_b128 %0, %4 offset:64\n ds_read_b128 %1, %4 offset:80\n ds_read_b128 %2, %4 offset:96\n ds_read_b128 %3, %4 offset:112\n"
                         "ds_read_b128 %0, %4 offset:128\n ds_read_b128 %1, %4 offset:144\n ds_read_b128 %2, %4 offset:160\n ds_read_b128 %3, %4 offset:176\n"
                         "ds_read_b128 %0, %4 offset:192\n ds_read_b128 %1, %4 offset:208\n ds_read_b128 %2, %4 offset:224\n ds_read_b128 %3, %4 offset:240\n s_waitcnt lgkmcnt(0)"
                         : "=v"(r0_), "=v"(r1_), "=v"(r2_), "=v"(r3_) : "v"(z_)); a += r0_.x + r1_.x + r2_.x + r3_.x;))
// one LDS round trip: write, wait, read, wait (dependent)
KERNEL(k_lds_roundtrip, unsigned la_ = threadIdx.x * 8;,
       REP16(asm volatile("ds_write_b64 %1, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(la_));))
KERNEL(k_lds_write_b128_1lane, unsigned la_ = 0; d2v w_; w_.x = 1.0; w_.y = 2.0;,
       REP16(if (threadIdx.x == 3) asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:16\n ds_write_b128 %0, %1 offset:32\n ds_write_b128 %0, %1 offset:48" : : "v"(la_), "v"(w_));))
KERNEL(k_lds_write_b64_1lane, unsigned la_ = 0; double w_ = 1.0;,
       REP16(if (threadIdx.x == 3) asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:8\n ds_write_b64 %0, %1 offset:16\n ds_write_b64 %0, %1 offset:24" : : "v"(la_), "v"(w_));))
KERNEL(k_lds_write2_b64_1lane, unsigned la_ = 0; double w_ = 1.0;,
       REP16(if (threadIdx.x == 3) asm volatile("ds_write2_b64 %0, %1, %1 offset0:0 offset1:1\n ds_write2_b64 %0, %1, %1 offset0:2 offset1:3\n ds_write2_b64 %0, %1, %1 offset0:4 offset1:5\n ds_write2_b64 %0, %1, %1 offset0:6 offset1:7" : : "v"(la_), "v"(w_));))
KERNEL(k_lds_write_b128_all, unsigned la_ = threadIdx.x * 64; d2v w_; w_.x = 1.0; w_.y = 2.0;,
       REP16(asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:16\n ds_write_b128 %0, %1 offset:32\n ds_write_b128 %0, %1 offset:48" : : "v"(la_), "v"(w_));))
KERNEL(k_lds_write_b32_1lane, unsigned la_ = 0; int w_ = 1;,
       REP16(if (threadIdx.x == 3) asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:4\n ds_write_b32 %0, %1 offset:8\n ds_write_b32 %0, %1 offset:12" : : "v"(la_), "v"(w_));))
KERNEL(k_lds_read_b64_bcast, double r0_; double r1_; double r2_; double r3_; unsigned z_ = 0;,
       REP4(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n"
                         "ds_read_b64 %0, %4 offset:32\n ds_read_b64 %1, %4 offset:40\n ds_read_b64 %2, %4 offset:48\n ds_read_b64 %3, %4 offset:56\n"
                         "ds_read_b64 %0, %4 offset:64\n ds_read_b64 %1, %4 offset:72\n ds_read_b64 %2, %4 offset:80\n ds_read_b64 %3, %4 offset:88\n"
                         "ds_read_b64 %0, %4 offset:96\n ds_read_b64 %1, %4 offset:104\n ds_read_b64 %2, %4 offset:112\n ds_read_b64 %3, %4 offset:120\n s_waitcnt lgkmcnt(0)"
                         : "=v"(r0_), "=v"(r1_), "=v"(r2_), "=v"(r3_) : "v"(z_)); a += r0_ + r1_ + r2_ + r3_;))
KERNEL(k_sbranch_taken, int s_ = 1;, REP16(asm volatile("s_cmp_lg_u32 %0, 77\n s_cbranch_scc1 1f\n s_add_i32 %0, %0, 2\n 1:\n s_nop 0" : "+s"(s_) : : "scc");); ia += s_;)
KERNEL(k_execz, , REP16(asm volatile("s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_mov_b32 %0, %0\n 1:\n s_or_b64 exec, exec, s[20:21]" : "+v"(ia) : : "s20", "s21", "vcc");))
KERNEL(k_salu, int s_ = 1;, REP16(asm volatile("s_add_i32 %0, %0, 3\n s_xor_b32 %0, %0, 5\n s_add_i32 %0, %0, 3\n s_xor_b32 %0, %0, 5" : "+s"(s_) : : "scc");); ia += s_;)
KERNEL(k_sbranch, int s_ = 1;, REP16(asm volatile("s_cmp_eq_u32 %0, 77\n s_cbranch_scc1 1f\n s_add_i32 %0, %0, 2\n 1:\n s_nop 0" : "+s"(s_) : : "scc");); ia += s_;)
KERNEL(k_bpermute, unsigned ad_ = ((threadIdx.x + 2) & 63) * 4;, REP16(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(ia) : "v"(ad_));))
KERNEL(k_rsq_f64, , REP16(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));))
typedef double v4d_ __attribute__((ext_vector_type(4)));
#define MFMA1 acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc_, 0, 0, 0);
KERNEL(k_mfma_f64, v4d_ acc_; acc_[0] = 0.0; acc_[1] = 0.0; acc_[2] = 0.0; acc_[3] = 0.0;, REP16(MFMA1) a += acc_[0];)

struct Entry { const char *name; void (*fn)(long long *, double *, int); int per_iter; const char *what; };

int main()
{
    long long *out; double *sink;
    hipMalloc(&out, 1024 * sizeof(long long));
    hipMalloc(&sink, 1024 * 64 * sizeof(double));
    std::vector<Entry> es = {
        {"v_fma_f64 independent", k_fma_indep, 64, "instr"}, {"v_fma_f64 dependent chain", k_fma_dep, 64, "instr"},
        {"v_mul_f64 independent", k_mul_f64, 64, "instr"}, {"v_add_f64 independent", k_add_f64, 64, "instr"},
        {"v_fma_f32 independent", k_fma_f32, 64, "instr"}, {"v_mov_b32", k_mov_b32, 64, "instr"}, {"v_cndmask_b32", k_cndmask, 64, "instr"},
        {"v_readlane_b32", k_readlane, 64, "instr"}, {"FMA -> 2 readlane -> FMA chain", k_readlane_chain, 16, "group"}, {"same, 4 chains interleaved", k_readlane_chain4, 64, "group"}, {"v_mov_b32_dpp", k_dpp_mov, 64, "instr"},
        {"dpp step (2 dpp mov + dependent fma_f64)", k_dpp_step, 16, "group"}, {"v_accvgpr_write/read", k_accvgpr, 64, "instr"},
        {"4 x ds_read_b128 broadcast + wait", k_lds_b128_bcast, 16, "group"}, {"16 x ds_read_b128 broadcast + wait", k_lds_b128_burst, 4, "group"},
        {"ds_write_b64 -> wait -> ds_read_b64 -> wait", k_lds_roundtrip, 16, "group"}, {"4 x ds_write_b128 by one lane (no wait)", k_lds_write_b128_1lane, 16, "group"},
        {"4 x ds_write_b64 by one lane", k_lds_write_b64_1lane, 16, "group"}, {"4 x ds_write2_b64 by one lane", k_lds_write2_b64_1lane, 16, "group"},
        {"4 x ds_write_b128 all lanes", k_lds_write_b128_all, 16, "group"}, {"4 x ds_write_b32 by one lane", k_lds_write_b32_1lane, 16, "group"},
        {"16 x ds_read_b64 broadcast + wait", k_lds_read_b64_bcast, 4, "group"},
        {"s_cmp + s_cbranch (taken)", k_sbranch_taken, 16, "group"}, {"saveexec + execz (not taken) + restore", k_execz, 16, "group"},
        {"s_add/s_xor dependent", k_salu, 64, "instr"}, {"s_cmp + s_cbranch (not taken) + s_add", k_sbranch, 16, "group"},
        {"ds_bpermute_b32 + wait", k_bpermute, 16, "group"}, {"v_rsq_f64 dependent", k_rsq_f64, 16, "instr"},
        {"v_mfma_f64_16x16x4 dependent", k_mfma_f64, 16, "instr"},
    };
    const int iters = 2000;
    for (int nb : {1, 1024}) {
        printf("---- %d workgroup(s) of one wave\n", nb);
        for (auto &e : es) {
            hipLaunchKernelGGL(e.fn, dim3(nb), dim3(64), 0, 0, out, sink, 10);
            hipLaunchKernelGGL(e.fn, dim3(nb), dim3(64), 0, 0, out, sink, iters);
            hipDeviceSynchronize();
            std::vector<long long> h(nb);
            hipMemcpy(h.data(), out, nb * sizeof(long long), hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            printf("%-48s %8.2f s_memtime ticks per %s\n", e.name, s / nb / iters / e.per_iter, e.what);
        }
    }
    return 0;
}
