// Second part of the instruction-issue measurement (see issue_rate_bench.hip): WHY an independent v_fma_f32 stream
// issues at half the rate of v_add_f32 on gfx950.  Whole timed loops are single asm blocks with explicit register
// numbers so operand banks (VGPR index mod 4) and operand kinds (VGPR / SGPR / literal) are under control.
// One JSON object per line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static constexpr int kTrips = 2048;   // 32 measured instructions per trip

#define I3(op, d, a, b) op " v" #d ", v" #d ", v" #a ", v" #b "\n"
#define I2(op, d, a) op " v" #d ", v" #d ", v" #a "\n"
#define X4(x) x x x x
// eight destination registers, 4 rounds = 32 instructions
#define BODY8(M, r0, r1, r2, r3, r4, r5, r6, r7) X4(M(r0) M(r1) M(r2) M(r3) M(r4) M(r5) M(r6) M(r7))
#define CLOB "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
             "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "s20", "s21", "s22", "s24", "s25", "scc", "vcc"
#define PRO "v_mov_b32 v4, %1\n v_mov_b32 v5, %2\n v_mov_b32 v40, %2\n v_mov_b32 v41, %1\n s_mov_b32 s21, 0x3f7fbe77\n s_mov_b32 s22, 0x38d1b717\n" \
            "v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n" \
            "v_mov_b32 v16, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v28, %0\n" \
            "v_mov_b32 v32, %0\n v_mov_b32 v36, %0\n v_cmp_lt_f32 vcc, v4, v5\n v_cmp_lt_f32 s[24:25], v5, v4\n s_nop 4\n s_mov_b32 s20, %3\n 1:\n"
#define EPI "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" \
            "v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10\n v_add_f32 %0, %0, v11\n v_add_f32 %0, %0, v12\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v20\n"

#define M_FMA_ROT(d) I3("v_fma_f32", d, 4, 5)            // src0 rotates over the four banks, m = v4 (bank 0), c = v5 (bank 1)
#define M_FMA_SAMEBANK(d) I3("v_fma_f32", d, 4, 40)      // with d in bank 0: all three sources in bank 0
#define M_FMAC(d) "v_fmac_f32 v" #d ", v4, v5\n"          // VOP2, still reads d
#define M_FMAAK(d) "v_fmaak_f32 v" #d ", v" #d ", v4, 0x38d1b717\n"   // literal addend: two VGPR reads
#define M_FMA_SGPR(d) "v_fma_f32 v" #d ", v" #d ", s21, v5\n"          // one SGPR source
#define M_FMA_2SAME(d) "v_fma_f32 v" #d ", v" #d ", v4, v4\n"          // two sources the same register
#define M_MUL(d) I2("v_mul_f32", d, 4)
#define M_ADD(d) I2("v_add_f32", d, 5)
#define M_MULADD(d) I2("v_mul_f32", d, 4) I2("v_add_f32", d, 5)        // what -ffp-contract=off makes of a*b+c (2 instructions: 64 per trip)
#define M_MAX3(d) I3("v_max3_f32", d, 4, 5)
#define M_CNDMASK(d) "v_cndmask_b32 v" #d ", v" #d ", v4, vcc\n"
#define M_CMP(d) "v_cmp_lt_f32 vcc, v" #d ", v4\n"
#define M_CNDMASK_SGPR(d) "v_cndmask_b32_e64 v" #d ", v" #d ", v4, s[24:25]\n"          // mask in an SGPR pair written once before the loop
#define M_CMP_CNDMASK(d) "v_cmp_lt_f32 vcc, v" #d ", v4\n v_cndmask_b32 v" #d ", v" #d ", v5, vcc\n"   // the pair real code issues (counted as 2)
#define M_CMP_SGPR(d) "v_cmp_lt_f32 s[24:25], v" #d ", v4\n"
#define M_MULHI(d) I2("v_mul_hi_u32", d, 4)
#define M_MAD64(d) "v_mad_u64_u32 v[" #d ":" #d "+1], vcc, v4, v5, v[" #d ":" #d "+1]\n"
#define M_LSHL(d) "v_lshlrev_b32 v" #d ", 1, v" #d "\n"
#define M_AND(d) I2("v_and_b32", d, 4)
#define M_ADD3(d) I3("v_add3_u32", d, 4, 5)
#define M_DIVSCALE(d) "v_div_scale_f32 v" #d ", vcc, v" #d ", v4, v" #d "\n"
#define M_DIVFIXUP(d) "v_div_fixup_f32 v" #d ", v" #d ", v4, v5\n"
#define M_DIVFMAS(d) "v_div_fmas_f32 v" #d ", v" #d ", v4, v5\n"

enum { V_FMA_ROT = 0, V_FMA_NOCONFLICT, V_FMA_SAMEBANK, V_FMAC, V_FMAAK, V_FMA_SGPR, V_FMA_2SAME, V_MUL, V_ADD, V_MULADD, V_MAX3, V_CNDMASK, V_CMP, V_MULHI,
       V_MAD64, V_LSHL, V_AND, V_ADD3, V_DIVSCALE, V_DIVFIXUP, V_DIVFMAS, V_LDS_CHASE, V_CNDMASK_SGPR, V_CMP_CNDMASK, V_CMP_SGPR, N_VAR };
static const char* var_name[N_VAR] = {
    "v_fma_f32 d,d,m,c  (d over banks 0-3, m bank 0, c bank 1)", "v_fma_f32 d,d,m,c  (d in banks 2/3 only: no two sources share a bank)",
    "v_fma_f32 d,d,m,c  (all three sources in bank 0)", "v_fmac_f32 d,m,c", "v_fmaak_f32 d,d,m,literal", "v_fma_f32 d,d,SGPR,c", "v_fma_f32 d,d,m,m",
    "v_mul_f32 d,d,m", "v_add_f32 d,d,c", "v_mul_f32 + v_add_f32 (a*b+c with contraction off; counted as 2)", "v_max3_f32 d,d,m,c", "v_cndmask_b32 d,d,m,vcc",
    "v_cmp_lt_f32 vcc,d,m", "v_mul_hi_u32 d,d,m", "v_mad_u64_u32", "v_lshlrev_b32 d,1,d", "v_and_b32 d,d,m", "v_add3_u32 d,d,m,c", "v_div_scale_f32", "v_div_fixup_f32",
    "v_div_fmas_f32", "ds_read_b32 dependent chain (LDS pointer chase; latency)", "v_cndmask_b32_e64 d,d,m,s[24:25] (mask written once)",
    "v_cmp_lt_f32 vcc + v_cndmask_b32 vcc pairs (counted as 2)", "v_cmp_lt_f32 s[24:25],d,m (VOP3 compare into an SGPR pair)"};
static const int var_instr_per_trip[N_VAR] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 64, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 64, 32};

template <int V>
__global__ void __launch_bounds__(256) k_var(float* out, unsigned long long* cycles, float seed) {
    __shared__ unsigned int lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (unsigned)(((i * 1237 + 64) & 4095) * 4);   // byte offsets, a permutation-ish walk
    __syncthreads();
    float x = seed + (float)threadIdx.x * 1e-3f;
    const float m = 0.999f + seed * 1e-9f, c = 1e-4f * seed;
    unsigned long long t0 = __builtin_readcyclecounter();
    if constexpr (V == V_FMA_ROT) asm volatile(PRO BODY8(M_FMA_ROT, 8, 9, 10, 11, 12, 13, 14, 15) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_FMA_NOCONFLICT) asm volatile(PRO BODY8(M_FMA_ROT, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_FMA_SAMEBANK) asm volatile(PRO BODY8(M_FMA_SAMEBANK, 8, 12, 16, 20, 24, 28, 32, 36) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_FMAC) asm volatile(PRO BODY8(M_FMAC, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_FMAAK) asm volatile(PRO BODY8(M_FMAAK, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_FMA_SGPR) asm volatile(PRO BODY8(M_FMA_SGPR, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_FMA_2SAME) asm volatile(PRO BODY8(M_FMA_2SAME, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_MUL) asm volatile(PRO BODY8(M_MUL, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_ADD) asm volatile(PRO BODY8(M_ADD, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_MULADD) asm volatile(PRO BODY8(M_MULADD, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_MAX3) asm volatile(PRO BODY8(M_MAX3, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_CNDMASK) asm volatile(PRO BODY8(M_CNDMASK, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_CMP) asm volatile(PRO BODY8(M_CMP, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_MULHI) asm volatile(PRO BODY8(M_MULHI, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_MAD64) asm volatile(PRO BODY8(M_MAD64, 8, 10, 12, 14, 16, 18, 20, 22) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_LSHL) asm volatile(PRO BODY8(M_LSHL, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_AND) asm volatile(PRO BODY8(M_AND, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_ADD3) asm volatile(PRO BODY8(M_ADD3, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_DIVSCALE) asm volatile(PRO BODY8(M_DIVSCALE, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_DIVFIXUP) asm volatile(PRO BODY8(M_DIVFIXUP, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_DIVFMAS) asm volatile(PRO BODY8(M_DIVFMAS, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_CNDMASK_SGPR) asm volatile(PRO BODY8(M_CNDMASK_SGPR, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_CMP_CNDMASK) asm volatile(PRO BODY8(M_CMP_CNDMASK, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_CMP_SGPR) asm volatile(PRO BODY8(M_CMP_SGPR, 10, 11, 14, 15, 18, 19, 22, 23) EPI : "+v"(x) : "v"(m), "v"(c), "n"(kTrips) : CLOB);
    else if constexpr (V == V_LDS_CHASE) {
        unsigned p = (threadIdx.x & 1023u) * 4u;
        for (int it = 0; it < kTrips; ++it) {
#pragma unroll
            for (int r = 0; r < 32; ++r) p = *(volatile unsigned*)((char*)lds + p);
        }
        x += (float)p;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = x;
    if ((threadIdx.x & 63) == 0) cycles[gid >> 6] = t1 - t0;
}

template <int V>
static void run_var(int n_cu, float* d_out, unsigned long long* d_cyc, std::vector<unsigned long long>& h_cyc) {
    for (int waves_per_simd : {1, 2, 4, 8}) {
        int blocks = n_cu * waves_per_simd;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_var<V>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1.0f);
        CK(hipDeviceSynchronize());
        const int reps = 5;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_var<V>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        size_t n_waves = (size_t)blocks * 4;
        CK(hipMemcpy(h_cyc.data(), d_cyc, n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double cyc = 0; for (size_t i = 0; i < n_waves; i++) cyc += (double)h_cyc[i];
        cyc /= (double)n_waves;
        double per_wave = (double)kTrips * var_instr_per_trip[V];
        double rate = per_wave * (double)n_waves / (ms * 1e-3);
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_instr_per_s_G\": %.1f, \"wave_cycles_per_instr\": %.3f}\n",
               var_name[V], waves_per_simd, ms, rate * 1e-9, cyc / per_wave);
        fflush(stdout);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
}

template <int V> static void run_all(int n_cu, float* d_out, unsigned long long* d_cyc, std::vector<unsigned long long>& h) {
    run_var<V>(n_cu, d_out, d_cyc, h);
    if constexpr (V + 1 < N_VAR) run_all<V + 1>(n_cu, d_out, d_cyc, h);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int n_cu = p.multiProcessorCount;
    printf("{\"arch\": \"%s\", \"cus\": %d, \"clock_MHz\": %d}\n", p.gcnArchName, n_cu, p.clockRate / 1000);
    size_t max_threads = (size_t)n_cu * 8 * 256;
    float* d_out; unsigned long long* d_cyc;
    CK(hipMalloc(&d_out, max_threads * sizeof(float)));
    CK(hipMalloc(&d_cyc, max_threads / 64 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h_cyc(max_threads / 64);
    run_all<0>(n_cu, d_out, d_cyc, h_cyc);
    return 0;
}
