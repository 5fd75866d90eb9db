// Instruction-issue ceiling of gfx950 (MI355X), measured: the constant DESIGN.md and bench.py price VALU-bound
// kernels against.  Standalone (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/issue_rate_bench.hip).
//
// For each instruction stream - independent v_fma_f32 chains, ONE dependent chain, v_rcp_f32, the IEEE division
// sequence hipcc emits for `a / b`, v_sqrt_f32 / IEEE sqrt, v_mul_lo_u32 / 64-bit integer multiply (the sampler's
// fixed point), ds_read_b128 - and for 1 / 2 / 4 / 8 waves per SIMD on every CU, it reports wave-instructions per
// second over the whole chip and cycles per wave-instruction per SIMD (from s_memtime, the shader clock).
// One JSON object per line on stdout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static constexpr int kIters = 2048;   // loop trips; every trip issues kUnroll instructions of the measured kind
static constexpr int kUnroll = 32;

enum Kind { FMA_INDEP8 = 0, FMA_DEP1, FMA_INDEP2, RCP, DIV_IEEE, SQRT_HW, SQRT_IEEE, MUL_LO_U32, MUL_U64, ADD_F32, LDS_B128, FMA_PK, N_KIND };
static const char* kind_name[N_KIND] = {"v_fma_f32 x8 independent chains", "v_fma_f32 one dependent chain", "v_fma_f32 x2 independent chains",
    "v_rcp_f32 (independent)", "IEEE f32 division a/b (independent)", "v_sqrt_f32 (independent)", "IEEE sqrtf (independent)",
    "v_mul_lo_u32 (independent)", "u64 multiply (independent)", "v_add_f32 x8 independent chains", "ds_read_b128 (independent)", "v_pk_fma_f32 x8 independent chains"};

template <int K>
__global__ void __launch_bounds__(256) k_issue(float* out, unsigned long long* cycles, float seed) {
    __shared__ float4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_float4(seed, seed * 0.5f, 1.0f, 2.0f);
    __syncthreads();
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
    unsigned int u[8];
#pragma unroll
    for (int i = 0; i < 8; i++) u[i] = (unsigned)threadIdx.x * 2654435761u + i;
    unsigned long long w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = 0x9E3779B97F4A7C15ull * (threadIdx.x + i + 1);
    const float m = 0.999f + seed * 1e-9f, c = 1e-4f * seed;
    const unsigned kmul = 747796405u + (unsigned)seed * 2u;
    const unsigned long long kmul64 = 6364136223846793005ull + (unsigned long long)seed * 2ull;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; it++) {
        if constexpr (K == FMA_INDEP8) {
#pragma unroll
            for (int r = 0; r < kUnroll / 8; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if constexpr (K == FMA_DEP1) {
#pragma unroll
            for (int r = 0; r < kUnroll; r++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
        } else if constexpr (K == FMA_INDEP2) {
#pragma unroll
            for (int r = 0; r < kUnroll / 2; r++) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(m), "v"(c)); }
        } else if constexpr (K == RCP) {
#pragma unroll
            for (int r = 0; r < kUnroll / 8; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if constexpr (K == DIV_IEEE) {
#pragma unroll
            for (int r = 0; r < kUnroll / 8; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) { a[i] = m / a[i]; asm volatile("" : "+v"(a[i])); }
        } else if constexpr (K == SQRT_HW) {
#pragma unroll
            for (int r = 0; r < kUnroll / 8; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        } else if constexpr (K == SQRT_IEEE) {
#pragma unroll
            for (int r = 0; r < kUnroll / 8; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) { a[i] = __builtin_sqrtf(a[i]); asm volatile("" : "+v"(a[i])); }
        } else if constexpr (K == MUL_LO_U32) {
#pragma unroll
            for (int r = 0; r < kUnroll / 8; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(kmul));
        } else if constexpr (K == MUL_U64) {
#pragma unroll
            for (int r = 0; r < kUnroll / 4; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) { w[i] = w[i] * kmul64; asm volatile("" : "+v"(w[i])); }
        } else if constexpr (K == ADD_F32) {
#pragma unroll
            for (int r = 0; r < kUnroll / 8; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if constexpr (K == LDS_B128) {
#pragma unroll
            for (int r = 0; r < kUnroll / 8; r++) {
                float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = lds[(threadIdx.x + 64 * i + it) & 1023];
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] += v[i].x + v[i].w;   // plus 16 v_add per 8 ds_read
            }
        } else if constexpr (K == FMA_PK) {
            typedef float float2v __attribute__((ext_vector_type(2)));
            float2v p[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { p[i].x = a[2 * i]; p[i].y = a[2 * i + 1]; }
            float2v mm = {m, m}, cc = {c, c};
#pragma unroll
            for (int r = 0; r < kUnroll / 4; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(mm), "v"(cc));
#pragma unroll
            for (int i = 0; i < 4; i++) { a[2 * i] = p[i].x; a[2 * i + 1] = p[i].y; }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + (float)u[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += (float)w[i];
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) cycles[gid >> 6] = t1 - t0;
}

template <int K>
static void run_kind(int n_cu, float* d_out, unsigned long long* d_cyc, std::vector<unsigned long long>& h_cyc) {
    // instructions of the measured kind per loop trip (what the compiler must keep: checked with the ISA dump in profiles/)
    int per_trip = kUnroll;
    if (K == DIV_IEEE || K == SQRT_IEEE || K == MUL_U64) per_trip = kUnroll;   // reported per SOURCE operation
    for (int waves_per_simd : {1, 2, 4, 8}) {
        int blocks = n_cu * waves_per_simd;          // 256 threads = 4 waves = one per SIMD of a CU
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_issue<K>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1.0f);   // warm-up
        CK(hipDeviceSynchronize());
        const int reps = 5;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_issue<K>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        size_t n_waves = (size_t)blocks * 4;
        CK(hipMemcpy(h_cyc.data(), d_cyc, n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double cyc = 0; for (size_t i = 0; i < n_waves; i++) cyc += (double)h_cyc[i];
        cyc /= (double)n_waves;
        double ops_per_wave = (double)kIters * per_trip;
        double total = ops_per_wave * (double)n_waves;
        double rate = total / (ms * 1e-3);                 // source operations per second, whole chip (wave granularity)
        // cycles of SIMD time per wave-operation: a wave's loop took `cyc` shader cycles while sharing its SIMD with
        // waves_per_simd - 1 others
        double cyc_per_op_simd = cyc / ops_per_wave / waves_per_simd;
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"blocks\": %d, \"ms\": %.4f, \"wave_ops_per_s_G\": %.1f, "
               "\"cycles_per_wave_op_per_simd\": %.3f, \"wave_cycles\": %.0f, \"eff_clock_GHz\": %.3f}\n",
               kind_name[K], waves_per_simd, blocks, ms, rate * 1e-9, cyc_per_op_simd, cyc, cyc / (ms * 1e-3) * 1e-9);
        fflush(stdout);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int n_cu = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_MHz\": %d}\n", p.name, p.gcnArchName, n_cu, p.clockRate / 1000);
    size_t max_threads = (size_t)n_cu * 8 * 256;
    float* d_out; unsigned long long* d_cyc;
    CK(hipMalloc(&d_out, max_threads * sizeof(float)));
    CK(hipMalloc(&d_cyc, max_threads / 64 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h_cyc(max_threads / 64);
    run_kind<FMA_INDEP8>(n_cu, d_out, d_cyc, h_cyc);
    run_kind<FMA_INDEP2>(n_cu, d_out, d_cyc, h_cyc);
    run_kind<FMA_DEP1>(n_cu, d_out, d_cyc, h_cyc);
    run_kind<ADD_F32>(n_cu, d_out, d_cyc, h_cyc);
    run_kind<FMA_PK>(n_cu, d_out, d_cyc, h_cyc);
    run_kind<RCP>(n_cu, d_out, d_cyc, h_cyc);
    run_kind<DIV_IEEE>(n_cu, d_out, d_cyc, h_cyc);
    run_kind<SQRT_HW>(n_cu, d_out, d_cyc, h_cyc);
    run_kind<SQRT_IEEE>(n_cu, d_out, d_cyc, h_cyc);
    run_kind<MUL_LO_U32>(n_cu, d_out, d_cyc, h_cyc);
    run_kind<MUL_U64>(n_cu, d_out, d_cyc, h_cyc);
    run_kind<LDS_B128>(n_cu, d_out, d_cyc, h_cyc);
    CK(hipFree(d_out)); CK(hipFree(d_cyc));
    return 0;
}
