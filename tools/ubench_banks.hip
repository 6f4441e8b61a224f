// VGPR bank conflicts on gfx950: does an instruction whose source registers share (register number mod 4) cost more?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_banks.hip -o tools/_build/ubench_banks && tools/_build/ubench_banks
// Hard-coded registers (values are garbage: only the issue rate is measured), 8 waves per SIMD, s_memtime clock.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

#define REP8(A, B, C) \
    asm volatile("v_bitop3_b32 v40, " A ", " B ", " C " bitop3:0x96\n v_bitop3_b32 v41, " A ", " B ", " C " bitop3:0xe8\n" \
                 "v_bitop3_b32 v42, " A ", " B ", " C " bitop3:0x96\n v_bitop3_b32 v43, " A ", " B ", " C " bitop3:0xe8\n" \
                 "v_bitop3_b32 v44, " A ", " B ", " C " bitop3:0x96\n v_bitop3_b32 v45, " A ", " B ", " C " bitop3:0xe8\n" \
                 "v_bitop3_b32 v46, " A ", " B ", " C " bitop3:0x96\n v_bitop3_b32 v47, " A ", " B ", " C " bitop3:0xe8\n" \
                 ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47")
#define REP8_2(A, B) \
    asm volatile("v_xor_b32_e32 v40, " A ", " B "\n v_and_b32_e32 v41, " A ", " B "\n" \
                 "v_xor_b32_e32 v42, " A ", " B "\n v_and_b32_e32 v43, " A ", " B "\n" \
                 "v_xor_b32_e32 v44, " A ", " B "\n v_and_b32_e32 v45, " A ", " B "\n" \
                 "v_xor_b32_e32 v46, " A ", " B "\n v_and_b32_e32 v47, " A ", " B "\n" \
                 ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47")

// mixtures of register-only (V) and SGPR-reading (S) instructions, 64 per iteration
#define V1 "v_bitop3_b32 v40, v0, v1, v2 bitop3:0x96\n"
#define V2 "v_bitop3_b32 v41, v3, v4, v5 bitop3:0xe8\n"
#define S1 "v_xor_b32_e32 v42, s20, v6\n"
#define S2 "v_bitop3_b32 v43, v7, v9, s21 bitop3:0xf6\n"
#define B1 "s_bfe_i32 s22, s24, 0x10005\n"
#define B2 "s_bfe_i32 s23, s24, 0x10015\n"
#define CLOB ::: "v40", "v41", "v42", "v43", "s22", "s23", "scc"
#define X4(a) a a a a
#define X8(a) a a a a a a a a
#define X16(a) X4(a) X4(a) X4(a) X4(a)

template <int MODE>
__global__ __launch_bounds__(1024, 8) void k(unsigned long long *stamps, int iters)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) REP8("v0", "v1", "v2");      // three banks
            if (MODE == 1) REP8("v0", "v4", "v8");      // one bank
            if (MODE == 2) REP8("v0", "v4", "v1");      // two share a bank
            if (MODE == 3) REP8("v0", "v1", "s20");     // two VGPRs (two banks) + SGPR
            if (MODE == 4) REP8("v0", "v4", "s20");     // two VGPRs (one bank) + SGPR
            if (MODE == 5) REP8_2("v0", "v1");
            if (MODE == 6) REP8_2("v0", "v4");
            if (MODE == 7) REP8_2("s20", "v0");
            if (MODE == 8) REP8("v0", "v0", "v1");      // same register twice
            if (MODE == 9) REP8("v0", "v1", "v6");      // banks 0 1 2, not adjacent
            if (MODE == 10) asm volatile(X4(V1 S1) CLOB);                 // V S V S ...
            if (MODE == 11) asm volatile(X4(B1 B2 V1 S1 V2 S2) CLOB);    // the pipelined step's pattern: 2 SALU, V S V S
            if (MODE == 12) asm volatile(X4(V1 V2 S1 S2) CLOB);          // V V S S
            if (MODE == 13) asm volatile(X4(V1 V2 V1 V2) X4(S1 S2 S1 S2) CLOB); // 16 V then 16 S (per 32)
            if (MODE == 14) asm volatile(X4(V1 S1 V2 S2) CLOB);           // V S V S, two kinds each
            if (MODE == 15) asm volatile(X4(B1 V1 B2 S1) CLOB);           // SALU between every vector instruction
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63u) == 0) {
        const uint32_t w = blockIdx.x * 16 + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int MODE> static void run(const char *name, unsigned long long *d_st)
{
    const int blocks = 512, iters = 20000;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d_st, iters);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d_st, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> st(2 * blocks * 16);
    CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (size_t w = 0; w < st.size() / 2; ++w) { cyc += st[2 * w]; real += st[2 * w + 1]; }
    const double ghz = cyc / real * 0.1;
    const double per_iter = (MODE == 10) ? 64 : (MODE == 11) ? 128 : (MODE == 12 || MODE == 14) ? 128 : (MODE == 13) ? 256 : (MODE == 15) ? 64 : 64;
    const double instr = double(iters) * per_iter * blocks * 16;
    printf("%-44s %8.3f ms  clock %.2f GHz  VALU/cycle/SIMD %.3f  %.2f G instr/s/SIMD\n", name, ms, ghz,
           instr / (ms * 1e-3 * ghz * 1e9 * 1024.0), instr / (ms * 1e-3) / 1024.0 / 1e9);
}

int main()
{
    unsigned long long *d_st;
    CHECK(hipMalloc(&d_st, 512 * 16 * 16));
    run<0>("bitop3 v0 v1 v2   (three banks)", d_st);
    run<9>("bitop3 v0 v1 v6   (three banks)", d_st);
    run<2>("bitop3 v0 v4 v1   (two share a bank)", d_st);
    run<1>("bitop3 v0 v4 v8   (one bank)", d_st);
    run<8>("bitop3 v0 v0 v1   (same register twice)", d_st);
    run<3>("bitop3 v0 v1 s20  (two banks + SGPR)", d_st);
    run<4>("bitop3 v0 v4 s20  (one bank + SGPR)", d_st);
    run<5>("xor/and v0 v1     (two banks)", d_st);
    run<6>("xor/and v0 v4     (one bank)", d_st);
    run<7>("xor/and s20 v0    (SGPR + VGPR)", d_st);
    run<10>("mix V S V S", d_st);
    run<14>("mix V S V S (two kinds each)", d_st);
    run<12>("mix V V S S", d_st);
    run<13>("mix 16 V then 16 S", d_st);
    run<11>("mix bfe bfe V S V S   (VALU count only)", d_st);
    run<15>("mix bfe V bfe S       (VALU count only)", d_st);
    return 0;
}
