// Instruction-rate microbenchmark for the ops of the scan kernel's inner loop (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
// Reports wave64 lane-ops/s per variant against the 78.6 T lane-op/s VALU peak (256 CU x 4 SIMD x 32 x 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t c[8];
    for (int i = 0; i < 8; ++i) c[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u + blockIdx.x;
    uint32_t acc = 64;
    uint32_t g = seed;
    for (int it = 0; it < iters; ++it) {
        g = g * 1664525u + 1013904223u; // scalar
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            uint32_t y, o, d;
            if (MODE == 0) { // xor only
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(y) : "s"(g), "v"(c[r]));
                acc ^= y;
            } else if (MODE == 1) { // bcnt only
                asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d) : "v"(c[r] ^ 0));
                acc ^= d;
            } else if (MODE == 2) { // sdwa or only
                asm volatile("v_or_b32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(o) : "v"(c[r]));
                acc ^= o;
            } else if (MODE == 3) { // xor + sdwa + bcnt + min (the scan mix)
                y = c[r] ^ g;
                asm("v_or_b32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(o) : "v"(y));
                d = __builtin_popcount(o);
                acc = d < acc ? d : acc;
            } else if (MODE == 4) { // xor + lshr + and_or + bcnt + min (no sdwa)
                y = c[r] ^ g;
                o = (y | (y >> 16)) & 0xFFFFu;
                d = __builtin_popcount(o);
                acc = d < acc ? d : acc;
            } else if (MODE == 5) { // plain add chain-free: v_add_u32 x1 per candidate
                acc += c[r] ^ g;
            }
        }
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc; // keep
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE> void run(const char *name, int ops_per_cand, uint32_t *d_out)
{
    const int iters = 20000, blocks = 256 * 8;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double cands = double(iters) * 8 * 256 * blocks;
    printf("%-34s %8.3f ms  %7.2f Tcand/s  ~%6.2f T lane-ops/s (%d VALU ops/cand assumed)\n", name, ms,
           cands / ms / 1e9, cands * ops_per_cand / ms / 1e9, ops_per_cand);
}

int main()
{
    uint32_t *d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * 4 + 64);
    run<5>("xor+add (2 ops)", 2, d_out);
    run<0>("v_xor + xor-acc (2 ops)", 2, d_out);
    run<1>("v_bcnt + xor-acc (2 ops)", 2, d_out);
    run<2>("v_or_sdwa + xor-acc (2 ops)", 2, d_out);
    run<3>("scan mix sdwa (3.5 ops)", 4, d_out);
    run<4>("scan mix no-sdwa (4.5 ops)", 5, d_out);
    return 0;
}
