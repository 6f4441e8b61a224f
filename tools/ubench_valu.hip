// Instruction-rate microbenchmark for the ops of the scan kernel's inner loop (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
// Reports wave64 lane-ops/s per variant against the 78.6 T lane-op/s VALU peak (256 CU x 4 SIMD x 32 x 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// ---- bit-sliced variant: a lane holds 32 candidates as 32 bit planes (16 positions x {low,high} bit) ----
__device__ __forceinline__ void fa(uint32_t a, uint32_t b, uint32_t c, uint32_t &sum, uint32_t &carry)
{
    sum = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);   // a ^ b ^ c
    carry = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); // majority
}
__device__ __forceinline__ void ha(uint32_t a, uint32_t b, uint32_t &sum, uint32_t &carry)
{
    sum = a ^ b;
    carry = a & b;
}
// planes c[0..15] = low bits, c[16..31] = high bits; g = guide scan word; returns plane of candidates with <= 4 mismatches
__device__ __forceinline__ uint32_t sliced_le4(const uint32_t (&c)[32], uint32_t g)
{
    uint32_t m[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const uint32_t s0 = 0u - ((g >> p) & 1u);
        const uint32_t s1 = 0u - ((g >> (16 + p)) & 1u);
        m[p] = (c[p] ^ s0) | (c[16 + p] ^ s1);
    }
    // weight-1 column: 16 inputs
    uint32_t s[6], k2[8], n0, n1, n2, k4[4], k8[2], t, u;
    fa(m[0], m[1], m[2], s[0], k2[0]);
    fa(m[3], m[4], m[5], s[1], k2[1]);
    fa(m[6], m[7], m[8], s[2], k2[2]);
    fa(m[9], m[10], m[11], s[3], k2[3]);
    fa(m[12], m[13], m[14], s[4], k2[4]);
    fa(s[0], s[1], s[2], t, k2[5]);
    fa(s[3], s[4], m[15], u, k2[6]);
    ha(t, u, n0, k2[7]);
    // weight-2 column: 8 inputs
    uint32_t a2, b2, c2;
    fa(k2[0], k2[1], k2[2], a2, k4[0]);
    fa(k2[3], k2[4], k2[5], b2, k4[1]);
    fa(k2[6], k2[7], a2, c2, k4[2]);
    ha(b2, c2, n1, k4[3]);
    // weight-4 column: 4 inputs
    uint32_t a4;
    fa(k4[0], k4[1], k4[2], a4, k8[0]);
    ha(a4, k4[3], n2, k8[1]);
    // count <= 4  <=>  no weight-8/16 bit and (n2 == 0 or n1 == n0 == 0)
    const uint32_t big = k8[0] | k8[1];
    return ~(big | (n2 & (n1 | n0)));
}

// same, with the guide's 32 broadcast masks read from memory (s_load) instead of 32 s_bfe per guide
__device__ __forceinline__ uint32_t sliced_le4_masks(const uint32_t (&c)[32], const uint32_t *__restrict__ mk)
{
    uint32_t m[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) m[p] = (c[p] ^ mk[p]) | (c[16 + p] ^ mk[16 + p]);
    uint32_t s[6], k2[8], n0, n1, n2, k4[4], k8[2], t, u;
    fa(m[0], m[1], m[2], s[0], k2[0]);
    fa(m[3], m[4], m[5], s[1], k2[1]);
    fa(m[6], m[7], m[8], s[2], k2[2]);
    fa(m[9], m[10], m[11], s[3], k2[3]);
    fa(m[12], m[13], m[14], s[4], k2[4]);
    fa(s[0], s[1], s[2], t, k2[5]);
    fa(s[3], s[4], m[15], u, k2[6]);
    ha(t, u, n0, k2[7]);
    uint32_t a2, b2, c2;
    fa(k2[0], k2[1], k2[2], a2, k4[0]);
    fa(k2[3], k2[4], k2[5], b2, k4[1]);
    fa(k2[6], k2[7], a2, c2, k4[2]);
    ha(b2, c2, n1, k4[3]);
    uint32_t a4;
    fa(k4[0], k4[1], k4[2], a4, k8[0]);
    ha(a4, k4[3], n2, k8[1]);
    const uint32_t big = k8[0] | k8[1];
    return ~(big | (n2 & (n1 | n0)));
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void k_masks(uint32_t *out, const uint32_t *__restrict__ masks, uint32_t seed, int iters)
{
    uint32_t pl[32];
    for (int i = 0; i < 32; ++i) pl[i] = seed * (threadIdx.x + 3) + i * 0x85EBCA6Bu + blockIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc |= sliced_le4_masks(pl, masks + (((it * 4 + u) & 1023) * 32));
    }
    out[blockIdx.x * (WPB * 64) + threadIdx.x] = acc;
}

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
            } else if (MODE == 7) { // bitop3 rate
                acc = __builtin_amdgcn_bitop3_b32(acc, c[r], g, 0x96);
            } else if (MODE == 8) { // bfi rate
                asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(acc) : "v"(c[r]), "s"(g));
            }
        }
    }
    if (MODE == 6) { // bit-sliced: 32 planes, 4 guides per iteration
        uint32_t pl[32];
        for (int i = 0; i < 32; ++i) pl[i] = seed * (threadIdx.x + 3) + i * 0x85EBCA6Bu + blockIdx.x;
        uint32_t gg = seed;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                gg = gg * 1664525u + 1013904223u;
                acc |= sliced_le4(pl, gg);
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
    run<7>("v_bitop3 chain (1 op)", 1, d_out);
    run<8>("v_bfi chain (1 op)", 1, d_out);
    {   // bit-sliced: each iteration = 4 guides x 32 candidates per lane = 128 comparisons per lane
        const int iters = 4000, blocks = 256 * 8;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 10);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        const double cmps = double(iters) * 128 * 256 * blocks;
        printf("%-34s %8.3f ms  %7.2f Tcand/s\n", "bit-sliced le4 (32 cand/lane-op)", ms, cmps / ms / 1e9);
    }
    {   // same with the masks read through the scalar cache
        const int iters = 4000, blocks = 256 * 8;
        uint32_t *d_masks; hipMalloc(&d_masks, 1024 * 32 * 4);
        uint32_t *h = new uint32_t[1024 * 32];
        for (int i = 0; i < 1024 * 32; ++i) h[i] = (i * 2654435761u >> 7) & 1u ? 0xFFFFFFFFu : 0u;
        hipMemcpy(d_masks, h, 1024 * 32 * 4, hipMemcpyHostToDevice);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k_masks<4>, dim3(blocks), dim3(256), 0, 0, d_out, d_masks, 12345u, 10);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_masks<4>, dim3(blocks), dim3(256), 0, 0, d_out, d_masks, 12345u, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        const double cmps = double(iters) * 128 * 256 * blocks;
        printf("%-34s %8.3f ms  %7.2f Tcand/s\n", "bit-sliced le4, masks via s_load", ms, cmps / ms / 1e9);
    }
    return 0;
}
