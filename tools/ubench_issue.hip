// What bounds the scan's inner loop on gfx950?  Instruction-issue microbenchmark, round 2.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o tools/_build/ubench_issue && tools/_build/ubench_issue
// Every kernel runs 2 x 1024-thread workgroups per CU (8 waves per SIMD, like k_scan) from registers only and stamps
// s_memtime / s_memrealtime, so that rates are reported per SIMD CLOCK CYCLE of the clock the chip actually held.
//   vop2 / vop3 / vop2lit : streams of independent v_xor_b32 (4-byte encoding), v_bitop3_b32 (8 bytes) and v_xor_b32
//                           with a 32-bit literal (8 bytes): same VALU work per instruction, different code bytes
//   sliced_bfe            : the scan's comparison of 32 candidates per lane with one guide, masks by s_bfe_i32 with a
//                           literal field descriptor (8-byte SALU encodings) -- what the compiler emits for k_scan
//   sliced_bfe_sgpr       : the same with the 32 field descriptors held in SGPRs (4-byte encodings)
//   sliced_const          : masks loop-invariant (no SALU in the loop): the pure 62-VALU stream
//   sliced_sload          : masks fetched by s_load_dwordx16 from a table the size of an item's guide list
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

__device__ __forceinline__ void fa(uint32_t a, uint32_t b, uint32_t c, uint32_t &sum, uint32_t &carry)
{
    sum = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
    carry = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
}
__device__ __forceinline__ void ha(uint32_t a, uint32_t b, uint32_t &sum, uint32_t &carry) { sum = a ^ b; carry = a & b; }

__device__ __forceinline__ uint32_t tree_le4(const uint32_t (&m)[16])
{
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
    return ~(k8[0] | k8[1] | (n2 & (n1 | n0)));
}

// ---- software-pipelined comparison -------------------------------------------------------------------------------
// VALU instructions that read an SGPR issue at half rate when they follow one another (vop2 / vop3 above: 0.24 per
// cycle) but at full rate when they alternate with register-only instructions (vop2 sgpr/vgpr 1:1: 0.44).  One guide
// costs 32 of the former (mismatch planes: c ^ guide mask) and 30 of the latter (the adder tree).  step() therefore
// finishes guide A's tree while it forms guide B's mismatch planes, written in strictly alternating order;
// sched_barrier(0) keeps the compiler's schedulers from regrouping them.
// The compiler's instruction selection does not keep the source order of independent operations (and sched_barrier only
// binds the later machine scheduler), so every vector instruction of the step is an `asm volatile` statement: those
// keep their order.  Register allocation, the scalar mask extraction and the waitcnt insertion stay with the compiler.
#define SB
__device__ __forceinline__ uint32_t op_sum3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t op_maj3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe8" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t op_xor(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm volatile("v_xor_b32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t op_and(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t op_xor_s(uint32_t mask, uint32_t v) // v ^ scalar mask
{
    uint32_t d;
    asm volatile("v_xor_b32_e32 %0, %1, %2" : "=v"(d) : "s"(mask), "v"(v));
    return d;
}
__device__ __forceinline__ uint32_t op_or_xor_s(uint32_t a, uint32_t b, uint32_t mask) // a | (b ^ scalar mask)
{
    uint32_t d;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xf6" : "=v"(d) : "v"(a), "v"(b), "s"(mask));
    return d;
}
__device__ __forceinline__ uint32_t op_and_or(uint32_t a, uint32_t b, uint32_t c) // a & (b | c)
{
    uint32_t d;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe0" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t op_nor3(uint32_t a, uint32_t b, uint32_t c) // ~(a | b | c)
{
    uint32_t d;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x01" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
#define SUM3(a, b, c) op_sum3(a, b, c)
#define MAJ3(a, b, c) op_maj3(a, b, c)
#define XM(p) const uint32_t x##p = op_xor_s(0u - ((gw >> p) & 1u), c[p])
#define MM(p) n[p] = op_or_xor_s(x##p, c[16 + p], 0u - ((gw >> (16 + p)) & 1u))
__device__ __forceinline__ uint32_t step_le4(const uint32_t (&c)[32], const uint32_t (&m)[16], uint32_t (&n)[16], uint32_t gw)
{
    XM(0); const uint32_t s0 = SUM3(m[0], m[1], m[2]); SB;
    MM(0);  const uint32_t k0 = MAJ3(m[0], m[1], m[2]); SB;
    XM(1); const uint32_t s1 = SUM3(m[3], m[4], m[5]); SB;
    MM(1);  const uint32_t k1 = MAJ3(m[3], m[4], m[5]); SB;
    XM(2); const uint32_t s2 = SUM3(m[6], m[7], m[8]); SB;
    MM(2);  const uint32_t k2 = MAJ3(m[6], m[7], m[8]); SB;
    XM(3); const uint32_t s3 = SUM3(m[9], m[10], m[11]); SB;
    MM(3);  const uint32_t k3 = MAJ3(m[9], m[10], m[11]); SB;
    XM(4); const uint32_t s4 = SUM3(m[12], m[13], m[14]); SB;
    MM(4);  const uint32_t k4 = MAJ3(m[12], m[13], m[14]); SB;
    XM(5); const uint32_t t = SUM3(s0, s1, s2); SB;
    MM(5);  const uint32_t k5 = MAJ3(s0, s1, s2); SB;
    XM(6); const uint32_t u = SUM3(s3, s4, m[15]); SB;
    MM(6);  const uint32_t k6 = MAJ3(s3, s4, m[15]); SB;
    XM(7); const uint32_t a2 = SUM3(k0, k1, k2); SB;
    MM(7);  const uint32_t q0 = MAJ3(k0, k1, k2); SB;
    XM(8); const uint32_t n0 = op_xor(t, u); SB;
    MM(8);  const uint32_t k7 = op_and(t, u); SB;
    XM(9); const uint32_t b2 = SUM3(k3, k4, k5); SB;
    MM(9);  const uint32_t q1 = MAJ3(k3, k4, k5); SB;
    XM(10); const uint32_t c2 = SUM3(k6, k7, a2); SB;
    MM(10); const uint32_t q2 = MAJ3(k6, k7, a2); SB;
    XM(11); const uint32_t n1 = op_xor(b2, c2); SB;
    MM(11); const uint32_t q3 = op_and(b2, c2); SB;
    XM(12); const uint32_t a4 = SUM3(q0, q1, q2); SB;
    MM(12); const uint32_t k8a = MAJ3(q0, q1, q2); SB;
    XM(13); const uint32_t n2 = op_xor(a4, q3); SB;
    MM(13); const uint32_t k8b = op_and(a4, q3); SB;
    XM(14); const uint32_t w = op_and_or(n2, n1, n0); SB;
    MM(14); const uint32_t ok = op_nor3(k8a, k8b, w); SB;
    XM(15); MM(15);
    return ok;
}

#define XK(p) const uint32_t x##p = op_xor_s(mk[p], c[p])
#define MK(p) n[p] = op_or_xor_s(x##p, c[16 + p], mk[16 + p])
__device__ __forceinline__ uint32_t step_le4_masks(const uint32_t (&c)[32], const uint32_t (&m)[16], uint32_t (&n)[16], const uint32_t (&mk)[32])
{
    XK(0); const uint32_t s0 = SUM3(m[0], m[1], m[2]); SB;
    MK(0);  const uint32_t k0 = MAJ3(m[0], m[1], m[2]); SB;
    XK(1); const uint32_t s1 = SUM3(m[3], m[4], m[5]); SB;
    MK(1);  const uint32_t k1 = MAJ3(m[3], m[4], m[5]); SB;
    XK(2); const uint32_t s2 = SUM3(m[6], m[7], m[8]); SB;
    MK(2);  const uint32_t k2 = MAJ3(m[6], m[7], m[8]); SB;
    XK(3); const uint32_t s3 = SUM3(m[9], m[10], m[11]); SB;
    MK(3);  const uint32_t k3 = MAJ3(m[9], m[10], m[11]); SB;
    XK(4); const uint32_t s4 = SUM3(m[12], m[13], m[14]); SB;
    MK(4);  const uint32_t k4 = MAJ3(m[12], m[13], m[14]); SB;
    XK(5); const uint32_t t = SUM3(s0, s1, s2); SB;
    MK(5);  const uint32_t k5 = MAJ3(s0, s1, s2); SB;
    XK(6); const uint32_t u = SUM3(s3, s4, m[15]); SB;
    MK(6);  const uint32_t k6 = MAJ3(s3, s4, m[15]); SB;
    XK(7); const uint32_t a2 = SUM3(k0, k1, k2); SB;
    MK(7);  const uint32_t q0 = MAJ3(k0, k1, k2); SB;
    XK(8); const uint32_t n0 = op_xor(t, u); SB;
    MK(8);  const uint32_t k7 = op_and(t, u); SB;
    XK(9); const uint32_t b2 = SUM3(k3, k4, k5); SB;
    MK(9);  const uint32_t q1 = MAJ3(k3, k4, k5); SB;
    XK(10); const uint32_t c2 = SUM3(k6, k7, a2); SB;
    MK(10); const uint32_t q2 = MAJ3(k6, k7, a2); SB;
    XK(11); const uint32_t n1 = op_xor(b2, c2); SB;
    MK(11); const uint32_t q3 = op_and(b2, c2); SB;
    XK(12); const uint32_t a4 = SUM3(q0, q1, q2); SB;
    MK(12); const uint32_t k8a = MAJ3(q0, q1, q2); SB;
    XK(13); const uint32_t n2 = op_xor(a4, q3); SB;
    MK(13); const uint32_t k8b = op_and(a4, q3); SB;
    XK(14); const uint32_t w = op_and_or(n2, n1, n0); SB;
    MK(14); const uint32_t ok = op_nor3(k8a, k8b, w); SB;
    XK(15); MK(15);
    return ok;
}

struct Stamp {
    unsigned long long c0, r0;
    __device__ void start() { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    __device__ void stop(unsigned long long *out)
    {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63u) == 0) {
            const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
            out[2 * w] = c1 - c0;
            out[2 * w + 1] = r1 - r0;
        }
    }
};

enum { VOP2 = 0, VOP3 = 1, VOP2LIT = 2, SL_BFE = 3, SL_BFE_SGPR = 4, SL_CONST = 5, SL_SLOAD = 6, VOP2VV = 7, VOP3VVV = 8, VOP2INL = 9, VOP2MIX = 10, VOP3_2S = 11, VOP2_WAVES4 = 12, SL_PIPE = 13, SL_PIPE_SLOAD = 14 };

template <int MODE>
__global__ __launch_bounds__(1024, 8) void k(uint32_t *out, unsigned long long *stamps, const uint32_t *__restrict__ table,
                                             uint32_t seed, int iters)
{
    Stamp st;
    uint32_t keep = 0;
    if (MODE <= VOP2LIT || (MODE >= VOP2VV && MODE < SL_PIPE)) {
        uint32_t a[8];
        for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u + blockIdx.x;
        uint32_t g = seed;
        st.start();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                if (MODE == VOP2) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[r & 7]) : "s"(g));
                if (MODE == VOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[r & 7]) : "v"(a[(r + 3) & 7]), "s"(g));
                if (MODE == VOP2LIT) asm volatile("v_xor_b32_e32 %0, 0x12345678, %0" : "+v"(a[r & 7]));
                if (MODE == VOP2VV) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[r & 7]) : "v"(a[(r + 3) & 7]));
                if (MODE == VOP3VVV) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[r & 7]) : "v"(a[(r + 3) & 7]), "v"(a[(r + 5) & 7]));
                if (MODE == VOP2INL) asm volatile("v_xor_b32_e32 %0, -1, %0" : "+v"(a[r & 7]));
                if (MODE == VOP2MIX) {
                    if (r & 1) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[r & 7]) : "s"(g));
                    else asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[r & 7]) : "v"(a[(r + 3) & 7]));
                }
                if (MODE == VOP3_2S) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a[r & 7]) : "s"(g));
            }
        }
        st.stop(stamps);
        for (int i = 0; i < 8; ++i) keep ^= a[i];
    } else {
        uint32_t pl[32];
        for (int i = 0; i < 32; ++i) pl[i] = seed * (threadIdx.x + 3) + i * 0x85EBCA6Bu + blockIdx.x;
        uint32_t gg = __builtin_amdgcn_readfirstlane(seed + blockIdx.x);
        uint32_t desc[32];
        if (MODE == SL_BFE_SGPR)
            for (int p = 0; p < 32; ++p) desc[p] = __builtin_amdgcn_readfirstlane((0x10000u | p) + (seed & 0u));
        if (MODE == SL_PIPE || MODE == SL_PIPE_SLOAD) {
            uint32_t m[16];
            for (int p = 0; p < 16; ++p) m[p] = ~0u;
            st.start();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    uint32_t n[16];
                    if (MODE == SL_PIPE) {
                        gg = gg * 1664525u + 1013904223u;
                        keep |= step_le4(pl, m, n, gg);
                    } else {
                        struct alignas(64) M16 { uint32_t w[16]; };
                        const M16 *src = reinterpret_cast<const M16 *>(table + (((it * 8 + u) & 511) * 32));
                        const M16 lo = src[0], hi = src[1];
                        uint32_t mk[32];
#pragma unroll
                        for (int q = 0; q < 16; ++q) { mk[q] = lo.w[q]; mk[16 + q] = hi.w[q]; }
                        keep |= step_le4_masks(pl, m, n, mk);
                    }
#pragma unroll
                    for (int p = 0; p < 16; ++p) m[p] = n[p];
                }
            }
            st.stop(stamps);
            for (int p = 0; p < 16; ++p) keep ^= m[p];
        } else {
        st.start();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                uint32_t m[16];
                if (MODE == SL_BFE) {
                    gg = gg * 1664525u + 1013904223u;
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        const uint32_t s0 = 0u - ((gg >> p) & 1u), s1 = 0u - ((gg >> (16 + p)) & 1u);
                        m[p] = (pl[p] ^ s0) | (pl[16 + p] ^ s1);
                    }
                } else if (MODE == SL_BFE_SGPR) {
                    gg = gg * 1664525u + 1013904223u;
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        uint32_t s0, s1;
                        asm volatile("s_bfe_i32 %0, %1, %2" : "=s"(s0) : "s"(gg), "s"(desc[p]) : "scc");
                        asm volatile("s_bfe_i32 %0, %1, %2" : "=s"(s1) : "s"(gg), "s"(desc[16 + p]) : "scc");
                        m[p] = (pl[p] ^ s0) | (pl[16 + p] ^ s1);
                    }
                } else if (MODE == SL_CONST) {
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        const uint32_t s0 = 0u - ((gg >> p) & 1u), s1 = 0u - ((gg >> (16 + p)) & 1u);
                        m[p] = (pl[p] ^ s0) | (pl[16 + p] ^ s1);
                    }
#pragma unroll
                    for (int q = 0; q < 32; ++q) asm volatile("" : "+v"(pl[q])); // keeps the eight comparisons apart (no CSE)
                } else { // SL_SLOAD: 512 guides x 32 masks = 64 KiB, every wave walks the same list
                    const uint32_t *mk = table + (((it * 8 + u) & 511) * 32);
#pragma unroll
                    for (int p = 0; p < 16; ++p) m[p] = (pl[p] ^ mk[p]) | (pl[16 + p] ^ mk[16 + p]);
                }
                keep |= tree_le4(m);
            }
        }
        st.stop(stamps);
        }
    }
    if (keep == 0x12345u) out[0] = keep;
    out[blockIdx.x * 1024 + threadIdx.x] = keep;
}

template <int MODE>
static void run(const char *name, double valu_per_iter, double code_bytes_per_iter, uint32_t *d_out, unsigned long long *d_st,
                const uint32_t *d_table, int iters)
{
    const int blocks = 512; // 2 resident per CU
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d_out, d_st, d_table, 12345u, iters / 8 + 1);
    CHECK(hipDeviceSynchronize());
    // a second of load first: the clock the chip holds under this stream, not the boost clock
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d_out, d_st, d_table, 12345u, iters);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d_out, d_st, d_table, 12345u, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> st(2 * blocks * 16);
    CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (size_t w = 0; w < st.size() / 2; ++w) { cyc += st[2 * w]; real += st[2 * w + 1]; }
    const double clock_ghz = cyc / real * 0.1;                 // s_memrealtime ticks at 100 MHz
    const double wave_iters = double(iters) * blocks * 16;      // iterations over all waves
    const double simd_cycles = ms * 1e-3 * clock_ghz * 1e9 * 1024.0;
    printf("%-20s %8.3f ms  clock %.2f GHz  VALU/cycle/SIMD %.3f (peak 0.5)  code %.1f B/cycle/CU\n", name, ms, clock_ghz,
           wave_iters * valu_per_iter / simd_cycles, wave_iters * code_bytes_per_iter / (simd_cycles / 4.0));
    if (MODE >= SL_BFE) printf("%-16s   = %.2f T comparisons/s\n", "", wave_iters * 8 * 2048 / (ms * 1e-3) / 1e12);
}

int main()
{
    uint32_t *d_out, *d_table;
    unsigned long long *d_st;
    CHECK(hipMalloc(&d_out, 512 * 1024 * 4 + 64));
    CHECK(hipMalloc(&d_st, 512 * 16 * 16));
    CHECK(hipMalloc(&d_table, 512 * 32 * 4));
    std::vector<uint32_t> h(512 * 32);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) >> 7) & 1u ? 0xFFFFFFFFu : 0u;
    CHECK(hipMemcpy(d_table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // code bytes per iteration: 64 instructions of 4 or 8 bytes; sliced loops per 8 guides: 46 VOP3 (8 B) + 16 VOP2 (4 B)
    // per guide = 432 B, plus the masks: 32 s_bfe of 8 B (literal) or 4 B (SGPR descriptor), 2 s_load of 8 B, or none
    run<VOP2>("vop2", 64, 64 * 4, d_out, d_st, d_table, 20000);
    run<VOP3>("vop3", 64, 64 * 8, d_out, d_st, d_table, 20000);
    run<VOP2LIT>("vop2lit", 64, 64 * 8, d_out, d_st, d_table, 20000);
    run<VOP2VV>("vop2 vgpr,vgpr", 64, 64 * 4, d_out, d_st, d_table, 20000);
    run<VOP3VVV>("vop3 3 vgprs", 64, 64 * 8, d_out, d_st, d_table, 20000);
    run<VOP2INL>("vop2 inline -1", 64, 64 * 4, d_out, d_st, d_table, 20000);
    run<VOP2MIX>("vop2 sgpr/vgpr 1:1", 64, 64 * 4, d_out, d_st, d_table, 20000);
    run<VOP3_2S>("vop3 same sgpr x2", 64, 64 * 8, d_out, d_st, d_table, 20000);
    run<SL_BFE>("sliced_bfe", 8 * 62, 8 * (432 + 256 + 12), d_out, d_st, d_table, 3000);
    run<SL_BFE_SGPR>("sliced_bfe_sgpr", 8 * 62, 8 * (432 + 128 + 12), d_out, d_st, d_table, 3000);
    run<SL_CONST>("sliced_const", 8 * 62, 8 * 432, d_out, d_st, d_table, 3000);
    run<SL_SLOAD>("sliced_sload", 8 * 62, 8 * (432 + 16), d_out, d_st, d_table, 3000);
    run<SL_PIPE>("sliced_pipelined", 8 * 62, 8 * (432 + 256 + 12), d_out, d_st, d_table, 3000);
    run<SL_PIPE_SLOAD>("sliced_pipe_sload", 8 * 62, 8 * (432 + 16), d_out, d_st, d_table, 3000);
    return 0;
}
