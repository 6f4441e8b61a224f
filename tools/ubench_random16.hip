// What does a random 16-byte read cost on gfx950, and how much does it pull from HBM?  (round 3: k_verify reads one
// 16-byte stream record per noted candidate at a random place of a 23 GB array.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_random16.hip -o tools/_build/ubench_random16
//   tools/_build/ubench_random16 [GiB of table] [million reads]
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o f -- tools/_build/ubench_random16   (bytes per read)
// Variants: plain global_load_dwordx4; the same with the nontemporal hint; 8-byte and 4-byte reads at the same places;
// reads whose 16 bytes are the first / last of their 128-byte line; two reads of the same 64-byte half from two lanes.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// mode 0: 16 B plain   1: 16 B nontemporal   2: 8 B   3: 4 B   4: 16 B at line offset 0   5: 16 B at line offset 112
// 6: pairs of lanes read the two 16-byte quarters of one 32-byte sector   7: 32 B (two dwordx4 of one sector) per lane
template <int MODE>
__global__ __launch_bounds__(128) void k_reads(const uint8_t *__restrict__ table, uint64_t n_slots16, uint64_t n_reads,
                                               uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_reads; i += stride) {
        uint64_t slot = mix(i ^ 0x9E3779B97F4A7C15ull) % n_slots16;
        if (MODE == 4) slot &= ~7ull;
        if (MODE == 5) slot |= 7ull;
        if (MODE == 6) slot = (mix((i >> 1) ^ 0x9E3779B97F4A7C15ull) % n_slots16 & ~1ull) | (i & 1ull);
        const uint8_t *p = table + slot * 16ull;
        if (MODE == 0 || MODE >= 4) {
            const v4u q = *reinterpret_cast<const v4u *>(MODE == 7 ? table + (slot & ~1ull) * 16ull : p);
            acc += q.x ^ q.w;
            if (MODE == 7) { const v4u r = *reinterpret_cast<const v4u *>(table + (slot | 1ull) * 16ull); acc += r.y; }
        } else if (MODE == 1) {
            const v4u q = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
            acc += q.x ^ q.w;
        } else if (MODE == 2) {
            const v2u q = *reinterpret_cast<const v2u *>(p);
            acc += q.x ^ q.y;
        } else {
            acc += *reinterpret_cast<const uint32_t *>(p);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc; // (never: keeps the loads)
}

// random WRITES of `BYTES` (16 or 32: one or two dwordx4 into one 32-byte sector) -- k_verify's slot records
template <int BYTES>
__global__ __launch_bounds__(128) void k_writes(uint8_t *__restrict__ table, uint64_t n_slots32, uint64_t n_writes)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_writes; i += stride) {
        const uint64_t slot = mix(i ^ 0x9E3779B97F4A7C15ull) % n_slots32;
        v4u *p = reinterpret_cast<v4u *>(table + slot * 32ull);
        const v4u val = {static_cast<unsigned int>(i), 1u, 2u, 3u};
        p[0] = val;
        if (BYTES == 32) p[1] = val;
    }
}

template <int BYTES> static void run_writes(const char *name, uint8_t *table, uint64_t n_slots32, uint64_t n_writes)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_writes<BYTES>, dim3(16384), dim3(128), 0, 0, table, n_slots32, n_writes / 8);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_writes<BYTES>, dim3(16384), dim3(128), 0, 0, table, n_slots32, n_writes);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %8.3f ms  %7.2f G writes/s\n", name, ms, n_writes / ms / 1e6);
}

template <int MODE> static void run(const char *name, const uint8_t *table, uint64_t n_slots16, uint64_t n_reads, uint32_t *sink)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const dim3 grid(16384), block(128);
    hipLaunchKernelGGL(k_reads<MODE>, grid, block, 0, 0, table, n_slots16, n_reads / 8, sink); // warm-up (TLB)
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_reads<MODE>, grid, block, 0, 0, table, n_slots16, n_reads, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %8.3f ms  %7.2f G reads/s  (x128 B = %6.2f TB/s, x64 B = %6.2f, x32 B = %6.2f)\n", name, ms, n_reads / ms / 1e6,
           n_reads * 128.0 / ms / 1e9, n_reads * 64.0 / ms / 1e9, n_reads * 32.0 / ms / 1e9);
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 22.0;
    const uint64_t n_reads = static_cast<uint64_t>((argc > 2 ? atof(argv[2]) : 17.0) * 1e6);
    const uint64_t bytes = static_cast<uint64_t>(gib * (1ull << 30)) & ~127ull;
    uint8_t *table = nullptr;
    uint32_t *sink = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void **>(&table), bytes));
    CHECK(hipMalloc(reinterpret_cast<void **>(&sink), 64));
    CHECK(hipMemset(table, 1, bytes));
    CHECK(hipDeviceSynchronize());
    printf("table %.1f GiB, %.1f M random reads per launch, one read per thread and step, 16384 x 128 threads\n", gib, n_reads / 1e6);
    const uint64_t slots = bytes / 16;
    run<0>("16 B (global_load_dwordx4)", table, slots, n_reads, sink);
    run<1>("16 B nontemporal", table, slots, n_reads, sink);
    run<2>("8 B", table, slots, n_reads, sink);
    run<3>("4 B", table, slots, n_reads, sink);
    run<4>("16 B, first of its 128-B line", table, slots, n_reads, sink);
    run<5>("16 B, last of its 128-B line", table, slots, n_reads, sink);
    run<6>("16 B, lane pairs share a 32-B sector", table, slots, n_reads, sink);
    run<7>("32 B per lane (both halves of a sector)", table, slots, n_reads, sink);
    run_writes<32>("random 32-B writes (one sector each), 22 GiB", table, bytes / 32, n_reads);
    run_writes<16>("random 16-B writes", table, bytes / 32, n_reads);
    run_writes<32>("random 32-B writes into 1.6 GB (the hit slots' extent)", table, (1600ull << 20) / 32, n_reads);
    return 0;
}
