#!/usr/bin/env python3
"""Writes a microbenchmark (single-asm-block kernels with fixed registers) for the scan's guide loop:

    python tools/gen_ubench_sel.py /tmp/ubench_sel.hip && hipcc --offload-arch=gfx950 -O3 /tmp/ubench_sel.hip -o tools/_build/ubench_sel

A: today's loop (32 s_bfe masks, 16 x (v_xor + v_bitop3), carry-save adder tree);  B/C: four precomputed planes per
position and a register-indexed v_mov per position (s_set_gpr_idx_on / _idx), with the indices extracted by s_bfe (B)
or already in SGPRs (C).  Measured on MI355X (256-thread blocks): A 21.2-23.5, B 20.0, C 25.8 T comparisons/s -- the
indexed moves save 16 of 62 VALU ops but cost a mode switch each; not worth a hand-allocated kernel."""
import sys
# generates ubench_sel.hip: single-asm-block kernels measuring the guide loop variants
def csa(m, tmp_base):
    """m: list of 16 register names holding mismatch planes; returns asm lines and the result register (ok plane, THR=4)."""
    L=[]; t=[f"v{tmp_base+i}" for i in range(24)]
    def fa(a,b,c,s,k): L.append(f"v_bitop3_b32 {s}, {a}, {b}, {c} bitop3:0x96"); L.append(f"v_bitop3_b32 {k}, {a}, {b}, {c} bitop3:0xe8")
    def ha(a,b,s,k): L.append(f"v_xor_b32 {s}, {a}, {b}"); L.append(f"v_and_b32 {k}, {a}, {b}")
    s0,s1,s2,s3,s4,tt,u,n0,n1,n2=t[0:10]; k2=t[10:18]; k4=t[18:22]; k8a,k8b=t[22],t[23]
    fa(m[0],m[1],m[2],s0,k2[0]); fa(m[3],m[4],m[5],s1,k2[1]); fa(m[6],m[7],m[8],s2,k2[2]); fa(m[9],m[10],m[11],s3,k2[3]); fa(m[12],m[13],m[14],s4,k2[4])
    fa(s0,s1,s2,tt,k2[5]); fa(s3,s4,m[15],u,k2[6]); ha(tt,u,n0,k2[7])
    a2,b2,c2=s0,s1,s2
    fa(k2[0],k2[1],k2[2],a2,k4[0]); fa(k2[3],k2[4],k2[5],b2,k4[1]); fa(k2[6],k2[7],a2,c2,k4[2]); ha(b2,c2,n1,k4[3])
    a4=s3
    fa(k4[0],k4[1],k4[2],a4,k8a); ha(a4,k4[3],n2,k8b)
    # ~(k8a | k8b | (n2 & (n1|n0)))
    L.append(f"v_or_b32 {s4}, {n1}, {n0}"); L.append(f"v_and_b32 {s4}, {s4}, {n2}"); L.append(f"v_or3_b32 {s4}, {s4}, {k8a}, {k8b}"); L.append(f"v_not_b32 {s4}, {s4}")
    return L, s4

def kernel_A():
    # planes c0[p]=v(32+p), c1[p]=v(48+p); guide word in s20 (changes per iteration); masks via s_bfe into s[40:71]
    L=["s_mov_b32 s20, 0x12345678", "v_mov_b32 v1, 0", ".LA%=:"]
    L.append("s_mul_i32 s20, s20, 0x19660d"); L.append("s_add_u32 s20, s20, 0x3c6ef35f")
    m=[]
    for p in range(16):
        L.append(f"s_bfe_i32 s{40+p}, s20, {0x10000|p}"); L.append(f"s_bfe_i32 s{56+p}, s20, {0x10000|(16+p)}")
    for p in range(16):
        L.append(f"v_xor_b32 v{64+p}, s{40+p}, v{32+p}")
        L.append(f"v_bitop3_b32 v{64+p}, v{48+p}, s{56+p}, v{64+p} bitop3:0xf6")  # (c1 ^ g1) | t   (table value irrelevant for timing)
        m.append(f"v{64+p}")
    c,res=csa(m,80)
    L+=c; L.append(f"v_or_b32 v1, v1, {res}")
    L+=["s_sub_u32 s21, s21, 1","s_cmp_lg_u32 s21, 0","s_cbranch_scc1 .LA%="]
    return L
def kernel_B():
    # 4 planes per position: M[p][x] = v(128+4p+x) (64 regs); per guide 16 selections into v(64+p); indices in s[40:55]
    L=["s_mov_b32 s20, 0x12345678", "v_mov_b32 v1, 0", ".LB%=:"]
    L.append("s_mul_i32 s20, s20, 0x19660d"); L.append("s_add_u32 s20, s20, 0x3c6ef35f")
    for p in range(16): L.append(f"s_bfe_u32 s{40+p}, s20, {0x20000|(2*p)}")   # index 0..3 (real code: loaded from memory)
    L.append("s_set_gpr_idx_on s40, 0x1")
    m=[]
    for p in range(16):
        if p: L.append(f"s_set_gpr_idx_idx s{40+p}")
        L.append(f"v_mov_b32 v{96+p}, v{32+4*p}")
        m.append(f"v{96+p}")
    L.append("s_set_gpr_idx_off")
    c,res=csa(m,112)
    L+=c; L.append(f"v_or_b32 v1, v1, {res}")
    L+=["s_sub_u32 s21, s21, 1","s_cmp_lg_u32 s21, 0","s_cbranch_scc1 .LB%="]
    return L
def kernel_C():
    # like B but indices preloaded (no s_bfe): measures the selection mechanism alone
    L=["v_mov_b32 v1, 0"]+[f"s_mov_b32 s{40+p}, {p%4}" for p in range(16)]+[".LC%=:"]
    L.append("s_set_gpr_idx_on s40, 0x1")
    m=[]
    for p in range(16):
        if p: L.append(f"s_set_gpr_idx_idx s{40+p}")
        L.append(f"v_mov_b32 v{96+p}, v{32+4*p}")
        m.append(f"v{96+p}")
    L.append("s_set_gpr_idx_off")
    c,res=csa(m,112)
    L+=c; L.append(f"v_or_b32 v1, v1, {res}")
    L+=["s_sub_u32 s21, s21, 1","s_cmp_lg_u32 s21, 0","s_cbranch_scc1 .LC%="]
    return L
def body(lines, nv):
    init=[]
    for r in range(32,nv):
        init.append(f"s_mov_b32 s22, {2654435761+r*97}")
        init.append(f"v_mul_lo_u32 v{r}, %[tid], s22")
    pre=["s_mov_b32 s21, %[iters]"]
    post=["v_lshlrev_b32 v2, 2, %[tid]","global_store_dword v2, v1, %[out]"]
    allv=",".join(f'"v{i}"' for i in range(1,nv))
    alls=",".join(f'"s{i}"' for i in range(20,72))
    txt="\\n\\t".join(init+pre+lines+post)
    return f'    asm volatile("{txt}" :: [tid]"v"(threadIdx.x + blockIdx.x * blockDim.x), [iters]"s"(iters), [out]"s"(out) : {allv}, {alls}, "memory", "scc", "m0");'
src='''#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
'''
for name,fn,nv in (("kA",kernel_A,104),("kB",kernel_B,136),("kC",kernel_C,136)):
    src+=f'__global__ __launch_bounds__(256) void {name}(uint32_t *out, int iters)\n{{\n{body(fn(),nv)}\n}}\n'
src+='''
template <typename K> void run(const char *name, K k, uint32_t *d, int wpb_note) {
    const int iters = 20000, blocks = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double cmps = double(iters) * 32 * 256 * blocks;
    printf("%-44s %8.3f ms  %7.2f Tcmp/s\\n", name, ms, cmps / ms / 1e9);
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run("A: s_bfe masks + xor/or + CSA (today)", kA, d, 0);
    run("B: 4 planes/position, gpr_idx select + s_bfe idx", kB, d, 0);
    run("C: 4 planes/position, gpr_idx select, idx ready", kC, d, 0);
    run("A again", kA, d, 0);
    return 0;
}
'''
open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/ubench_sel.hip', 'w').write(src)
