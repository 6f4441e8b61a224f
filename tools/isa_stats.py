#!/usr/bin/env python3
"""Compile issl_kernels.hip to gfx950 assembly and print, per kernel matching a pattern, its register / scratch / LDS
budget and the counts of the instructions that make up the scan's hot loop (development aid; runs without a GPU).
    tools/isa_stats.py [name-substring, default k_scan] [--keep /tmp/dir]"""
import pathlib, re, subprocess, sys, tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
pat = next((a for a in sys.argv[1:] if not a.startswith("--")), "k_scan")
keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
out = pathlib.Path(keep or tempfile.mkdtemp()) / "issl_kernels.s"
out.parent.mkdir(parents=True, exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-S",
                "--cuda-device-only", f"-I{ROOT}/include", "-o", str(out), str(ROOT / "crackling_amd/csrc/issl_kernels.hip")],
               check=True, stderr=subprocess.DEVNULL)
s = out.read_text()
demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    get = lambda key: (re.search(rf"\.amdhsa_{key} (\d+)", body) or [None, "?"])[1]
    cnt = lambda p: len(re.findall(p, body))
    print(f"{demangle(name).split('(')[0]}: vgpr {get('next_free_vgpr')} sgpr {get('next_free_sgpr')} "
          f"scratch {get('private_segment_fixed_size')} B lds {get('group_segment_fixed_size')} B | "
          f"v_bitop3 {cnt(r'v_bitop3_b32')} v_xor {cnt(r'v_xor_b32')} v_perm {cnt(r'v_perm_b32')} s_bfe_i32 {cnt(r's_bfe_i32')} "
          f"s_load_x8 {cnt(r's_load_dwordx8')} global_load_x4 {cnt(r'global_load_dwordx4')} ds_read_b128 {cnt(r'ds_read_b128')} "
          f"scratch ops {cnt(r'scratch_(load|store)')} | lines {body.count(chr(10))}")
if keep:
    print("assembly kept at", out)
