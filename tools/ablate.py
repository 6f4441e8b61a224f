#!/usr/bin/env python3
"""Builds variants of libissl_hip.so with pieces of a kernel cut out (timing experiments, never shipped):

    python tools/ablate.py build            # here (hipcc cross-compiles): build/ablate/libissl_hip_<name>.so
    python tools/ablate.py run [bench args] # on the GPU box: bench.py once per variant, stage times side by side

A variant is a list of (old, new) text replacements applied to a COPY of csrc/issl_kernels.hip; the results of a cut
kernel are wrong by construction -- only its duration is read."""
import json, os, pathlib, subprocess, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
CSRC = ROOT / "crackling_amd" / "csrc"
OUT = ROOT / "build" / "ablate"

VARIANTS = {
    "base": [],
    # k_scan, full units of the pruned plan (near_plane12): what limits the loop -- the scalar pipe (24 s_bfe_i32 per pass),
    # scalar operands of vector instructions, or the hit path?
    "scan_half_bfe": [("        const uint32_t g1 = 0u - ((gw >> (12 + p)) & 1u);\n        m[p] = (c[p] ^ g0) | (c[12 + p] ^ g1);",
                       "        m[p] = (c[p] ^ g0) | (c[12 + p] ^ g0);")],
    "scan_no_bfe": [("        const uint32_t g0 = 0u - ((gw >> p) & 1u);\n        const uint32_t g1 = 0u - ((gw >> (12 + p)) & 1u);\n        m[p] = (c[p] ^ g0) | (c[12 + p] ^ g1);",
                     "        m[p] = (c[p] ^ gw) | (c[12 + p] ^ __builtin_rotateleft32(gw, 7));")],
    "scan_vgpr_masks": [("    uint32_t m[12];\n#pragma unroll\n    for (int p = 0; p < 12; ++p) {\n        const uint32_t g0",
                         "    uint32_t m[12];\n    uint32_t vg = c[3], vg2 = c[17];\n#pragma unroll\n    for (int p = 0; p < 12; ++p) {\n        const uint32_t g0"),
                        ("        const uint32_t g0 = 0u - ((gw >> p) & 1u);\n        const uint32_t g1 = 0u - ((gw >> (12 + p)) & 1u);\n        m[p] = (c[p] ^ g0) | (c[12 + p] ^ g1);",
                         "        if (p == 0) { asm volatile(\"v_xor_b32 %0, %2, %0\\n v_xor_b32 %1, %2, %1\" : \"+v\"(vg), \"+v\"(vg2) : \"s\"(gw)); }\n        m[p] = (c[p] ^ vg) | (c[12 + p] ^ vg2);")],
    "scan_no_hit_path_all": [("                            if (__ballot(ok != 0u) != 0ull) {\n                                note_candidates(fine_dup(ok, prev, dup_filter), g + uu,",
                              "                            if (__ballot(ok == 0x9E3779B9u) != 0ull) {\n                                note_candidates(fine_dup(ok, prev, dup_filter), g + uu,"),
                             ("                    if (__ballot(ok != 0u) != 0ull) {\n                        note_candidates(fine_dup(ok, prev, dup_filter), g + uu,",
                              "                    if (__ballot(ok == 0x9E3779B9u) != 0ull) {\n                        note_candidates(fine_dup(ok, prev, dup_filter), g + uu,"),
                             ("                        if (__ballot(ok != 0u) != 0ull) {\n                            note_candidates(fine_dup(ok, prev, dup_filter), gb + i * per,",
                              "                        if (__ballot(ok == 0x9E3779B9u) != 0ull) {\n                            note_candidates(fine_dup(ok, prev, dup_filter), gb + i * per,")],
    "scan_no_hits_no_fetch": [("                            if (__ballot(ok != 0u) != 0ull) {\n                                note_candidates(fine_dup(ok, prev, dup_filter), g + uu,",
                              "                            if (__ballot(ok == 0x9E3779B9u) != 0ull) {\n                                note_candidates(fine_dup(ok, prev, dup_filter), g + uu,"),
                             ("                    if (__ballot(ok != 0u) != 0ull) {\n                        note_candidates(fine_dup(ok, prev, dup_filter), g + uu,",
                              "                    if (__ballot(ok == 0x9E3779B9u) != 0ull) {\n                        note_candidates(fine_dup(ok, prev, dup_filter), g + uu,"),
                             ("                        if (__ballot(ok != 0u) != 0ull) {\n                            note_candidates(fine_dup(ok, prev, dup_filter), gb + i * per,",
                              "                        if (__ballot(ok == 0x9E3779B9u) != 0ull) {\n                            note_candidates(fine_dup(ok, prev, dup_filter), gb + i * per,"),
                              ("                    const uint4 a0 = src[q0 * 64u], a1 = src[q1 * 64u], a2 = src[q2 * 64u];\n                    const uint4 b0 = src[(4u + q0) * 64u], b1 = src[(4u + q1) * 64u], b2 = src[(4u + q2) * 64u];",
                               "                    const uint4 a0 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u), a1 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u), a2 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u), b0 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u), b1 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u), b2 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u); (void)src;"),
                              ("                const uint4 a0 = src[q0 * 64u], a1 = src[q1 * 64u], a2 = src[q2 * 64u];\n                const uint4 b0 = src[(4u + q0) * 64u], b1 = src[(4u + q1) * 64u], b2 = src[(4u + q2) * 64u];",
                               "                const uint4 a0 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u), a1 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u), a2 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u), b0 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u), b1 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u), b2 = make_uint4(lane * 2654435761u + q0, lane * 40503u + q1, (lane << 7) ^ q2 ^ 0x5bd1e995u, lane * 97u + 13u); (void)src;")],
    "scan_no_hit_path": [("                            if (__ballot(ok != 0u) != 0ull) {\n                                note_candidates(fine_dup(ok, prev, dup_filter), g + uu,",
                          "                            if (__ballot(ok == 0x9E3779B9u) != 0ull) {\n                                note_candidates(fine_dup(ok, prev, dup_filter), g + uu,")],
    # VERDICT r04 item 1(b), scan side only: one 16-byte record per lane with its whole result plane (k_verify switched off)
    "scan_wide_record": [("    uint64_t who = __ballot(ok != 0u);\n    if (who == 0ull) return;\n    // One round per candidate of the lane that has most -- almost always ONE: the loop is laid out with its first round as the\n    // straight path (a taken branch drains the wave's instruction buffer, and more than half of the passes come through\n    // here: k_scan<4> -3 % same-box, profiles/r05_ab_scan_peel_lanes3.log).\n    do {\n        const uint32_t n = static_cast<uint32_t>(__builtin_popcountll(who));\n        if (__builtin_expect(w.fill + n > kChunkRecs, 0)) {\n            raw_retire(w, lane, raw, raw_used);\n            raw_acquire(w, raw, max_chunks, counters, lane);\n        }\n        if (ok != 0u) {\n            const uint32_t q = static_cast<uint32_t>(__builtin_ctz(ok));\n            ok &= ok - 1u;\n            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),\n                                                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));\n            w.chunk[w.fill + rank] = raw_record(gslot + (q >> w_log), tile, off0 + (q & ((1u << w_log) - 1u)));\n        }\n        w.fill += n;\n        who = __ballot(ok != 0u);\n    } while (__builtin_expect(who != 0ull, 0));\n}\n", '    // TIMING ONLY (tools/ablate.py: scan_wide_record): one 16-byte record per lane with its whole result plane instead of one\n    // 8-byte record per candidate -- no bit scan, no second round; k_verify of this build does nothing.\n    const uint64_t who = __ballot(ok != 0u);\n    if (who == 0ull) return;\n    const uint32_t n = 2u * static_cast<uint32_t>(__builtin_popcountll(who)); // in 8-byte slots\n    if (__builtin_expect(((w.fill + 1u) & ~1u) + n > kChunkRecs, 0)) {\n        raw_retire(w, lane, raw, raw_used);\n        raw_acquire(w, raw, max_chunks, counters, lane);\n    }\n    if (ok != 0u) {\n        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),\n                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));\n        uint4 rec;\n        rec.x = ok; rec.y = gslot | (w_log << 27); rec.z = tile; rec.w = off0;\n        *reinterpret_cast<uint4 *>(w.chunk + ((w.fill + 1u) & ~1u) + 2u * rank) = rec;\n    }\n    w.fill = ((w.fill + 1u) & ~1u) + n;\n}\n'), ('    short_kernel_priority();\n    const int max_dist = p.max_dist;\n    const bool calc_mit = p.method == ISSL_METHOD_MIT || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||\n                          p.method == ISSL_METHOD_AVG;\n    const bool calc_cfd = p.method == ISSL_METHOD_CFD || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||\n                          p.method == ISSL_METHOD_AVG;\n    const uint32_t prune_mode = ws.plan->fine; // what the scan of this batch worked through', "    short_kernel_priority();\n    if (p.max_dist >= -100) return; // TIMING ONLY (scan_wide_record): the records of this build are not k_verify's\n    const int max_dist = p.max_dist;\n    const bool calc_mit = p.method == ISSL_METHOD_MIT || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||\n                          p.method == ISSL_METHOD_AVG;\n    const bool calc_cfd = p.method == ISSL_METHOD_CFD || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||\n                          p.method == ISSL_METHOD_AVG;\n    const uint32_t prune_mode = ws.plan->fine; // what the scan of this batch worked through")],
    "scan_wide_record12": [("    uint64_t who = __ballot(ok != 0u);\n    if (who == 0ull) return;\n    // One round per candidate of the lane that has most -- almost always ONE: the loop is laid out with its first round as the\n    // straight path (a taken branch drains the wave's instruction buffer, and more than half of the passes come through\n    // here: k_scan<4> -3 % same-box, profiles/r05_ab_scan_peel_lanes3.log).\n    do {\n        const uint32_t n = static_cast<uint32_t>(__builtin_popcountll(who));\n        if (__builtin_expect(w.fill + n > kChunkRecs, 0)) {\n            raw_retire(w, lane, raw, raw_used);\n            raw_acquire(w, raw, max_chunks, counters, lane);\n        }\n        if (ok != 0u) {\n            const uint32_t q = static_cast<uint32_t>(__builtin_ctz(ok));\n            ok &= ok - 1u;\n            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),\n                                                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));\n            w.chunk[w.fill + rank] = raw_record(gslot + (q >> w_log), tile, off0 + (q & ((1u << w_log) - 1u)));\n        }\n        w.fill += n;\n        who = __ballot(ok != 0u);\n    } while (__builtin_expect(who != 0ull, 0));\n}\n", '    // TIMING ONLY (tools/ablate.py: scan_wide_record): one 16-byte record per lane with its whole result plane instead of one\n    // 8-byte record per candidate -- no bit scan, no second round; k_verify of this build does nothing.\n    const uint64_t who = __ballot(ok != 0u);\n    if (who == 0ull) return;\n    const uint32_t n = 2u * static_cast<uint32_t>(__builtin_popcountll(who)); // in 8-byte slots\n    if (__builtin_expect(((w.fill + 1u) & ~1u) + n > kChunkRecs, 0)) {\n        raw_retire(w, lane, raw, raw_used);\n        raw_acquire(w, raw, max_chunks, counters, lane);\n    }\n    if (ok != 0u) {\n        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),\n                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));\n        uint3 rec;\n        rec.x = ok; rec.y = gslot | (w_log << 27) | ((off0 & 24u) << 27); rec.z = (tile << 6) | (off0 >> 5);\n        *reinterpret_cast<uint3 *>(reinterpret_cast<uint32_t *>(w.chunk + ((w.fill + 1u) & ~1u)) + 3u * rank) = rec;\n    }\n    w.fill = ((w.fill + 1u) & ~1u) + n;\n}\n'), ('    short_kernel_priority();\n    const int max_dist = p.max_dist;\n    const bool calc_mit = p.method == ISSL_METHOD_MIT || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||\n                          p.method == ISSL_METHOD_AVG;\n    const bool calc_cfd = p.method == ISSL_METHOD_CFD || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||\n                          p.method == ISSL_METHOD_AVG;\n    const uint32_t prune_mode = ws.plan->fine; // what the scan of this batch worked through', "    short_kernel_priority();\n    if (p.max_dist >= -100) return; // TIMING ONLY (scan_wide_record): the records of this build are not k_verify's\n    const int max_dist = p.max_dist;\n    const bool calc_mit = p.method == ISSL_METHOD_MIT || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||\n                          p.method == ISSL_METHOD_AVG;\n    const bool calc_cfd = p.method == ISSL_METHOD_CFD || p.method == ISSL_METHOD_AND || p.method == ISSL_METHOD_OR ||\n                          p.method == ISSL_METHOD_AVG;\n    const uint32_t prune_mode = ws.plan->fine; // what the scan of this batch worked through")],
    # an alternative (results stay right): 8 bytes per guide slot {guide, group} instead of 16 {guide, group, signature} -- k_fine_scatter
    # writes 40 % fewer bytes, k_verify fetches the guide's signature from the batch (one more dependent load)
    "fmeta8": [("        for (uint32_t k2 = c; k2 < static_cast<uint32_t>(slots); ++k2) { fmeta[slot_at + k2] = FineMeta{kNoGuide, 0u, 0ull}; fword[slot_at + k2] = kPadGuideWord; }",
                "        for (uint32_t k2 = c; k2 < static_cast<uint32_t>(slots); ++k2) { reinterpret_cast<uint2 *>(fmeta)[slot_at + k2] = make_uint2(kNoGuide, 0u); fword[slot_at + k2] = kPadGuideWord; }"),
               ("            fmeta[slot] = FineMeta{guide, (b << 8) | ww, gsig};",
                "            reinterpret_cast<uint2 *>(fmeta)[slot] = make_uint2(guide, (b << 8) | ww);"),
               ("            if (prune_mode) { const FineMeta m = ws.fmeta[gslot]; guide = m.guide; where = m.where; gsig = m.gsig; }",
                "            if (prune_mode) { const uint2 m = reinterpret_cast<const uint2 *>(ws.fmeta)[gslot]; guide = m.x; where = m.y; if (guide != kNoGuide) gsig = guides[guide]; }")],
    # alternatives (results stay right): wave priorities for the two workgroups a CU starts the scan with -- the arbiter favours one of
    # them (it ends after 500 us where the other takes 910, DESIGN 3.1 "what the 13 % are")
    "scan_prio_second": [("    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out",
                          "    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out\n    if ((blockIdx.x >> 8) & 1u) __builtin_amdgcn_s_setprio(1);")],
    "scan_prio_first": [("    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out",
                         "    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out\n    if (!((blockIdx.x >> 8) & 1u)) __builtin_amdgcn_s_setprio(1);")],
    "scan_prio_late": [("    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out",
                        "    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out\n    if (blockIdx.x >= 512u) __builtin_amdgcn_s_setprio(1);")],
    "scan_prio_ramp": [("    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out",
                       "    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out\n    { const uint32_t q = blockIdx.x >> 8; if (q == 1u) __builtin_amdgcn_s_setprio(1); else if (q == 2u) __builtin_amdgcn_s_setprio(2); else if (q >= 3u) __builtin_amdgcn_s_setprio(3); }")],
    "scan_prio_late3": [("    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out",
                       "    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out\n    if (blockIdx.x >= 512u) __builtin_amdgcn_s_setprio(3);")],
    "scan_prio_last": [("    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out",
                       "    if (threadIdx.x == 0) atomicMin(span, t_start); // the launch's own span: first workgroup in, last one out\n    if (blockIdx.x >= 768u) __builtin_amdgcn_s_setprio(2); else if (blockIdx.x >= 512u) __builtin_amdgcn_s_setprio(1);")],
    # k_verify
    "verify_no_atomic": [("            if (live) rank = atomicAdd(&ws.gcount[guide], 1u);", "            if (live) rank = lane;"),
                         ("if (live && !continues) base = atomicAdd(&ws.gcount[guide], next - lane);", "if (live && !continues) base = next - lane;")],
    "verify_no_terms": [("            score_terms(v, hit_gsig, hit_ot, hit_occ, calc_mit, calc_cfd, mit_term, cfd_term, dist);\n            const uint64_t at",
                         "            dist = 0;\n            const uint64_t at")],
    "verify_no_srec": [("if (in_use && v.srec) sr_early = v.srec[static_cast<uint64_t>(tile) * kTileCands + offset];",
                        "if (in_use && v.srec) { sr_early.sig = rec * 0x9E3779B97F4A7C15ull; sr_early.id = static_cast<uint32_t>(rec); }")],
    "verify_srec_only": [("        if (guide != kNoGuide) {\n            const uint32_t bucket = prune_mode ? where >> 8 : where;",
                          "        if (guide != kNoGuide && sr_early.sig == 0x123456789ull) {\n            const uint32_t bucket = prune_mode ? where >> 8 : where;")],
    "verify_no_slot_store": [("            ws.slots[at] = r;", "            if (rank == 0xFFFFFFFFu) ws.slots[at] = r;")],
    # not cuts but alternatives (results stay right): streaming hints on the random accesses of k_verify
    "verify_srec_nontemporal": [("if (in_use && v.srec) sr_early = v.srec[static_cast<uint64_t>(tile) * kTileCands + offset];",
                                 "if (in_use && v.srec) { typedef unsigned int v4u __attribute__((ext_vector_type(4))); const v4u q4 = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(v.srec + static_cast<uint64_t>(tile) * kTileCands + offset)); sr_early = *reinterpret_cast<const StreamRec *>(&q4); }")],
    "verify_slot_nontemporal": [("            ws.slots[at] = r;",
                                 "            { typedef unsigned int v4u __attribute__((ext_vector_type(4))); const v4u *r4 = reinterpret_cast<const v4u *>(&r); v4u *d4 = reinterpret_cast<v4u *>(ws.slots + at); __builtin_nontemporal_store(r4[0], d4); __builtin_nontemporal_store(r4[1], d4 + 1); }")],
    "verify_no_key_store": [("        if (!in_use) continue;\n        recs[t] = key;", "        if (!in_use) continue;\n        if (key != kDeadKey) recs[t] = key;")],
}


def build():
    OUT.mkdir(parents=True, exist_ok=True)
    src = (CSRC / "issl_kernels.hip").read_text()
    flags = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", f"-I{CSRC}", f"-I{ROOT / 'include'}"]
    objs = []
    for f in ("issl_extract.hip", "issl_build.hip", "issl_capi.cpp", "issl_node.cpp", "issl_host.cpp", "issl_text.cpp"):  # once for all variants
        o = OUT / (f + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", "-o", str(o), str(CSRC / f)])
        objs.append(str(o))
    only = os.environ.get("ABLATE_ONLY")
    for name, edits in VARIANTS.items():
        if only and name not in only.split(","):
            continue
        text = src
        for old, new in edits:
            assert text.count(old) == 1, (name, old[:60], text.count(old))
            text = text.replace(old, new)
        tmp = OUT / f"issl_kernels_{name}.hip"
        tmp.write_text(text)
        print(name, flush=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", "-o", str(OUT / f"k_{name}.o"), str(tmp)])
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-shared", "-o", str(OUT / f"libissl_hip_{name}.so"), str(OUT / f"k_{name}.o")] + objs + ["-lpthread", "-ldl"])
        tmp.unlink()


def run(args):
    only = os.environ.get("ABLATE_ONLY")
    for name in VARIANTS:
        if only and name not in only.split(","):
            continue
        lib = OUT / f"libissl_hip_{name}.so"
        env = dict(os.environ, ISSL_HIP_LIBRARY=str(lib))
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-extras", "--no-cpu-baseline", "--steps", "5", "--warmup", "2"] + args,
                           env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{name:28s} step {d['ms_per_step']:7.3f} ms  " + "  ".join(f"{k} {v:.3f}" for k, v in d["kernel_ms"].items()), flush=True)
        except Exception as e:  # a cut kernel may upset a later stage: report and go on
            print(f"{name:28s} failed: {e}; {r.stderr.strip().splitlines()[-1:] }", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:])
