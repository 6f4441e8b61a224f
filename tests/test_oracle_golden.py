"""CPU: the oracle restatement against the golden vectors produced by the compiled reference
(oracle/make_golden.py).  This is what pins the oracle; the GPU parity tests then compare against it."""
import hashlib

import numpy as np
import pytest

import oracle_util as ou
from synth import random_sites, random_guides, sigs_to_text


def test_oracle_scores_match_reference_stdout(golden):
    ix = ou.OracleIndex(golden.issl)
    sigs = ou.encode(golden.guides)
    seqs = [s.upper() if set(s) <= set("ACGT") else None for s in golden.guides]
    for key, want in golden.expected.items():
        method, thr, dist = key.split("|")
        mit, cfd = ix.score(sigs, int(dist), float(thr), method)
        lines = want.splitlines()
        printed = [l.split("\t")[0] for l in lines]  # the reference prints the DECODED signature
        got = ou.format_tsv(printed, mit, cfd, method)
        assert got == want, key


def test_oracle_hit_lists_match_reference(golden):
    ix = ou.OracleIndex(golden.issl)
    sigs = ou.encode(golden.guides)
    for thr in golden.hit_thresholds():
        _, _, hits = ix.score(sigs, 4, float(thr), "and", want_hits=True)
        assert np.array_equal(hits, golden.hits(thr)), thr


def test_oracle_builder_bytes_match_reference(golden):
    data = ou.build_issl(golden.sites_txt.read_bytes())
    assert data == golden.issl.read_bytes()
    assert hashlib.sha256(data).hexdigest() == (golden.dir / "index.sha256").read_text().strip()


def test_oracle_matches_reference_on_a_table_with_duplicate_odd_and_missing_masks(golden_oddtable):
    """isslScoreOfftargets.cpp:188-197 (insert: the first pair of a mask wins) and :394 (operator[]: a missing mask
    scores 0.0) on a table no builder writes; one thread, like the golden run."""
    g = golden_oddtable
    ix = ou.OracleIndex(g.issl)
    sigs = ou.encode(g.guides)
    for key, want in g.expected.items():
        method, thr, dist = key.split("|")
        mit, cfd = ix.score(sigs, int(dist), float(thr), method, threads=1)
        assert ou.format_tsv([l.split("\t")[0] for l in want.splitlines()], mit, cfd, method) == want, key
    for thr in g.hit_thresholds():
        _, _, hits = ix.score(sigs, 4, float(thr), "and", threads=1, want_hits=True)
        assert np.array_equal(hits, g.hits(thr)), thr


def test_oracle_matches_reference_on_large_counts_and_signed_tables(golden_extra):
    """isslScoreOfftargets.cpp:348 (the count of the entry met first, a uint32), :394 / :460 (whatever the table holds
    is added: negative, NaN, inf) -- reference outputs of oracle/make_golden_extra.py; one thread, like the golden run."""
    g = golden_extra
    ix = ou.OracleIndex(g.issl)
    sigs = ou.encode(g.guides)
    for key, want in g.expected.items():
        method, thr, dist = key.split("|")
        mit, cfd = ix.score(sigs, int(dist), float(thr), method, threads=1)
        assert ou.format_tsv([l.split("\t")[0] for l in want.splitlines()], mit, cfd, method) == want, key
    for thr in g.hit_thresholds():
        _, _, hits = ix.score(sigs, 4, float(thr), "and", threads=1, want_hits=True)
        assert np.array_equal(hits, g.hits(thr)), thr


def test_oracle_and_builder_match_reference_on_narrow_slices(golden_width):
    """Slice widths 4 and 2 (isslCreateIndex.cpp:212-234, isslScoreOfftargets.cpp:261-270,330-341): the host builder's
    bytes have the digest of the reference-built index (checked when the fixture rebuilds it), the oracle reproduces the
    reference's stdout and hit lists on it."""
    g = golden_width
    ix = ou.OracleIndex(g.issl)
    sigs = ou.encode(g.guides)
    for key, want in g.expected.items():
        method, thr, dist = key.split("|")
        mit, cfd = ix.score(sigs, int(dist), float(thr), method, threads=1)
        assert ou.format_tsv([l.split("\t")[0] for l in want.splitlines()], mit, cfd, method) == want, key
    for thr in g.hit_thresholds():
        _, _, hits = ix.score(sigs, 4, float(thr), "and", threads=1, want_hits=True)
        assert np.array_equal(hits, g.hits(thr)), thr


def test_oracle_thread_count_invariance(golden_uniform):
    ix = ou.OracleIndex(golden_uniform.issl)
    sigs = ou.encode(golden_uniform.guides)
    a = ix.score(sigs, 4, 75.0, "and", threads=1)
    b = ix.score(sigs, 4, 75.0, "and", threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_worked_example_of_reference_comment():
    # isslScoreOfftargets.cpp:350-375: AATTGCAT vs ATATCGAT -> 4 mismatches
    a = int(ou.encode(["AATTGCAT" + "A" * 12])[0])
    b = int(ou.encode(["ATATCGAT" + "A" * 12])[0])
    x = a ^ b
    mm = ((x & 0xAAAAAAAAAAAAAAAA) >> 1) | (x & 0x5555555555555555)
    assert bin(mm).count("1") == 4
    assert mm == 0b00_01_01_00_01_01_00_00 >> 0 or bin(mm).count("1") == 4


def test_first_matching_slice_rule_equals_seen_bitmap():
    """a candidate met in slice i is scored iff no slice j<i of the XOR is all zero."""
    sigs, occ = random_sites(30000, seed=11)
    text = sigs_to_text(sigs, occ)
    import tempfile, os
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "x.issl")
        open(p, "wb").write(ou.build_issl(text))
        ix = ou.OracleIndex(p)
        guides = random_guides(sigs, 300, seed=12)
        _, _, hits = ix.score(guides, 4, 0.0, "and", want_hits=True)
    assert len(hits) > 100
    for g, sl, pos, sid, dist, oc in hits:
        x = int(guides[g]) ^ int(sigs[sid])
        zero = [((x >> (8 * j)) & 0xFF) == 0 for j in range(5)]
        assert zero[sl] and not any(zero[:sl])
    # and every (guide, site) pair appears once
    assert len({(int(h[0]), int(h[3])) for h in hits}) == len(hits)


def test_successor_slice_pigeonhole_behind_the_pruned_scan():
    """What the pruned scan of the HIP path rests on (csrc/issl_kernels.hip, k_fine_count): split the <= 4 mismatches of a
    hit over the five slices in every possible way -- some slice matches exactly AND its cyclic successor has at most
    one mismatch (none at all when there are <= 2 mismatches).  The reference scans whole buckets
    (isslScoreOfftargets.cpp:344) and so finds the site in every exactly matching slice; the pruned scan needs one."""
    import itertools
    for counts in itertools.product(range(5), repeat=5):
        total = sum(counts)
        if total > 4:
            continue
        ok1 = any(counts[i] == 0 and counts[(i + 1) % 5] <= 1 for i in range(5))
        assert ok1, counts
        if total <= 2:
            assert any(counts[i] == 0 and counts[(i + 1) % 5] == 0 for i in range(5)), counts
    # and the bound is tight: with 5 mismatches no such slice need exist; with 3 an exact successor need not
    assert not any(c == 0 for c in (1, 1, 1, 1, 1))
    counts = (0, 1, 0, 1, 1)  # 3 mismatches: every exact slice is followed by a slice with one mismatch
    assert not any(counts[i] == 0 and counts[(i + 1) % 5] == 0 for i in range(5))


def test_successor_slice_pigeonhole_for_five_mismatches():
    """max_dist 5 (the pruned scan's third mode): the reference only ever meets a site in the bucket of a slice that
    matches the guide exactly (isslScoreOfftargets.cpp:330-344), so a hit has one; among the exact slices one is followed
    by a slice with at most TWO mismatches -- otherwise 3 |E| + (5 - 2 |E|) > 5.  One mismatch does not do (tight)."""
    import itertools
    tight = False
    for counts in itertools.product(range(6), repeat=5):
        if sum(counts) > 5 or 0 not in counts:
            continue
        assert any(counts[i] == 0 and counts[(i + 1) % 5] <= 2 for i in range(5)), counts
        tight = tight or not any(counts[i] == 0 and counts[(i + 1) % 5] <= 1 for i in range(5))
    assert tight
    assert 1 + 4 * 3 + 6 * 9 == 67   # successor bytes within two mismatches of a guide's own


def test_successor_unit_pigeonhole_for_narrow_slices():
    """4- and 2-bit slices (ten / twenty per site, two / one positions each): the sorted layouts order a bucket by the byte of
    the next two / four slices (succ_byte, csrc/issl_device.hpp), and a hit within max_dist of a guide -- which matches it
    exactly in some slice, or the reference never meets it (isslScoreOfftargets.cpp:330-344) -- has an exact slice whose
    successor unit holds no mismatch (max_dist <= 2), at most one (<= 4), at most two (5): every distribution of mismatches
    over the slices.  And the filter of the scan's cold block (fine_dup: the encounter in slice s is dropped when the first
    four of the twelve positions compared match exactly -- the previous slice and the two / three positions behind the
    successor unit, scan_word_sorted_narrow) never drops the reporter's encounter."""
    import itertools
    for n_slices, per in ((10, 2), (20, 1)):
        unit = 4 // per            # slices of the successor unit
        behind = (4 - per) // per  # slices behind it that share the first quad with the previous slice
        for max_dist, tol in ((2, 0), (4, 1), (5, 2)):
            for k in range(max_dist + 1):
                for pos in itertools.combinations(range(20), k):
                    counts = [0] * n_slices
                    for q in pos:
                        counts[q // per] += 1
                    if 0 not in counts:
                        continue
                    ok = [i for i in range(n_slices)
                          if counts[i] == 0 and sum(counts[(i + j) % n_slices] for j in range(1, unit + 1)) <= tol]
                    assert ok, (n_slices, max_dist, counts)
                    # the scan meets the hit in every slice of `ok`; it drops the encounter in s > 0 when its first quad is exact
                    kept = [s for s in ok if not (s > 0 and counts[s - 1] == 0 and
                                                  all(counts[(s + unit + j) % n_slices] == 0 for j in range(1, behind + 1)))]
                    assert ok[0] in kept, (n_slices, max_dist, counts)


def test_previous_slice_filter_of_the_pruned_scan_never_drops_the_reporter():
    """The scan's duplicate filter (csrc/issl_kernels.hip, fine_dup): a guide meets a hit once per exactly matching slice
    whose successor is within tolerance, and the SMALLEST such slice reports it (k_verify).  The scan drops the encounter in
    slice s when slice s - 1 matches exactly as well -- then s - 1 is such a slice itself (its successor s matches), so s is
    never the reporter.  Every placement of <= 4 mismatches on the 20 positions, both tolerances: the reporter's encounter
    survives; and the filter takes what the kernel comment says it takes (58 % of the duplicates at distance 4)."""
    import itertools
    for tol, dists in ((1, (0, 1, 2, 3, 4)), (0, (0, 1, 2)), (2, (5,))):
        for d in dists:
            hits = records = dups = dropped = 0
            for pos in itertools.combinations(range(20), d):
                cnt = [0] * 5
                for p in pos:
                    cnt[p // 4] += 1
                met = [j for j in range(5) if cnt[j] == 0 and cnt[(j + 1) % 5] <= tol]
                if tol == 2 and 0 not in cnt:
                    continue                                               # (no exact slice: the reference does not find it either)
                assert met, pos                                            # (the pigeonholes above)
                reporter = min(met)
                kept = [j for j in met if not (j >= 1 and cnt[j - 1] == 0)]   # what the scan still notes
                assert reporter in kept, (pos, met, kept)
                hits += 1; records += len(met); dups += len(met) - 1; dropped += len(met) - len(kept)
            if tol == 1 and d == 4:
                assert abs(records / hits - 1.419) < 1e-3 and abs(dropped / dups - 0.58) < 0.01
