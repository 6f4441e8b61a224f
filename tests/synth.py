"""Seeded synthetic inputs shared by tests and bench: uniform random 20-mers with ~2.5 % duplicated
lines, sorted as text; guides = 80 % a site with 0-4 substitutions, 20 % random."""
import numpy as np


def text_order_key(sig, seq_len=20):
    """Key whose numeric order equals the text order of the 20-mer (A<C<G<T, position 0 first)."""
    sig = sig.astype(np.uint64)
    key = np.zeros_like(sig)
    for j in range(seq_len):
        key |= ((sig >> np.uint64(2 * j)) & np.uint64(3)) << np.uint64(2 * (seq_len - 1 - j))
    return key


def random_sites(n_lines, seed, dup_frac=0.025, seq_len=20):
    """-> (sigs sorted in text order & de-duplicated, occurrences), like the state of isslCreateIndex.cpp:199-200."""
    rng = np.random.default_rng(seed)
    n_base = int(n_lines / (1 + dup_frac))
    base = rng.integers(0, 1 << (2 * seq_len), size=n_base, dtype=np.uint64)
    dup = base[rng.integers(0, n_base, size=n_lines - n_base)]
    allsig = np.concatenate([base, dup])
    key = text_order_key(allsig, seq_len)
    order = np.argsort(key, kind="stable")
    allsig = allsig[order]
    key = key[order]
    first = np.ones(len(key), dtype=bool)
    first[1:] = key[1:] != key[:-1]
    idx = np.flatnonzero(first)
    occ = np.diff(np.append(idx, len(key))).astype(np.uint32)
    return allsig[idx], occ


def random_guides(sigs, n_guides, seed, seq_len=20):
    rng = np.random.default_rng(seed)
    g = np.empty(n_guides, dtype=np.uint64)
    pick = sigs[rng.integers(0, len(sigs), size=n_guides)]
    rnd = rng.integers(0, 1 << (2 * seq_len), size=n_guides, dtype=np.uint64)
    nsub = rng.integers(0, 5, size=n_guides)
    is_rand = (np.arange(n_guides) % 5) == 4
    for i in range(n_guides):
        if is_rand[i]:
            g[i] = rnd[i]
            continue
        s = int(pick[i])
        for p in rng.choice(seq_len, size=int(nsub[i]), replace=False):
            old = (s >> (2 * int(p))) & 3
            new = (old + int(rng.integers(1, 4))) & 3
            s = (s & ~(3 << (2 * int(p)))) | (new << (2 * int(p)))
        g[i] = s
    return g


def sigs_to_text(sigs, occ=None, seq_len=20):
    """Sorted site list text for the index builders (occurrences expanded)."""
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    sigs = np.asarray(sigs, dtype=np.uint64)
    if occ is not None:
        sigs = np.repeat(sigs, occ)
    arr = np.empty((len(sigs), seq_len + 1), dtype=np.uint8)
    for j in range(seq_len):
        arr[:, j] = letters[((sigs >> np.uint64(2 * j)) & np.uint64(3)).astype(np.int64)]
    arr[:, seq_len] = ord("\n")
    return arr.tobytes()
