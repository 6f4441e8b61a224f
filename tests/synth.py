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


def random_sites_fast(n_lines, seed, threads=16, dup_frac=0.025):
    """Multi-billion-site variant of random_sites (not the same stream): 20-mers drawn directly in text order as 256
    independent chunks (one per leading four bases) on a thread pool; ~dup_frac of the sites occur twice."""
    from concurrent.futures import ThreadPoolExecutor
    per = int(n_lines / (1 + dup_frac)) // 256

    def chunk(c):
        rng = np.random.default_rng([seed, c])
        low = rng.integers(0, 1 << 32, size=per, dtype=np.uint64)
        low.sort()
        keep = np.ones(per, dtype=bool)
        keep[1:] = low[1:] != low[:-1]
        low = low[keep]
        occ = np.ones(len(low), dtype=np.uint32)
        occ[rng.random(len(low)) < dup_frac] = 2
        return text_order_key((np.uint64(c) << np.uint64(32)) | low), occ   # the key map is its own inverse

    with ThreadPoolExecutor(max_workers=threads) as pool:
        parts = list(pool.map(chunk, range(256)))
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


class DeviceSites:
    """A site table that lives in device memory (torch tensors), for tests at sizes whose host arrays would not fit the
    box: len() and fancy indexing (-> numpy, a gather on the device) are all the checkers need of it."""

    def __init__(self, tensor, dtype=np.uint64):
        self.d, self.dtype = tensor, dtype

    def __len__(self):
        return int(self.d.numel())

    def __getitem__(self, idx):
        import torch
        idx = torch.as_tensor(np.asarray(idx, dtype=np.int64), device=self.d.device)
        return self.d[idx].cpu().numpy().view(self.dtype)


def random_sites_device(n_lines, seed, device="cuda:0", threads=16, dup_frac=0.025):
    """random_sites_fast() -- the SAME sites for the same arguments -- delivered as device tensors: the 256 chunks are
    drawn on the host thread pool and copied into place one by one, so the host never holds more than the chunks in
    flight (12 B/site x 3 G lines is what kept BASELINE configs[4] out of the default suite).
    -> (d_sigs int64 view of the packed signatures, d_occ int32), text order, distinct."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    per = int(n_lines / (1 + dup_frac)) // 256
    d_sigs = torch.empty(per * 256, dtype=torch.int64, device=device)
    d_occ = torch.empty(per * 256, dtype=torch.int32, device=device)

    def chunk(c):  # (identical to random_sites_fast's)
        rng = np.random.default_rng([seed, c])
        low = rng.integers(0, 1 << 32, size=per, dtype=np.uint64)
        low.sort()
        keep = np.ones(per, dtype=bool)
        keep[1:] = low[1:] != low[:-1]
        low = low[keep]
        occ = np.ones(len(low), dtype=np.uint32)
        occ[rng.random(len(low)) < dup_frac] = 2
        return text_order_key((np.uint64(c) << np.uint64(32)) | low), occ

    at = 0
    with ThreadPoolExecutor(max_workers=threads) as pool:
        pending = []
        for c in range(256):  # at most threads + 4 chunks exist on the host at any time
            pending.append(pool.submit(chunk, c))
            if len(pending) >= threads + 4 or c == 255:
                for f in (pending if c == 255 else pending[:1]):
                    sig, occ = f.result()
                    d_sigs[at:at + len(sig)] = torch.from_numpy(sig.view(np.int64))
                    d_occ[at:at + len(sig)] = torch.from_numpy(occ.view(np.int32))
                    at += len(sig)
                pending = [] if c == 255 else pending[1:]
    return d_sigs[:at], d_occ[:at]


def neighbours_device(d_sigs, guide, max_dist, chunk=1 << 27):
    """Indices of the sites within max_dist mismatches of `guide`: brute force over the whole device-resident table with
    plain torch arithmetic (a checker: nothing of the product is involved)."""
    import torch
    g = int(guide)
    g = g - (1 << 64) if g >= (1 << 63) else g
    even, m2, m4, ones = 0x5555555555555555, 0x3333333333333333, 0x0F0F0F0F0F0F0F0F, 0x0101010101010101
    found = []
    for lo in range(0, d_sigs.numel(), chunk):
        x = d_sigs[lo:lo + chunk] ^ g
        x = (x | (x >> 1)) & even                      # one flag per mismatching position, on the even bits (bits 0..39)
        x = (x & m2) + ((x >> 2) & m2)
        x = (x + (x >> 4)) & m4
        cnt = (x * ones) >> 56                         # 40-bit inputs: the byte sums stay far below 256
        hit = torch.nonzero(cnt <= max_dist).flatten()
        if hit.numel():
            found.append((hit + lo).cpu().numpy())
    return np.concatenate(found) if found else np.empty(0, dtype=np.int64)


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


def random_guides_fast(sigs, n_guides, seed, seq_len=20):
    """Same recipe as random_guides (80 % = a site with 0-4 substitutions at distinct positions, 20 % random 20-mers),
    vectorised for the 100k-1M guide batches of bench.py; not the same random stream."""
    rng = np.random.default_rng(seed)
    g = sigs[rng.integers(0, len(sigs), size=n_guides)].astype(np.uint64)
    nsub = rng.integers(0, 5, size=n_guides)
    # four distinct positions per guide: the first columns of a random permutation of the positions
    pos = np.argsort(rng.random((n_guides, seq_len)), axis=1)[:, :4].astype(np.uint64)
    delta = rng.integers(1, 4, size=(n_guides, 4), dtype=np.uint64)   # old base + 1..3 (mod 4) is never the old base
    for k in range(4):
        use = nsub > k
        sh = np.uint64(2) * pos[:, k]
        old = (g >> sh) & np.uint64(3)
        new = (old + delta[:, k]) & np.uint64(3)
        g = np.where(use, (g & ~(np.uint64(3) << sh)) | (new << sh), g)
    is_rand = (np.arange(n_guides) % 5) == 4
    rnd = rng.integers(0, 1 << (2 * seq_len), size=n_guides, dtype=np.uint64)
    return np.where(is_rand, rnd, g).astype(np.uint64)


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


def markov_sites(n_lines, seed, seq_len=20, at_bias=0.62, order=3):
    """Skewed site list: order-3 Markov chain with an AT-rich stationary bias and context-dependent
    transition rows drawn once from a Dirichlet.  Buckets of AT-rich slices come out several times larger than
    GC-rich ones and near-duplicate neighbourhoods are frequent.  Returns (sigs, occ) like random_sites()."""
    rng = np.random.default_rng(seed)
    base_p = np.array([at_bias / 2, (1 - at_bias) / 2, (1 - at_bias) / 2, at_bias / 2])  # A C G T
    n_ctx = 4 ** order
    rows = rng.dirichlet(base_p * 6.0, size=n_ctx)          # one transition row per context
    cum = np.cumsum(rows, axis=1)
    cum[:, -1] = 1.0
    codes = np.empty((seq_len, n_lines), dtype=np.uint8)
    ctx = np.zeros(n_lines, dtype=np.int64)
    for p in range(seq_len):
        u = rng.random(n_lines)
        c = cum[ctx] if p >= order else np.broadcast_to(np.cumsum(base_p), (n_lines, 4))
        b = (u[:, None] > c[:, :3]).sum(axis=1).astype(np.uint8)
        codes[p] = b
        ctx = ((ctx * 4) + b) % n_ctx
    sig = np.zeros(n_lines, dtype=np.uint64)
    for p in range(seq_len):
        sig |= codes[p].astype(np.uint64) << np.uint64(2 * p)
    key = text_order_key(sig, seq_len)
    order_ix = np.argsort(key, kind="stable")
    sig = sig[order_ix]
    key = key[order_ix]
    first = np.ones(len(key), dtype=bool)
    first[1:] = key[1:] != key[:-1]
    idx = np.flatnonzero(first)
    occ = np.diff(np.append(idx, len(key))).astype(np.uint32)
    return sig[idx], occ


def markov_sites_fast(n_lines, seed, threads=16, seq_len=20, at_bias=0.62, order=3):
    """markov_sites() for hundreds of millions of lines: the same chain (same transition rows for the same seed), drawn
    in chunks on a thread pool with a 256-step inverse-CDF table per context (probabilities rounded to 1/256: synthetic
    data), sorted per leading four bases.  Not the same random stream as markov_sites()."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(seed)
    base_p = np.array([at_bias / 2, (1 - at_bias) / 2, (1 - at_bias) / 2, at_bias / 2])  # A C G T
    n_ctx = 4 ** order
    rows = rng.dirichlet(base_p * 6.0, size=n_ctx)

    def table(p):  # 256 equally likely bytes -> base, by the cumulative probabilities
        edges = np.round(np.cumsum(p) * 256).astype(np.int64)
        edges[-1] = 256
        return np.repeat(np.arange(4, dtype=np.uint8), np.diff(np.concatenate([[0], edges])))
    first_tab = table(base_p)
    ctx_tab = np.concatenate([table(r) for r in rows])          # [ctx * 256 + byte]
    n_chunks = max(threads * 4, 1)
    per = (n_lines + n_chunks - 1) // n_chunks

    def chunk(c):
        m = min(per, n_lines - c * per)
        if m <= 0:
            return np.empty(0, dtype=np.uint64)
        r = np.random.default_rng([seed, 7, c])
        key = np.zeros(m, dtype=np.uint64)                      # text-order key: position 0 in the top bits
        ctx = np.zeros(m, dtype=np.int64)
        for pos in range(seq_len):
            u = r.integers(0, 256, size=m, dtype=np.uint8)
            b = first_tab[u] if pos < order else ctx_tab[ctx * 256 + u]
            key |= b.astype(np.uint64) << np.uint64(2 * (seq_len - 1 - pos))
            ctx = ((ctx * 4) + b) % n_ctx
        key.sort()
        return key

    with ThreadPoolExecutor(max_workers=threads) as pool:
        parts = list(pool.map(chunk, range(n_chunks)))

        def lead(c):  # all keys whose leading four bases are c, sorted, distinct, with their multiplicities
            lo, hi = np.uint64(c) << np.uint64(32), np.uint64(c + 1) << np.uint64(32)
            k = np.concatenate([p[np.searchsorted(p, lo):np.searchsorted(p, hi)] for p in parts])
            k.sort()
            first = np.ones(len(k), dtype=bool)
            first[1:] = k[1:] != k[:-1]
            idx = np.flatnonzero(first)
            return text_order_key(k[idx], seq_len), np.diff(np.append(idx, len(k))).astype(np.uint32)  # the key map is its own inverse
        out = list(pool.map(lead, range(256)))
    return np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out])


def check_comparisons(ix, guides, prune=None):
    """Counters of the last call: what the reference would compare, what was planned, what the scan kernel counted."""
    st = ix.stats()
    expected = ix.count_candidates(guides)
    assert st["reference_comparisons"] == expected
    if prune == 0:
        assert st["pruned"] == 0
    if st["pruned"] == 0:
        assert st["candidates"] == expected == st["planned_comparisons"]
    else:   # a group's first and last tile also hold its neighbours' candidates
        assert st["planned_comparisons"] <= st["candidates"]
    return st
