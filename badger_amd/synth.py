"""Synthetic inputs for tests and bench.py.

There is no 10x whitelist or ONT data in the image (no network), so both are
generated: a whitelist of W distinct uniform 16-mers, and ONT-like reads
  junk U[0,40] + R1 + barcode + UMI + T*30 + cDNA
with total length lognormal(ln 900, 0.5) clipped to [200, 8000], half of them
reverse-complemented, then iid per-base errors (sub 3 %, ins 2 %, del 3 %).
Written with torch ops so the 1M-read bench workload is built on the GPU; the
same code runs on CPU for the small test cases.
"""
import math

import numpy as np
import torch

R1 = "CTACACGACGCTCTTCCGATCT"          # reference barcode_callers.py:154
_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}  # reference common.py:11-14 (rank encoding)
_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_whitelist(W, seed=20250711):
    """W distinct uniform 16-mers as rank-packed uint32, sorted ascending."""
    rng = np.random.default_rng(seed)
    got = np.empty(0, dtype=np.uint32)
    while len(got) < W:
        draw = rng.integers(0, 1 << 32, size=int(W * 1.1) + 16, dtype=np.uint64).astype(np.uint32)
        cat = np.concatenate([got, draw])
        _, first = np.unique(cat, return_index=True)
        got = cat[np.sort(first)]            # first-occurrence order, de-duplicated
    return np.sort(got[:W])


def rank_to_str(rk):
    rk = int(rk)
    return "".join("ACGT"[(rk >> (2 * i)) & 3] for i in range(16))


def str_to_rank(s):
    rk = 0
    for i, ch in enumerate(s[:16]):
        rk |= _CODE[ch] << (2 * i)
    return rk


def make_reads(n, whitelist, seed=1, device="cpu", umi_len=12, n_cells=5000,
               p_sub=0.03, p_ins=0.02, p_del=0.03, chunk=50000, with_truth=False):
    """-> (bases uint8[total] ASCII, off int64[n+1]) on `device` (+ truth dict)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    wl = torch.from_numpy(np.ascontiguousarray(whitelist).astype(np.int64)).to(dev)
    n_cells = min(n_cells, len(wl))
    cells = wl[torch.randperm(len(wl), generator=g, device=dev)[:n_cells]]
    weights = torch.exp(torch.randn(n_cells, generator=g, device=dev))
    r1 = torch.tensor([_CODE[c] for c in R1], dtype=torch.uint8, device=dev)
    ascii_lut = torch.from_numpy(_ASCII.copy()).to(dev)

    out_bases, out_len, truth_bc, truth_rc = [], [], [], []
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        is_cell = torch.rand(m, generator=g, device=dev) < 0.95
        draw = torch.multinomial(weights, m, replacement=True, generator=g)
        rnd_bc = torch.randint(0, 1 << 32, (m,), generator=g, device=dev, dtype=torch.int64)
        bc = torch.where(is_cell, cells[draw], rnd_bc)
        junk = torch.randint(0, 41, (m,), generator=g, device=dev, dtype=torch.int64)
        total = torch.exp(math.log(900.0) + 0.5 * torch.randn(m, generator=g, device=dev))
        total = total.clamp(200, 8000).round().to(torch.int64)
        fixed = junk + len(R1) + 16 + umi_len + 30
        L = torch.maximum(total, fixed)
        rc = torch.rand(m, generator=g, device=dev) < 0.5
        off = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(L, 0)
        N = int(off[-1])
        rid = torch.repeat_interleave(torch.arange(m, device=dev), L)
        pos = torch.arange(N, device=dev) - off[rid]
        codes = torch.randint(0, 4, (N,), generator=g, device=dev, dtype=torch.uint8)
        rel = pos - junk[rid]
        in_r1 = (rel >= 0) & (rel < 22)
        codes[in_r1] = r1[rel[in_r1]]
        in_bc = (rel >= 22) & (rel < 38)
        codes[in_bc] = ((bc[rid[in_bc]] >> (2 * (rel[in_bc] - 22))) & 3).to(torch.uint8)
        t0 = 38 + umi_len
        in_t = (rel >= t0) & (rel < t0 + 30)
        codes[in_t] = 3
        # reverse-complement half of the reads
        src = torch.where(rc[rid], off[rid] + L[rid] - 1 - pos, torch.arange(N, device=dev))
        codes = torch.where(rc[rid], 3 - codes[src], codes)
        # sequencing errors
        u = torch.rand(N, generator=g, device=dev)
        is_del = u < p_del
        is_sub = (u >= p_del) & (u < p_del + p_sub)
        is_ins = (u >= p_del + p_sub) & (u < p_del + p_sub + p_ins)
        shift = torch.randint(1, 4, (N,), generator=g, device=dev, dtype=torch.uint8)
        codes = torch.where(is_sub, (codes + shift) & 3, codes)
        ins_code = torch.randint(0, 4, (N,), generator=g, device=dev, dtype=torch.uint8)
        keep = ~is_del
        cnt = keep.to(torch.int64) + is_ins.to(torch.int64)
        cum = torch.zeros(N + 1, dtype=torch.int64, device=dev)
        cum[1:] = torch.cumsum(cnt, 0)
        out = torch.empty(int(cum[-1]), dtype=torch.uint8, device=dev)
        out[cum[:-1][keep]] = codes[keep]
        out[(cum[:-1] + keep.to(torch.int64))[is_ins]] = ins_code[is_ins]
        new_off = cum[off]
        out_bases.append(ascii_lut[out.to(torch.int64)])
        out_len.append(new_off[1:] - new_off[:-1])
        if with_truth:
            truth_bc.append(bc)
            truth_rc.append(rc)
    bases = torch.cat(out_bases)
    lens = torch.cat(out_len)
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(lens, 0)
    if with_truth:
        return bases, off, {"barcode": torch.cat(truth_bc), "revcomp": torch.cat(truth_rc)}
    return bases, off


def reads_to_list(bases, off):
    """Concatenated buffer -> list[str] (host, small inputs only)."""
    b = bases.cpu().numpy().tobytes()
    o = off.cpu().numpy()
    return [b[int(o[i]):int(o[i + 1])].decode("ascii") for i in range(len(o) - 1)]


def list_to_reads(seqs):
    """list[str] -> (uint8 ndarray, uint64 ndarray[n+1])."""
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    bases = np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)
    return bases, off
