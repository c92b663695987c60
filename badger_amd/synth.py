"""Synthetic inputs for tests and bench.py.

There is no 10x whitelist or ONT data in the image (no network), so both are
generated: a whitelist of W distinct uniform 16-mers, and ONT-like reads
  junk U[0,40] + R1 + barcode + UMI + T*30 + cDNA
with total length lognormal(ln 900, 0.5) clipped to [200, 8000], half of them
reverse-complemented, then iid per-base errors (sub 3 %, ins 2 %, del 3 %).
Written with torch ops so the 1M-read bench workload is built on the GPU; every random draw is a
hash of (seed, purpose, index) in integer arithmetic, so the same call gives the same bytes on the CPU
(where the workload can be regenerated without a GPU) and on any GPU.
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


# ---- a counter-based generator: value = hash(seed, stream, index), integer arithmetic only, so that the same reads come out on
# ---- the CPU and on the GPU (a torch device generator gives different streams per device; numpy's cannot run on the GPU)
_M64 = (1 << 64) - 1


def _i64(x):
    """python int (mod 2^64) -> the int64 with the same bits"""
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(x, k):
    """logical right shift of an int64 tensor"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _hash64(seed, stream, idx):
    """splitmix64 finaliser of (seed, stream, idx); idx: int64 tensor -> int64 tensor of 64 random bits (wrapping multiplies)"""
    x = idx * _i64(0x9E3779B97F4A7C15) + _i64((int(seed) * 0xD1342543DE82EF95 + int(stream) * 0xAF251AF3B0F025B5 + 0x2545F4914F6CDD1D))
    x = (x ^ _lsr(x, 30)) * _i64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _i64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def _rand_below(seed, stream, idx, k):
    """uniform integers in [0, k) (k < 2^31): top 32 bits of the hash times k, shifted down"""
    return (_lsr(_hash64(seed, stream, idx), 32) * int(k)) >> 32


def _rand_u24(seed, stream, idx):
    return _lsr(_hash64(seed, stream, idx), 40)                   # 24 random bits: p < x  <=>  u24 < x * 2^24


def _table_lengths(seed):
    """65,536 quantiles of lognormal(ln 900, 0.5) clipped to [200, 8000]: total read lengths are drawn by index"""
    q = (np.arange(65536, dtype=np.float64) + 0.5) / 65536.0
    # inverse normal CDF by bisection-free rational approximation (Acklam), pure numpy, deterministic
    a = [-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00]
    b = [-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01]
    c = [-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00]
    d = [7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00]
    z = np.empty_like(q)
    lo, hi = q < 0.02425, q > 1 - 0.02425
    mid = ~(lo | hi)
    t = np.sqrt(-2 * np.log(q[lo]))
    z[lo] = (((((c[0] * t + c[1]) * t + c[2]) * t + c[3]) * t + c[4]) * t + c[5]) / ((((d[0] * t + d[1]) * t + d[2]) * t + d[3]) * t + 1)
    t = np.sqrt(-2 * np.log(1 - q[hi]))
    z[hi] = -(((((c[0] * t + c[1]) * t + c[2]) * t + c[3]) * t + c[4]) * t + c[5]) / ((((d[0] * t + d[1]) * t + d[2]) * t + d[3]) * t + 1)
    t = q[mid] - 0.5
    r = t * t
    z[mid] = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * t / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1)
    return np.clip(np.rint(np.exp(math.log(900.0) + 0.5 * z)), 200, 8000).astype(np.int64)


def make_reads(n, whitelist, seed=1, device="cpu", umi_len=12, n_cells=5000,
               p_sub=0.03, p_ins=0.02, p_del=0.03, chunk=50000, with_truth=False):
    """-> (bases uint8[total] ASCII, off int64[n+1]) on `device` (+ truth dict).  The same (n, whitelist, seed) give
    the same bytes on every device: every random draw is a hash of (seed, purpose, index) in 64-bit integer arithmetic,
    tables (cell choice, read lengths) are built with numpy on the host."""
    dev = torch.device(device)
    wl_np = np.ascontiguousarray(whitelist).astype(np.int64)
    n_cells = min(n_cells, len(wl_np))
    host = np.random.default_rng(int(seed))                             # host-side tables only: which cells, how big
    cells = torch.from_numpy(wl_np[host.permutation(len(wl_np))[:n_cells]]).to(dev)
    w = np.exp(host.standard_normal(n_cells))                            # lognormal(sigma = 1) cell sizes
    cum = np.floor(np.cumsum(w) / w.sum() * (1 << 32)).astype(np.int64)
    cum[-1] = 1 << 32
    cell_cum = torch.from_numpy(cum).to(dev)
    len_tab = torch.from_numpy(_table_lengths(seed)).to(dev)
    r1 = torch.tensor([_CODE[c] for c in R1], dtype=torch.uint8, device=dev)
    ascii_lut = torch.from_numpy(_ASCII.copy()).to(dev)
    T_DEL, T_SUB, T_INS = int(p_del * (1 << 24)), int((p_del + p_sub) * (1 << 24)), int((p_del + p_sub + p_ins) * (1 << 24))

    out_bases, out_len, truth_bc, truth_rc = [], [], [], []
    base_index = 0                                                       # running index of pre-error bases: the per-base streams
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        ridx = torch.arange(c0, c0 + m, device=dev, dtype=torch.int64)   # global read index: the per-read streams
        is_cell = _rand_u24(seed, 1, ridx) < int(0.95 * (1 << 24))
        draw = torch.searchsorted(cell_cum, _lsr(_hash64(seed, 2, ridx), 32), right=True).clamp(max=n_cells - 1)
        rnd_bc = _lsr(_hash64(seed, 3, ridx), 32)
        bc = torch.where(is_cell, cells[draw], rnd_bc)
        junk = _rand_below(seed, 4, ridx, 41)
        total = len_tab[_lsr(_hash64(seed, 5, ridx), 48)]
        fixed = junk + len(R1) + 16 + umi_len + 30
        L = torch.maximum(total, fixed)
        rc = (_hash64(seed, 6, ridx) & 1) == 1
        off = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(L, 0)
        N = int(off[-1])
        rid = torch.repeat_interleave(torch.arange(m, device=dev), L)
        pos = torch.arange(N, device=dev) - off[rid]
        gidx = torch.arange(base_index, base_index + N, device=dev, dtype=torch.int64)
        base_index += N
        h = _hash64(seed, 7, gidx)                                       # one hash per base: bits 62-63 the base, 38-61 the error
        codes = _lsr(h, 62).to(torch.uint8)                              # draw, 16-31 the substitution, 32-33 the inserted base
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
        u = (h >> 38) & 0xFFFFFF
        is_del = u < T_DEL
        is_sub = (u >= T_DEL) & (u < T_SUB)
        is_ins = (u >= T_SUB) & (u < T_INS)
        shift = (1 + (((h >> 16) & 0xFFFF) * 3 >> 16)).to(torch.uint8)  # 1..3
        codes = torch.where(is_sub, (codes + shift) & 3, codes)
        ins_code = ((h >> 32) & 3).to(torch.uint8)
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
