"""BASELINE-size runs on the GPU, checked through size-independent properties and oracle samples
(the oracle cannot finish the full sizes in seconds): config 2 (1M reads vs the 737,280-entry
whitelist), config 3 (graph at thr 1 over 500K distinct barcodes) and one GPU's share of config 5
(4.9M-entry whitelist, thr 2 graph from row blocks) and of config 4 (12.5M reads in one batch)."""
import numpy as np
import pytest
import torch

from badger_amd import _native, synth

pytestmark = pytest.mark.gpu

N_READS = 1000000
N_WL = 737280


@pytest.fixture(scope="module")
def world():
    from oracle import pyoracle as orc
    dev = torch.device("cuda", 0)
    ctx = _native.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    wl = synth.make_whitelist(N_WL)
    bases, off, truth = synth.make_reads(N_READS, wl, seed=1, device=dev, with_truth=True)
    total = int(off[-1])
    bases = torch.cat([bases, torch.zeros(64, dtype=torch.uint8, device=dev)])
    recs = torch.zeros((N_READS, 8), dtype=torch.int32, device=dev)
    ctx.extract_batch_dev(bases, off.contiguous(), N_READS, total, 12, recs)
    rc, bad, nwin = ctx.extract_status()
    if rc == _native.E_CAPACITY:
        ctx.extract_batch_dev(bases, off.contiguous(), N_READS, total, 12, recs)
        rc, bad, nwin = ctx.extract_status()
    assert rc == 0
    torch.cuda.synchronize()
    return {"orc": orc, "ctx": ctx, "dev": dev, "wl": wl, "bases": bases, "off": off, "truth": truth,
            "recs": recs.cpu().numpy().view(_native.REC_DTYPE).reshape(-1), "total": total}


def test_config2_extract_properties_and_samples(world):
    recs, orc = world["recs"], world["orc"]
    valid = recs["valid"] == 1
    assert 0.98 < valid.mean() < 1.0
    # structural invariants of barcode_callers.py:204-229
    v = recs[valid]
    assert (v["bc_start"] == v["r1_end"] + 1).all() and (v["umi_start"] == v["r1_end"] + 17).all()
    assert (v["r1_score"] >= 9).all() and (v["r1_score"] <= 22).all()
    assert ((v["polyT"] == -1) | (v["polyT"] - v["r1_end"] >= 16)).all()
    assert (v["umi_end"] > v["umi_start"]).all()
    assert ((recs["strand"] == 0) == (recs["polyT"] == -1)).all()
    iv = recs[~valid]
    assert (iv["r1_end"] == -1).all() and (iv["r1_score"] == 0).all() and (iv["bc_start"] == -1).all()
    # the simulator's truth: strand almost always recovered, and the barcode exactly as often as 16 clean bases occur
    rev = (recs["flags"] & 1).astype(bool)
    assert (rev == world["truth"]["revcomp"].cpu().numpy())[valid].mean() > 0.99
    ok = (recs["flags"] & 2) != 0
    exact = recs["bc_rank"][ok] == world["truth"]["barcode"].cpu().numpy().astype(np.uint32)[ok]
    assert 0.22 < exact.mean() < 0.30          # 0.92 ** 16 = 0.263
    # oracle on three contiguous samples (start, middle, end of the batch)
    off = world["off"].cpu().numpy().astype(np.uint64)
    for lo in (0, N_READS // 2, N_READS - 3000):
        hi = lo + 3000
        b = world["bases"][int(off[lo]):int(off[hi])].cpu().numpy()
        want = orc.extract_batch(b, off[lo:hi + 1] - off[lo], 12, threads=8)
        assert (recs[lo:hi] == want).all()


def test_config2_rerun_is_deterministic_and_partition_invariant(world):
    ctx, dev = world["ctx"], world["dev"]
    off = world["off"]
    lo, hi = 400000, 650000
    n = hi - lo
    sub_off = (off[lo:hi + 1] - off[lo]).contiguous()
    a0, a1 = int(off[lo]), int(off[hi])
    # the sub-buffer must start 16-byte aligned: copy it
    sub = torch.cat([world["bases"][a0:a1], torch.zeros(64, dtype=torch.uint8, device=dev)])
    out = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    ctx.extract_batch_dev(sub, sub_off, n, a1 - a0, 12, out)
    assert ctx.extract_status()[0] == 0
    got = out.cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
    assert (got == world["recs"][lo:hi]).all()


def test_config2_nearest_properties_and_samples(world):
    ctx, dev, wl, orc = world["ctx"], world["dev"], world["wl"], world["orc"]
    recs = world["recs"]
    q = torch.from_numpy(recs["bc_rank"].astype(np.int64)).to(dev).to(torch.int32)
    ctx.whitelist_load(wl)
    bi = torch.zeros(N_READS, dtype=torch.int32, device=dev)
    be = torch.zeros(N_READS, dtype=torch.uint8, device=dev)
    bt = torch.zeros(N_READS, dtype=torch.int16, device=dev)
    ctx.nearest16_set_algo(2)
    ctx.nearest16_dev(q, N_READS, 2, bi, be, bt)
    torch.cuda.synchronize()
    idx, ed, ties = bi.cpu().numpy().astype(np.uint32), be.cpu().numpy(), bt.cpu().numpy().astype(np.uint16)
    hit = ed != 255
    assert ((idx == 0xFFFFFFFF) == ~hit).all() and ((ties == 0) == ~hit).all() and (ed[hit] <= 2).all()
    qh = recs["bc_rank"]
    assert (wl[idx[ed == 0]] == qh[ed == 0]).all()                                    # distance 0 is membership
    x = wl[idx[ed == 1]] ^ qh[ed == 1]
    assert (np.array([bin(int((v | (v >> 1)) & 0x55555555)).count("1") for v in x[:20000]]) == 1).all()   # distance 1 = one substitution
    # the record-strided entry point (what bench.py calls): same answers for usable barcodes, "no hit" for the others
    d_recs = torch.from_numpy(recs.view(np.int32).reshape(-1, 8).copy()).to(dev)
    ri = torch.zeros(N_READS, dtype=torch.int32, device=dev)
    re = torch.zeros(N_READS, dtype=torch.uint8, device=dev)
    rt = torch.zeros(N_READS, dtype=torch.int16, device=dev)
    ctx.nearest16_recs_dev(d_recs, N_READS, 2, ri, re, rt)
    torch.cuda.synchronize()
    usable = (recs["flags"] & 2) != 0
    assert 0.95 < usable.mean() < 1.0
    r_idx, r_ed, r_ties = ri.cpu().numpy().astype(np.uint32), re.cpu().numpy(), rt.cpu().numpy().astype(np.uint16)
    assert (r_idx[usable] == idx[usable]).all() and (r_ed[usable] == ed[usable]).all() and (r_ties[usable] == ties[usable]).all()
    assert (r_idx[~usable] == 0xFFFFFFFF).all() and (r_ed[~usable] == 255).all() and (r_ties[~usable] == 0).all()
    # idempotence: a whitelist entry calls itself
    ctx.nearest16_dev(torch.from_numpy(wl[:100000].astype(np.int64)).to(dev).to(torch.int32), 100000, 2, bi, be, bt)
    torch.cuda.synchronize()
    assert (bi[:100000].cpu().numpy() == np.arange(100000)).all() and (be[:100000].cpu().numpy() == 0).all()
    # exhaustive Myers scan of the whole whitelist (the other device path) on a sample; the oracle on a smaller one
    sel = np.random.default_rng(0).integers(0, N_READS, 1024)
    qs = torch.from_numpy(qh[sel].astype(np.int64)).to(dev).to(torch.int32)
    ctx.nearest16_set_algo(1)
    ctx.nearest16_dev(qs, 1024, 2, bi, be, bt)
    torch.cuda.synchronize()
    assert (bi[:1024].cpu().numpy().astype(np.uint32) == idx[sel]).all()
    assert (be[:1024].cpu().numpy() == ed[sel]).all() and (bt[:1024].cpu().numpy().astype(np.uint16) == ties[sel]).all()
    wi, we, wt = orc.nearest16(qh[sel[:24]], wl, 2, threads=8)
    assert (wi == idx[sel[:24]]).all() and (we == ed[sel[:24]]).all() and (wt == ties[sel[:24]]).all()
    ctx.nearest16_set_algo(0)
    # the oracle on 150,000 of the million calls (its neighbourhood-probe form, pinned to the exhaustive scan by
    # test_nearest16_probe_form_equals_exhaustive_scan): index, distance and tie count of every one
    big = np.random.default_rng(1).choice(N_READS, 150000, replace=False)
    wi, we, wt = orc.nearest16(qh[big], wl, 2, threads=16, probe=True)
    assert (wi == idx[big]).all() and (we == ed[big]).all() and (wt == ties[big]).all()


def test_config3_graph_probe_equals_scan_and_oracle_rows(world):
    ctx, orc = world["ctx"], world["orc"]
    recs = world["recs"]
    ranks = np.unique(recs["bc_rank"][(recs["flags"] & 2) != 0])
    assert len(ranks) > 500000
    ranks = ranks[:500000]
    ctx.graph_set_algo(2)
    e_probe = ctx.graph_edges(ranks, 1, 5)
    sub = ranks[:120000]
    ctx.graph_set_algo(2)
    e_sub_probe = ctx.graph_edges(sub, 1, 5)
    ctx.graph_set_algo(1)
    e_sub_scan = ctx.graph_edges(sub, 1, 5)
    ctx.graph_set_algo(0)
    assert len(e_sub_probe) == len(e_sub_scan) and (e_sub_probe == e_sub_scan).all()
    assert (e_probe["a"] < e_probe["b"]).all() and (e_probe["dist"] <= 1).all()
    key = e_probe["a"].astype(np.uint64) << np.uint64(32) | e_probe["b"].astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) > 0).all()                                   # sorted, no duplicates
    # complete rows.  dmin(a, b) <= 1 puts b in a's 176-string ball: a with one substitution (ed(a, b) = 1), a[:15] with one
    # base inserted (ed(a[:-1], b) = 1), or a with one base deleted followed by any base (ed(a, b[:-1]) = 1) - built here
    # from strings, independently of the kernel's rank arithmetic.  Every ball member that is a later row must be an
    # edge iff the oracle's S and dmin say so, and the kernel may report nothing outside the ball.
    rng = np.random.default_rng(4)
    present = set(ranks.tolist())
    for a in ranks[rng.integers(0, len(ranks), 40)]:
        sa = synth.rank_to_str(int(a))
        ball = set()
        for p in range(16):
            for c in "ACGT":
                ball.add(sa[:p] + c + sa[p + 1:])                    # substitution
                ball.add(sa[:15][:p] + c + sa[:15][p:])              # insertion into a[:15] (slots 0..14)
                ball.add(sa[:p] + sa[p + 1:] + c)                    # deletion, then a free last base
        for c in "ACGT":
            ball.add(sa[:15] + c)                                    # insertion at slot 15 of a[:15]
        ball.discard(sa)
        want = []
        for sb in ball:
            b = synth.str_to_rank(sb)
            if b > int(a) and b in present and orc.dmin3(int(a), b) <= 1 and orc.qgram_S(int(a), b) >= 5:
                want.append((b, orc.dmin3(int(a), b)))
        mine = sorted((int(x["b"]), int(x["dist"])) for x in e_probe[e_probe["a"] == a])
        assert mine == sorted(want)
    # and the whole list against the oracle's bucket method on the same 500,000 rows
    want, _, _ = orc.graph_edges_sampled(ranks, 1, 1, 5, threads=16, cap=len(e_probe) + 1)
    assert len(want) == len(e_probe) and (want == e_probe).all()
    # the one-deletion join (what thr 1 runs on at this size) gives the same list, as a whole and in 8 shares
    for algo in (6, 0):
        ctx.graph_set_algo(algo)
        e_join = ctx.graph_edges(ranks, 1, 5)
        assert len(e_join) == len(want) and (e_join == want).all(), algo
    import torch
    d_ranks = torch.from_numpy(ranks.view(np.int32)).cuda()
    cap = len(want) + 1024
    d_out = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    parts = []
    for part in range(8):
        ctx.graph_edges_part_dev(d_ranks, len(ranks), part, 8, 1, 5, d_out, cap, d_cnt)
        torch.cuda.synchronize()
        parts.append(d_out[:int(d_cnt[0])].cpu().numpy().view(np.uint32).copy())
    assert max(len(x) for x in parts) < 1.2 * min(len(x) for x in parts)
    e = np.concatenate(parts)
    e = e[np.lexsort((e[:, 1], e[:, 0]))]
    assert len(e) == len(want) and (e[:, 0] == want["a"]).all() and (e[:, 1] == want["b"]).all() and (e[:, 2] == want["dist"]).all()


def test_config2_distinct_on_device_matches_host_counting(world):
    """bdg_distinct_dev on the 1M records == numpy's unique/first-index/counts (barcode_graph.py:192-204)."""
    ctx, dev = world["ctx"], world["dev"]
    recs = world["recs"]
    d_recs = torch.from_numpy(recs.view(np.int32).reshape(-1, 8).copy()).to(dev)
    uniq = torch.zeros(N_READS, dtype=torch.int32, device=dev)
    cnt = torch.zeros(N_READS, dtype=torch.int32, device=dev)
    first = torch.zeros(N_READS, dtype=torch.int32, device=dev)
    dn = torch.zeros(2, dtype=torch.int32, device=dev)
    ctx.distinct_dev(d_recs, N_READS, uniq, cnt, first, dn)
    torch.cuda.synchronize()
    nu = int(dn[0])
    ok = (recs["valid"] == 1) & ((recs["flags"] & 2) != 0)
    idx = np.nonzero(ok)[0]
    wu, wf, wc = np.unique(recs["bc_rank"][ok], return_index=True, return_counts=True)
    assert nu == len(wu) and int(dn[1]) == 0
    assert (uniq[:nu].cpu().numpy().astype(np.uint32) == wu).all()
    assert (cnt[:nu].cpu().numpy() == wc).all()
    assert (first[:nu].cpu().numpy() == idx[wf]).all()
    # first-occurrence order of the reference's counts dict
    order = np.argsort(first[:nu].cpu().numpy(), kind="stable")
    assert (np.diff(first[:nu].cpu().numpy()[order]) > 0).all()


def test_config5_visium_scale_whitelist_and_thr2_graph(world):
    """BASELINE config 5 on one GPU's share: nearest16 against a 4.9M-entry whitelist (pair-table buckets are chains of
    several blocks there, the deletion map is 7 % full) and the thr = 2 graph (all-pairs sweep with the lossy q-gram
    statistic) built from row blocks the way 8 GPUs would split it."""
    from badger_amd import dist as bdist
    orc, dev, recs = world["orc"], world["dev"], world["recs"]
    rng = np.random.default_rng(55)
    wl = synth.make_whitelist(4900000)
    ctx = _native.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.whitelist_load(wl)
    # queries: whitelist entries as they are, with 1-3 substitutions, with a deletion (tail shifts in), and random 16-mers
    nq = 200000
    src = wl[rng.integers(0, len(wl), nq)].astype(np.uint64)
    kind = rng.integers(0, 5, nq)
    q = src.copy()
    for rounds, sel in ((1, kind == 1), (2, kind == 2), (3, kind == 3)):
        for _ in range(rounds):
            q = np.where(sel, q ^ (rng.integers(1, 4, nq).astype(np.uint64) << (2 * rng.integers(0, 16, nq).astype(np.uint64))), q)
    pos = rng.integers(0, 16, nq).astype(np.uint64)
    low = (np.uint64(1) << (np.uint64(2) * pos)) - np.uint64(1)
    dele = (q & low) | ((q >> np.uint64(2)) & ~low & np.uint64(0x3FFFFFFF)) | (rng.integers(0, 4, nq).astype(np.uint64) << np.uint64(30))
    q = np.where(kind == 4, dele, q)
    q = np.where(rng.random(nq) < 0.1, rng.integers(0, 1 << 32, nq, dtype=np.uint64), q).astype(np.uint32)
    d_q = torch.from_numpy(q.view(np.int32)).to(dev)
    bi = torch.zeros(nq, dtype=torch.int32, device=dev)
    be = torch.zeros(nq, dtype=torch.uint8, device=dev)
    bt = torch.zeros(nq, dtype=torch.int16, device=dev)
    ctx.nearest16_dev(d_q, nq, 2, bi, be, bt)
    torch.cuda.synchronize()
    idx, ed, ties = bi.cpu().numpy().view(np.uint32), be.cpu().numpy(), bt.cpu().numpy().view(np.uint16)
    hit = ed != 255
    assert hit.mean() > 0.9 and (ed[hit] <= 2).all()        # the radius-2 ball (~2,100 strings) times 4.9M entries covers the space ~2.4 times
    assert ((idx == 0xFFFFFFFF) == ~hit).all()
    assert (wl[idx[ed == 0]] == q[ed == 0]).all() and (ed[kind == 0][q[kind == 0] == src[kind == 0].astype(np.uint32)] == 0).all()
    # the exhaustive device path on a sample, the oracle on a smaller one
    sel = rng.integers(0, nq, 512)
    ctx.nearest16_set_algo(1)
    ctx.nearest16_dev(torch.from_numpy(q[sel].view(np.int32)).to(dev), 512, 2, bi, be, bt)
    torch.cuda.synchronize()
    assert (bi[:512].cpu().numpy().view(np.uint32) == idx[sel]).all() and (be[:512].cpu().numpy() == ed[sel]).all()
    assert (bt[:512].cpu().numpy().view(np.uint16) == ties[sel]).all()
    ctx.nearest16_set_algo(0)
    wi, we, wt = orc.nearest16(q[sel[:12]], wl, 2, threads=8)
    assert (wi == idx[sel[:12]]).all() and (we == ed[sel[:12]]).all() and (wt == ties[sel[:12]]).all()
    # ... and ALL 200,000 answers against the oracle's neighbourhood-probe form on the 4.9 M-entry list (postprocessing's
    # argmin, barcode_graph.py:370-385; the form is pinned to the exhaustive one by test_nearest16_probe_form_equals_exhaustive_scan)
    wi, we, wt = orc.nearest16(q, wl, 2, threads=16, probe=True)
    assert (wi == idx).all() and (we == ed).all() and (wt == ties).all()

    # thr = 2 graph at SURVEY 8d's config-5 size: 500K observed barcodes.  The deletion-variant join (what thr 2 runs on)
    # must give the all-pairs sweep's edge list and the q-gram join's, its 8 shares (as 8 GPUs would take them) and the
    # q-gram join's 8 row blocks must tile the list, the whole list must be the oracle's, and the oracle's graph of a subset
    # must be the restriction of the full graph.
    ranks = np.unique(recs["bc_rank"][(recs["flags"] & 2) != 0])[:500000]
    n, T = len(ranks), orc.qgram_threshold(2)
    assert n == 500000 and T == 4
    ctx.graph_set_algo(0)
    whole = ctx.graph_edges(ranks, 2, T)
    assert len(whole) > 100000 and (whole["dist"] <= 2).all() and (whole["a"] < whole["b"]).all()
    key = whole["a"].astype(np.uint64) << np.uint64(32) | whole["b"].astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) > 0).all()                                   # sorted, no duplicates
    ctx.graph_set_algo(1)
    sweep = ctx.graph_edges(ranks, 2, T)
    ctx.graph_set_algo(3)
    qjoin = ctx.graph_edges(ranks, 2, T)
    ctx.graph_set_algo(0)
    assert len(sweep) == len(whole) and (sweep == whole).all()
    assert len(qjoin) == len(whole) and (qjoin == whole).all()
    d_ranks = torch.from_numpy(ranks.view(np.int32)).to(dev)
    cap = len(whole) + 1024
    d_out = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    parts = []
    for part in range(8):                                   # the shares 8 GPUs would take, as bench.py asks for them
        ctx.graph_edges_part_dev(d_ranks, n, part, 8, 2, T, d_out, cap, d_cnt)
        torch.cuda.synchronize()
        parts.append(d_out[:int(d_cnt[0])].cpu().numpy().view(np.uint32).copy())
    sizes = [len(x) for x in parts]
    assert max(sizes) < 1.2 * min(sizes)                    # (the edges of a share follow its 14-mer groups: even within a few percent)
    ctx.graph_set_algo(3)                                   # the q-gram join by row blocks of equal pair counts
    rows = []
    for lo, hi in bdist.graph_row_blocks(n, 8, bdist.graph_balance(2)):
        ctx.graph_edges_rows_dev(d_ranks, n, lo, hi, 2, T, d_out, cap, d_cnt)
        torch.cuda.synchronize()
        rows.append(d_out[:int(d_cnt[0])].cpu().numpy().view(np.uint32).copy())
    ctx.graph_set_algo(0)
    er = np.concatenate(rows)
    er = er[np.lexsort((er[:, 1], er[:, 0]))]
    assert len(er) == len(whole) and (er[:, 0] == whole["a"]).all() and (er[:, 1] == whole["b"]).all() and (er[:, 2] == whole["dist"]).all()
    e = np.concatenate(parts)
    e = e[np.lexsort((e[:, 1], e[:, 0]))]
    assert len(e) == len(whole) and (e[:, 0] == whole["a"]).all() and (e[:, 1] == whole["b"]).all() and (e[:, 2] == whole["dist"]).all()
    # the whole list against the oracle's bucket method on the same 500,000 rows (a few seconds of host time)
    want, _, _ = orc.graph_edges_sampled(ranks, 2, 1, T, threads=16, cap=len(whole) + 1)
    assert len(want) == len(whole) and (want == whole).all()
    # the edge condition depends on the pair only: the oracle's graph of a subset == the full graph restricted to it
    sub = np.sort(ranks[rng.permutation(n)[:6000]])
    want = orc.graph_edges(sub, 2, T, threads=8)
    keep = np.isin(whole["a"], sub) & np.isin(whole["b"], sub)
    got = whole[keep]
    assert len(got) == len(want) and (got == want).all()


def test_offsets_across_the_32_bit_boundaries():
    """The scan kernel addresses a task's vectors relative to its first one and rebuilds 64-bit offsets from 32-bit halves
    (readfirstlane, DPP): the same reads placed so that their offsets straddle 2^31, 2^32 and 2^33 give the same records."""
    dev = torch.device("cuda", 0)
    wl = synth.make_whitelist(1000)
    b, o = synth.make_reads(3000, wl, seed=3)
    b, o = b.to(dev), o.to(dev)
    n, tot = len(o) - 1, int(o[-1])
    ctx = _native.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    want = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    ctx.extract_batch_dev(b, o, n, tot, 12, want)
    assert ctx.extract_status()[0] == 0
    for shift in ((1 << 31) - 1000003, (1 << 31) + 17, (1 << 32) - 999, (1 << 32) + 5000, (1 << 33) - 1000000):
        big = torch.full((shift + tot + 64,), 65, dtype=torch.uint8, device=dev)
        big[shift:shift + tot] = b[:tot]
        got = torch.zeros((n, 8), dtype=torch.int32, device=dev)
        ctx.extract_batch_dev(big, (o.to(torch.int64) + shift).contiguous(), n, shift + tot, 12, got)
        rc, bad, _ = ctx.extract_status()
        assert rc == 0, (shift, rc, bad)
        assert bool((got == want).all()), shift
        del big


def test_config4_one_gpu_share_of_100m_reads():
    """BASELINE config 4: 100M reads over 8 GPUs = 12.5M reads (12.6 GB of bases) per GPU in ONE batch.  Records of
    sampled ranges must equal what the same reads give when extracted alone (no dependence on batch size, queue
    segment or task placement) and what the oracle gives."""
    from oracle import pyoracle as orc
    dev = torch.device("cuda", 0)
    n = 12500000
    wl = synth.make_whitelist(N_WL)
    bases, off = synth.make_reads(n, wl, seed=4, device=dev)
    total = int(off[-1])
    assert total > 12 * 10 ** 9
    bases = torch.cat([bases, torch.zeros(64, dtype=torch.uint8, device=dev)])
    off = off.contiguous()
    ctx = _native.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    recs = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    for _ in range(2):
        ctx.extract_batch_dev(bases, off, n, total, 12, recs)
        rc, bad, nwin = ctx.extract_status()
        if rc != _native.E_CAPACITY:
            break
    assert rc == 0 and nwin > n
    valid = recs[:, 6].contiguous().view(torch.uint8).reshape(-1, 4)[:, 2]          # byte 26 of the record
    assert 0.98 < float(valid.float().mean()) < 1.0
    small = torch.zeros((4000, 8), dtype=torch.int32, device=dev)
    for lo in (0, 6249000, n - 4000):
        sub_off = off[lo:lo + 4001].contiguous()
        ctx.extract_batch_dev(bases, sub_off, 4000, total, 12, small)
        rc, _, _ = ctx.extract_status()
        assert rc == 0
        assert bool((small == recs[lo:lo + 4000]).all())
        b0, b1 = int(sub_off[0]), int(sub_off[-1])
        hb = bases[b0:b1].cpu().numpy()
        ho = (sub_off.cpu().numpy() - b0).astype(np.uint64)
        want = orc.extract_batch(hb, ho[:1001], 12, threads=8)
        got = small[:1000].cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
        assert (got == want).all()
