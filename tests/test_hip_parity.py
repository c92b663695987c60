"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden fixtures.
Bit-exact: every quantity is an integer.  Needs a real MI355X: `pytest -m gpu`."""
import gzip
import json
import os

import numpy as np
import pytest

from badger_amd import _native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _native.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import pyoracle
    return pyoracle


def _row(rid, seq, rec, orc):
    s = orc.revcomp(seq) if rec["flags"] & 1 else seq
    if rec["valid"]:
        bc, umi, score = s[rec["bc_start"]:rec["bc_start"] + 16], s[rec["umi_start"]:rec["umi_end"]], 0
    else:
        bc, umi, score = "*", "*", -1
    strand = {1: "+", -1: "-", 0: "."}[int(rec["strand"])]
    return "%s\t%s\t%s\t%d\t%s\t%s\t%d\t%d" % (rid, bc, umi, score, False, strand, rec["polyT"], rec["r1_end"])


def _diff(got, want):
    bad = np.nonzero(got != want)[0]
    return "first mismatch at %d: got %s want %s (%d total)" % (bad[0], got[bad[0]], want[bad[0]], len(bad)) if len(bad) else ""


# --------------------------------------------------------------------------- extraction
def test_extract_golden_rows(ctx, orc, golden_dir):
    ext = json.load(open(os.path.join(golden_dir, "extract_rows.json")))
    seqs = [r["seq"] for r in ext["reads"]]
    bases, off = synth.list_to_reads(seqs)
    for umi_len, key, skey in ((12, "row_v3", "r1_score_v3"), (10, "row_v2", "r1_score_v2")):
        recs = ctx.extract_batch(bases, off, umi_len)
        for r, rec in zip(ext["reads"], recs):
            assert _row(r["id"], r["seq"], rec, orc) == r[key], r["id"]
            assert int(rec["r1_score"]) == r[skey], r["id"]
        want = orc.extract_batch(bases, off, umi_len, threads=4)
        assert (recs == want).all(), _diff(recs, want)


def test_extract_no_polya_rule(ctx, orc, golden_dir):
    """bdg_extract_set_strand_rule(NO_POLYA) against the rows the reference's find_barcode_umi_no_polya produced
    (barcode_callers.py:231-248), against the oracle on a synthetic batch, and through the host mirror's method."""
    from badger_amd import _native
    from badger_amd.barcode_extraction.barcode_callers import TenXBarcodeExtractorV3
    ext = json.load(open(os.path.join(golden_dir, "extract_rows.json")))
    alt = json.load(open(os.path.join(golden_dir, "extract_rows_no_polya.json")))
    seqs = [r["seq"] for r in ext["reads"]]
    bases, off = synth.list_to_reads(seqs)
    ctx.extract_set_strand_rule(_native.STRAND_RULE_NO_POLYA)
    try:
        for umi_len, key, skey in ((12, "row_v3", "r1_score_v3"), (10, "row_v2", "r1_score_v2")):
            recs = ctx.extract_batch(bases, off, umi_len)
            for r, a, rec in zip(ext["reads"], alt["reads"], recs):
                assert _row(r["id"], r["seq"], rec, orc) == a[key], r["id"]
                assert int(rec["r1_score"]) == a[skey], r["id"]
        wl = synth.make_whitelist(1000)
        b, o = synth.make_reads(4000, wl, seed=11, p_sub=0.08, p_ins=0.04, p_del=0.04)
        b, o = b.numpy(), o.numpy().astype(np.uint64)
        got = ctx.extract_batch(b, o, 12)
        want = orc.extract_batch(b, o, 12, threads=8, rule=orc.RULE_NO_POLYA)
        assert (got == want).all(), _diff(got, want)
        assert (got != orc.extract_batch(b, o, 12, threads=8)).any()          # the rules do differ on this batch
    finally:
        ctx.extract_set_strand_rule(_native.STRAND_RULE_DEFAULT)
    det = TenXBarcodeExtractorV3()
    chunk = [(r["id"], r["seq"]) for r in ext["reads"][:60]]
    for res, a in zip(det.find_barcode_umi_no_polya_batch(chunk), alt["reads"]):
        assert str(res) == a["row_v3"]
    assert str(det.find_barcode_umi_no_polya(*chunk[3])) == alt["reads"][3]["row_v3"]
    assert str(det.find_barcode_umi(*chunk[3])) == ext["reads"][3]["row_v3"]      # the shared context went back to the default rule


def test_extract_config1_tsv(ctx, orc, golden_dir):
    seqs, ids = [], []
    with gzip.open(os.path.join(golden_dir, "c1_reads.fa.gz"), "rt") as f:
        for line in f:
            (ids if line.startswith(">") else seqs).append(line[1:].strip() if line.startswith(">") else line.strip())
    bases, off = synth.list_to_reads(seqs)
    recs = ctx.extract_batch(bases, off, 12)
    want = open(os.path.join(golden_dir, "c1_expected.tsv")).read().split("\n")
    rows = [_row(i, s, r, orc) for i, s, r in zip(ids, seqs, recs)]
    assert rows == want[1:1 + len(rows)]


@pytest.mark.parametrize("n,seed,errs", [(5000, 3, (0.03, 0.02, 0.03)), (3000, 4, (0.0, 0.0, 0.0)), (3000, 5, (0.10, 0.05, 0.05))])
def test_extract_vs_oracle_synthetic(ctx, orc, n, seed, errs):
    wl = synth.make_whitelist(1000)
    bases, off = synth.make_reads(n, wl, seed=seed, p_sub=errs[0], p_ins=errs[1], p_del=errs[2])
    b, o = bases.numpy(), off.numpy().astype(np.uint64)
    got = ctx.extract_batch(b, o, 12)
    want = orc.extract_batch(b, o, 12, threads=8)
    assert (got == want).all(), _diff(got, want)


def test_extract_adversarial(ctx, orc):
    rng = np.random.default_rng(99)
    R1 = "CTACACGACGCTCTTCCGATCT"
    rnd = lambda k: "".join("ACGT"[i] for i in rng.integers(0, 4, k))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))
    seqs = ["", "A", "ACGTA", "T" * 16, "T" * 17, "A" * 16, "A" * 17, "T" * 5000, "A" * 5000, R1, rc(R1), R1 * 50, rc(R1 * 50),
            "N" * 100, rnd(1007), rnd(1008), rnd(1009), rnd(2016), rnd(2017), rnd(8000), rnd(70000)]
    for k in range(200):    # reads with R1 / polyT planted at every alignment relative to the 16-byte vectors
        pre = rnd(int(rng.integers(0, 70)))
        mid = R1 if rng.random() < 0.8 else R1[:int(rng.integers(8, 22))]
        tail = "T" * int(rng.integers(10, 40)) if rng.random() < 0.8 else rnd(5)
        s = pre + mid + rnd(int(rng.integers(0, 45))) + tail + rnd(int(rng.integers(0, 1200)))
        if rng.random() < 0.2:
            p = int(rng.integers(0, len(s)))
            s = s[:p] + "N" + s[p + 1:]
        seqs.append(s if rng.random() < 0.5 else rc(s))
    for k in range(100):    # short reads around every length threshold
        seqs.append(rnd(int(rng.integers(0, 80))))
    bases, off = synth.list_to_reads(seqs)
    for umi_len in (10, 12):
        got = ctx.extract_batch(bases, off, umi_len)
        want = orc.extract_batch(bases, off, umi_len, threads=8)
        assert (got == want).all(), _diff(got, want)
    # a sub-range of the buffer (off[0] != 0) gives the same records
    got2 = ctx.extract_batch(bases, off[20:], 12)
    assert (got2 == orc.extract_batch(bases, off, 12, threads=8)[20:]).all()


def test_extract_reads_built_around_sw_ties(ctx, orc):
    """Reads whose adapter region is made of the tie cases of tests/golden/ssw_tie_kats.json (tandem tails, repeated
    heads, two full copies, N inside, truncated copies): the end / begin cell the packed 16-bit Smith-Waterman of
    k_sw_clusters reports decides R1_end and with it the barcode slice; every record must equal the oracle's."""
    import json
    rng = np.random.default_rng(321)
    R1 = "CTACACGACGCTCTTCCGATCT"
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))
    rnd = lambda k: "".join("ACGT"[i] for i in rng.integers(0, 4, k))
    kats = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ssw_tie_kats.json")))["kats"]
    pieces = [k["window"] for k in kats if "N" not in k["window"] or len(k["window"]) > 4] + \
             [R1 + "CCGATCT", R1 + "TCT", R1 + "CT" * 3, "CT" * 4 + R1, R1[:-1], R1[:-2] + "CTCT", R1[3:], R1[:11] + R1[9:], R1[:15] + "N" + R1[16:],
              R1 + R1[-9:], R1[:8] + R1, "TCTT" + R1[2:], R1[:-4] + "ATCTATCT"]
    seqs = []
    for k in range(3000):
        mid = pieces[int(rng.integers(0, len(pieces)))]
        if rng.random() < 0.3:                          # one sequencing error inside
            p = int(rng.integers(0, len(mid)))
            mid = mid[:p] + ("" if rng.random() < 0.4 else "ACGT"[int(rng.integers(0, 4))]) + mid[p + (0 if rng.random() < 0.3 else 1):]
        tail = "T" * int(rng.integers(12, 34)) if rng.random() < 0.85 else rnd(8)
        s = rnd(int(rng.integers(0, 48))) + mid + rnd(16) + rnd(12) + tail + rnd(int(rng.integers(0, 300)))
        seqs.append(s if rng.random() < 0.5 else rc(s))
    bases, off = synth.list_to_reads(seqs)
    got = ctx.extract_batch(bases, off, 12)
    want = orc.extract_batch(bases, off, 12, threads=8)
    assert (got == want).all(), _diff(got, want)
    assert 0.5 < want["valid"].mean() < 1.0


def test_synthetic_reads_are_the_same_on_cpu_and_gpu():
    """bench.py builds its workload on the GPU; the same call on the CPU must give the same bytes (the workload can be
    regenerated without a GPU; tests/test_host_mirror.py pins the CPU bytes)."""
    import torch
    wl = synth.make_whitelist(5000)
    for seed in (1, 7):
        bc, oc, tc = synth.make_reads(30000, wl, seed=seed, with_truth=True)
        bg, og, tg = synth.make_reads(30000, wl, seed=seed, device="cuda", with_truth=True)
        assert bool((oc == og.cpu()).all()) and bool((bc == bg.cpu()).all())
        assert bool((tc["barcode"] == tg["barcode"].cpu()).all()) and bool((tc["revcomp"] == tg["revcomp"].cpu()).all())


def test_overlap_mode_gives_the_same_calls(orc):
    """bdg_set_overlap: the whitelist match of batch i runs on the context's auxiliary stream beside the extraction of
    batch i + 1 (two record buffers).  Five different batches through the pipelined loop must give, batch by batch, the
    records and calls of the plain sequential loop."""
    import torch
    dev = torch.device("cuda", 0)
    wl = synth.make_whitelist(20000)
    batches = []
    for k in range(5):
        bases, off = synth.make_reads(30000 + 1000 * k, wl, seed=50 + k, device=dev)
        batches.append((torch.cat([bases, torch.zeros(64, dtype=torch.uint8, device=dev)]), off.contiguous(), len(off) - 1, int(off[-1])))
    nmax = max(b[2] for b in batches)
    results = {}
    for overlap in (False, True):
        c = _native.Context(0)
        c.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        c.whitelist_load(wl)
        c.set_overlap(overlap)
        nb = 2 if overlap else 1
        recs = [torch.zeros((nmax, 8), dtype=torch.int32, device=dev) for _ in range(nb)]
        bi = [torch.zeros(nmax, dtype=torch.int32, device=dev) for _ in range(nb)]
        be = [torch.zeros(nmax, dtype=torch.uint8, device=dev) for _ in range(nb)]
        bt = [torch.zeros(nmax, dtype=torch.int16, device=dev) for _ in range(nb)]
        out = []
        for k, (bases, off, n, total) in enumerate(batches):
            b = k % nb
            c.extract_batch_dev(bases, off, n, total, 12, recs[b])
            c.nearest16_recs_dev(recs[b], n, 2, bi[b], be[b], bt[b])
            c.synchronize()                      # results of a match are complete after bdg_synchronize (both streams)
            out.append((recs[b][:n].cpu().numpy().copy(), bi[b][:n].cpu().numpy().copy(), be[b][:n].cpu().numpy().copy(), bt[b][:n].cpu().numpy().copy()))
        results[overlap] = out
        assert c.extract_status()[0] == 0
        c.close()
    for a, b in zip(results[False], results[True]):
        assert all((x == y).all() for x, y in zip(a, b))
    # and without a synchronisation between the batches (the way bench.py runs it): the last two batches' buffers at the end
    c = _native.Context(0)
    c.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    c.whitelist_load(wl)
    c.set_overlap(True)
    recs = [torch.zeros((nmax, 8), dtype=torch.int32, device=dev) for _ in range(2)]
    bi = [torch.zeros(nmax, dtype=torch.int32, device=dev) for _ in range(2)]
    be = [torch.zeros(nmax, dtype=torch.uint8, device=dev) for _ in range(2)]
    bt = [torch.zeros(nmax, dtype=torch.int16, device=dev) for _ in range(2)]
    for rep in range(3):
        for k, (bases, off, n, total) in enumerate(batches):
            c.extract_batch_dev(bases, off, n, total, 12, recs[k % 2])
            c.nearest16_recs_dev(recs[k % 2], n, 2, bi[k % 2], be[k % 2], bt[k % 2])
    c.synchronize()
    for k in (3, 4):
        n = batches[k][2]
        want = results[False][k]
        assert (recs[k % 2][:n].cpu().numpy() == want[0]).all() and (bi[k % 2][:n].cpu().numpy() == want[1]).all()
        assert (be[k % 2][:n].cpu().numpy() == want[2]).all() and (bt[k % 2][:n].cpu().numpy() == want[3]).all()
    c.close()
    # A match is only noted when it is asked for (it is queued behind the NEXT extraction's scan).  Whatever ends the wait
    # must give the same answers: a synchronisation with no extraction behind the match, a second match on the other buffer
    # (queues the first), and a whitelist change (the waiting match meant the OLD list and must run against it).
    c = _native.Context(0)
    c.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    c.whitelist_load(wl)
    c.set_overlap(True)
    (b0, o0, n0, t0), (b1, o1, n1, t1) = batches[0], batches[1]
    c.extract_batch_dev(b0, o0, n0, t0, 12, recs[0])
    c.nearest16_recs_dev(recs[0], n0, 2, bi[0], be[0], bt[0])
    c.synchronize()                              # nothing behind the match
    assert (bi[0][:n0].cpu().numpy() == results[False][0][1]).all() and (bt[0][:n0].cpu().numpy() == results[False][0][3]).all()
    c.extract_batch_dev(b1, o1, n1, t1, 12, recs[1])
    c.nearest16_recs_dev(recs[1], n1, 2, bi[1], be[1], bt[1])
    bi[0].zero_(); be[0].zero_(); bt[0].zero_()
    c.nearest16_recs_dev(recs[0], n0, 2, bi[0], be[0], bt[0])      # a second match: the first is queued now, this one waits
    other = synth.make_whitelist(3000, seed=77)
    c.whitelist_load(other)                      # ... until the list changes: it meant `wl`
    c.synchronize()
    for k in (0, 1):
        n = batches[k][2]
        assert (bi[k][:n].cpu().numpy() == results[False][k][1]).all() and (be[k][:n].cpu().numpy() == results[False][k][2]).all()
        assert (bt[k][:n].cpu().numpy() == results[False][k][3]).all()
    c.nearest16_recs_dev(recs[0], n0, 2, bi[0], be[0], bt[0])      # and the new list answers afterwards
    c.synchronize()
    got = bi[0][:n0].cpu().numpy().copy()
    c.set_overlap(False)
    c.nearest16_recs_dev(recs[0], n0, 2, bi[1], be[1], bt[1])
    c.synchronize()
    assert (got == bi[1][:n0].cpu().numpy()).all() and not (got == results[False][0][1]).all()
    c.close()


def test_extract_queue_overflow_is_contained_and_recovered(orc):
    """Adapter-dense reads (concatemers) against a deliberately tiny candidate queue.  The host-buffer call loops until
    the workspace fits and returns the oracle's records.  The device-resident call cannot loop by itself: an overflowing
    batch must leave placeholder records only ({valid 0, BDG_FLAG_INCOMPLETE}), so that bdg_nearest16_recs_dev /
    bdg_distinct_dev running behind it on the stream use nothing, bdg_extract_status says BDG_E_CAPACITY, and the
    caller's rerun loop ends with the oracle's records."""
    import torch
    rng = np.random.default_rng(123)
    R1 = "CTACACGACGCTCTTCCGATCT"
    rnd = lambda k: "".join("ACGT"[i] for i in rng.integers(0, 4, k))
    seqs = []
    for k in range(600):
        unit = rnd(int(rng.integers(0, 12))) + R1 + rnd(16) + rnd(12) + "T" * int(rng.integers(14, 32)) + rnd(int(rng.integers(0, 60)))
        seqs.append(unit * int(rng.integers(1, 12)))
    bases, off = synth.list_to_reads(seqs)
    want = orc.extract_batch(bases, off, 12, threads=8)
    assert want["valid"].mean() > 0.9
    # host-buffer call
    c = _native.Context(0)
    c.extract_set_queue_capacity(16)
    got = c.extract_batch(bases, off, 12)
    assert (got == want).all(), _diff(got, want)
    c.close()
    # device-resident call
    c = _native.Context(0)
    dev = torch.device("cuda", 0)
    c.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    wl = np.unique(want["bc_rank"][(want["flags"] & 2) != 0])
    c.whitelist_load(wl)
    n, total = len(seqs), int(off[-1])
    d_bases = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_recs = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    bi = torch.zeros(n, dtype=torch.int32, device=dev)
    be = torch.zeros(n, dtype=torch.uint8, device=dev)
    bt = torch.zeros(n, dtype=torch.int16, device=dev)
    uq = torch.zeros(n, dtype=torch.int32, device=dev)
    uc = torch.zeros(n, dtype=torch.int32, device=dev)
    uf = torch.zeros(n, dtype=torch.int32, device=dev)
    dn = torch.zeros(2, dtype=torch.int32, device=dev)
    c.extract_set_queue_capacity(16)
    passes = 0
    while True:
        c.extract_batch_dev(d_bases, d_off, n, total, 12, d_recs)
        c.nearest16_recs_dev(d_recs, n, 2, bi, be, bt)                 # enqueued before anybody looked at the status
        c.distinct_dev(d_recs, n, uq, uc, uf, dn)
        rc, bad, _ = c.extract_status()
        passes += 1
        recs = d_recs.cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
        if rc != _native.E_CAPACITY:
            break
        assert passes < 8
        assert (recs["valid"] == 0).all() and (recs["flags"] == _native.FLAG_INCOMPLETE).all()
        assert (be.cpu().numpy() == 255).all() and (bi.cpu().numpy().view(np.uint32) == 0xFFFFFFFF).all()
        assert int(dn[0]) == 0
    assert rc == 0 and passes >= 2
    assert (recs == want).all(), _diff(recs, want)
    ok = (want["flags"] & 2) != 0
    assert (be.cpu().numpy()[ok] == 0).all() and int(dn[0]) == len(wl)
    c.close()


def test_extract_empty_and_errors(ctx):
    assert len(ctx.extract_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64))) == 0
    bases, off = synth.list_to_reads(["ACGT" * 30, "ACGTacgt" * 10, "ACGT" * 5])
    with pytest.raises(_native.BadgerHipError) as e:
        ctx.extract_batch(bases, off, 12)
    assert e.value.code == _native.E_BADBASE and "read 1" in str(e.value)
    with pytest.raises(_native.BadgerHipError):
        ctx.extract_batch(bases, np.array([0, 10, 5, 20], np.uint64), 12)


def test_extract_partition_invariance(ctx, orc):
    """Sharding reads across GPUs = running contiguous sub-batches: records must not depend on the split."""
    wl = synth.make_whitelist(1000)
    bases, off = synth.make_reads(4000, wl, seed=21)
    b, o = bases.numpy(), off.numpy().astype(np.uint64)
    whole = ctx.extract_batch(b, o, 12)
    parts = [ctx.extract_batch(b, o[s:e + 1], 12) for s, e in ((0, 1000), (1000, 1001), (1001, 2500), (2500, 4000))]
    assert (np.concatenate(parts) == whole).all()


# --------------------------------------------------------------------------- nearest16
def _near_queries(wl, rng, n):
    q = []
    for _ in range(n):
        w = int(wl[int(rng.integers(0, len(wl)))])
        u = rng.random()
        if u < 0.2:
            pass
        elif u < 0.5:
            w ^= int(rng.integers(1, 4)) << (2 * int(rng.integers(0, 16)))
        elif u < 0.7:
            w ^= int(rng.integers(1, 4)) << (2 * int(rng.integers(0, 16)))
            w ^= int(rng.integers(1, 4)) << (2 * int(rng.integers(0, 16)))
        elif u < 0.9:      # delete one base, insert one elsewhere
            s = list(synth.rank_to_str(w))
            del s[int(rng.integers(0, 16))]
            s.insert(int(rng.integers(0, 16)), "ACGT"[int(rng.integers(0, 4))])
            w = synth.str_to_rank("".join(s))
        else:
            w = int(rng.integers(0, 1 << 32))
        q.append(w)
    return np.array(q, dtype=np.uint32)


@pytest.mark.parametrize("algo", [1, 2])
@pytest.mark.parametrize("max_ed", [0, 1, 2])
def test_nearest16_vs_oracle(ctx, orc, algo, max_ed):
    rng = np.random.default_rng(17)
    wl = synth.make_whitelist(3000)
    # a dense cluster so that ties and multi-hit neighbourhoods occur
    base = int(wl[0])
    cluster = {base ^ (x << (2 * p)) for p in range(16) for x in (1, 2, 3)} | {base ^ (1 << 2) ^ (2 << 10), base ^ (3 << 6) ^ (1 << 20)}
    wl = np.unique(np.concatenate([wl, np.array(sorted(cluster), dtype=np.uint32)]))
    rng.shuffle(wl)                      # caller order != rank order: indices must refer to the caller's order
    q = np.concatenate([_near_queries(wl, rng, 1500), np.array([base, base ^ 3, base ^ (1 << 30)], dtype=np.uint32)])
    ctx.nearest16_set_algo(algo)
    gi, ge, gt = ctx.nearest16(q, wl, max_ed)
    wi, we, wt = orc.nearest16(q, wl, max_ed, threads=8)
    assert (ge == we).all(), _diff(ge, we)
    assert (gi == wi).all(), _diff(gi, wi)
    assert (gt == wt).all(), _diff(gt, wt)
    ctx.nearest16_set_algo(0)


def test_nearest16_scan_large_max_ed(ctx, orc):
    rng = np.random.default_rng(5)
    wl = synth.make_whitelist(5000)
    q = rng.integers(0, 1 << 32, 300, dtype=np.uint64).astype(np.uint32)
    gi, ge, gt = ctx.nearest16(q, wl, 16)
    wi, we, wt = orc.nearest16(q, wl, 16, threads=8)
    assert (ge == we).all() and (gi == wi).all() and (gt == wt).all()


def test_nearest16_picks_the_algorithm_by_size(orc):
    """automatic mode: a small job (stage 2's --high_sens pass: some thousand centres) takes the exhaustive scan and builds no
    index at all; a whitelist-sized list takes the probe path, whose deletion-variant part only appears with max_ed = 2.
    Results equal the oracle's either way."""
    c = _native.Context(0)
    rng = np.random.default_rng(23)
    wl = synth.make_whitelist(5000)
    q = _near_queries(wl, rng, 4000)
    gi, ge, gt = c.nearest16(q, wl, 2)
    assert c.nearest16_index_bytes() == 0
    wi, we, wt = orc.nearest16(q, wl, 2, threads=8)
    assert (gi == wi).all() and (ge == we).all() and (gt == wt).all()
    wl = synth.make_whitelist(737280)
    q = _near_queries(wl, rng, 6000)
    gi, ge, gt = c.nearest16(q, wl, 1)
    pairs_only = c.nearest16_index_bytes()
    assert 0 < pairs_only < 150 << 20
    wi, we, wt = orc.nearest16(q, wl, 1, threads=8, probe=True)
    assert (gi == wi).all() and (ge == we).all() and (gt == wt).all()
    gi, ge, gt = c.nearest16(q, wl, 2)
    assert c.nearest16_index_bytes() > pairs_only
    wi, we, wt = orc.nearest16(q, wl, 2, threads=8, probe=True)
    assert (gi == wi).all() and (ge == we).all() and (gt == wt).all()
    c.nearest16(q[:10], wl, 2)                               # the index exists: a small job uses it too
    c.close()


def test_nearest16_edge_cases(ctx):
    wl = synth.make_whitelist(10)
    gi, ge, gt = ctx.nearest16(np.zeros(0, np.uint32), wl, 2)
    assert len(gi) == 0
    gi, ge, gt = ctx.nearest16(wl[:3], np.zeros(0, np.uint32), 2)
    assert (gi == 0xFFFFFFFF).all() and (ge == 255).all() and (gt == 0).all()
    with pytest.raises(_native.BadgerHipError):
        ctx.nearest16(wl[:3], np.array([5, 5, 7], np.uint32), 2)      # whitelist must be distinct


# --------------------------------------------------------------------------- graph
def _ranks_of(barcodes):
    out = []
    for s in barcodes:
        if len(s) == 17:
            s = s[:-1]
        if len(s) == 16:
            out.append(synth.str_to_rank(s))
    return np.unique(np.array(out, dtype=np.uint32))


@pytest.mark.parametrize("algo", [1, 2, 3, 4, 5, 6])
def test_graph_golden(ctx, golden_dir, algo):
    g = json.load(open(os.path.join(golden_dir, "graph.json")))
    for key in ("c1_thr1", "c1_thr2", "cells60_thr1", "cells60_thr2"):
        thr = int(key[-1])
        if algo in (2, 6) and thr != 1:
            continue
        ctx.graph_set_algo(algo)
        case = g[key]
        e = ctx.graph_edges(_ranks_of(case["barcodes"]), thr, case["qgram_T"])
        got = [[int(x["a"]), int(x["b"]), int(x["dist"])] for x in e]
        assert got == case["edges"], key
    ctx.graph_set_algo(0)


def _observed_barcodes(n_cells, n_obs, seed):
    rng = np.random.default_rng(seed)
    cells = rng.integers(0, 1 << 32, n_cells, dtype=np.uint64)
    out = np.empty(n_obs, dtype=np.uint64)
    for k in range(n_obs):
        s = list(synth.rank_to_str(int(cells[int(rng.integers(0, n_cells))])))
        i = 0
        while i < len(s):
            u = rng.random()
            if u < 0.03: s[i] = "ACGT"[int(rng.integers(0, 4))]
            elif u < 0.05: del s[i]; continue
            elif u < 0.07: s.insert(i, "ACGT"[int(rng.integers(0, 4))]); i += 1
            i += 1
        s = ("".join(s) + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 4)))[:16]
        out[k] = synth.str_to_rank(s)
    return np.unique(out.astype(np.uint32))


@pytest.mark.parametrize("algo,thr", [(1, 1), (2, 1), (1, 2), (1, 3), (3, 1), (3, 2), (3, 3), (4, 1), (4, 2), (0, 2), (0, 3), (5, 1), (5, 2), (6, 1)])
def test_graph_vs_oracle(ctx, orc, algo, thr):
    ranks = _observed_barcodes(300, 12000, 31)
    ctx.graph_set_algo(algo)
    T = orc.qgram_threshold(thr)
    e = ctx.graph_edges(ranks, thr, T)
    w = orc.graph_edges(ranks, thr, T, threads=8)
    assert len(e) == len(w) and len(e) > 100
    assert (e == w).all()
    ctx.graph_set_algo(0)


@pytest.mark.parametrize("algo,thr", [(2, 1), (1, 1), (1, 2), (3, 2), (4, 2), (5, 2)])
def test_graph_row_blocks_partition_the_edges(ctx, orc, algo, thr):
    """a GPU owns a block of rows of the sorted rank array and emits the edges whose smaller rank lies in
    it; the blocks of any partition give disjoint lists whose union is the full list (blocks cut inside 256-row tiles,
    empty blocks and one-row blocks included)."""
    import torch
    from badger_amd import dist as bdist
    ranks = _observed_barcodes(300, 9000, 33)
    n = len(ranks)
    T = orc.qgram_threshold(thr)
    want = orc.graph_edges(ranks, thr, T, threads=8)
    want = want[np.lexsort((want["b"], want["a"]))]
    d_ranks = torch.from_numpy(ranks.view(np.int32)).cuda()
    cap = 4 * n + 1024
    d_out = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.graph_set_algo(algo)
    cuts = [(0, 0), (0, 1), (1, 300), (300, 1000), (1000, 1000), (1000, n)]
    for blocks in (cuts, bdist.graph_row_blocks(n, 3, "pairs"), bdist.graph_row_blocks(n, 8, "rows")):
        got = []
        for lo, hi in blocks:
            ctx.graph_edges_rows_dev(d_ranks, n, lo, hi, thr, T, d_out, cap, d_cnt)
            ctx.synchronize()
            k = int(d_cnt[0])
            assert k <= cap
            e = d_out[:k].cpu().numpy().view(np.uint32)
            if k:
                assert (e[:, 0] >= ranks[lo]).all() and (e[:, 0] <= ranks[hi - 1]).all() and (e[:, 0] < e[:, 1]).all()
            got.append(e)
        e = np.concatenate(got)
        e = e[np.lexsort((e[:, 1], e[:, 0]))]
        assert len(e) == len(want) and len(e) > 100
        assert (e[:, 0] == want["a"]).all() and (e[:, 1] == want["b"]).all() and (e[:, 2] == want["dist"]).all()
    ctx.graph_set_algo(0)


@pytest.mark.parametrize("algo,thr", [(0, 1), (0, 2), (0, 3), (3, 2), (5, 2), (5, 1), (1, 2), (6, 1)])
def test_graph_parts_partition_the_edges(ctx, orc, algo, thr):
    """bdg_graph_edges_part_dev: the nparts shares of any cut are disjoint and their union is the oracle's list, whichever
    path serves the threshold (row blocks for the probes / the q-gram join / the sweep, shares of the 14-mer groups for the
    deletion-variant join); a part index outside the cut is an error."""
    import torch
    ranks = _observed_barcodes(300, 9000, 35)
    n = len(ranks)
    T = orc.qgram_threshold(thr)
    want = orc.graph_edges(ranks, thr, T, threads=8)
    want = want[np.lexsort((want["b"], want["a"]))]
    d_ranks = torch.from_numpy(ranks.view(np.int32)).cuda()
    cap = 4 * n + 1024
    d_out = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.graph_set_algo(algo)
    for nparts in (1, 3, 8):
        got = []
        for part in range(nparts):
            ctx.graph_edges_part_dev(d_ranks, n, part, nparts, thr, T, d_out, cap, d_cnt)
            ctx.synchronize()
            k = int(d_cnt[0])
            assert k <= cap
            got.append(d_out[:k].cpu().numpy().view(np.uint32).copy())
        assert nparts == 1 or min(len(g) for g in got) > 0
        e = np.concatenate(got)
        e = e[np.lexsort((e[:, 1], e[:, 0]))]
        assert len(e) == len(want) and len(e) > 100
        assert (e[:, 0] == want["a"]).all() and (e[:, 1] == want["b"]).all() and (e[:, 2] == want["dist"]).all()
    with pytest.raises(_native.BadgerHipError):
        ctx.graph_edges_part_dev(d_ranks, n, 3, 3, thr, T, d_out, cap, d_cnt)
    ctx.graph_set_algo(0)


def test_graph_deletion_variant_join_in_rounds(ctx, orc, monkeypatch):
    """a large input is taken in several rounds over shares of the 14-mer groups (the cut that gives GPUs their parts):
    forced here on small inputs - 1, 3 and 7 rounds, alone and inside 2 parts - the list stays the oracle's"""
    import torch
    for ranks in (_observed_barcodes(300, 9000, 37), _low_complexity_barcodes(4000, 44)):
        n = len(ranks)
        T = orc.qgram_threshold(2)
        want = orc.graph_edges(ranks, 2, T, threads=8)
        d_ranks = torch.from_numpy(ranks.view(np.int32)).cuda()
        cap = len(want) + 1024
        d_out = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
        d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        ctx.graph_set_algo(5)
        for rounds in ("1", "3", "7"):
            monkeypatch.setenv("BADGER_AMD_D2_ROUNDS", rounds)
            for nparts in (1, 2):
                got = []
                for part in range(nparts):
                    ctx.graph_edges_part_dev(d_ranks, n, part, nparts, 2, T, d_out, cap, d_cnt)
                    ctx.synchronize()
                    got.append(d_out[:int(d_cnt[0])].cpu().numpy().view(np.uint32).copy())
                e = np.concatenate(got)
                e = e[np.lexsort((e[:, 1], e[:, 0]))]
                assert len(e) == len(want) and (e[:, 0] == want["a"]).all() and (e[:, 1] == want["b"]).all() and (e[:, 2] == want["dist"]).all(), (rounds, nparts)
        monkeypatch.delenv("BADGER_AMD_D2_ROUNDS")
        ctx.graph_set_algo(0)


def _low_complexity_barcodes(n, seed):
    """16-mers made of short repeats and homopolymer runs with a few edits: six-mers repeat inside a barcode
    (S counts products of multiplicities, index.py:80-93), buckets are very uneven, many pairs have S >= T."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        unit = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(1, 5))))
        s = list((unit * 16)[:16])
        for _ in range(int(rng.integers(0, 4))):
            s[int(rng.integers(0, 16))] = "ACGT"[int(rng.integers(0, 4))]
        if rng.random() < 0.3:
            k = int(rng.integers(0, 16))
            s = (s[:k] + s[k + 1:] + ["ACGT"[int(rng.integers(0, 4))]])
        out.append(synth.str_to_rank("".join(s)))
    return np.unique(np.array(out, dtype=np.uint32))


@pytest.mark.parametrize("thr", [1, 2, 3])
def test_graph_qjoin_low_complexity(ctx, orc, thr):
    """q-gram join (and its closed-form fallback) against the all-pairs sweep and the oracle where six-mers repeat."""
    ranks = _low_complexity_barcodes(6000, 40 + thr)
    T = orc.qgram_threshold(thr)
    w = orc.graph_edges(ranks, thr, T, threads=8)
    assert len(w) > 1000
    for algo in (1, 3, 4) + ((5,) if thr <= 2 else ()) + ((6,) if thr == 1 else ()):      # (5 / 6: the deletion-variant joins; repeats make a row's 14-mers collide)
        ctx.graph_set_algo(algo)
        e = ctx.graph_edges(ranks, thr, T)
        assert len(e) == len(w) and (e == w).all(), algo
    if thr == 3:
        ctx.graph_set_algo(5)
        with pytest.raises(_native.BadgerHipError):
            ctx.graph_edges(ranks, thr, T)                     # complete for thr <= 2 only
    if thr >= 2:
        ctx.graph_set_algo(6)
        with pytest.raises(_native.BadgerHipError):
            ctx.graph_edges(ranks, thr, T)                     # one deletion: thr <= 1 only
    ctx.graph_set_algo(0)


def test_graph_qjoin_dense_slices(ctx, orc):
    """Few distinct cells, many near-identical variants: the bucket tails of a row exceed the join's table
    several times over, so slices are cut into counted hash parts (P > 1)."""
    rng = np.random.default_rng(77)
    cell = int(rng.integers(0, 1 << 32))
    obs = np.full(40000, cell, dtype=np.uint64)
    for _ in range(4):                       # up to four substitutions, all in bases 6..15: bases 0..5 (one six-mer) never change
        obs = obs ^ (rng.integers(0, 4, len(obs)).astype(np.uint64) << (2 * rng.integers(6, 16, len(obs)).astype(np.uint64)))
    ranks = np.unique(obs.astype(np.uint32))
    assert len(ranks) > 6000                 # the shared bucket's tail alone exceeds the 4096-entry pass for the early rows
    T = orc.qgram_threshold(2)
    w = orc.graph_edges(ranks, 2, T, threads=8)
    for algo in (3, 1, 5):
        ctx.graph_set_algo(algo)
        e = ctx.graph_edges(ranks, 2, T)
        assert len(e) == len(w) and (e == w).all(), algo
    ctx.graph_set_algo(0)


def test_graph_edge_cases(ctx):
    assert len(ctx.graph_edges(np.zeros(0, np.uint32), 1, 5)) == 0
    assert len(ctx.graph_edges(np.array([7], np.uint32), 1, 5)) == 0
    e = ctx.graph_edges(np.array([0, 1], np.uint32), 1, 5)
    assert len(e) == 1 and tuple(e[0]) == (0, 1, 1)
    with pytest.raises(_native.BadgerHipError):
        ctx.graph_edges(np.array([3, 3], np.uint32), 1, 5)


# --------------------------------------------------------------------------- fuzz
def _fragment_reads(rng, n):
    """Reads glued from adapter pieces, T/A runs, N and junk: many tiny reads, every branch boundary."""
    R1 = "CTACACGACGCTCTTCCGATCT"
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))
    rnd = lambda k: "".join("ACGT"[i] for i in rng.integers(0, 4, k))

    def mutate(s, p):
        out = []
        for ch in s:
            u = rng.random()
            if u < p: out.append("ACGT"[int(rng.integers(0, 4))])
            elif u < 1.5 * p: continue
            elif u < 2 * p: out.append(ch + "ACGT"[int(rng.integers(0, 4))])
            else: out.append(ch)
        return "".join(out)

    seqs = []
    for _ in range(n):
        parts = []
        for _ in range(int(rng.integers(0, 7))):
            kind = int(rng.integers(0, 8))
            if kind == 0: parts.append(rnd(int(rng.integers(0, 60))))
            elif kind == 1: parts.append(mutate(R1, float(rng.choice([0.0, 0.05, 0.15]))))
            elif kind == 2: parts.append(R1[int(rng.integers(0, 10)):int(rng.integers(12, 23))])
            elif kind == 3: parts.append(mutate("T" * int(rng.integers(3, 45)), float(rng.choice([0.0, 0.1, 0.3]))))
            elif kind == 4: parts.append(mutate("A" * int(rng.integers(3, 45)), float(rng.choice([0.0, 0.1, 0.3]))))
            elif kind == 5: parts.append("N" * int(rng.integers(1, 4)))
            elif kind == 6: parts.append(rc(mutate(R1, 0.05)) + rnd(int(rng.integers(0, 30))))
            else: parts.append(rnd(16) + rnd(12) + "T" * 30)
        s = "".join(parts)
        seqs.append(s if rng.random() < 0.5 else rc(s))
    return seqs


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_extract_fuzz_fragments(ctx, orc, seed):
    rng = np.random.default_rng(1000 + seed)
    seqs = _fragment_reads(rng, 12000)
    bases, off = synth.list_to_reads(seqs)
    umi = 12 if seed != 2 else 10
    got = ctx.extract_batch(bases, off, umi)
    want = orc.extract_batch(bases, off, umi, threads=8)
    assert (got == want).all(), _diff(got, want)
    assert 0.02 < got["valid"].mean() < 0.9          # both outcomes are well represented


def test_extract_long_reads_and_r1_repeats(ctx, orc):
    """Reads spanning many steps of the scan kernel, adapter concatemers (every position a hit) and
    homopolymers (every window T- or A-rich): exercises queue overflow paths and the candidate compaction."""
    rng = np.random.default_rng(77)
    R1 = "CTACACGACGCTCTTCCGATCT"
    rnd = lambda k: "".join("ACGT"[i] for i in rng.integers(0, 4, k))
    seqs = [rnd(100000), R1 * 3000, "T" * 50000, "A" * 50000, "AT" * 20000, "TTTA" * 10000,
            rnd(30000) + R1 + rnd(16) + rnd(12) + "T" * 30 + rnd(30000), (R1 + rnd(20)) * 500]
    seqs += [rnd(int(rng.integers(1, 40))) for _ in range(64)]
    bases, off = synth.list_to_reads(seqs)
    got = ctx.extract_batch(bases, off, 12)
    want = orc.extract_batch(bases, off, 12, threads=8)
    assert (got == want).all(), _diff(got, want)


def test_extract_task_and_step_boundaries(ctx, orc):
    """The scan kernel takes reads in tasks of 16 and a task's vectors 63 per step, with the loads of the next two steps in
    flight across step and task ends: tasks without vectors (runs of empty reads), tasks of exactly one or two steps in a row,
    vector totals at every residue around a multiple of 63, batch sizes around a multiple of 16, reads that end exactly on a
    16-byte boundary, N and adapter copies in the first / last vector of a task."""
    rng = np.random.default_rng(4242)
    R1 = "CTACACGACGCTCTTCCGATCT"
    rnd = lambda k: "".join("ACGT"[i] for i in rng.integers(0, 4, k))
    cdna = lambda k: rnd(int(rng.integers(0, 30))) + R1 + rnd(28) + "T" * 30 + rnd(k)

    def check(seqs, lead=0):
        bases, off = synth.list_to_reads(seqs)
        if lead:                                   # the first read does not start on a 16-byte boundary
            bases = np.concatenate([np.frombuffer(rnd(lead).encode(), dtype=np.uint8), bases])
            off = off + np.uint64(lead)
        got = ctx.extract_batch(bases, off, 12)
        want = orc.extract_batch(bases, off, 12, threads=8)
        assert (got == want).all(), _diff(got, want)

    for n in (1, 2, 15, 16, 17, 31, 32, 33, 48):                         # batch sizes around the task size
        check([cdna(int(rng.integers(0, 900))) for _ in range(n)], lead=int(rng.integers(0, 16)))
    # whole tasks without a single vector between ordinary ones
    seqs = [cdna(500) for _ in range(16)] + [""] * 16 + [cdna(300) for _ in range(5)] + [""] * 40 + [cdna(700) for _ in range(20)] + [""] * 16
    check(seqs)
    check([""] * 100)
    # tasks of one step (16 reads in <= 63 vectors) in a row, then long ones, then one-step tasks again
    short = lambda: rnd(int(rng.integers(20, 60)))
    check([short() for _ in range(64)] + [cdna(4000) for _ in range(16)] + [short() for _ in range(48)] + [R1 + rnd(16) + rnd(12) + "T" * 30])
    # a task's vector total at every residue around multiples of 63: 15 fixed reads + one that sets the total
    for total in (62, 63, 64, 65, 125, 126, 127, 189, 190):
        fixed = [rnd(16 * 3) for _ in range(15)]                      # 45 vectors (reads start on 16-byte boundaries here)
        last = 16 * (total - 45)
        check(fixed + [cdna(0)[:last] if last <= 110 else cdna(last - 110)[:last]] + [cdna(200) for _ in range(16)])
    # reads of every length around the 16-byte grid, with N at the ends and adapter copies cut by task boundaries
    seqs = []
    for L in list(range(0, 70)) + [95, 96, 97, 1007, 1008, 1009]:
        s = cdna(L)[:max(L, 0)] if L < 120 else cdna(L - 110)
        seqs.append(s)
        if len(s) > 2:
            seqs.append("N" + s[1:-1] + "N")
    rng.shuffle(seqs)
    check(seqs, lead=7)
    check([R1 * 3 + "T" * 40] * 40 + [rnd(1008)] * 8 + [(rnd(10) + R1)[-16 * k:] for k in range(1, 3)] * 8)


@pytest.mark.parametrize("seed", [11, 12])
def test_nearest16_and_graph_fuzz_clustered(ctx, orc, seed):
    """Dense barcode neighbourhoods: many ties, multi-hit deletion variants, homopolymer-rich strings."""
    rng = np.random.default_rng(seed)
    seeds = [0, 0xFFFFFFFF, 0x55555555, 0x0F0F0F0F] + [int(x) for x in rng.integers(0, 1 << 32, 20, dtype=np.uint64)]
    wl = set()
    for s0 in seeds:
        for _ in range(150):
            s = list(synth.rank_to_str(s0))
            for _ in range(int(rng.integers(0, 4))):
                u = rng.random()
                p = int(rng.integers(0, len(s)))
                if u < 0.5: s[p] = "ACGT"[int(rng.integers(0, 4))]
                elif u < 0.75 and len(s) > 1: del s[p]
                else: s.insert(p, "ACGT"[int(rng.integers(0, 4))])
            s = ("".join(s) + "ACGTACGT")[:16]
            wl.add(synth.str_to_rank(s))
    wl = np.array(sorted(wl), dtype=np.uint32)
    rng.shuffle(wl)
    q = np.concatenate([_near_queries(wl, rng, 3000), wl[:200]])
    for algo in (1, 2):
        ctx.nearest16_set_algo(algo)
        gi, ge, gt = ctx.nearest16(q, wl, 2)
        wi, we, wt = orc.nearest16(q, wl, 2, threads=8)
        assert (ge == we).all() and (gi == wi).all() and (gt == wt).all(), (algo, _diff(gt, wt))
    ctx.nearest16_set_algo(0)
    ranks = np.unique(np.concatenate([wl, q]))
    for algo, thr in ((1, 1), (2, 1), (1, 2)):
        ctx.graph_set_algo(algo)
        e = ctx.graph_edges(ranks, thr, orc.qgram_threshold(thr))
        w = orc.graph_edges(ranks, thr, threads=8)
        assert len(e) == len(w) and (e == w).all(), (algo, thr)
    ctx.graph_set_algo(0)


def _records_of(ranks, usable=None):
    recs = np.zeros(len(ranks), dtype=_native.REC_DTYPE)
    recs["bc_rank"] = ranks
    ok = np.ones(len(ranks), bool) if usable is None else usable
    recs["valid"] = ok
    recs["flags"] = np.where(ok, _native.FLAG_RANK_OK | _native.FLAG_BC16, 0)
    return recs


@pytest.mark.parametrize("case", ["one_rank", "two_ranks_low_bit", "dense_range", "hot_among_random", "two_levels", "narrow_prefix"])
def test_distinct_dev_on_adversarial_counts(ctx, case):
    """bdg_distinct_dev where the bucket a block groups in LDS overflows: one barcode seen 300,000 times, two that differ in
    their last bit, 40,000 consecutive ranks, hot barcodes among random ones, an input large enough for the second bucket
    level, and ranks that share their top 20 bits - always numpy's unique / first index / counts
    (index_bc_single_thread, barcode_graph.py:192-204)."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(hash(case) % 1000)
    if case == "one_rank":
        ranks = np.full(300000, 0x9ABCDEF1, dtype=np.uint32)
    elif case == "two_ranks_low_bit":
        ranks = (np.uint32(0x12345678) ^ rng.integers(0, 2, 70000).astype(np.uint32))
    elif case == "dense_range":
        ranks = (np.uint32(0x40000000) + rng.permutation(40000).astype(np.uint32))
        ranks = np.concatenate([ranks, ranks[:5000]])
    elif case == "hot_among_random":
        hot = rng.integers(0, 1 << 32, 5, dtype=np.uint64).astype(np.uint32)
        ranks = np.concatenate([rng.integers(0, 1 << 32, 200000, dtype=np.uint64).astype(np.uint32), np.repeat(hot, 9000)])
        rng.shuffle(ranks)
    elif case == "two_levels":
        ranks = rng.integers(0, 1 << 32, 2500000, dtype=np.uint64).astype(np.uint32)
        ranks[::7] = ranks[3::7][:len(ranks[::7])]
    else:
        ranks = (np.uint32(0xABCDE000) | rng.integers(0, 1 << 12, 30000).astype(np.uint32))
    usable = rng.random(len(ranks)) < 0.97
    recs = _records_of(ranks, usable)
    n = len(recs)
    d_recs = torch.from_numpy(recs.view(np.int32).reshape(-1, 8).copy()).to(dev)
    uniq, cnt, first = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3))
    dn = torch.zeros(2, dtype=torch.int32, device=dev)
    ctx.set_stream(0)
    ctx.distinct_dev(d_recs, n, uniq, cnt, first, dn)
    torch.cuda.synchronize()
    wu, wf, wc = np.unique(ranks[usable], return_index=True, return_counts=True)
    nu = int(dn[0])
    assert nu == len(wu)
    assert (uniq[:nu].cpu().numpy().view(np.uint32) == wu).all()
    assert (cnt[:nu].cpu().numpy() == wc).all()
    assert (first[:nu].cpu().numpy() == np.nonzero(usable)[0][wf]).all()
    assert int(dn[1]) == 0


def test_distinct_dev_small_and_edge_cases(ctx, orc):
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    seqs = _fragment_reads(rng, 3000)
    bases, off = synth.list_to_reads(seqs)
    recs = ctx.extract_batch(bases, off, 12)
    n = len(recs)
    d_recs = torch.from_numpy(recs.view(np.int32).reshape(-1, 8).copy()).to(dev)
    uniq, cnt, first = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3))
    dn = torch.zeros(2, dtype=torch.int32, device=dev)
    ctx.set_stream(0)
    ctx.distinct_dev(d_recs, n, uniq, cnt, first, dn)
    torch.cuda.synchronize()
    ok = (recs["valid"] == 1) & ((recs["flags"] & _native.FLAG_RANK_OK) != 0)
    wu, wf, wc = np.unique(recs["bc_rank"][ok], return_index=True, return_counts=True)
    nu = int(dn[0])
    assert nu == len(wu) > 10
    assert (uniq[:nu].cpu().numpy().astype(np.uint32) == wu).all() and (cnt[:nu].cpu().numpy() == wc).all()
    assert (first[:nu].cpu().numpy() == np.nonzero(ok)[0][wf]).all()
    # barcodes of 16 bases holding an N are counted separately (the reference raises KeyError on them)
    n16 = ((recs["valid"] == 1) & ((recs["flags"] & _native.FLAG_BC16) != 0) & ((recs["flags"] & _native.FLAG_RANK_OK) == 0)).sum()
    assert int(dn[1]) == int(n16)
    ctx.distinct_dev(d_recs, 0, uniq, cnt, first, dn)
    torch.cuda.synchronize()
    assert int(dn[0]) == 0


def test_rows_of_dev_and_device_arrays(ctx):
    """bdg_rows_of_dev (ranks -> positions in the sorted distinct array, strided values, absent -> NONE) and the
    library's own device buffers (bdg_mem_alloc / bdg_mem_to_host), against numpy."""
    rng = np.random.default_rng(21)
    for n, m in ((0, 5), (1, 7), (1000, 0), (1000, 4097), (200000, 300001)):
        srt = np.unique(rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32))
        n_u = len(srt)
        vals = rng.integers(0, 1 << 32, (m, 3), dtype=np.uint64).astype(np.uint32)
        if n_u and m:
            take = rng.random(m) < 0.7
            vals[take, 0] = srt[rng.integers(0, n_u, int(take.sum()))]
            vals[:, 1] = srt[rng.integers(0, n_u, m)]
            vals[0, 0], vals[m - 1, 1] = srt[0], srt[-1]
        d_srt = _native.DeviceArray.from_host(ctx, srt)
        d_vals = _native.DeviceArray.from_host(ctx, vals)
        d_rows = _native.DeviceArray(ctx, (2, max(m, 1)), np.uint32)
        assert (d_rows.to_host() == 0).all()                                   # allocations arrive zero-filled
        ctx.rows_of_dev(d_srt, n_u, d_vals, m, 3, d_rows.data_ptr(), 0)
        ctx.rows_of_dev(d_srt, n_u, d_vals, m, 3, d_rows.data_ptr() + 4 * max(m, 1), 1)
        got = d_rows.to_host()
        for col in (0, 1):
            v = vals[:, col]
            pos = np.searchsorted(srt, v)
            hit = (pos < n_u) & (srt[np.minimum(pos, max(n_u - 1, 0))] == v) if n_u else np.zeros(m, bool)
            want = np.where(hit, pos, 0xFFFFFFFF).astype(np.uint32)
            assert (got[col, :m] == want).all(), (n, m, col)
        assert (d_vals.to_host(min(m, 5)) == vals[:5]).all()
        for d in (d_srt, d_vals, d_rows):
            d.free()


@pytest.mark.parametrize("thr", [1, 2])
def test_deletion_variant_join_with_oversize_buckets(ctx, orc, monkeypatch, thr):
    """The joins' cold paths, forced: without the second bucket level (BADGER_AMD_DJ_L2MAX=0) a fine bucket is a whole coarse
    one - thousands of entries - so every bucket is beyond a wave's 256 entries and goes to the block kernel through the
    overflow list, and the larger ones beyond the block's 2048 are taken in shares of the low key bits.  Same edge lists as
    the oracle's, alone and in parts."""
    import torch
    for ranks in (_observed_barcodes(300, 16000, 41), _low_complexity_barcodes(6000, 46)):
        n = len(ranks)
        T = orc.qgram_threshold(thr)
        want = orc.graph_edges(ranks, thr, T, threads=8)
        d_ranks = torch.from_numpy(ranks.view(np.int32)).cuda()
        cap = len(want) + 1024
        d_out = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
        d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        ctx.graph_set_algo(5 if thr == 2 else 6)
        monkeypatch.setenv("BADGER_AMD_DJ_L2MAX", "0")
        for nparts in (1, 3):
            got = []
            for part in range(nparts):
                ctx.graph_edges_part_dev(d_ranks, n, part, nparts, thr, T, d_out, cap, d_cnt)
                ctx.synchronize()
                ctx.graph_status()
                got.append(d_out[:int(d_cnt[0])].cpu().numpy().view(np.uint32).copy())
            e = np.concatenate(got)
            e = e[np.lexsort((e[:, 1], e[:, 0]))]
            assert len(e) == len(want) and (e[:, 0] == want["a"]).all() and (e[:, 1] == want["b"]).all() and (e[:, 2] == want["dist"]).all(), nparts
        monkeypatch.delenv("BADGER_AMD_DJ_L2MAX")
        ctx.graph_set_algo(0)
