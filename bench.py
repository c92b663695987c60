#!/usr/bin/env python3
"""Headline benchmark: barcode calls/s against the 737K-entry whitelist (BASELINE.json).

One step = one pass of the hot path over one device-resident batch of synthetic ONT reads:
  K1 extract (k_scan_reads, k_sw_clusters, k_strict_filter, k_sw_singles, k_finalize_reads)  -> 32-byte record per read
  K2 nearest16 (probe path, max_ed 2) of every extracted barcode against the whitelist
A "call" is one read taken through both.  Reads are sharded per GPU (weak scaling, no
collective on the data path); the process group is only used for the barrier and the
max-over-ranks clock.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--whitelist W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 from a plain `python bench.py` starts torch.distributed.run itself, as a child process and
before this process has touched a GPU, and exits with the child's code.

`--config 3` / `--config 5` time the other GPU configurations of BASELINE.json instead (K3 graph edges over 500K
distinct barcodes at threshold 1 / 2) and print a line of the same shape with their own unit and roofline block.

Before the W warm-up steps the step runs untimed for a quarter of a second (`config.clock_ramp`; `--no-ramp` skips it): a
step lasts a millisecond, and a handful of warm-up steps is over before the GPU has left its idle clocks.

After the timed region the run is checked: the library's status word (queue overflow, bad base) and a sample of the
final records and calls against the CPU oracle; a mismatch exits non-zero and prints no line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from badger_amd import _native, synth  # noqa: E402
from badger_amd import dist as bdist  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


def cores_visible():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def host_cores():
    """CPUs this job may actually use: the cgroup's quota (a GPU box shows 256 CPUs and gives a job 16), else the affinity mask.
    The CPU legs run this many OpenMP threads and report this number as `cores`."""
    vis = cores_visible()
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1") and int(period) > 0:
                return max(1, min(vis, int(round(int(quota) / int(period)))))
        except Exception:
            continue
    return vis


def cpu_baseline(bases_dev, off_dev, wl, device_out=None):
    """The CPU oracle (a port of the reference algorithm) timed on this box's host cores, on a bounded sample of the
    same workload.  Only the checker is timed here; nothing the product path produces depends on it.
    Legs, each on all cores and on one core:
      extract  orc_extract_batch: the reference's per-read path (barcode_callers.py:165-229), OpenMP over reads like the
               reference's ProcessPoolExecutor over chunks
      nearest  the whitelist match.  `probe`: neighbourhood enumeration against a hash set, the algorithm class the GPU
               path uses (a tuned CPU implementation).  `exhaustive`: Levenshtein against every entry, which is what the
               reference's postprocessing loop does against its ~5K centres (barcode_graph.py:376-384) - against 737K
               entries it is a lower bound nobody would run, reported for completeness only.
    value = calls/s of extract + probe on all cores.
    device_out = (records, best_idx, best_ed, n_ties) of the last step: what the oracle computes for the timing is compared
    with it record by record (the sample the CPU leg runs is the largest piece of oracle output this run has)."""
    from oracle import pyoracle as orc
    cores = host_cores()
    n_all, n_one = 100000, 3000
    n_all = min(n_all, off_dev.numel() - 1)
    end = int(off_dev[n_all])
    b = bases_dev[:end].cpu().numpy()
    o = off_dev[:n_all + 1].cpu().numpy().astype(np.uint64)
    orc.extract_batch(b[:int(o[64])], o[:65], 12, threads=cores)          # warm
    t0 = time.perf_counter()
    recs = orc.extract_batch(b, o, 12, threads=cores)
    ext_all = n_all / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    orc.extract_batch(b[:int(o[n_one])], o[:n_one + 1], 12, threads=1)
    ext_one = n_one / (time.perf_counter() - t0)
    q = recs["bc_rank"][(recs["flags"] & 2) != 0]
    orc.nearest16(q[:64], wl, 2, threads=cores, probe=True)                # warm (builds nothing persistent, pages the code in)
    t0 = time.perf_counter()
    wi, we, wt = orc.nearest16(q, wl, 2, threads=cores, probe=True)
    near_all = len(q) / (time.perf_counter() - t0)
    checked = None
    if device_out is not None:
        d_recs, d_idx, d_ed, d_ties = device_out
        ok = (recs["flags"] & 2) != 0
        got = d_recs[:n_all].cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
        gi = d_idx[:n_all].cpu().numpy().view(np.uint32)[ok]
        ge = d_ed[:n_all].cpu().numpy()[ok]
        gt = d_ties[:n_all].cpu().numpy().view(np.uint16)[ok]
        if not ((got == recs).all() and (gi == wi).all() and (ge == we).all() and (gt == wt).all()):
            raise SystemExit("parity failed: the first %d records / calls differ from the oracle's" % n_all)
        checked = "the device's first %d records and %d calls equal the oracle's" % (n_all, len(q))
    q1 = q[:20000]
    t0 = time.perf_counter()
    orc.nearest16(q1, wl, 2, threads=1, probe=True)
    near_one = len(q1) / (time.perf_counter() - t0)
    qe = q[:48]
    t0 = time.perf_counter()
    orc.nearest16(qe, wl, 2, threads=cores)
    near_exh = len(qe) / (time.perf_counter() - t0)
    return {"value": 1.0 / (1.0 / ext_all + 1.0 / near_all), "unit": "calls/s", "cores": cores, "cores_usable": cores,
            "cores_visible": cores_visible(), "kind": "port",
            "sample": "%d reads through the oracle's extract_batch + their %d barcodes through the oracle's neighbourhood-probe "
                      "nearest16 against the %d-entry whitelist, OpenMP over the %d CPUs of the job's quota; 1-core legs on %d reads / %d barcodes; "
                      "exhaustive-scan leg on %d barcodes" % (n_all, len(q), len(wl), cores, n_one, len(q1), len(qe)),
            "extract_reads_per_s": ext_all, "extract_reads_per_s_1core": ext_one,
            "nearest_probe_calls_per_s": near_all, "nearest_probe_calls_per_s_1core": near_one,
            "nearest_exhaustive_calls_per_s": near_exh,
            "value_1core": 1.0 / (1.0 / ext_one + 1.0 / near_one), "checked": checked}


def kernels_hash(lib_version):
    """the device-side source hash inside bdg_version() ("... kernels <hash> host <hash>")"""
    parts = lib_version.split()
    return parts[parts.index("kernels") + 1] if "kernels" in parts[:-1] else lib_version


def profile_counters(kernel):
    """HBM bytes / vector instructions per launch of `kernel` from the committed rocprofv3 PMC passes
    (tools/pmc_profile.sh -> tools/summarize_profile.py), only if they were taken from this very build of the
    library; otherwise None (a stale profile must not be printed beside live timings)."""
    out = {"traffic": None, "valu": None, "lds": None, "source": None}
    try:
        lib_version = _native.load().bdg_version().decode()
    except Exception:
        return out
    kernels = kernels_hash(lib_version)
    for name, key in (("traffic.json", "traffic"), ("valu.json", "valu"), ("lds.json", "lds")):
        f = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(f):
            continue
        try:
            d = json.load(open(f))
        except Exception:
            continue
        meta = d.get("_meta", {})
        if meta.get("kernels") != kernels:              # (per-launch counters belong to the device code: the host side may differ)
            out["source"] = "profiles/%s is from kernels %r, these are %r: not reported" % (name, meta.get("kernels"), kernels)
            continue
        out[key] = d.get(kernel)
        out["source"] = "profiles/{traffic,valu,lds}.json@%s (rocprofv3 --pmc, %s)" % (meta.get("tag"), lib_version)
    return out


def parity_sample(ctx, bases, off_u, n, wl, recs, best_idx, best_ed, n_ties):
    """A sample of the final step's output against the CPU oracle: 3 x 1000 records (start, middle, end of the
    batch), the calls of those reads against the oracle's probe form, 24 of them against its exhaustive scan."""
    from oracle import pyoracle as orc
    cores = min(host_cores(), 32)
    off = off_u.cpu().numpy().astype(np.uint64)
    m = min(1000, n)
    for lo in sorted({0, max(0, n // 2 - m // 2), n - m}):
        hi = lo + m
        b = bases[int(off[lo]):int(off[hi])].cpu().numpy()
        want = orc.extract_batch(b, off[lo:hi + 1] - off[lo], 12, threads=cores)
        got = recs[lo:hi].cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
        if not (got == want).all():
            return "records %d..%d differ from the oracle" % (lo, hi)
        ok = (want["flags"] & _native.FLAG_RANK_OK) != 0
        gi = best_idx[lo:hi].cpu().numpy().view(np.uint32)
        ge = best_ed[lo:hi].cpu().numpy()
        gt = n_ties[lo:hi].cpu().numpy().view(np.uint16)
        wi, we, wt = orc.nearest16(want["bc_rank"][ok], wl, 2, threads=cores, probe=True)
        if not ((gi[ok] == wi).all() and (ge[ok] == we).all() and (gt[ok] == wt).all()):
            return "calls of reads %d..%d differ from the oracle" % (lo, hi)
        if not ((gi[~ok] == 0xFFFFFFFF).all() and (ge[~ok] == 255).all() and (gt[~ok] == 0).all()):
            return "calls of unusable barcodes in %d..%d are not 'none'" % (lo, hi)
        sel = np.nonzero(ok)[0][:8]
        xi, xe, xt = orc.nearest16(want["bc_rank"][sel], wl, 2, threads=cores)
        if not ((gi[sel] == xi).all() and (ge[sel] == xe).all() and (gt[sel] == xt).all()):
            return "calls of reads %d.. differ from the oracle's exhaustive scan" % lo
    return "ok"


def spawn_ranks(ngpus):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU through torch.distributed.run as a CHILD
    process (this process has not touched the GPU and never will) and hand back its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ngpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def observed_barcodes(n_distinct, wl, seed=3, n_cells=5000):
    """n distinct 16-mers as extraction would see them: cell barcodes with substitutions and a
    deletion, numpy.default_rng(seed)."""
    rng = np.random.default_rng(seed)
    cells = wl[rng.permutation(len(wl))[:n_cells]].astype(np.uint64)
    out = np.zeros(0, dtype=np.uint32)
    while len(out) < n_distinct:
        m = 2 * n_distinct
        r = cells[rng.integers(0, len(cells), m)]
        for _ in range(2):
            hit = rng.random(m) < 0.35
            r = np.where(hit, r ^ (rng.integers(1, 4, m).astype(np.uint64) << (2 * rng.integers(0, 16, m).astype(np.uint64))), r)
        hit = rng.random(m) < 0.25
        pos = rng.integers(0, 16, m).astype(np.uint64)
        low = (np.uint64(1) << (2 * pos)) - np.uint64(1)
        d = (r & low) | ((r >> np.uint64(2)) & ~low & np.uint64(0xFFFFFFFF)) | (rng.integers(0, 4, m).astype(np.uint64) << np.uint64(30))
        r = np.where(hit, d, r)
        out = np.unique(np.concatenate([out, (r & np.uint64(0xFFFFFFFF)).astype(np.uint32)]))
    rng.shuffle(out)
    return np.sort(out[:n_distinct])


RAMP_S = 0.25


def clock_ramp(step, dev, seconds=RAMP_S):
    """Untimed steps for a fixed wall time before the W warm-up steps.  A step of this workload takes about a millisecond,
    so W = 2..5 warm-up steps are over before the GPU has left its idle clocks (measured: the first timed steps of a cold
    run are 3-5 % slower than the steady state a batch pipeline runs in).  Returns the number of steps it ran."""
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            step()
        torch.cuda.synchronize(dev)
        k += 8
    return k


GRAPH_PATHS = {"k_graph_probe": ("neighbourhood probes", "blocks of equal rows"),
               "k_d1_pairs": ("one-deletion join", "shares of the 15-mer groups"),
               "k_d1_emit": ("one-deletion join", "shares of the 15-mer groups"),
               "k_graph_qjoin_w": ("q-gram join", "row blocks of equal pair counts"),
               "k_graph_qjoin": ("q-gram join, closed form", "row blocks of equal pair counts"),
               "k_graph_scan": ("all-pairs sweep", "row blocks of equal pair counts")}


def measure_graph(args, rank, world, dev, ctx, thr, n, wl, steps, warmup, ramp, cpu_check):
    """One graph configuration of BASELINE.json (config 3: thr 1, config 5: thr 2): K3 over n distinct observed barcodes,
    every rank its share of the edge list (bdg_graph_edges_part_dev, no collective).  Returns the block of figures rank 0
    prints: time per pass, rows/s, edges, every kernel of the pass (whatever its name: nothing on this path is a library
    call any more), the roofline block of the one that takes longest, the whole pass against the HBM roofline, and - at one
    GPU - the CPU oracle on the same rows, whose whole edge list is the checker."""
    from oracle import pyoracle as orc
    from badger_amd.barcode_graph import qgram_threshold
    ranks = observed_barcodes(n, wl)
    T = qgram_threshold(thr, 16)                       # (the product path's own: index.py:22-24)
    d_ranks = torch.from_numpy(ranks.view(np.int32)).to(dev)
    cap = 32 * n
    d_edges = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
    d_n = torch.zeros(1, dtype=torch.int64, device=dev)
    ctx.graph_set_algo(args.graph_algo)

    def step():
        ctx.graph_edges_part_dev(d_ranks, n, rank, world, thr, T, d_edges, cap, d_n)

    ramp_steps = clock_ramp(step, dev) if ramp else 0
    for _ in range(max(1, warmup)):
        step()
    ctx.profile(True)
    ctx.profile_reset()
    elapsed = bdist.timed(step, steps, dev)
    prof = ctx.profile_read()
    ctx.profile(False)
    ctx.graph_status()                                 # (what the asynchronous calls could not return)
    ne = int(d_n[0])
    if ne > cap:
        raise SystemExit("edge capacity too small: %d > %d" % (ne, cap))
    # check (every rank, its own share): no edge twice, and 3,000 sampled edges are edges by the oracle's S and dmin with the
    # distance they carry; the whole list is compared with the oracle's below (one GPU) or by its length (several)
    e = d_edges[:ne].cpu().numpy().view(np.uint32)
    rng = np.random.default_rng(7 + rank)
    status = "ok"
    key = e[:, 0].astype(np.uint64) << np.uint64(32) | e[:, 1].astype(np.uint64)
    if len(np.unique(key)) != ne or not (e[:, 0] < e[:, 1]).all():
        status = "an edge is listed twice or not as (a < b)"
    for a, b, d in (e[rng.integers(0, ne, 3000)] if ne else []):
        if orc.qgram_S(int(a), int(b)) < T or orc.dmin3(int(a), int(b)) != int(d) or int(d) > thr:
            status = "edge (%d, %d, %d) is not an edge by the oracle" % (int(a), int(b), int(d))
    ne_all = int(bdist.all_sum(ne, dev))
    fails = bdist.all_max(0.0 if status == "ok" else 1.0, dev)
    if fails:
        if status != "ok":
            sys.stderr.write("rank %d: %s\n" % (rank, status))
        raise SystemExit("parity sample failed (graph, threshold %d)" % thr)
    if rank != 0:
        return None
    per_launch_ms = {k: v[1] / max(1, v[0]) for k, v in prof.items() if v[0]}
    dom = max(per_launch_ms, key=per_launch_ms.get)    # the kernel that takes longest, whatever it is called
    path, share = GRAPH_PATHS.get(dom, ("deletion-variant join", "shares of the 14-mer groups"))
    ms_per_pass = elapsed / steps * 1e3
    alg = 4 * n + 9 * ne                               # 4n in + 9E out (rank's own edges)
    achieved = alg / (per_launch_ms[dom] * 1e-3) / 1e9
    pc = profile_counters(dom)
    # HBM is not what binds these kernels (L2-resident gathers, LDS counters, integer issue): say how close they are to the
    # ceilings that could - vector-instruction issue (1024 SIMDs, one wave64 instruction per 4 clocks) and the LDS arrays
    # (one cycle per lane group and CU, MI355X_MICROARCH.md "LDS") - from the committed counter passes of this build
    t_dom = per_launch_ms[dom] * 1e-3
    issue_peak, lds_peak = 1024 * 2.4e9 / 4.0, 256 * 2.4e9
    int_issue = None if pc["valu"] is None else {"valu_insts_per_launch": pc["valu"], "achieved": pc["valu"] / t_dom, "peak": issue_peak,
                                                 "unit": "wave-instr/s", "frac": pc["valu"] / t_dom / issue_peak}
    lds_use = None if pc["lds"] is None else {"lds_cycles_per_launch": pc["lds"]["cycles"], "bank_conflict_cycles": pc["lds"]["conflict_cycles"],
                                              "achieved": pc["lds"]["cycles"] / t_dom, "peak": lds_peak, "unit": "LDS-array cycles/s (256 CUs)",
                                              "frac": pc["lds"]["cycles"] / t_dom / lds_peak}
    whole = alg / (ms_per_pass * 1e-3) / 1e9
    block = {
        "threshold": thr, "rows": n, "qgram_T": T, "path": path, "ms_per_pass": ms_per_pass, "rows_per_s": n / (elapsed / steps),
        "steps": steps, "warmup": warmup, "edges_rank0": ne, "edges_all_ranks": ne_all,
        "parallelism": "one share of the edge list per GPU (bdg_graph_edges_part_dev: %s), no collectives" % share,
        "clock_ramp": "%d untimed passes before the %d warm-up passes" % (ramp_steps, warmup),
        "kernels_ms_per_pass": {k: round(v, 4) for k, v in sorted(per_launch_ms.items())},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pc["traffic"], "traffic_source": pc["source"],
                     "algorithmic_bytes_per_launch": alg, "kernel_ms": per_launch_ms[dom], "int_issue": int_issue, "lds": lds_use},
        "whole_pass": {"algorithmic_bytes": alg, "achieved": whole, "unit": "GB/s", "frac_of_hbm_peak": whole / HBM_PEAK_GBS,
                       "traffic_all_kernels": profile_counters_sum(per_launch_ms)},
        "parity_sample": "ok",
    }
    if cpu_check and world == 1:
        # the whole job on the host cores: same rows, same index, and its edge list is the checker for the device's
        cores = host_cores()
        t0 = time.perf_counter()
        want, t_index, t_rows = orc.graph_edges_sampled(ranks, thr, 1, T, threads=cores, cap=ne + 1)
        t_all = time.perf_counter() - t0
        got = e[np.lexsort((e[:, 1], e[:, 0]))]
        same = len(want) == ne and np.array_equal(got[:, 0], want["a"]) and np.array_equal(got[:, 1], want["b"]) \
            and np.array_equal(got[:, 2], want["dist"])
        if not same:
            raise SystemExit("parity failed: the device's edge list differs from the oracle's (%d against %d edges)" % (ne, len(want)))
        block["parity_sample"] = "ok (all %d edges equal the oracle's)" % ne
        block["cpu_baseline"] = {"value": n / (t_index + t_rows), "unit": "rows/s", "cores": cores, "cores_usable": cores,
                                 "cores_visible": cores_visible(), "kind": "port",
                                 "sample": "oracle graph_edges (QGramIndex buckets + 3 Levenshtein per candidate, "
                                           "barcode_graph.py:207-249) on the same %d rows, OpenMP over %d CPUs: index %.2f s, "
                                           "rows %.2f s (sorting the edge list for the comparison, %.2f s more, is not counted)"
                                           % (n, cores, t_index, t_rows, t_all - t_index - t_rows)}
    elif cpu_check:
        # several GPUs: the shares must add up to the oracle's list (each rank checked its own for repeats and soundness)
        want, _, _ = orc.graph_edges_sampled(ranks, thr, 1, T, threads=host_cores(), cap=ne_all + 1)
        if len(want) != ne_all:
            raise SystemExit("parity failed: the ranks' shares hold %d edges, the oracle's list %d" % (ne_all, len(want)))
        block["parity_sample"] = "ok (the %d shares hold the oracle's %d edges; every rank: no repeats, 3000 sampled edges sound)" % (world, ne_all)
    return block


def profile_counters_sum(per_launch_ms):
    """HBM bytes of all the kernels of a pass together (committed counter passes of this very build), or None"""
    total = 0.0
    for k in per_launch_ms:
        t = profile_counters(k)["traffic"]
        if t is None:
            return None
        total += t if not isinstance(t, (list, tuple)) else t[-1]
    return total


def bench_graph(args, rank, world, dev, local_dev):
    """`--config 3` / `--config 5`: the graph configuration alone, as a line of the headline's shape."""
    thr = 1 if args.config == 3 else 2
    n = args.rows
    wl = synth.make_whitelist(args.whitelist)
    ctx = _native.Context(local_dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    blk = measure_graph(args, rank, world, dev, ctx, thr, n, wl, args.steps, args.warmup, not args.no_ramp, not args.no_cpu_baseline)
    if rank == 0:
        line = {
            "metric": "graph rows/sec, threshold=%d, %d distinct barcodes" % (thr, n), "value": blk["rows_per_s"],
            "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": blk["ms_per_pass"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BASELINE config %d: barcode_graph edges, threshold %d, %d distinct observed barcodes (%s)"
                                   % (args.config, thr, n, blk["path"]),
                       "rows": n, "edges_rank0": blk["edges_rank0"], "edges_all_ranks": blk["edges_all_ranks"], "qgram_T": blk["qgram_T"],
                       "parallelism": blk["parallelism"], "clock_ramp": blk["clock_ramp"]},
            "roofline": blk["roofline"], "whole_pass": blk["whole_pass"],
            "kernels_ms_per_step": blk["kernels_ms_per_pass"],
            "parity_sample": blk["parity_sample"],
        }
        if "cpu_baseline" in blk:
            line["cpu_baseline"] = blk["cpu_baseline"]
        print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-ramp", action="store_true", help="skip the fixed-time untimed steps that bring the GPU off its idle clocks before the warm-up")
    ap.add_argument("--reads", type=int, default=1000000, help="reads per GPU (weak scaling)")
    ap.add_argument("--whitelist", type=int, default=737280)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 5),
                    help="BASELINE.json config: 2 = the headline (K1 + K2), 3 = graph thr 1, 5 = graph thr 2")
    ap.add_argument("--rows", type=int, default=500000, help="distinct barcodes for --config 3 / 5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="headline only: skip the graph passes (configs 3 and 5) behind the config-2 measurement")
    ap.add_argument("--graph-algo", type=int, default=0, help="--config 3 / 5: bdg_graph_set_algo (0: the library's choice)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="no batch pipelining.  By default (bdg_set_overlap) the whitelist match of batch i runs on a second stream beside "
                         "the alignment kernels of batch i+1 - queued behind that batch's scan, so k_scan_reads (the roofline block) still "
                         "runs alone; the other kernels share the chip and their own durations in the per-kernel table grow")
    ap.add_argument("--overlap", action="store_true", help="(the default now; accepted for older command lines)")
    ap.add_argument("--rehearse", action="store_true",
                    help="the multi-rank plumbing without GPUs (self-launch, rendezvous, row / read split, barrier, max-over-ranks clock, "
                         "rank 0's line) with a stub in place of the step; prints a line marked \"rehearsal\": true - never a measurement")
    args = ap.parse_args()

    rank, local_rank, world = bdist.env_rank()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus))           # before anything here touches a GPU
        args.gpus = world
    if args.rehearse:
        return rehearse(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("BADGER_DIST_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 path with ranks sharing a GPU
    if local_rank >= ndev and backend == "nccl":
        raise SystemExit("rank %d has no GPU (%d visible)" % (local_rank, ndev))
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    bdist.init(backend=backend, device=dev)           # RCCL; only the barrier and the clock use it
    try:
        if args.config in (3, 5):
            bench_graph(args, rank, world, dev, local_dev)
        else:
            bench_calls(args, rank, world, dev, local_dev)
    finally:
        if world > 1 and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


def rehearse(args, rank, world):
    """Everything of a --gpus N run except the GPU: the ranks rendezvous over gloo, take their share of the work exactly as the
    real run cuts it (reads: N x --reads, weak; graph: part `rank` of `world` of the edge list, cut by the library), time
    a stub step between the barriers with the max-over-ranks clock, and rank 0 prints the line from the totals over ranks.
    The SCALE leg of the driver is then not the first time this code runs."""
    bdist.init(backend="gloo")
    try:
        if args.config in (3, 5):
            thr = 1 if args.config == 3 else 2
            lo, hi = bdist.partition(args.rows, world, rank)          # (for the totals: the shares themselves are cut by the library)
            units, unit, scaling = hi - lo, "rows/s", "strong"
            share = {"part": rank, "nparts": world, "threshold": thr,
                     "cut_by": "bdg_graph_edges_part_dev (thr 1: blocks of equal rows; thr 2: shares of the 14-mer groups)"}
        else:
            units, unit, scaling = args.reads, "calls/s", "weak"
            share = {"reads_per_gpu": args.reads, "seed_this_rank": 1 + rank}
        elapsed = bdist.timed(lambda: time.sleep(0.002 * (1 + rank)), args.steps, None)
        total = bdist.all_sum(units)
        if rank == 0:
            print(json.dumps({"rehearsal": True, "metric": "stub step, no GPU work", "value": total / (elapsed / args.steps), "unit": unit,
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                              "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "data": "none",
                              "config": dict(share, workload="rehearsal of BASELINE config %d" % args.config, units_all_ranks=int(total))}))
        bdist.barrier(None)
    finally:
        if world > 1 and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


def bench_calls(args, rank, world, dev, local_dev):
    # ---- inputs, resident in HBM before the clock starts
    wl = synth.make_whitelist(args.whitelist)
    bases, off = synth.make_reads(args.reads, wl, seed=1 + rank, device=dev)
    n = args.reads
    total_bytes = int(off[-1])
    pad = torch.zeros(64, dtype=torch.uint8, device=dev)
    bases = torch.cat([bases, pad])[:total_bytes + 64]
    off_u = off.to(torch.int64).contiguous()          # same bits as uint64
    # two sets of per-batch outputs: with overlap on, the whitelist match of step i (auxiliary stream) runs beside the
    # extraction of step i + 1, which therefore writes the other set
    nbuf = 1 if args.no_overlap else 2
    recs_b = [torch.zeros((n, 8), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    idx_b = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(nbuf)]
    ed_b = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    ties_b = [torch.zeros(n, dtype=torch.int16, device=dev) for _ in range(nbuf)]

    ctx = _native.Context(local_dev)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    ctx.whitelist_load(wl)
    ctx.set_overlap(not args.no_overlap)
    count = [0]

    def step():
        b = count[0] % nbuf
        count[0] += 1
        ctx.extract_batch_dev(bases, off_u, n, total_bytes, 12, recs_b[b])
        ctx.nearest16_recs_dev(recs_b[b], n, 2, idx_b[b], ed_b[b], ties_b[b])      # every record's barcode against the whitelist

    # Warm-up with every kernel timed: it says which kernel dominates.  In the timed region only that kernel carries events
    # (a pair of events per kernel costs the step ~6 % when all eight kernels have one); the per-kernel table of the line
    # comes from a few more steps behind the timed region, all kernels timed again.
    ramp_steps = clock_ramp(step, dev) if not args.no_ramp else 0
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(max(1, args.warmup)):
        step()
    rc, bad, nwin = ctx.extract_status()
    if rc == _native.E_CAPACITY:                       # window queue grown: one more warm-up pass
        ctx.profile_reset()
        step()
        rc, bad, nwin = ctx.extract_status()
    if rc != 0:
        raise SystemExit("extract failed: rc=%d bad_read=%d" % (rc, bad))
    stats = ctx.extract_counters()
    warm = ctx.profile_read()
    dom = max((k for k, v in warm.items() if v[0]), key=lambda k: warm[k][1] / warm[k][0])
    ctx.profile_only(dom)
    ctx.profile_reset()

    # barrier + sync, K steps, sync (device-wide: both streams) + barrier, max over ranks; with batch pipelining the match of
    # the last warm-up step is queued before the clock starts and the match of the last timed step before it stops
    elapsed = bdist.timed(step, args.steps, dev, settle=None if args.no_overlap else ctx.synchronize)
    prof = ctx.profile_read()                       # the dominant kernel, measured live over the timed region
    ctx.profile_only(None)
    ctx.profile_reset()
    table_steps = min(5, args.steps)
    for _ in range(table_steps):
        step()
    table = ctx.profile_read()
    ctx.profile(False)
    last = (count[0] - 1) % nbuf
    recs, best_idx, best_ed, n_ties = recs_b[last], idx_b[last], ed_b[last], ties_b[last]

    # ---- the timed steps must have been clean, and their output right (every rank checks its own)
    rc, bad, nwin = ctx.extract_status()
    status = "ok" if rc == 0 else "extract status after the timed region: rc=%d bad_read=%d" % (rc, bad)
    if status == "ok":
        status = parity_sample(ctx, bases, off_u, n, wl, recs, best_idx, best_ed, n_ties)
    fails = bdist.all_max(0.0 if status == "ok" else 1.0, dev)
    if status != "ok":
        sys.stderr.write("rank %d: %s\n" % (rank, status))
    if fails:
        raise SystemExit("parity sample failed")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n / (elapsed / args.steps)
        # roofline of the dominant kernel: algorithmic bytes of the unit it serves / its own launch time
        k1 = ("k_scan_reads", "k_sw_clusters", "k_sw_singles", "k_strict_filter", "k_finalize_reads")
        per_launch_ms = {k: v[1] / max(1, v[0]) for k, v in prof.items() if v[0]}
        table_ms = {k: v[1] / max(1, v[0]) for k, v in table.items() if v[0]}
        k1_bytes = total_bytes + 40 * n                # sum(L_i) + 8 (offset) + 32 (record) per read
        k2_bytes = 11 * n + 4 * len(wl)                # 4 (query) + 7 (idx, ed, ties) per call + whitelist once
        alg = k1_bytes if dom in k1 else k2_bytes
        achieved = alg / (per_launch_ms[dom] * 1e-3) / 1e9
        pc = profile_counters(dom)
        # the path is integer-issue bound, so say how close the kernel is to THAT ceiling:
        # 1024 SIMDs, one wave64 VALU instruction per 4 clocks each, at the 2.4 GHz peak clock
        issue_peak = 1024 * 2.4e9 / 4.0
        valu = pc["valu"]
        int_issue = None if valu is None else {"valu_insts_per_launch": valu, "achieved": valu / (per_launch_ms[dom] * 1e-3),
                                               "peak": issue_peak, "unit": "wave-instr/s",
                                               "frac": valu / (per_launch_ms[dom] * 1e-3) / issue_peak}
        line = {
            "metric": "barcode calls/sec vs 737K 10x whitelist", "value": value, "unit": "calls/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "%d synthetic ONT reads per GPU (mean %.0f bp) vs %d-entry whitelist: K1 extract + K2 nearest16(max_ed=2)"
                                   % (n, total_bytes / n, len(wl)),
                       "reads_per_gpu": n, "whitelist": len(wl), "sw_windows_per_step": int(nwin), "pipeline_counts": stats,
                       "parallelism": "reads sharded per GPU, no collectives",
                       "clock_ramp": "%d untimed steps (%.2f s) before the %d warm-up steps" % (ramp_steps, RAMP_S, args.warmup),
                       "batch_pipelining": "off" if args.no_overlap else "K2 of batch i on a second stream beside the alignment kernels of batch i+1, behind its scan (two record buffers)"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pc["traffic"], "traffic_source": pc["source"],
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": per_launch_ms[dom], "int_issue": int_issue},
            # the whole step against the same roofline: the algorithmic bytes of K1 and K2 together / the step's time per GPU
            "whole_step": {"algorithmic_bytes": k1_bytes + k2_bytes, "achieved": (k1_bytes + k2_bytes) / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                           "frac_of_hbm_peak": (k1_bytes + k2_bytes) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(table_ms.items())},
            "kernels_ms_per_step_source": "%d steps behind the timed region with every kernel timed; roofline.kernel_ms is the dominant "
                                          "kernel's mean over the timed region itself, where only it carries events%s" % (
                                              table_steps, "" if args.no_overlap else "; with batch pipelining k_nearest_* run BESIDE k_sw_* / "
                                              "k_strict_filter / k_finalize_reads of the next batch: those durations overlap and do not add up to the step "
                                              "(k_scan_reads runs alone)"),
            "parity_sample": "ok",
        }
        if not args.no_cpu_baseline and world == 1:       # the CPU leg runs at N = 1 only (the host threads would fight the other ranks)
            line["cpu_baseline"] = cpu_baseline(bases, off_u, wl, (recs, best_idx, best_ed, n_ties))
            line["parity_sample"] = "ok (3 x 1000 records and calls after the timed region; " + line["cpu_baseline"].pop("checked") + ")"
    # BASELINE configs 3 and 5 in the same line (one GPU): the graph passes at 500 K rows, thr 1 and thr 2, each with its
    # kernel table, roofline block, CPU oracle on the same rows and the whole edge list compared
    if world == 1 and not args.no_graph:
        for key, thr in (("graph_thr1", 1), ("graph_thr2", 2)):
            blk = measure_graph(args, rank, world, dev, ctx, thr, args.rows, wl, args.steps, args.warmup, not args.no_ramp, not args.no_cpu_baseline)
            line[key] = blk
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        bdist.barrier(dev)


if __name__ == "__main__":
    main()
