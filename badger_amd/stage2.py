"""Stage 2 on arrays: the same results as badger_amd.barcode_graph.BarcodeGraph (the dictionary mirror of the reference's
barcode_graph.py), computed with numpy over rank-packed barcodes instead of Python dictionaries of strings.

    counts            distinct barcodes, multiplicities, first-occurrence order   (index_bc_single_thread, :192-204)
    graph             edge list from the GPU (bdg_graph_edges / bdg_graph_edges_dev)                  (:207-249)
    cluster centres   get_cluster_centers                                                             (:252-277)
    cluster           two breadth-first levels, conflicts dropped                                      (:279-301)
    assignment        assign_by_cluster (+ postprocessing with --high_sens: bdg_nearest16)     (:322-329,:370-385)
    output            <out>_output_file.tsv                                                            (:388-410)

Why it can be done level by level without visiting order: a barcode reached on one level by two different centres
belongs to nobody whatever the order; so level 1 is "non-centres adjacent to exactly one centre" and
level 2 "unclustered barcodes adjacent to level-1 members of exactly one centre".  tests/test_stage2_arrays.py checks this
module against the dictionary mirror on random graphs, tests/test_cli_gpu.py against the reference's own output files.
"""
import logging
from statistics import mean

import numpy as np

from . import _native
from .barcode_graph import qgram_threshold
from .common import RANK, BarcodeRanks, rank, rank_many, rank_valid_many

logger = logging.getLogger("BarcodeGraph")
NONE = np.uint32(0xFFFFFFFF)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def unrank_many(ranks, length=16):
    """uint32 ranks -> list[str] (common.py:27-38, vectorised)"""
    r = np.asarray(ranks, dtype=np.uint64)
    if len(r) == 0:
        return []
    codes = ((r[:, None] >> (2 * np.arange(length, dtype=np.uint64))) & np.uint64(3)).astype(np.intp)
    raw = _ACGT[codes].tobytes().decode("ascii")
    return [raw[i * length:(i + 1) * length] for i in range(len(r))]


def observed_from_strings(barcodes_per_read, bc_len=16):
    """Per read: the observed barcode as the reference sees it (badger.py:104-111: '*' = none; a bc_len + 1 string loses
    its last base) -> (rank uint32[n], usable bool[n]).  usable = a bc_len-base barcode; a usable barcode with a base
    outside ACGT raises KeyError like the reference's rank()."""
    n = len(barcodes_per_read)
    usable = np.zeros(n, dtype=bool)
    kept, where = [], []
    for i, s in enumerate(barcodes_per_read):
        if s == "*":
            continue
        if len(s) == bc_len + 1:
            s = s[:-1]
        if len(s) == bc_len:
            kept.append(s)
            where.append(i)
    ranks = np.zeros(n, dtype=np.uint32)
    if kept:
        idx = np.array(where, dtype=np.intp)
        ranks[idx] = rank_many(kept, bc_len).astype(np.uint32)
        usable[idx] = True
    return ranks, usable


def graph_contexts(gpus, first):
    """the contexts a --gpus N edge build runs on: `first` (where the distinct barcodes already are) and one per further
    device; BADGER_AMD_CONTEXTS_ON_ONE_DEVICE=1 rehearses N on a one-GPU box with N independent contexts of device 0"""
    import os
    if gpus <= 1:
        return [first]
    if os.environ.get("BADGER_AMD_CONTEXTS_ON_ONE_DEVICE") == "1":
        return [first] + [_native.default_context(0, instance=100 + g) for g in range(1, gpus)]
    have = _native.device_count()
    if gpus > have:
        raise SystemExit("--gpus %d: this node shows %d device(s)" % (gpus, have))
    others = [d for d in range(have) if d != first.device][:gpus - 1]
    return [first] + [_native.default_context(d) for d in others]


class Stage2:
    def __init__(self, threshold, device=0):
        self.threshold = threshold
        self.device = device
        self.uniq = np.zeros(0, np.uint32)       # distinct barcodes, ascending
        self.count = np.zeros(0, np.int64)
        self.first = np.zeros(0, np.int64)       # position of the first occurrence among the counted barcodes
        self._ea = self._eb = np.zeros(0, np.uint32)  # edges as indices into uniq (on the host only if something asks for them)
        self.owner = np.zeros(0, np.int64)       # per distinct barcode: index of its centre, -1 conflict, -2 unclustered
        self.centers = []

    def _ctx(self):
        return _native.default_context(self.device)

    # the edges as positions in uniq.  After build_edges() they live on the device (self._dev); the host copy is made when
    # something reads it (the numpy clustering, tests) - the command line never does.
    def _edges_to_host(self):
        dev = getattr(self, "_dev", None)
        if self._ea is None:
            if dev is None:
                raise RuntimeError("the edges went back with the device buffers (release_device) before anything read them")
            rows = dev["rows"].to_host()
            self._ea, self._eb = rows[0, :dev["m"]], rows[1, :dev["m"]]

    @property
    def ea(self):
        self._edges_to_host()
        return self._ea

    @ea.setter
    def ea(self, v):
        self._ea = v

    @property
    def eb(self):
        self._edges_to_host()
        return self._eb

    @eb.setter
    def eb(self, v):
        self._eb = v

    # ------------------------------------------------------------------ counting
    def count_host(self, obs_rank, usable):
        """distinct barcodes of the usable reads, in read order (counts in first-occurrence order = argsort(first)).
        One sort of (rank << 32 | position) gives the distinct ranks, their first positions, their counts and - kept for
        per_read() - every usable read's place among the distinct ones."""
        r = obs_rank[usable]
        n = len(r)
        if n == 0 or n >= 1 << 32:
            self.uniq, self.first, self.count = np.unique(r, return_index=True, return_counts=True)
            self.uniq = self.uniq.astype(np.uint32)
            self._place = None
            return
        key = r.astype(np.uint64) << np.uint64(32) | np.arange(n, dtype=np.uint64)
        key.sort()
        rk = (key >> np.uint64(32)).astype(np.uint32)
        idx = (key & np.uint64(0xFFFFFFFF)).astype(np.int64)
        new_run = np.empty(n, dtype=bool)
        new_run[0] = True
        np.not_equal(rk[1:], rk[:-1], out=new_run[1:])
        start = np.flatnonzero(new_run)
        self.uniq = rk[start]
        self.first = idx[start]                               # (positions ascend inside a run: its first is the earliest read)
        self.count = np.diff(np.append(start, n)).astype(np.int64)
        place = np.empty(n, dtype=np.int64)
        place[idx] = np.cumsum(new_run, dtype=np.int64) - 1
        self._place = (obs_rank, usable, place)

    def count_device(self, ctx):
        """the same from the extraction records the context kept on the device (bdg_distinct_dev)"""
        ptr, n = ctx.kept_records()
        m = max(n, 1)
        # device arrays through the library's own allocator (bdg_mem_alloc): the command line runs without torch
        uniq = _native.DeviceArray(ctx, m, np.uint32)
        cnt = _native.DeviceArray(ctx, m, np.uint32)
        first = _native.DeviceArray(ctx, m, np.uint32)
        dn = _native.DeviceArray(ctx, 2, np.uint32)
        if n:
            ctx.distinct_dev(ptr, n, uniq, cnt, first, dn)
        nu, nbad = (int(x) for x in dn.to_host())
        if nbad:
            raise KeyError("%d extracted barcodes hold a base outside ACGT" % nbad)      # reference: rank() raises KeyError
        self.uniq = uniq.to_host(nu)
        self.count = cnt.to_host(nu).astype(np.int64)
        self.first = first.to_host(nu).astype(np.int64)
        self._d_uniq = uniq                                    # stays on the device for the edge build
        return nu

    # ------------------------------------------------------------------ graph
    def build_edges(self, ctx=None, on_device=False, gpus=1):
        """edges as pairs of positions in uniq.  The distinct barcodes are on the device already after count_device();
        after count_host() they are sent there.  Either way the edge list is built there (bdg_graph_edges_dev), its ranks
        are turned into positions there (bdg_rows_of_dev) and only the positions come back.
        gpus > 1: the reference's `-tr N` fan-out of compare_chunk (barcode_graph.py:164-189) over devices - every context
        holds the distinct barcodes and builds part g of N of the edge list (bdg_graph_edges_part_dev: disjoint shares whose
        union is the list, cut by the library), the shares' positions are put side by side on the first device, where the
        clustering runs.  No exchange between the devices."""
        nu = len(self.uniq)
        T = qgram_threshold(self.threshold, 16)
        if nu < 2:
            self._ea = self._eb = np.zeros(0, np.uint32)
            return
        ctx = ctx or self._ctx()
        d_uniq = self._d_uniq if on_device else _native.DeviceArray.from_host(ctx, self.uniq)
        if gpus > 1:
            return self._build_edges_parts(ctx, d_uniq, nu, T, gpus)
        cap = max(1024, 8 * nu)
        while True:
            d_edges = _native.DeviceArray(ctx, (cap, 3), np.uint32)
            d_tot = _native.DeviceArray(ctx, 1, np.uint64)
            ctx.graph_edges_dev(d_uniq, nu, self.threshold, T, d_edges, cap, d_tot)
            tot = int(d_tot.to_host()[0])
            ctx.graph_status()
            if tot <= cap:
                break
            cap = tot
            d_edges.free()
        d_rows = _native.DeviceArray(ctx, (2, max(tot, 1)), np.uint32)
        ctx.rows_of_dev(d_uniq, nu, d_edges, tot, 3, d_rows.data_ptr(), 0)
        ctx.rows_of_dev(d_uniq, nu, d_edges, tot, 3, d_rows.data_ptr() + 4 * max(tot, 1), 1)
        for d in (d_edges, d_tot):
            d.free()
        # the positions and the distinct barcodes stay on the device: the clustering levels, the count badger.py prints and the
        # per-read assignment run there (50 M edges are 400 MB that the host would only hold)
        self._ea = self._eb = None
        self._dev = {"ctx": ctx, "rows": d_rows, "m": tot, "uniq": d_uniq}

    def _build_edges_parts(self, ctx, d_uniq, nu, T, gpus):
        ctxs = [ctx] + [c for c in graph_contexts(gpus, ctx) if c is not ctx][:gpus - 1]
        n = len(ctxs)
        cap = max(1024, 8 * nu // n + 4096)
        work = []
        for g, c in enumerate(ctxs):                         # every device starts on its share before anybody waits
            du = d_uniq if c is ctx else _native.DeviceArray.from_host(c, self.uniq)
            d_edges = _native.DeviceArray(c, (cap, 3), np.uint32)
            d_tot = _native.DeviceArray(c, 1, np.uint64)
            c.graph_edges_part_dev(du, nu, g, n, self.threshold, T, d_edges, cap, d_tot)
            work.append([c, du, d_edges, d_tot, cap])
        shares = []
        for g, w in enumerate(work):
            c, du, d_edges, d_tot, cp = w
            tot = int(d_tot.to_host()[0])
            c.graph_status()
            while tot > cp:                                      # (a share larger than its room: once more with room)
                d_edges.free()
                cp = tot
                d_edges = _native.DeviceArray(c, (cp, 3), np.uint32)
                c.graph_edges_part_dev(du, nu, g, n, self.threshold, T, d_edges, cp, d_tot)
                tot = int(d_tot.to_host()[0])
                c.graph_status()
            d_rows = _native.DeviceArray(c, (2, max(tot, 1)), np.uint32)
            c.rows_of_dev(du, nu, d_edges, tot, 3, d_rows.data_ptr(), 0)
            c.rows_of_dev(du, nu, d_edges, tot, 3, d_rows.data_ptr() + 4 * max(tot, 1), 1)
            rows = d_rows.to_host()
            shares.append((rows[0, :tot], rows[1, :tot]))
            for d in (d_edges, d_tot, d_rows) + (() if c is ctx else (du,)):
                d.free()
        tot = sum(len(a) for a, _ in shares)
        both = np.zeros((2, max(tot, 1)), np.uint32)
        if tot:
            both[0, :tot] = np.concatenate([a for a, _ in shares])
            both[1, :tot] = np.concatenate([b for _, b in shares])
        self._ea = self._eb = None
        self._dev = {"ctx": ctx, "rows": _native.DeviceArray.from_host(ctx, both), "m": tot, "uniq": d_uniq}
        self.edge_shares = [len(a) for a, _ in shares]          # (for logs and tests: how the list was cut)

    # ------------------------------------------------------------------ centres
    def get_cluster_centers(self, true_barcodes, bc_len, barcode_list, n_cells, interval):
        """reference :252-277, on arrays.  Returns the centres as ranks, in the reference's order.  The reference sorts every
        distinct barcode by count; its loops only ever walk the barcodes above the cutoff (a prefix of that order), so only
        those are sorted here, and the rest of the order is produced if a loop really walks into it."""
        nu = len(self.uniq)
        cnt, first = self.count, self.first
        # counts of the first n_cells barcodes in insertion order (the counts dict, :253-254)
        if nu > n_cells:
            head = np.argpartition(first, n_cells)[:n_cells]
        else:
            head = np.arange(nu)
        cutoff = max(mean([int(x) for x in cnt[head]]) / 5.0, 5)      # (the mean of a set of integers: order does not matter)
        hi, lo = n_cells + n_cells * interval * 0.01, n_cells - n_cells * interval * 0.01
        above = np.flatnonzero(cnt > cutoff)
        # sorted(..., reverse=True) keeps ties in insertion order: by falling count, then by first occurrence
        order = [above[np.lexsort((first[above], -cnt[above]))]]

        def by_count(i):
            if i >= len(order[0]) and len(order[0]) < nu:                # a loop walks past the barcodes above the cutoff
                ins = np.argsort(first, kind="stable")
                order[0] = ins[np.argsort(-cnt[ins], kind="stable")]
            return int(order[0][i])

        tbcs, n, i = [], 0, 0
        if true_barcodes:
            tbcs = [rank(bc, bc_len) for bc in true_barcodes]
        elif barcode_list:
            wl_ranks = barcode_list.ranks if isinstance(barcode_list, BarcodeRanks) else rank_valid_many(barcode_list, bc_len).astype(np.uint32)
            top = order[0]
            listed = np.isin(self.uniq[top], wl_ranks)
            while i < len(top) and n <= hi:                               # (every barcode of `top` is above the cutoff)
                if listed[i]:
                    tbcs.append(int(self.uniq[top[i]]))
                    n += 1
                i += 1
        else:
            while cnt[by_count(i)] > cutoff and n <= hi:
                tbcs.append(int(self.uniq[by_count(i)]))
                i += 1
                n += 1
        while n < lo:
            tbcs.append(int(self.uniq[by_count(i)]))
            i += 1
            n += 1
        return tbcs

    # ------------------------------------------------------------------ clustering
    def cluster(self, true_barcodes, barcode_list, n_cells, bc_len, interval):
        self.centers = self.get_cluster_centers(true_barcodes, bc_len, barcode_list, n_cells, interval)
        nu = len(self.uniq)
        owner = np.full(nu, -2, dtype=np.int64)
        cr = np.array(self.centers, dtype=np.uint32)
        pos = np.searchsorted(self.uniq, cr)
        present = ((pos < nu) & (self.uniq[np.minimum(pos, nu - 1)] == cr)) if nu else np.zeros(len(cr), bool)
        cidx = pos[present]                                               # centres that were observed (the others have no edges)
        owner[cidx] = cidx
        dev = getattr(self, "_dev", None)
        if dev is not None and self._ea is None:
            # the edges are on the device: both levels there (bdg_cluster_dev), the same rule as the array code below
            print(1)
            print(2)                                                      # the reference prints the level numbers (:289)
            ctx = dev["ctx"]
            own32 = owner.astype(np.int32)
            d_owner = _native.DeviceArray.from_host(ctx, own32)
            m = dev["m"]
            ctx.cluster_dev(dev["rows"].data_ptr(), dev["rows"].data_ptr() + 4 * max(m, 1), m, nu, d_owner)
            self.owner = d_owner.to_host(nu).astype(np.int64) if nu else owner
            d_owner.free()
            return
        u = np.concatenate([self.ea, self.eb]).astype(np.intp)
        v = np.concatenate([self.eb, self.ea]).astype(np.intp)            # every edge in both directions: u expands, v is reached
        for level in (1, 2):
            print(level)                                                  # the reference prints the level number (:289)
            if level == 1:
                expanding = np.zeros(nu, bool)
                expanding[cidx] = True
            else:
                expanding = reached_once                                  # level-1 members that belong to a centre
            sel = expanding[u] & (owner[v] == -2)
            if not sel.any():
                reached_once = np.zeros(nu, bool)
                continue
            node, cen = v[sel], owner[u[sel]]
            pairs = np.unique(node * np.int64(nu) + cen)                  # distinct (barcode, centre) contacts
            node_u = pairs // nu
            first_of = np.concatenate([[True], node_u[1:] != node_u[:-1]])
            cnt_c = np.add.reduceat(np.ones(len(pairs), np.int64), np.nonzero(first_of)[0])
            nodes = node_u[first_of]
            cen_first = (pairs % nu)[first_of]
            single = cnt_c == 1
            owner[nodes[single]] = cen_first[single]
            owner[nodes[~single]] = -1                                    # reached by two centres on this level: nobody's
            reached_once = np.zeros(nu, bool)
            reached_once[nodes[single]] = True
        self.owner = owner

    def disconnected(self):
        """what badger.py prints at the end (:131-132): len(counts) - len(edges).  The reference's `edges` is a defaultdict:
        besides the barcodes that have an edge it holds a key for every barcode whose neighbours were looked up while
        clustering, i.e. for every centre (observed or not)."""
        nu = len(self.uniq)
        cr = np.unique(np.array(self.centers, dtype=np.uint32))                # a dict key exists once however often it is looked up
        pos = np.searchsorted(self.uniq, cr)
        present = ((pos < nu) & (self.uniq[np.minimum(pos, nu - 1)] == cr)) if nu else np.zeros(len(cr), bool)
        dev = getattr(self, "_dev", None)
        if dev is not None and self._ea is None:
            # the edges are on the device: counted there (bdg_touched_count_dev)
            ctx, m = dev["ctx"], dev["m"]
            cidx = np.ascontiguousarray(pos[present], dtype=np.uint32)
            d_c = _native.DeviceArray.from_host(ctx, cidx) if len(cidx) else None
            touched = ctx.touched_count_dev(dev["rows"].data_ptr(), dev["rows"].data_ptr() + 4 * max(m, 1), m, nu,
                                            d_c.data_ptr() if d_c is not None else 0, len(cidx))
            if d_c is not None:
                d_c.free()
            return nu - (touched + int((~present).sum()))
        key = np.zeros(nu, bool)
        key[self.ea] = True
        key[self.eb] = True
        key[pos[present]] = True
        return nu - (int(key.sum()) + int((~present).sum()))

    # ------------------------------------------------------------------ assignment and output
    def assigned(self, high_sens=False):
        """per distinct barcode: (rank of the barcode it is corrected to, whether it has one)"""
        out = np.zeros(len(self.uniq), dtype=np.uint32)
        ok = self.owner >= 0
        out[ok] = self.uniq[self.owner[ok]]
        has = ok.copy()
        if high_sens:
            centers = np.unique(out[ok])                                  # the centres in use, ascending: ties -> lowest rank
            todo = np.nonzero(~ok)[0]
            if len(centers) and len(todo):
                idx, ed, _ = self._ctx().nearest16(self.uniq[todo], centers, 2)
                near = ed < 3
                out[todo[near]] = centers[idx[near]]
                has[todo[near]] = True
        return out, has

    def assigned_rank(self, high_sens=False):
        """per distinct barcode: rank of the barcode it is corrected to, NONE if unassigned"""
        out, has = self.assigned(high_sens)
        out = out.copy()
        out[~has] = NONE
        return out

    def per_read(self, obs_rank, usable, high_sens):
        """per read: (rank of the corrected barcode, whether there is one), on the host"""
        assigned, has = self.assigned(high_sens)
        n = len(obs_rank)
        rank, got = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint8)
        if usable.any():
            kept = getattr(self, "_place", None)
            if kept is not None and kept[0] is obs_rank and kept[1] is usable:
                pos = kept[2]                                 # (count_host saw these very arrays)
            else:
                pos = np.searchsorted(self.uniq, obs_rank[usable])
            rank[usable] = assigned[pos]
            got[usable] = has[pos]
        return rank, got

    def output_file(self, read_ids, obs_rank, usable, out, high_sens):
        """<out>_output_file.tsv with columns readID, barcode (reference :388-410), written natively (bdg_write_assignments).
        read_ids: a list of str or an _native.IdStore."""
        rank, got = self.per_read(obs_rank, usable, high_sens)
        ids = read_ids if isinstance(read_ids, _native.IdStore) else _native.IdStore(read_ids)
        _native.write_assignments(ids, rank, got, out + "_output_file.tsv")

    def output_file_from_device(self, ids, ctx, out, high_sens):
        """the same straight from the extraction records the context kept: per read, the position of its barcode among the
        distinct ones and what that was corrected to, on the device (bdg_assign_reads_dev); the host only writes the file"""
        assigned, has = self.assigned(high_sens)
        ptr, n = ctx.kept_records()
        dev = self._dev
        d_assigned = _native.DeviceArray.from_host(ctx, assigned)
        d_has = _native.DeviceArray.from_host(ctx, has.astype(np.uint8))
        d_rank = _native.DeviceArray(ctx, max(n, 1), np.uint32)
        d_got = _native.DeviceArray(ctx, max(n, 1), np.uint8)
        if n:
            ctx.assign_reads_dev(ptr, n, dev["uniq"], len(self.uniq), d_assigned, d_has, d_rank, d_got)
        rank, got = d_rank.to_host(n), d_got.to_host(n)
        for d in (d_assigned, d_has, d_rank, d_got):
            d.free()
        _native.write_assignments(ids, rank, got, out + "_output_file.tsv")

    def release_device(self):
        dev = getattr(self, "_dev", None)
        if dev is not None:
            dev["rows"].free()
            if dev["uniq"] is not getattr(self, "_d_uniq", None):
                dev["uniq"].free()
            self._dev = None
