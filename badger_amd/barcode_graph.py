"""Host-side mirror of the reference's BarcodeGraph (barcode_graph.py), with the edge build
(graph_construction, :207-249) and the nearest-center pass (postprocessing, :370-385) on the
MI355X through the C ABI.  Clustering (:252-301) is an O(V+E) dictionary walk and stays on
the host, restated so that the final TSV matches the reference's.

Post-conditions of graph_construction are the reference's: counts (rank -> n, first-occurrence
order), edges (rank -> [rank], both directions), dists ((a,b) -> d, both directions).
"""
import logging
from collections import defaultdict
from statistics import mean

import numpy as np

from . import _native
from .common import rank, rank_many, unrank

logger = logging.getLogger("BarcodeGraph")


def qgram_threshold(threshold, bc_len, q=6):
    """reference index.py:22-24"""
    t = bc_len - q + 1 - q * threshold
    return t if t > 0 else 4


class BarcodeGraph:
    def __init__(self, threshold, device=0):
        self.threshold = threshold
        self.device = device
        self.counts = {}
        self.edges = defaultdict(list)
        self.dists = {}
        self.clusters = defaultdict(list)
        self.clustering = dict()
        self.clustered = defaultdict(bool)

    def _ctx(self):
        return _native.default_context(self.device)

    # ------------------------------------------------------------------ graph
    def index_barcodes(self, barcodes, bc_len):
        """counts in first-occurrence order (reference :192-204): 17-char strings lose their
        last base, anything that is not then bc_len long is dropped."""
        kept = []
        for s in barcodes:
            if len(s) == bc_len + 1:
                s = s[:-1]
            if len(s) == bc_len:
                kept.append(s)
        ranks = rank_many(kept, bc_len)
        uniq, first, cnt = np.unique(ranks, return_index=True, return_counts=True)
        order = np.argsort(first, kind="stable")
        self.counts = {int(uniq[i]): int(cnt[i]) for i in order}
        return uniq.astype(np.uint32)

    def _take_edges(self, a, b, d):
        for x, y, z in zip(a, b, d):
            self.edges[x].append(y)
            self.edges[y].append(x)
            self.dists[(x, y)] = z
            self.dists[(y, x)] = z

    def graph_construction(self, barcodes, bc_len, threads=1):
        if bc_len != 16:
            raise ValueError("only 16-base barcodes (tenX) are supported")
        uniq = self.index_barcodes(barcodes, bc_len)
        e = self._ctx().graph_edges(uniq, self.threshold, qgram_threshold(self.threshold, bc_len))
        self._take_edges(e["a"].tolist(), e["b"].tolist(), e["dist"].tolist())

    def graph_construction_from_device(self, ctx, bc_len=16):
        """Same post-conditions as graph_construction, from the extraction records the context kept on the device
        (extract_keep_records): distinct barcodes and their counts by sort + run-length on the GPU
        (index_bc_single_thread, reference :192-204), edges straight from that sorted array; only the distinct ranks,
        counts, first-occurrence indices and the edge list come back to the host.  counts is ordered by first
        occurrence like the reference's dict (get_cluster_centers depends on it, :253-255)."""
        import torch
        if bc_len != 16:
            raise ValueError("only 16-base barcodes (tenX) are supported")
        ptr, n = ctx.kept_records()
        dev = torch.device("cuda", ctx.device)
        if n == 0:
            return
        m = max(n, 1)
        uniq = torch.zeros(m, dtype=torch.int32, device=dev)
        cnt = torch.zeros(m, dtype=torch.int32, device=dev)
        first = torch.zeros(m, dtype=torch.int32, device=dev)
        dn = torch.zeros(2, dtype=torch.int32, device=dev)
        ctx.distinct_dev(ptr, n, uniq, cnt, first, dn)
        ctx.synchronize()
        nu, nbad = int(dn[0]), int(dn[1])
        if nbad:
            raise KeyError("%d extracted barcodes hold a base outside ACGT" % nbad)      # reference: rank() raises KeyError
        h_uniq = uniq[:nu].cpu().numpy().view(np.uint32)
        h_cnt = cnt[:nu].cpu().numpy()
        order = np.argsort(first[:nu].cpu().numpy().view(np.uint32), kind="stable")
        self.counts = {int(h_uniq[i]): int(h_cnt[i]) for i in order}
        if nu < 2:
            return
        T = qgram_threshold(self.threshold, bc_len)
        cap = max(1024, 8 * nu)
        while True:
            d_edges = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
            d_tot = torch.zeros(1, dtype=torch.int64, device=dev)
            ctx.graph_edges_dev(uniq, nu, self.threshold, T, d_edges, cap, d_tot)
            ctx.synchronize()
            ctx.graph_status()
            tot = int(d_tot[0])
            if tot <= cap:
                break
            cap = tot
        e = d_edges[:tot].cpu().numpy().view(np.uint32)
        e = e[np.lexsort((e[:, 1], e[:, 0]))]               # the order the host-buffer call delivers (neighbour order is free anyway)
        self._take_edges(e[:, 0].tolist(), e[:, 1].tolist(), e[:, 2].tolist())

    # ------------------------------------------------------------------ clustering (host)
    def get_cluster_centers(self, true_barcodes, bc_len, barcode_list, n_cells, interval):
        """reference :252-277"""
        by_count = [k for k, _ in sorted(self.counts.items(), key=lambda kv: kv[1], reverse=True)]
        cutoff = max(mean(list(self.counts.values())[:n_cells]) / 5.0, 5)
        hi, lo = n_cells + n_cells * interval * 0.01, n_cells - n_cells * interval * 0.01
        tbcs, n, i = [], 0, 0
        if true_barcodes:
            tbcs = [rank(bc, bc_len) for bc in true_barcodes]
        elif barcode_list:
            while i < len(by_count) and self.counts[by_count[i]] > cutoff and n <= hi:
                if unrank(by_count[i], bc_len) in barcode_list:
                    tbcs.append(by_count[i])
                    n += 1
                i += 1
        else:
            while self.counts[by_count[i]] > cutoff and n <= hi:
                tbcs.append(by_count[i])
                i += 1
                n += 1
        while n < lo:
            tbcs.append(by_count[i])
            i += 1
            n += 1
        return tbcs

    def cluster(self, true_barcodes, barcode_list, n_cells, bc_len, interval):
        """Two breadth-first levels from every center; a barcode reached by two centers on the
        same level belongs to nobody (reference :279-301)."""
        for tbc in self.get_cluster_centers(true_barcodes, bc_len, barcode_list, n_cells, interval):
            self.clusters[tbc] = [tbc]
            self.clustering[tbc] = (tbc, 0)
            self.clustered[tbc] = True
        for level in (1, 2):
            print(level)                      # the reference prints the level number (:289)
            for center in self.clusters.keys():
                members = self.clusters[center]
                for k in range(len(members)):
                    for nb in self.edges[members[k]]:
                        if not self.clustered[nb]:
                            members.append(nb)
                            self.clustering[nb] = (center, level)
                            self.clustered[nb] = True
                        else:
                            owner, lvl = self.clustering[nb]
                            if owner != center and owner != -1 and lvl == level:
                                self.clusters[owner].remove(nb)
                                self.clustering[nb] = (-1, -1)

    def assign_by_cluster(self, bc_len):
        out = defaultdict(str)
        for node in self.counts.keys():
            if self.clustered[node] and self.clustering[node][0] != -1:
                out[unrank(node, bc_len)] = unrank(self.clustering[node][0], bc_len)
        return out

    def postprocessing(self, assignments, bc_len):
        """Unassigned barcodes go to the nearest cluster center if it is closer than 3 edits
        (reference :370-385).  The reference walks a set of strings, so which of several
        equally near centers wins depends on the hash seed; here the lowest rank wins."""
        centers = sorted({rank(c, bc_len) for c in set(assignments.values()) if len(c) == bc_len})
        todo = [r for r in self.counts.keys() if assignments[unrank(r, bc_len)] in ("", "*")]
        if not centers or not todo:
            return assignments
        idx, ed, _ = self._ctx().nearest16(np.array(todo, dtype=np.uint32), np.array(centers, dtype=np.uint32), 2)
        for r, i, d in zip(todo, idx.tolist(), ed.tolist()):
            if d < 3:
                assignments[unrank(r, bc_len)] = unrank(centers[i], bc_len)
        return assignments

    def output_file(self, read_assignment, out, true_barcodes, bc_len, post):
        """<out>_output_file.tsv with columns readID, barcode (reference :388-410)."""
        assignments = self.assign_by_cluster(bc_len)
        if post:
            assignments = self.postprocessing(assignments, bc_len)
        with open(out + "_output_file.tsv", "w") as f:
            f.write("readID\tbarcode\n")
            for read in read_assignment:
                observed, assigned = read[1], "*"
                if observed != "*":
                    assigned = assignments[observed] or "*"
                f.write("%s\t%s\n" % (read[0], assigned))
