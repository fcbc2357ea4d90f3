"""CPU stand-ins for the HIP objects bench.py drives -- TEST INFRASTRUCTURE, imported by bench.py only under --rehearse-cpu
(tests/test_bench_launcher.py: gloo, world size 2, no GPU in the container).  The launcher, the sharding, the collective and the
cross-rank checks of bench.py are then the code that runs on the 8-GPU node; the compute here is the CPU oracle, and the line
such a run prints is marked invalid.  The product path never imports this module."""
import time

import numpy as np

import oracle_lib as O


def cpu_one(mode, kind, p, t, scoring):
    if mode == "affine":
        return (O.ref_affine_score if kind == "reference" else O.affine_score)(p, t, *scoring)
    if mode == "nwdist":
        return O.ref_nw_distance(p, t, *scoring) if kind == "reference" else O.nw_distance(p, t, *scoring)[0]
    return (O.ref_align(mode, p, t, *scoring) if kind == "reference" else O.align(mode, p, t, *scoring))["score"]


class RehearsalBatch:
    """Stands in for bioinformatics_algorithms_amd.Batch (scores-only pass)."""

    def __init__(self, mode, seqs, pa, pb, scoring):
        self._mode, self._seqs, self._pa, self._pb, self._sc = mode, seqs, pa, pb, scoring
        self.n_pairs = len(pa)
        self._out = None
        self._times = []

    def info(self):
        cells = sum(len(self._seqs[a]) * len(self._seqs[b]) for a, b in zip(self._pa.tolist(), self._pb.tolist()))
        return dict(cells=cells, padded_cells=cells, n_tasks=self.n_pairs, kernel="cpu-oracle-rehearsal")

    def set_out(self, tensor):
        self._out = tensor

    def run(self, stream=None):
        t0 = time.perf_counter()
        sc = [cpu_one(self._mode, "port", self._seqs[a], self._seqs[b], self._sc) for a, b in zip(self._pa.tolist(), self._pb.tolist())]
        self._scores = np.asarray(sc, dtype=np.int32)
        if self._out is not None:
            import torch
            self._out.copy_(torch.from_numpy(self._scores))
        self._times.append((time.perf_counter() - t0) * 1e3)

    def fetch_into(self, arr):
        arr[:] = self._scores

    def last_ms(self):
        return self._times[-1]

    def run_times(self, cap=64):
        return self._times[-cap:]

    def close(self):
        pass


class RehearsalContext:
    """Stands in for bioinformatics_algorithms_amd.Context in the `g` / `gb` workloads (full alignments of a pair block)."""

    def set_score_band(self, on):
        pass

    def align_batch_arrays(self, mode, packed, pa, pb, match, mismatch, gap, out=None):
        _, _, seqs = packed
        t0 = time.perf_counter()
        res = [O.align(mode, seqs[int(a)], seqs[int(b)], match, mismatch, gap, compact=True) for a, b in zip(pa, pb)]
        n = len(res)
        ops_off = np.zeros(max(n, 1), dtype=np.uint64)
        caps = [len(seqs[int(a)]) + len(seqs[int(b)]) for a, b in zip(pa, pb)]
        if n > 1:
            ops_off[1:n] = np.cumsum(caps[:-1])
        ops = np.zeros(sum(caps) + 1, dtype=np.uint8)
        for k, r in enumerate(res):
            ops[int(ops_off[k]):int(ops_off[k]) + len(r["ops"])] = np.frombuffer(r["ops"], dtype=np.uint8)
        self._ms = (time.perf_counter() - t0) * 1e3
        return dict(scores=np.array([r["score"] for r in res], dtype=np.int32), n_ops=np.array([len(r["ops"]) for r in res], dtype=np.uint64),
                    ops=ops, ops_off=ops_off)

    def align_stats(self):
        return dict(fill_ms=self._ms, traceback_ms=0.0, band_bytes=0)
