#!/usr/bin/env python3
"""bench.py -- GCUPS of the hw2.cpp DP hot path on MI355X (contract: one JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W [--workload c3|c4|c2|c5]

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM.
Default workload (BASELINE.json configs[2], the config the >=1 TCUPS target is quoted on):
  c3  batched Smith-Waterman, linear gap, int32, scores only: 4096 patterns (150 bp) x 256 texts
      (10 kbp) = 1 048 576 pairs = 1.573e12 DP cells per step and per GPU, scoring 1/-1/-1.
      N > 1: every rank gets its own 256 texts (weak scaling), one RCCL all-gather of the per-pair
      int32 scores per step.
Other workloads (reported the same way, not the headline line):
  c4  all-pairs NW of 1024 sequences x 1000 bp (523 776 pairs), pairs sharded over the ranks
      (strong scaling) + all-gather;   c2 / c5  one SW 10k x 10k / NW 100k x 100k pair with the
      traceback band written to HBM (does not shard: replicas only).

Inputs come from the counter-based generator of SURVEY.md 8(d) (splitmix64), restated here in numpy.
The oracle / compiled reference are used ONLY for the cpu_baseline leg and to verify a sample of the
GPU results after the timed region.
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "bioinformatics-algorithms_amd")

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU ceiling = the architectural one of MI355X_MICROARCH.md: 4 SIMD-32 per CU, a wave64 VALU instruction issues over
# 2 cycles  =>  256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s (the same rate as the guide's 157.3 TFLOP/s
# of vector FP32 FMA).  `roofline.frac` is quoted against THIS figure, so it is <= 1 by construction.  Only the
# adder / logic / move class of instructions reaches that rate; v_max3_i32, SDWA, v_perm_b32, compares, shifts and
# every v_pk_* issue at half of it (tools/valu_issue.hip -> profiles/r02_valu_issue_microbench.txt), which is why a
# DP cell of max3 + sdwa-add + perm cannot get much past 0.5 -- `roofline.issue_model` prices the actual mix.
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
# whole-workload checksums (sum of all per-pair scores of rank 0's shard at N = 1), asserted after the timed region.
# c3: independently re-derived by the round-1 judge from 3 000 random pairs through the CPU oracle (33.739 +- 0.047 per
# pair against 35 376 135 / 1 048 576 = 33.737) and asserted by tests/test_gpu_parity.py::test_full_size_c3_batch_properties
EXPECTED_CHECKSUM = {"c3": 35376135}
# VALU instructions per evaluated DP cell of each instantiation: SQ_INSTS_VALU x 64 / padded cells from the
# rocprofv3 PMC pass (c3: 5.02, c4: 2.53; profiles/), the others counted in the gfx950 ISA of the column block
VALU_PER_CELL = {"SC_PERM": 10.75, "SC_CMP": 12.0, "PACKED": 7.03,   # batch_nwdist_kernel<R,SCORE> / packed keys (hw4)
                 "BM_AFFS,SC_PERM": 5.6, "BM_AFFS,SC_CMP": 7.6, "BM_AFF,SC_PERM": 7.6, "BM_AFF,SC_CMP": 9.6,
                 "BM_SWS,SC_PERM": 4.06, "BM_SWS,SC_CMP": 6.06,
                 "BM_SWS,SC_PERM,LANES": 4.11, "BM_SWS,SC_CMP,LANES": 5.6,   # per-lane texts: 1248 / 1702 VALU per 304 cells (ISA)
                 "BM_NWG,SC_PERM,LANES": 2.58,
                 "BM_SW,SC_PERM": 5.02, "BM_SW,SC_CMP": 6.9, "BM_NW,SC_PERM": 4.5, "BM_NW,SC_CMP": 6.5,
                 "BM_NWG,SC_PERM": 2.53, "BM_NWG,SC_CMP": 4.5}


# Issue cost of the column loop per cell and SIMD lane group, from the per-instruction costs measured by
# tools/valu_class.hip (profiles/r01_valu_class_microbench.txt, >= 2 waves per SIMD): add / sub(+clamp) / and / xor / mov
# 3.15 / 3.16 / 2.68 / 2.65 / 2.64 cycles, max / max3 / sdwa / perm / cmp / cndmask ~4.45.  The flat "4 cycles per
# instruction" of VALU_PEAK_TOPS is the mean of the two classes; this is the finer model.
ISSUE_CYCLES_PER_CELL = {
    "BM_SWS,SC_PERM": 4.45 + 1.5 * 4.46 + 3.16 + 0.25 * 4.49 + 0.25 * 2.65,          # sdwa add, 1.5 max3, sub clamp, table
    "BM_SWS,SC_PERM,LANES": 4.45 + 1.5 * 4.46 + 3.16 + (0.25 + 4 / 304) * 4.49 + 0.25 * 2.65,
    "BM_NWG,SC_PERM": 4.45 + 4.46 + 0.25 * 4.49 + 0.25 * 2.65,                       # sdwa add, max3, table
    "BM_NWG,SC_PERM,LANES": 4.45 + 4.46 + (0.25 + 4 / 608) * 4.49 + 0.25 * 2.65,
    "PACKED": 3 * 3.15 + 2.68 + 4.45 + 4.45 + 4.46,                                   # 3 add, and, cmp, cndmask, max3
}


def load_pkg():
    name = "bioinformatics_algorithms_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# ----------------------------------------------------------------------------- synthetic input
def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def gen(seed, stream, ident, length):
    """base(seed,stream,id,pos) = "ACGT"[splitmix64(key + pos) >> 62]."""
    with np.errstate(over="ignore"):
        key = _splitmix64(_splitmix64(np.uint64(seed)) ^ (np.uint64(stream) << np.uint64(56)) ^ np.uint64(ident))
        v = _splitmix64(key + np.arange(length, dtype=np.uint64))
    return np.frombuffer(b"ACGT", dtype=np.uint8)[(v >> np.uint64(62)).astype(np.int64)].tobytes()


assert gen(1, 0, 0, 32) == b"GCAAAATTTCCTCTACCCAATTGGACGCATGC"   # SURVEY.md 8(d) self-check


class DevPtr:
    """Expose a raw device pointer to torch (for the RCCL all-gather) without copying."""

    def __init__(self, ptr, n, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def attach_traffic(roofline, workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic_<workload>.json, written by
    tools/make_traffic.py from the rocprofv3 summaries), corrected as MI355X_MICROARCH.md prescribes."""
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        roofline["traffic"] = tj.get("hbm_bytes_per_launch")
        roofline["traffic_source"] = tj.get("source")
        if tj.get("kernel_avg_ms") is not None:
            roofline["traffic_note"] = "per dispatch of %s (rocprof avg %.4f ms per dispatch)" % (tj.get("kernel"), tj["kernel_avg_ms"])


# ----------------------------------------------------------------------------- CPU baseline leg
def _cpu_one(mode, O, kind, p, t, scoring):
    if mode == "affine":
        return (O.ref_affine_score if kind == "reference" else O.affine_score)(p, t, *scoring)
    if mode == "nwdist":
        return O.ref_nw_distance(p, t, *scoring) if kind == "reference" else O.nw_distance(p, t, *scoring)[0]
    return (O.ref_align(mode, p, t, *scoring) if kind == "reference" else O.align(mode, p, t, *scoring))["score"]


def cpu_baseline(mode, pairs, seqs, scoring, budget_s=25.0, max_pairs=2048):
    """Time the unmodified reference (oracle/_ref, kind "reference") or the C restatement (kind
    "port") on a bounded sample of the same workload (`pairs`: a seeded random sample drawn over the WHOLE pair
    list, so that every pattern and text can be hit), 1 thread; then all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    kind = "reference" if (O.have_ref3() if mode == "affine" else O.have_ref4() if mode == "nwdist" else O.have_ref()) else "port"
    fn = lambda p, t: _cpu_one(mode, O, kind, p, t, scoring)   # noqa: E731
    t0 = time.perf_counter()
    cells, scores = 0, []
    for (a, b) in pairs[:max_pairs]:
        scores.append(fn(seqs[a], seqs[b]))
        cells += len(seqs[a]) * len(seqs[b])
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": kind,
           "sample": "%d pairs drawn at random (seed 481) over the whole pair list (%.3g cells, %.1f s), %s, g++ -O2"
                     % (len(scores), cells, dt, "six int matrices as hw3.cpp" if mode == "affine"
                        else "int+char matrices + traceback strings as hw4.cpp" if mode == "nwdist"
                        else "full int+char matrices as hw2.cpp")}
    # all host cores: one process per core over disjoint shards (the reference itself is single-threaded)
    try:
        ncore = min(len(os.sched_getaffinity(0)), 16)   # the 1-GPU box's CPU share
        import multiprocessing as mp
        per = max(1, min(len(scores), max_pairs) // 2)
        shards = [[pairs[(w * per + k) % len(pairs)] for k in range(per)] for w in range(ncore)]
        with mp.get_context("fork").Pool(ncore) as pool:
            t1 = time.perf_counter()
            res = pool.starmap(_cpu_shard, [(mode, sh, seqs, scoring, kind) for sh in shards])
            dt_all = time.perf_counter() - t1
        out["all_cores"] = {"value": sum(res) / dt_all / 1e9, "cores": ncore}
    except Exception as e:   # noqa: BLE001 -- the extra figure is optional
        out["all_cores"] = {"error": str(e)}
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu_model"] = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:   # noqa: BLE001
        pass
    return out, scores


def _cpu_shard(mode, shard, seqs, scoring, kind):
    import oracle_lib as O
    cells = 0
    for (a, b) in shard:
        _cpu_one(mode, O, kind, seqs[a], seqs[b], scoring)
        cells += len(seqs[a]) * len(seqs[b])
    return cells


# ----------------------------------------------------------------------------- workloads
def build_c3(rank, n_patterns=4096, n_texts=256, plen=150, tlen=10000):
    pats = [gen(1, 0, p, plen) for p in range(n_patterns)]
    txts = [gen(1, 1, rank * n_texts + t, tlen) for t in range(n_texts)]
    seqs = pats + txts
    pa = np.repeat(np.arange(n_patterns, dtype=np.uint32), n_texts)
    pb = np.tile(np.arange(n_texts, dtype=np.uint32) + np.uint32(n_patterns), n_patterns)
    desc = {"workload": "c3: batched SW scores-only, %d patterns x %d bp  X  %d texts x %d bp per GPU, 1/-1/-1"
                        % (n_patterns, plen, n_texts, tlen),
            "pairs_per_gpu": int(n_patterns * n_texts), "scoring": [1, -1, -1]}
    return "sw", seqs, pa, pb, (1, -1, -1), desc


def build_c3i(rank, n_pairs=131072, plen=150, tlen=2000):
    """Index-paired list, every pair its own text: the shape of the reference's own loop (hw2.cpp:328-338)."""
    pats = [gen(1, 0, p, plen) for p in range(n_pairs)]
    txts = [gen(1, 1, rank * n_pairs + t, tlen) for t in range(n_pairs)]
    seqs = pats + txts
    pa = np.arange(n_pairs, dtype=np.uint32)
    pb = pa + np.uint32(n_pairs)
    desc = {"workload": "c3i: SW scores-only, %d index-paired (pattern i, text i) pairs %d x %d bp per GPU, 1/-1/-1"
                        % (n_pairs, plen, tlen), "pairs_per_gpu": int(n_pairs), "scoring": [1, -1, -1]}
    return "sw", seqs, pa, pb, (1, -1, -1), desc


def build_long(rank, n_pairs=64, slen=10000):
    """Few LONG pairs (the C2 shape, 64 times): what hw2.cpp's loop hands over when a user aligns a handful of long sequences."""
    pats = [gen(1, 0, rank * n_pairs + p, slen) for p in range(n_pairs)]
    txts = [gen(1, 1, rank * n_pairs + t, slen) for t in range(n_pairs)]
    seqs = pats + txts
    pa = np.arange(n_pairs, dtype=np.uint32)
    pb = pa + np.uint32(n_pairs)
    desc = {"workload": "long: SW scores-only, %d index-paired pairs %d x %d bp per GPU (few long pairs: routed to the stripe engine), 1/-1/-1"
                        % (n_pairs, slen, slen), "pairs_per_gpu": int(n_pairs), "scoring": [1, -1, -1]}
    return "sw", seqs, pa, pb, (1, -1, -1), desc


def build_c4(rank, world, n_seq=1024, slen=1000):
    seqs = [gen(1, 2, i, slen) for i in range(n_seq)]
    ii, jj = np.triu_indices(n_seq, k=1)
    n_pairs = len(ii)
    load_pkg()
    from bioinformatics_algorithms_amd import shard
    lo, hi, per = shard.block(n_pairs, world, rank)   # contiguous equal-count blocks, last one padded
    pa = ii[lo:hi].astype(np.uint32)
    pb = jj[lo:hi].astype(np.uint32)
    desc = {"workload": "c4: all-pairs NW scores, %d seq x %d bp (%d pairs) sharded over %d GPU(s), 1/-1/-1"
                        % (n_seq, slen, n_pairs, world),
            "pairs_total": int(n_pairs), "scoring": [1, -1, -1], "_per": int(per)}
    return "nw", seqs, pa, pb, (1, -1, -1), desc


def build_hw3(rank, world, n_seq=1024, slen=1000):
    """hw3.cpp all-pairs affine score pass (232-241) on generator sequences, scoring of README.txt:31."""
    mode, seqs, pa, pb, _, desc = build_c4(rank, world, n_seq, slen)
    desc["workload"] = ("hw3: all-pairs AFFINE score pass (hw3.cpp:232-241), %d seq x %d bp (%d pairs) sharded over %d GPU(s), "
                        "5:-4:-16:-4" % (n_seq, slen, desc["pairs_total"], world))
    desc["scoring"] = [5, -4, -16, -4]
    return "affine", seqs, pa, pb, (5, -4, -16, -4), desc


def build_hw4(rank, world, n_seq=1024, slen=1000):
    """hw4.cpp all-pairs step (138-159): NW with hw4's tie-break + gap/mismatch column count."""
    mode, seqs, pa, pb, _, desc = build_c4(rank, world, n_seq, slen)
    desc["workload"] = ("hw4: all-pairs NW + traceback-derived distance (hw4.cpp:138-159), %d seq x %d bp (%d pairs) sharded "
                        "over %d GPU(s), 1/-1/-1" % (n_seq, slen, desc["pairs_total"], world))
    return "nwdist", seqs, pa, pb, (1, -1, -1), desc


# ----------------------------------------------------------------------------- launcher
def self_launch(args):
    """`python bench.py --gpus N` without an external launcher: start N ranks (one process per GPU) as CHILD processes
    and exit with their status.  Runs before anything in this process has touched HIP or torch -- no exec of a process
    that initialised the GPU -- and stops the remaining ranks by their exact PIDs if one fails."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, alive = 0, set(range(args.gpus))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    sys.exit(rc)


def rehearsal():
    """--rehearse-cpu ONLY: the CPU stand-ins of tests/rehearsal_batch.py (test infrastructure: gloo + the oracle as compute, so that the
    launcher, the sharding and the collective of this script can be driven without a GPU; the line is marked invalid)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rehearsal_batch
    return rehearsal_batch


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c3", "c3i", "long", "c4", "c2", "c2b", "c5", "hw3", "hw4", "g", "gb"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plen", type=int, default=150, help="pattern length of the g / gb workloads (150 = the C3 shape)")
    ap.add_argument("--pairs", type=int, default=0, help="g / gb: pairs per GPU (default 4096; other values are experiments, the line says so)")
    ap.add_argument("--nw", action="store_true", help="c3i: global (NW) instead of local scores")
    ap.add_argument("--small", action="store_true", help="reduced sizes (functional check only; line is marked invalid)")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="launcher / sharding / collective rehearsal without a GPU: gloo + the CPU oracle as compute "
                         "(needs --small; the line is marked invalid)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.rehearse_cpu and not args.small:
        raise SystemExit("--rehearse-cpu is a functional rehearsal: it needs --small")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)   # N child ranks; does not return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with matching values (or without a launcher)" % (args.gpus, world))
    if os.environ.get("BENCH_TEST_FAIL_RANK") == str(rank) and args.rehearse_cpu:
        raise SystemExit("rank %d: failure injected by tests/test_bench_launcher.py" % rank)
    dist = None
    torch = None
    # BENCH_FORCE_DIST=1 takes the distributed code path (RCCL init, explicit stream, all-gather) even with
    # one rank: rehearsal on a 1-GPU box under `python -m torch.distributed.run --nproc-per-node 1 ...`
    use_dist = world > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))
    cpu = args.rehearse_cpu
    dist_info = None
    if use_dist:
        import torch   # noqa: F811
        import torch.distributed as dist   # noqa: F811
        # ONE JSON line on stdout: RCCL's own messages (banner at INFO, "NCCL WARN Could not read node" on some boxes) go to stderr
        os.environ["NCCL_DEBUG"] = os.environ.get("BENCH_NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        if "RANK" not in os.environ:   # BENCH_FORCE_DIST=1 without a launcher: a one-rank group on this process
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
            sk.close()
        if cpu:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
        # which device every rank drives: gathered once, reported in the line (proof that N ranks on N devices took part)
        dev = torch.tensor([-1 if cpu else torch.cuda.current_device()], dtype=torch.int32, device="cpu" if cpu else "cuda")
        devs = [torch.empty_like(dev) for _ in range(dist.get_world_size())]
        dist.all_gather(devs, dev)
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "launcher": "bench.py --gpus N (child processes)" if os.environ.get("BENCH_SELF_LAUNCHED") else ("single process (BENCH_FORCE_DIST)" if "TORCHELASTIC_RUN_ID" not in os.environ and world == 1 else "external (torch.distributed.run)"),
                     "device_of_rank": [int(d.item()) for d in devs]}

    pkg = ctx = None
    if not cpu:
        pkg = load_pkg()
        ctx = pkg.Context(local_rank)

    if args.workload in ("g", "gb"):
        return bench_global_batch(args, pkg, ctx, rank, world, dist if use_dist else None, torch, dist_info, cpu)
    if args.workload in ("c2", "c2b", "c5"):
        return bench_single_pair(args, pkg, ctx, rank, world, dist if use_dist else None, torch)

    if args.workload == "c3":
        kw = dict(n_patterns=256, n_texts=16, tlen=2000) if args.small else {}
        if cpu:
            kw = dict(n_patterns=24, n_texts=4, tlen=300)
        mode, seqs, pa, pb, scoring, desc = build_c3(rank, **kw)
        scaling = "weak"
    elif args.workload == "c3i":
        kw = dict(n_pairs=4096) if args.small else {}
        mode, seqs, pa, pb, scoring, desc = build_c3i(rank, **kw)
        if args.nw:
            mode = "nw"
            desc["workload"] = desc["workload"].replace("SW scores-only", "NW scores-only")
        scaling = "weak"
    elif args.workload == "long":
        kw = dict(n_pairs=8, slen=2000) if args.small else dict(n_pairs=args.pairs or 64)
        mode, seqs, pa, pb, scoring, desc = build_long(rank, **kw)
        scaling = "weak"
    elif args.workload in ("hw3", "hw4"):
        kw = dict(n_seq=128) if args.small else {}
        mode, seqs, pa, pb, scoring, desc = (build_hw3 if args.workload == "hw3" else build_hw4)(rank, world, **kw)
        scaling = "strong"
    else:
        kw = dict(n_seq=128) if args.small else {}
        if cpu:
            kw = dict(n_seq=12, slen=120)
        mode, seqs, pa, pb, scoring, desc = build_c4(rank, world, **kw)
        scaling = "strong"

    def make_batch(packed=None):   # packed: the blob + offsets form a compiled host already holds (pack_sequences)
        src = packed if packed is not None else seqs
        if cpu:
            return rehearsal().RehearsalBatch(mode, seqs, pa, pb, scoring)
        if mode == "affine":
            return ctx.batch_affine(src, pa, pb, *scoring)
        if mode == "nwdist":
            return ctx.batch_distances(src, pa, pb, *scoring)
        return ctx.batch(mode, src, pa, pb, *scoring)

    batch = make_batch()
    info = batch.info()
    n_pairs = len(pa)
    scores = np.zeros(max(n_pairs, 1), dtype=np.int32)[:n_pairs]   # host copy of this rank's per-pair scores, refreshed every step

    stream = None
    mine = None
    shard = None
    state = {"gathered": None}
    if use_dist:
        if cpu:
            load_pkg_shard = importlib.util.spec_from_file_location("pwa_shard", os.path.join(PKG_DIR, "shard.py"))
            shard = importlib.util.module_from_spec(load_pkg_shard)
            load_pkg_shard.loader.exec_module(shard)
            mine = torch.empty(max(n_pairs, 1), dtype=torch.int32)[:n_pairs]
            batch.set_out(mine)
        else:
            from bioinformatics_algorithms_amd import shard
            # a real (non-default) torch stream: the kernels are enqueued on it through the C ABI and the
            # collective, issued under the same current stream, is ordered behind them
            tstream = torch.cuda.Stream()
            torch.cuda.set_stream(tstream)
            stream = tstream.cuda_stream
            assert stream != 0
            mine = torch.empty(max(n_pairs, 1), dtype=torch.int32, device="cuda")[:n_pairs]
            batch.set_d_scores(mine.data_ptr())   # kernels write the scores straight into the tensor the collective sends
    n_total = desc.get("pairs_total", n_pairs * world)
    per = desc.get("_per", n_pairs)

    def step():
        # one pass of the hot path: kernels, the collective (N > 1), and the per-pair results back in host memory --
        # "kernel + transfer of per-pair results included"
        batch.run(stream)
        if use_dist:   # RCCL all-gather of the per-pair int32 scores over xGMI (the path's only collective)
            if args.workload in ("c4", "hw3", "hw4"):
                state["gathered"] = shard.all_gather_scores(mine, n_total, per, dist)
            else:       # weak scaling: every rank contributes n_pairs scores of its own texts
                if state["gathered"] is None:
                    state["gathered"] = torch.empty(world * n_pairs, dtype=torch.int32, device=mine.device)
                dist.all_gather_into_tensor(state["gathered"], mine)
        batch.fetch_into(scores)   # waits for this step's kernels (their event), then D2H of this rank's scores

    def sync():
        if use_dist:
            dist.barrier()
            if not cpu:
                torch.cuda.synchronize()
        else:
            batch.last_ms()   # event-synchronises the library's stream

    for _ in range(args.warmup):
        step()
    if args.warmup == 0:
        batch.run(stream)   # make sync() well defined
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = batch.run_times(min(args.steps, 64))   # HIP events on the launch stream, per step
    cells = info["cells"]
    if use_dist:
        tdev = "cpu" if cpu else "cuda"
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([float(cells)], dtype=torch.float64, device=tdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        cells_total = float(c.item())
        # every rank holds the same gathered vector: compare a digest of it across ranks (sum and a position-weighted sum)
        g = state["gathered"].to(torch.int64)
        dig = torch.stack([g.sum(), (g * (torch.arange(g.numel(), device=g.device) % 1009 + 1)).sum()]).to(tdev)
        digs = [torch.empty_like(dig) for _ in range(world)]
        dist.all_gather(digs, dig)
        gathered_same = all(bool((d == digs[0]).all().item()) for d in digs)
        gathered_sum = int(digs[0][0].item())
    else:
        cells_total = float(cells)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    gcups = cells_total * args.steps / elapsed / 1e9
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    kern = info["kernel"]
    kern0 = kern.split(" + ")[0]   # a split batch names both engines' kernels; the first one is the strip kernel, whose instruction mix is priced
    key = ",".join(kern0[kern0.index("<") + 1:-1].split(",")[1:]) if "<" in kern0 else ""
    ops = VALU_PER_CELL.get(key)
    padded = info["padded_cells"] or cells
    alg_bytes = sum(len(s) for s in seqs) + 4 * n_pairs            # inputs once + 4 B per pair
    kernel_gcups = cells / (k_ms * 1e-3) / 1e9
    roofline = {
        # int32 VALU issue bound: the scores-only pass moves ~5e-6 B/cell over HBM, so the
        # contract's "hbm" / "mfma" bounds do not describe it; the HBM view is kept beside it under "hbm"
        "bound": "valu",
        "kernel": kern,
        "achieved": (padded / (k_ms * 1e-3)) * ops / 1e12 if ops else None,
        "peak": VALU_PEAK_TOPS,
        "peak_source": "MI355X_MICROARCH.md: 4 SIMD-32 per CU, wave64 VALU over 2 cycles: 256 x 4 x 32 lanes x 2.4 GHz",
        "unit": "Tiop/s",
        "frac": ((padded / (k_ms * 1e-3)) * ops / 1e12) / VALU_PEAK_TOPS if ops else None,
        "valu_ops_per_cell": ops,
        "issue_model": ({"cycles_per_cell_model": ISSUE_CYCLES_PER_CELL[key],
                         "cycles_per_cell_measured": (k_ms * 1e-3) * 2.4e9 * 1024 / (padded / 64.0),
                         "ratio": ISSUE_CYCLES_PER_CELL[key] / ((k_ms * 1e-3) * 2.4e9 * 1024 / (padded / 64.0)),
                         "note": "model = sum over the cell's instructions of their measured per-class issue cost (full-rate "
                                 "class ~2.7-3.2 cycles, half-rate class ~4.45; >= 2 waves per SIMD) at the NOMINAL 2.4 GHz; "
                                 "a ratio near 1 says the loop issues at the per-class rates, it is not a roofline fraction",
                         "source": "profiles/r01_valu_class_microbench.txt, profiles/r02_valu_issue_microbench.txt"}
                        if key in ISSUE_CYCLES_PER_CELL else None),
        "kernel_ms": k_ms,
        "kernel_gcups": kernel_gcups,
        "padded_cells_per_launch": padded,
        "traffic": None,
        "hbm": {"algorithmic_bytes_per_launch": alg_bytes, "achieved_GBs": alg_bytes / (k_ms * 1e-3) / 1e9,
                "peak_GBs": HBM_PEAK_GBS, "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "notional_4B_per_cell_frac": kernel_gcups * 4 / HBM_PEAK_GBS,
                "note": "scores-only pass is not HBM-bound; the notional figure prices the int32 score band "
                        "as if it were written (north-star accounting), it is NOT traffic"},
    }
    if not (args.workload == "c3i" and mode == "nw"):
        attach_traffic(roofline, args.workload)

    line = {
        "metric": "GCUPS (billion DP cells/s) SW linear-gap, 1/2/4/8xMI355X; bit-exact vs hw2.cpp"
                  if args.workload in ("c3", "c3i", "long") and mode == "sw" else ("GCUPS (billion DP cells/s) affine-gap all-pairs score pass; bit-exact vs hw3.cpp"
                                                 if args.workload == "hw3" else
                                                 "GCUPS (billion DP cells/s) NW + traceback-derived distance, all pairs; bit-exact vs hw4.cpp"
                                                 if args.workload == "hw4" else
                                                 "GCUPS (billion DP cells/s) NW linear-gap all-pairs; bit-exact vs hw2.cpp"),
        "value": gcups, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {k: v for k, v in desc.items() if not k.startswith("_")},
        "step_includes": "kernels" + (" + all-gather of the per-pair scores" if use_dist else "") + " + D2H of this rank's per-pair scores",
        "roofline": roofline,
    }
    if dist_info is not None:
        line["dist"] = dict(dist_info, gathered_identical_on_all_ranks=gathered_same, gathered_score_sum=gathered_sum)
        if not gathered_same:
            line["invalid"] = "the gathered score vectors differ between ranks"
    if args.small:
        line["invalid"] = "reduced sizes (--small): functional check only"
    if cpu:
        line["invalid"] = "CPU rehearsal of the launcher / sharding / collective (gloo + oracle): not a measurement"

    line["checksum"] = int(np.asarray(scores, dtype=np.int64).sum())
    want_sum = EXPECTED_CHECKSUM.get(args.workload) if (mode == "sw" and not args.small and not cpu) else None
    if want_sum is not None:
        line["checksum_expected"] = want_sum
        if line["checksum"] != want_sum:
            line["invalid"] = "checksum of rank 0's scores is %d, expected %d" % (line["checksum"], want_sum)

    if world == 1 and not args.no_cpu_baseline and not cpu:
        # verification + CPU baseline on the SAME sample: 2048 pairs drawn over the whole pair list (all patterns, all texts), ~13 s of CPU
        rs = np.random.RandomState(481)
        pick = np.sort(rs.choice(n_pairs, size=min(n_pairs, 2048), replace=False))
        pairs = list(zip(pa[pick].tolist(), pb[pick].tolist()))
        base, ref_scores = cpu_baseline(mode, pairs, seqs, scoring)
        line["cpu_baseline"] = base
        ok = all(int(scores[pick[k]]) == ref_scores[k] for k in range(len(ref_scores)))
        line["verified_vs_cpu"] = {"pairs": len(ref_scores), "distinct_patterns": int(len(set(pa[pick[:len(ref_scores)]].tolist()))),
                                   "distinct_texts": int(len(set(pb[pick[:len(ref_scores)]].tolist()))), "bit_exact": bool(ok)}
        if not ok:
            line["invalid"] = "GPU scores differ from the CPU baseline"
    if world == 1 and not cpu:
        # whole host call on HOST buffers (never `value`): sequence upload over PCIe, host-side wave-task
        # scheduling, kernel, score download
        batch.close()   # a caller in steady state: the context hands the previous batch's hand-off workspace to the next one
        packed = pkg.pack_sequences(seqs) if hasattr(pkg, "pack_sequences") else None
        t1 = time.perf_counter()
        b2 = make_batch(packed)
        t2 = time.perf_counter()
        b2.run()
        b2.fetch_into(scores)
        t3 = time.perf_counter()
        b2.close()
        dt = time.perf_counter() - t1
        line["host_call_inclusive"] = {"gcups": cells / dt / 1e9, "ms": dt * 1e3,
                                       "ms_create": (t2 - t1) * 1e3, "ms_run_fetch": (t3 - t2) * 1e3, "ms_destroy": (time.perf_counter() - t3) * 1e3,
                                       "what": "pwa_batch_create (PCIe upload + host scheduling of the pair list) + run + fetch + destroy, on host buffers"}
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if "invalid" in line and not (args.small or cpu):
        sys.exit(3)   # a full-size line that failed its own checks must not look like a result


def bench_global_batch(args, pkg, ctx, rank=0, world=1, dist=None, torch=None, dist_info=None, cpu=False):
    """The `-g` shape: FULL alignments (fill + traceback band + walk) of many index-paired short patterns x long
    texts (hw2.cpp:328-338 with global = true).  N > 1: the pair list of 4096 N pairs is dealt in contiguous blocks of
    4096 (weak scaling: every rank its own patterns), one all-gather of the per-pair scores and op counts per step."""
    n_pairs = (8 if cpu else 256) if args.small else (args.pairs or 4096)
    tlen, n_txt = (200, 4) if cpu else (10000, 256)
    bands = args.workload == "gb"   # gb: Smith-Waterman with the int32 score band ALSO written (5 B/cell)
    mode = "sw" if bands else "nw"
    if cpu:
        ctx = rehearsal().RehearsalContext()
    ctx.set_score_band(bands)
    plen = 40 if cpu else args.plen
    pats = [gen(1, 0, rank * n_pairs + p, plen) for p in range(n_pairs)]   # rank r: block r of the global pair list
    txts = [gen(1, 1, t, tlen) for t in range(n_txt)]
    seqs = pats + txts
    # host buffers as a compiled host holds them (hw2_amd: the FASTA reader's blob + offsets, the pair list, one op buffer that is
    # reused): built once, outside the timed region -- a step is pwa_align_batch on them, results back in host memory
    packed = (None, None, seqs) if cpu else pkg.pack_sequences(seqs)
    pa = np.arange(n_pairs, dtype=np.uint32)
    pb = (n_pairs + (np.arange(n_pairs) % n_txt)).astype(np.uint32)
    out = None
    state = {"gathered": None}
    dev = "cpu" if cpu else "cuda"

    def step(out):
        out = ctx.align_batch_arrays(mode, packed, pa, pb, 1, -1, -1, out)
        if dist is not None:   # the path's only exchange: per-pair scores + op counts of every rank's block (hw2.cpp:342-357 selects over all pairs)
            mine = torch.from_numpy(np.concatenate([out["scores"][:n_pairs].astype(np.int32), out["n_ops"][:n_pairs].astype(np.int32)])).to(dev)
            if state["gathered"] is None:   # rank r's block: [scores of its n_pairs pairs | their op counts]
                state["gathered"] = torch.empty(world * 2 * n_pairs, dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(state["gathered"], mine)
        return out

    for _ in range(args.warmup):
        out = step(out)
    if dist is not None:
        dist.barrier()
        if not cpu:
            torch.cuda.synchronize()
    t0 = time.perf_counter()
    fill, tb = [], []
    for _ in range(args.steps):
        out = step(out)
        st = ctx.align_stats()
        fill.append(st["fill_ms"])
        tb.append(st["traceback_ms"])
    if dist is not None:
        dist.barrier()
        if not cpu:
            torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gathered_same, gathered_sum = None, None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        g = state["gathered"].to(torch.int64).flatten()
        dig = torch.stack([g.sum(), (g * (torch.arange(g.numel(), device=g.device) % 1009 + 1)).sum()])
        digs = [torch.empty_like(dig) for _ in range(world)]
        dist.all_gather(digs, dig)
        gathered_same = all(bool((d == digs[0]).all().item()) for d in digs)
        gathered_sum = int(state["gathered"].view(world, 2, n_pairs)[:, 0, :].to(torch.int64).sum().item())
        if rank != 0:
            dist.destroy_process_group()
            return
    res = [dict(score=int(out["scores"][k]), ops=out["ops"][int(out["ops_off"][k]):int(out["ops_off"][k]) + int(out["n_ops"][k])])
           for k in range(n_pairs)]
    # a seeded sample of the batch against the CPU oracle, op for op (after the timed region)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    rs = np.random.RandomState(7)
    verified = True
    for k in rs.choice(n_pairs, size=min(n_pairs, 24), replace=False):
        want = O.align(mode, seqs[int(pa[k])], seqs[int(pb[k])], 1, -1, -1, compact=True)
        verified = verified and want["score"] == res[k]["score"] and want["ops"] == res[k]["ops"].tobytes()
    cells = float(n_pairs) * plen * tlen
    st = ctx.align_stats()
    k_ms = float(np.mean(fill))
    # the algorithmic bytes of a fill are 1 B (traceback code) -- or 5 B with the int32 score band -- per CELL of the
    # matrices, plus the inputs once; what the kernels actually write (band padding included) is reported next to it, never as `achieved`
    alg_bytes = cells * (5 if bands else 1) + sum(len(x) for x in seqs)
    written = float(st["band_bytes"])
    rl = next((r for r in (4, 6, 8, 10, 12, 16) if plen <= 16 * r), None)
    stripes_only = os.environ.get("PWA_TB_ENGINE") == "0"
    if rl and not stripes_only:       # four pairs per wave
        kern = "mini_fill_kernel<RL=%d,%s>" % (rl, "SW,SBAND" if bands else "NW,GAP0")
    elif plen <= 1024 and n_pairs >= 256 and not stripes_only:   # one pair per wave
        kern = "mini_fill_kernel<RL=%d,LN=64,%s>" % (6 if plen <= 384 else 8 if plen <= 512 else 12 if plen <= 768 else 16, "SW,SBAND" if bands else "NW,GAP0")
    else:
        kern = "pair_fill_kernel<RL=%d,W=%d,%s,TB%s,PERM%s>" % (4 if 128 < plen <= 256 or plen > 32768 else 2, 1 if plen <= 256 else 4, mode.upper(),
                                                                 ",SBAND" if bands else "", "" if bands else ",GAP0")
    line = {
        "metric": "GCUPS (billion DP cells/s) %s full alignments of a pair batch (-g shape); bit-exact vs hw2.cpp" % mode.upper(),
        "value": cells * world * args.steps / elapsed / 1e9, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": ("gb: %d pairs %d x %d per GPU, SW fill writing int32 score band + traceback band (5 B/cell) + walk"
                                if bands else "g: %d pairs %d x %d per GPU, NW fill + traceback band + walk, host buffers in and ops out") % (n_pairs, plen, tlen),
                   "scoring": [1, -1, -1]},
        "step_includes": "H2D of the block's sequences, fill, walk, D2H of scores and op lists" + (" + all-gather of per-pair scores and op counts" if dist is not None else ""),
        "roofline": {"bound": "hbm", "kernel": kern,
                     "launches_per_step": "the batch runs as ranges of <= 6 GiB (10 GiB with the score band) of band: achieved / kernel_ms are sums over the step's fill launches",
                     "achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_step": alg_bytes,
                     "algorithmic_bytes_rule": "SURVEY.md 8(d): n * m * %d B per pair (the reference's own %s) + the inputs once" % (5 if bands else 1, "int + char matrices, hw2.cpp:193-194" if bands else "char matrix, hw2.cpp:120"),
                     "written_bytes_per_step": written, "written_over_algorithmic": written / alg_bytes if alg_bytes else None,
                     "written_bytes_note": "band bytes the fill kernels store, padding rows / steps of the band geometry included (from the library's own layout; PMC WRITE_SIZE: profiles/traffic_%s.json)" % args.workload,
                     "traffic": None, "kernel_ms": k_ms, "traceback_ms": float(np.mean(tb)), "kernel_gcups": cells / (k_ms * 1e-3) / 1e9},
        "result": {"score_sum": int(sum(r["score"] for r in res)), "ops_total": int(sum(len(r["ops"]) for r in res))},
        "verified_vs_cpu": {"pairs": int(min(n_pairs, 24)), "what": "score and op list against the CPU oracle", "bit_exact": bool(verified)},
    }
    if dist_info is not None:
        line["dist"] = dict(dist_info, gathered_identical_on_all_ranks=gathered_same, gathered_score_sum=gathered_sum)
        if not gathered_same:
            line["invalid"] = "the gathered vectors differ between ranks"
    attach_traffic(line["roofline"], args.workload)
    if not verified:
        line["invalid"] = "GPU alignments differ from the CPU oracle"
    if args.small:
        line["invalid"] = "reduced sizes (--small): functional check only"
    if cpu:
        line["invalid"] = "CPU rehearsal of the launcher / sharding / collective (gloo + oracle): not a measurement"
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def bench_single_pair(args, pkg, ctx, rank, world, dist, torch):
    """c2 / c5: one big pair, full fill + traceback band in HBM.  Does not shard: replicas only."""
    if args.workload in ("c2", "c2b"):
        n = m = 2000 if args.small else 10000
        mode, label = "sw", "c2: SW 1 pair %dx%d, traceback band in HBM" % (n, m)
        if args.workload == "c2b":   # SURVEY.md 8(d) row C2 accounting: 4 B int32 score band + 1 B traceback band
            ctx.set_score_band(True)
            label = "c2b: SW 1 pair %dx%d, int32 score band + traceback band in HBM (5 B/cell)" % (n, m)
    else:
        n = m = 5000 if args.small else 100000
        mode, label = "nw", "c5: NW 1 pair %dx%d, traceback band in HBM" % (n, m)
    p, t = gen(1, 0, 0, n), gen(1, 1, 0, m)
    for _ in range(args.warmup):
        ctx.align(mode, p, t, 1, -1, -1, raw=True)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    fill, tb = [], []
    for _ in range(args.steps):
        r = ctx.align(mode, p, t, 1, -1, -1, raw=True)
        st = ctx.align_stats()
        fill.append(st["fill_ms"])
        tb.append(st["traceback_ms"])
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        dist.destroy_process_group()
        return
    st = ctx.align_stats()
    cells = float(n) * m
    k_ms = float(np.mean(fill))
    written = float(st["band_bytes"])
    band = cells * (5 if args.workload == "c2b" else 1) + n + m   # n * m * (1 | 5) B + the inputs; `written` counts the band's padding too
    line = {
        "metric": "GCUPS (billion DP cells/s) single pair with traceback; bit-exact vs hw2.cpp",
        "value": cells * world * args.steps / elapsed / 1e9, "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": label + " (replicas only: a single pair does not shard)", "scoring": [1, -1, -1],
                   "includes": "H2D of the pair, fill, traceback walk, D2H of the ops"},
        "roofline": {"bound": "hbm", "kernel": ("pair_fill_kernel<RL=2,W=4,SW,TB,PERM>" if mode == "sw" else "pair_fill_kernel<RL=4,W=4,NW,TB,PERM,GAP0>"),
                     "achieved": band / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": band / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": band, "written_bytes_per_launch": written, "written_over_algorithmic": written / band,
                     "kernel_ms": k_ms, "traceback_ms": float(np.mean(tb)),
                     "kernel_gcups": cells / (k_ms * 1e-3) / 1e9},
        "result": {"score": r["score"], "n_ops": len(r["ops"])},
    }
    attach_traffic(line["roofline"], args.workload)
    if not args.small:
        # the alignment itself against what the UNMODIFIED reference produced for this very pair (tests/golden/kat.json: C2,
        # kat_c5.json: C5 -- one 50 GB run of hw2.cpp): score, CIGAR and MD:Z by sha256, after the timed region
        import hashlib
        fmt = pkg.format_alignment(p, t, r["ops"], r["end"])
        gold = None
        for name in ("kat_c5.json", "kat.json"):
            with open(os.path.join(ROOT, "tests", "golden", name)) as f:
                for rec in json.load(f):
                    if rec["mode"] == mode and rec["gen_p"] == [1, 0, 0, n] and rec["gen_t"] == [1, 1, 0, m] and rec["scoring"] == [1, -1, -1]:
                        gold = rec
        if gold is not None:
            ok = (r["score"] == gold["score"] and hashlib.sha256(fmt["cigar"]).hexdigest() == gold["cigar_sha256"]
                  and hashlib.sha256(fmt["mdz"]).hexdigest() == gold["mdz_sha256"] and fmt["overlap"] == gold["overlap"])
            line["verified_vs_reference"] = {"what": "score, overlap, sha256 of CIGAR and MD:Z of the unmodified hw2.cpp for this pair (committed fixture)",
                                             "cigar_chars": len(fmt["cigar"]), "bit_exact": bool(ok)}
            if not ok:
                line["invalid"] = "the alignment differs from the reference's"
        if args.workload in ("c2", "c2b") and not args.no_cpu_baseline and rank == 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            kind = "reference" if O.have_ref() else "port"
            t1 = time.perf_counter()
            w = (O.ref_align if kind == "reference" else O.align)(mode, p, t, 1, -1, -1)
            dt = time.perf_counter() - t1
            line["cpu_baseline"] = {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": kind,
                                    "sample": "this pair in full (%.2f s), int + char matrices as hw2.cpp, g++ -O2" % dt}
            if w["score"] != r["score"] or w["cigar"] != fmt["cigar"]:
                line["invalid"] = "GPU alignment differs from the CPU baseline"
    if args.small:
        line["invalid"] = "reduced sizes (--small): functional check only"
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
