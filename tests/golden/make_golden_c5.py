#!/usr/bin/env python3
"""Generate tests/golden/kat_c5.json: the known answer of BASELINE.json config C5 (NW 100 000 x 100 000, 1/-1/-1) and
of its SW twin from the UNMODIFIED reference functions (oracle/_ref/libhw2_ref.so = g++ -std=c++17 -O2 of
/root/reference/Local_Global_Alignment/hw2.cpp behind oracle/ref_shim.cpp).  Dev container only.

The reference keeps an int and a char matrix of (n+1)(m+1) cells (hw2.cpp:119-120): 50 GB of RSS and about five
minutes per run, page-fault bound -- so this is a separate script, run once, not part of
make_golden.py.  Nothing else that needs memory may run beside it (the container has 62 GB and no swap).

What is recorded is DATA: the generator arguments of the two sequences and the five AlignmentResult fields the
reference returns (hw2.cpp:17-23) plus overlapLongestExactMatch (267-278): numbers in clear, strings as length + sha256.

Usage:  python tests/golden/make_golden_c5.py [--n 100000] [--modes g,l]
"""
import argparse
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def sha(b):
    return hashlib.sha256(b).hexdigest()


def run_ref(mode, p, t, scoring):
    """One call of the reference's own function (hw2.cpp:118 / 192) through oracle/_ref/libhw2_ref.so."""
    t0 = time.time()
    r = O.ref_align(mode, p, t, *scoring)
    return dict(score=r["score"], overlap=r["overlap"], aligned_len=len(r["aligned_pattern"]),
                aligned_pattern_sha256=sha(r["aligned_pattern"]), aligned_reference_sha256=sha(r["aligned_reference"]),
                cigar_len=len(r["cigar"]), cigar_sha256=sha(r["cigar"]), cigar_head=r["cigar"][:64].decode(),
                mdz_len=len(r["mdz"]), mdz_sha256=sha(r["mdz"]), mdz_head=r["mdz"][:64].decode(),
                reference_seconds=round(time.time() - t0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--modes", default="g,l")
    ap.add_argument("--out", default=os.path.join(HERE, "kat_c5.json"))
    args = ap.parse_args()
    assert O.have_ref(), "run `make -C oracle ref` first (needs /root/reference)"
    n = args.n
    gen_p, gen_t, scoring = (1, 0, 0, n), (1, 1, 0, n), (1, -1, -1)
    p, t = O.gen(*gen_p), O.gen(*gen_t)
    out = []
    if os.path.exists(args.out):
        out = [r for r in json.load(open(args.out)) if not (r["gen_p"] == list(gen_p) and r["flag"] in ["-" + m for m in args.modes.split(",")])]
    for m in args.modes.split(","):
        rec = dict(flag="-" + m, mode="nw" if m == "g" else "sw", gen_p=list(gen_p), gen_t=list(gen_t), scoring=list(scoring))
        rec.update(run_ref(rec["mode"], p, t, scoring))
        print(rec, flush=True)
        out.append(rec)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
