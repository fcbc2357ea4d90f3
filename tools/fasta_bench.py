#!/usr/bin/env python3
"""Host-side throughput of FASTA ingest: pwa_fasta_read (mmap + parallel parse into blob + offsets)
against the reference's own readFasta (hw2.cpp:25-57, compiled from the unmodified source into oracle/_ref) on the
same synthetic file.  No GPU work.  usage: tools/fasta_bench.py [MiB] [dir]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    d = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
    path = os.path.join(d, "fasta_bench_%d.fasta" % os.getpid())
    rec = b"".join(bench.gen(1, 1, 0, 9920)[i:i + 80] + b"\n" for i in range(0, 9920, 80))   # 124 lines of 80
    n_rec = mib * (1 << 20) // (len(rec) + 12)
    with open(path, "wb") as f:
        for r in range(n_rec):
            f.write(b">record%05d\n" % (r % 100000))
            f.write(rec)
    size = os.path.getsize(path)
    pkg = bench.load_pkg()
    L = pkg.lib()
    print("file: %.1f MiB, %d records of 9920 bases in 80-column lines; host cores: %d" % (size / 2**20, n_rec, os.cpu_count()))
    try:
        arr = (C.c_char_p * 1)(os.fsencode(path))
        for threads in (1, 2, 4, 8, 16):
            best = 1e9
            for _ in range(3):
                h, bad = C.c_void_p(), C.c_int(-1)
                t0 = time.perf_counter()
                rc = L.pwa_fasta_read(arr, 1, threads, C.byref(h), C.byref(bad))
                dt = time.perf_counter() - t0
                assert rc == 0 and L.pwa_fasta_n_seq(h) == n_rec
                L.pwa_fasta_free(h)
                best = min(best, dt)
            print("pwa_fasta_read   %2d threads: %7.1f ms  %6.2f GB/s" % (threads, best * 1e3, size / best / 1e9))
        if O.have_ref():
            lib = O.ref()
            t0 = time.perf_counter()
            f = lib.ref_read_fasta(os.fsencode(path))
            dt = time.perf_counter() - t0
            assert f.contents.count == n_rec
            lib.ref_free_fasta(f)
            print("reference readFasta (hw2.cpp:25-57, g++ -O2), 1 thread, incl. packing its vector<string>: %7.1f ms  %6.2f GB/s"
                  % (dt * 1e3, size / dt / 1e9))
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
