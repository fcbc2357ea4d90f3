// fasta.cpp -- FASTA ingest for the pair engine: readFasta (hw2.cpp:25-57) semantics on large
// inputs, straight into the layout the C ABI consumes (one byte blob + offsets), no per-record std::string.
//
// Semantics kept verbatim (hw2.cpp:33-54): lines end at '\n' (a last line without one still counts); trailing
// '\r' / isspace bytes are stripped (35-39); lines empty after that are skipped (40-42) and do NOT end a record;
// a line whose first byte is '>' ends the current record if it has any bytes (43-47) -- its text is dropped;
// every other line is appended (49); a non-empty record at end of file is kept (52-54).  So records with an
// empty body vanish, and bytes before the first header form a record of their own.
//
// Why this parallelises: the blob is simply every kept line, trimmed, in file order; headers only mark where
// it is cut.  Pass 1 (one thread per file chunk, chunks cut at line starts) counts, per chunk, the kept bytes
// before its first header and after each header.  A sequential merge over the chunk summaries (microseconds)
// turns these into record offsets and gives every chunk its output position.  Pass 2 copies the lines.
// The file is mmap'ed (read() fallback for pipes), so nothing is buffered twice; the blob is an anonymous
// mapping that the copying threads touch first (no zero-fill pass).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/pwalign.h"

struct pwa_fasta {
    // The blob is an anonymous mapping sized for the worst case (every input byte kept), never zero-filled by us:
    // the parser threads touch their own parts first (parallel page faults, huge pages where the kernel allows),
    // and pages past the last kept byte are never touched at all.
    uint8_t* bytes = nullptr;
    size_t mapped = 0;
    uint64_t n_bytes = 0;
    std::vector<uint64_t> off;        // n_seq + 1
    std::vector<uint32_t> first_seq;  // n_paths + 1
    ~pwa_fasta() {
        if (bytes) munmap(bytes, mapped);
    }
    bool reserve(size_t n) {
        mapped = std::max<size_t>(n, 1);
        void* m = mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) {
            mapped = 0;
            return false;
        }
#ifdef MADV_HUGEPAGE
        (void)madvise(m, mapped, MADV_HUGEPAGE);
#endif
        bytes = static_cast<uint8_t*>(m);
        return true;
    }
};

namespace {

inline bool ref_isspace(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }   // "C" locale isspace

struct Mapped {   // whole file in memory: mmap, or a heap copy when the path is not mappable
    const uint8_t* p = nullptr;
    size_t n = 0;
    void* map = nullptr;
    std::vector<uint8_t> heap;
    Mapped() = default;
    Mapped(const Mapped&) = delete;
    Mapped& operator=(const Mapped&) = delete;
    ~Mapped() {
        if (map) munmap(map, n);
    }
    bool open(const char* path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || S_ISDIR(st.st_mode)) {
            ::close(fd);
            return false;
        }
        if (S_ISREG(st.st_mode) && st.st_size > 0) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);   // one kernel pass, no per-page traps
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                map = m;
                p = static_cast<const uint8_t*>(m);
                n = (size_t)st.st_size;
                ::close(fd);
                return true;
            }
        }
        uint8_t buf[1 << 16];   // pipes, /dev/stdin, empty files
        for (;;) {
            const ssize_t r = ::read(fd, buf, sizeof buf);
            if (r < 0) {
                ::close(fd);
                return false;
            }
            if (r == 0) break;
            heap.insert(heap.end(), buf, buf + r);
        }
        ::close(fd);
        p = heap.data();
        n = heap.size();
        return true;
    }
};

struct ChunkSummary {
    size_t begin = 0, end = 0;            // byte range, begin at a line start
    uint64_t before_first = 0;            // kept bytes ahead of the chunk's first header
    std::vector<uint64_t> after_header;   // kept bytes after each header (up to the next header / chunk end)
    uint64_t kept = 0;                    // all kept bytes of the chunk
    uint64_t out_pos = 0;                 // where the chunk's first kept byte goes in the blob
};

// Calls line(begin, trimmed_end) for every line of [b, e) that is non-empty after trimming.
template <class F>
inline void for_each_line(const uint8_t* p, size_t b, size_t e, F&& line) {
    size_t pos = b;
    while (pos < e) {
        const void* nl = std::memchr(p + pos, '\n', e - pos);
        const size_t le = nl ? (size_t)(static_cast<const uint8_t*>(nl) - p) : e;
        size_t t = le;
        while (t > pos && ref_isspace(p[t - 1])) --t;   // '\r' is one of them (hw2.cpp:36)
        if (t > pos) line(pos, t);
        pos = le + 1;
    }
}

void summarize(const uint8_t* p, ChunkSummary& c) {
    uint64_t cur = 0;
    bool seen_header = false;
    for_each_line(p, c.begin, c.end, [&](size_t b, size_t t) {
        if (p[b] == '>') {
            if (seen_header) c.after_header.push_back(cur);
            else c.before_first = cur;
            seen_header = true;
            cur = 0;
        } else {
            cur += t - b;
            c.kept += t - b;
        }
    });
    if (seen_header) c.after_header.push_back(cur);
    else c.before_first = cur;
}

void copy_lines(const uint8_t* p, const ChunkSummary& c, uint8_t* out) {
    uint8_t* o = out + c.out_pos;
    for_each_line(p, c.begin, c.end, [&](size_t b, size_t t) {
        if (p[b] != '>') {
            std::memcpy(o, p + b, t - b);
            o += t - b;
        }
    });
}

int parse_one(const Mapped& mf, int n_threads, pwa_fasta& f) {
    const uint8_t* p = mf.p;
    const size_t n = mf.n;
    int T = n_threads > 0 ? n_threads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    size_t min_chunk = 4u << 20;   // a thread per >= 4 MiB of file; PWA_FASTA_MIN_CHUNK: tests force chunking of tiny files
    if (const char* e = std::getenv("PWA_FASTA_MIN_CHUNK")) min_chunk = std::max<size_t>(1, (size_t)std::atoll(e));
    T = (int)std::max<size_t>(1, std::min<size_t>({(size_t)T, n / min_chunk + 1, n}));   // T <= n: every cut is at byte >= 1
    std::vector<ChunkSummary> ch((size_t)T);
    for (int c = 0; c < T; ++c) {   // cut at line starts
        size_t b = n / (size_t)T * (size_t)c;
        if (c > 0) {
            const void* nl = std::memchr(p + b - 1, '\n', n - (b - 1));
            b = nl ? (size_t)(static_cast<const uint8_t*>(nl) - p) + 1 : n;
        }
        ch[(size_t)c].begin = b;
        if (c > 0) ch[(size_t)c - 1].end = b;
    }
    ch[(size_t)T - 1].end = n;
    auto run = [&](auto&& fn) {
        std::vector<std::thread> th;
        for (int c = 1; c < T; ++c) th.emplace_back([&, c] { fn(ch[(size_t)c]); });
        fn(ch[0]);
        for (auto& t : th) t.join();
    };
    run([&](ChunkSummary& c) { summarize(p, c); });

    // sequential merge: record cuts and output positions
    const uint64_t base = f.n_bytes;
    uint64_t pos = base, open_bytes = 0, open_start = base;
    for (auto& c : ch) {
        c.out_pos = pos;
        open_bytes += c.before_first;
        uint64_t at = pos + c.before_first;
        for (const uint64_t g : c.after_header) {   // a header: close the open record if it holds anything (hw2.cpp:44-47)
            if (open_bytes > 0) {
                if (f.off.size() >= 0xfffffffeull) return PWA_E_CAPACITY;
                f.off.push_back(open_start + open_bytes);
            }
            open_start = at;
            open_bytes = g;
            at += g;
        }
        pos += c.kept;
    }
    if (open_bytes > 0) {   // hw2.cpp:52-54
        if (f.off.size() >= 0xfffffffeull) return PWA_E_CAPACITY;
        f.off.push_back(open_start + open_bytes);
    }
    f.n_bytes = pos;   // <= base + file size: fits the reservation
    uint8_t* out = f.bytes;
    run([&](ChunkSummary& c) { copy_lines(p, c, out); });
    return PWA_OK;
}

}  // namespace

extern "C" {

int pwa_fasta_read(const char* const* paths, int n_paths, int n_threads, pwa_fasta** out, int* failed_path) try {
    if (failed_path) *failed_path = -1;
    if (!out || n_paths < 0 || (n_paths && !paths)) return PWA_E_INVALID;
    *out = nullptr;
    pwa_fasta* f = new (std::nothrow) pwa_fasta();
    if (!f) return PWA_E_NOMEM;
    struct Guard {
        pwa_fasta* f;
        ~Guard() { delete f; }
    } guard{f};
    f->off.push_back(0);
    f->first_seq.push_back(0);
    // open every file first: the reference stops at the first one it cannot open (hw2.cpp:28-31), before any work
    std::vector<Mapped> files((size_t)n_paths);
    size_t total = 0;
    for (int i = 0; i < n_paths; ++i) {
        if (!paths[i] || !files[(size_t)i].open(paths[i])) {
            if (failed_path) *failed_path = i;
            return paths[i] ? PWA_E_IO : PWA_E_INVALID;
        }
        total += files[(size_t)i].n;
    }
    if (!f->reserve(total)) return PWA_E_NOMEM;
    for (int i = 0; i < n_paths; ++i) {
        const int rc = parse_one(files[(size_t)i], n_threads, *f);
        if (rc != PWA_OK) {
            if (failed_path) *failed_path = i;
            return rc;
        }
        f->first_seq.push_back((uint32_t)(f->off.size() - 1));
    }
    guard.f = nullptr;
    *out = f;
    return PWA_OK;
} catch (const std::bad_alloc&) {
    return PWA_E_NOMEM;
} catch (...) {
    return PWA_E_IO;   // nothing may propagate across the C ABI
}

uint32_t pwa_fasta_n_seq(const pwa_fasta* f) { return f ? (uint32_t)(f->off.size() - 1) : 0; }
const uint8_t* pwa_fasta_bytes(const pwa_fasta* f) { return f ? f->bytes : nullptr; }
const uint64_t* pwa_fasta_offsets(const pwa_fasta* f) { return f ? f->off.data() : nullptr; }
const uint32_t* pwa_fasta_first_seq(const pwa_fasta* f) { return f ? f->first_seq.data() : nullptr; }
void pwa_fasta_free(pwa_fasta* f) { delete f; }

}  // extern "C"
