#include "er_generator.hpp"

#include <parallel/algorithm>

#include <omp.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>

#include "spmat_local.hpp"

namespace hnh {

std::vector<uint64_t> erdos_renyi_keys(uint64_t m, uint64_t n, uint64_t draws, uint64_t seed) {
    std::vector<uint64_t> keys(draws);
    const uint64_t G = 0x9E3779B97F4A7C15ull;
#pragma omp parallel for
    for (uint64_t k = 0; k < draws; k++) {
        const uint64_t base = seed + (2 * k) * G;
        const uint64_t r = splitmix64(base) % m, c = splitmix64(base + G) % n;
        keys[k] = r * n + c;
    }
    __gnu_parallel::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    return keys;
}

std::vector<uint64_t> rmat_keys(int logm, uint64_t edges, double a, double b, double c, uint64_t seed, bool scramble) {
    std::vector<uint64_t> keys(edges);
    const uint64_t G = 0x9E3779B97F4A7C15ull, n = 1ull << logm, mask = n - 1;
    const double ab = a + b, abc = a + b + c;
#pragma omp parallel for
    for (uint64_t k = 0; k < edges; k++) {
        uint64_t r = 0, col = 0;
        for (int l = 0; l < logm; l++) {
            const double u = (double)(splitmix64(seed + (k * (uint64_t)logm + (uint64_t)l) * G) >> 11) * 0x1.0p-53;
            const uint64_t rb = (u >= ab) ? 1 : 0;
            const uint64_t cb = (u >= a && u < ab) || (u >= abc) ? 1 : 0;
            r = (r << 1) | rb;
            col = (col << 1) | cb;
        }
        if (scramble) {
            r = (r * 0x9E3779B1ull + 0x7F4A7C15ull) & mask;
            col = (col * 0x9E3779B1ull + 0x7F4A7C15ull) & mask;
        }
        keys[k] = r * n + col;
    }
    __gnu_parallel::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    return keys;
}

std::vector<uint64_t> vertex_permutation(uint64_t n, uint64_t seed) {
    std::vector<std::pair<uint64_t, uint64_t>> order(n);
    const uint64_t G = 0x9E3779B97F4A7C15ull;
#pragma omp parallel for
    for (uint64_t v = 0; v < n; v++) order[v] = {splitmix64(seed + v * G), v};
    __gnu_parallel::sort(order.begin(), order.end());
    std::vector<uint64_t> label(n);
#pragma omp parallel for
    for (uint64_t k = 0; k < n; k++) label[order[k].second] = k;
    return label;
}

// Parses a MatrixMarket coordinate file (general / symmetric; pattern, integer or real) into 0-based tuples, mirrored
// entries of a symmetric file included, duplicates NOT yet merged.  The file is memory-mapped and its body cut at line
// boundaries into one piece per OpenMP thread, each parsed into its own vector (a SuiteSparse graph like com-Orkut is
// ~1.7 GB of text; indices are read by a plain digit loop — strtoull spends most of its time elsewhere — and values by
// strtod).  `parts` keeps the pieces in file order, so a caller can move them on without concatenating 24 bytes per tuple.
namespace {
struct MappedFile {
    const char* data = nullptr;
    size_t size = 0;
    std::string fallback;  // when mmap is not available (special files)
    int fd = -1;
    explicit MappedFile(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) fatal("Error, cannot open matrix file " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) fatal("Error, cannot stat matrix file " + path);
        size = (size_t)st.st_size;
        if (size > 0) {
            void* m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                data = static_cast<const char*>(m);
                ::madvise(m, size, MADV_SEQUENTIAL);
                return;
            }
        }
        std::ifstream in(path, std::ios::binary);
        fallback.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
        data = fallback.data();
        size = fallback.size();
    }
    ~MappedFile() {
        if (data != nullptr && fallback.empty() && size > 0) ::munmap(const_cast<char*>(data), size);
        if (fd >= 0) ::close(fd);
    }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
};

// decimal digits at p (after optional blanks); false when there are none
inline bool parse_index(const char*& p, const char* hi, uint64_t& out) {
    while (p < hi && (*p == ' ' || *p == '\t')) p++;
    const char* q = p;
    uint64_t v = 0;
    while (q < hi && (unsigned)(*q - '0') <= 9u) v = v * 10 + (uint64_t)(*q++ - '0');
    if (q == p) return false;
    out = v;
    p = q;
    return true;
}
}  // namespace

void parse_matrix_market_parts(const std::string& path, uint64_t& m, uint64_t& n, std::vector<std::vector<spcoord_t>>& parts) {
    MappedFile file(path);
    const char* text = file.data;
    const char* end = text + file.size;
    auto line_end = [&](const char* p) {
        const void* nl = p < end ? std::memchr(p, '\n', (size_t)(end - p)) : nullptr;
        return nl ? static_cast<const char*>(nl) : end;
    };
    const char* le = line_end(text);
    std::string line(text, le);
    if (line.rfind("%%MatrixMarket", 0) != 0) fatal("Error, " + path + " is not a MatrixMarket file");
    std::string lower = line;
    std::transform(lower.begin(), lower.end(), lower.begin(), ::tolower);
    if (lower.find("coordinate") == std::string::npos) fatal("Error, only coordinate MatrixMarket files are supported");
    const bool pattern = lower.find("pattern") != std::string::npos;
    const bool symmetric = lower.find("symmetric") != std::string::npos;
    // comment lines, then the size line
    uint64_t entries = 0;
    for (;;) {
        if (le >= end) fatal("Error, bad MatrixMarket size line in " + path);
        const char* start = le + 1;
        le = line_end(start);
        line.assign(start, le);
        if (line.empty() || line[0] == '%' || line == "\r") continue;
        std::istringstream hdr(line);
        if (!(hdr >> m >> n >> entries)) fatal("Error, bad MatrixMarket size line in " + path);
        break;
    }
    const char* body = le < end ? le + 1 : end;
    const int nthreads = std::max(1, omp_get_max_threads());
    parts.assign((size_t)nthreads, {});
    std::vector<int> bad((size_t)nthreads, 0);
    const uint64_t mm = m, nn = n;
#pragma omp parallel num_threads(nthreads)
    {
        const int t = omp_get_thread_num();
        const size_t len = (size_t)(end - body);
        const char* lo = body + len * (size_t)t / (size_t)nthreads;
        const char* hi = body + len * (size_t)(t + 1) / (size_t)nthreads;
        if (t > 0) {  // start at the first line that begins inside the piece
            while (lo < end && lo[-1] != '\n') lo++;
        }
        while (hi < end && hi[-1] != '\n') hi++;  // ... and finish the line that straddles its end
        std::vector<spcoord_t>& out = parts[(size_t)t];
        out.reserve((size_t)((double)(entries / (uint64_t)nthreads + 16) * (symmetric ? 2.0 : 1.0) * 1.05));
        const char* p = lo;
        while (p < hi) {
            while (p < hi && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) p++;
            if (p >= hi) break;
            if (*p == '%') {  // comment inside the body
                while (p < hi && *p != '\n') p++;
                continue;
            }
            uint64_t r = 0, c = 0;
            if (!parse_index(p, hi, r) || !parse_index(p, hi, c)) { bad[(size_t)t] = 1; break; }
            double v = 1.0;
            const void* nl = std::memchr(p, '\n', (size_t)(end - p));
            if (!pattern) {
                char* q = nullptr;
                // the value has to be on THIS line: strtod skips leading white space including line feeds, so a line without a
                // value would silently take the next line's first token (and read past the mapping on a file's last line)
                while (p < end && (*p == ' ' || *p == '\t')) p++;
                if (p >= end || *p == '\r' || *p == '\n') { bad[(size_t)t] = 1; break; }
                if (nl != nullptr) {  // strtod stops at the line feed at the latest
                    v = std::strtod(p, &q);
                    if (q == p) { bad[(size_t)t] = 1; break; }
                    p = q;
                } else {  // last line of a file that does not end in a line feed: the mapping is not NUL-terminated
                    const std::string tail(p, end);
                    v = std::strtod(tail.c_str(), &q);
                    if (q == tail.c_str()) { bad[(size_t)t] = 1; break; }
                    p += q - tail.c_str();
                }
            }
            p = nl != nullptr ? static_cast<const char*>(nl) : hi;  // ignore anything else on the line (complex files are not supported anyway)
            if (r < 1 || r > mm || c < 1 || c > nn) { bad[(size_t)t] = 2; break; }
            out.push_back({r - 1, c - 1, v});
            if (symmetric && r != c) out.push_back({c - 1, r - 1, v});
        }
    }
    size_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        if (bad[(size_t)t] == 1) fatal("Error, malformed line in MatrixMarket file " + path);
        if (bad[(size_t)t] == 2) fatal("Error, MatrixMarket index out of range in " + path);
        total += parts[(size_t)t].size();
    }
    if (!symmetric && total != entries) fatal("Error, truncated MatrixMarket file " + path);
    if (symmetric && total < entries) fatal("Error, truncated MatrixMarket file " + path);
}

void parse_matrix_market(const std::string& path, uint64_t& m, uint64_t& n, std::vector<spcoord_t>& tuples) {
    std::vector<std::vector<spcoord_t>> parts;
    parse_matrix_market_parts(path, m, n, parts);
    size_t total = 0;
    for (auto& v : parts) total += v.size();
    tuples.clear();
    tuples.reserve(total);  // (no resize: value-initialising 24 bytes per tuple first would touch every page twice)
    for (auto& v : parts) {
        tuples.insert(tuples.end(), v.begin(), v.end());
        std::vector<spcoord_t>().swap(v);
    }
}

// duplicates -> maximum (the reference reads with `maximum<double>()`, SpmatLocal.hpp:487); host version
void merge_duplicates_max(std::vector<spcoord_t>& tuples) {
    __gnu_parallel::sort(tuples.begin(), tuples.end(), [](const spcoord_t& a, const spcoord_t& b) { return row_major(a, b); });
    size_t out = 0;
    for (size_t e = 0; e < tuples.size(); e++) {
        if (out > 0 && tuples[out - 1].r == tuples[e].r && tuples[out - 1].c == tuples[e].c)
            tuples[out - 1].value = std::max(tuples[out - 1].value, tuples[e].value);
        else
            tuples[out++] = tuples[e];
    }
    tuples.resize(out);
}

void read_matrix_market(const std::string& path, uint64_t& m, uint64_t& n, std::vector<spcoord_t>& tuples) {
    parse_matrix_market(path, m, n, tuples);
    merge_duplicates_max(tuples);
}

}  // namespace hnh

// SpmatLocal::loadTuples — same signature as SpmatLocal.hpp:467-470.  Synthetic: M = N = 2^logM,
// M * nnz_per_row draws, seed 12345 (env HNH_ER_SEED overrides), values 1.0.  Every rank evaluates the
// same generator / reads the same file and keeps the strided slice {e : e % p == rank}.
void SpmatLocal::loadTuples(bool readFromFile, int logM, int nnz_per_row, std::string filename) {
    const int p = world->size, rank = world->rank;
    coords.clear();
    if (readFromFile) {
        std::vector<spcoord_t> all;
        uint64_t m = 0, n = 0;
        if (device_setup()) {
            // every rank parses the file (in parallel on its host cores), then the GPU orders the tuples, merges duplicate
            // coordinates with `maximum` (SpmatLocal.hpp:487) and keeps this rank's strided slice: the tuples are
            // device-resident from here on, like the generated ones
            std::vector<std::vector<spcoord_t>> parts;  // one per parsing thread, in file order: uploaded piece by piece
            hnh::parse_matrix_market_parts(filename, m, n, parts);
            if (m >> 32 || n >> 32) hnh::fatal("Error, matrices with more than 2^32 rows or columns are not supported!");
            M = m;
            N = n;
            size_t n_raw = 0;
            for (auto& v : parts) n_raw += v.size();
            hnh::DeviceArray raw(world, std::max<size_t>(n_raw, 1) * sizeof(spcoord_t));
            hnh_tuple* rt = static_cast<hnh_tuple*>(raw.ptr());
            size_t at = 0;
            for (auto& v : parts) {
                world->copy(rt + at, v.data(), v.size() * sizeof(spcoord_t), HNH_COPY_H2D, HNH_STREAM_COMPUTE);
                at += v.size();
            }
            world->sync(HNH_STREAM_COMPUTE);  // the pieces are pageable host memory: done with them before they go
            std::vector<std::vector<spcoord_t>>().swap(parts);
            hnh_tuple_key key{};
            key.kind = HNH_KEY_ROW_COL;
            int row_bits = 1;
            while (row_bits < 32 && ((uint64_t)1 << row_bits) < m) row_bits++;
            world->check(world->be->hnh_tuples_sort(world->ctx, rt, (int64_t)n_raw, &key, 32 + row_bits, HNH_STREAM_COMPUTE), "hnh_tuples_sort");
            int64_t unique = 0;
            world->check(world->be->hnh_tuples_dedup_max(world->ctx, rt, (int64_t)n_raw, &unique, HNH_STREAM_COMPUTE), "hnh_tuples_dedup_max");
            dist_nnz = (uint64_t)unique;
            n_resident = unique > rank ? (size_t)((unique - rank + p - 1) / p) : 0;
            dcoords = hnh::DeviceArray(world, std::max<size_t>(n_resident, 1) * sizeof(spcoord_t));
            world->check(world->be->hnh_tuples_take_strided(world->ctx, rt, rank, p, dptr(), (int64_t)n_resident, HNH_STREAM_COMPUTE),
                         "hnh_tuples_take_strided");
            world->sync(HNH_STREAM_COMPUTE);
            resident = true;
            if (rank == 0) std::cout << "File reader read " << dist_nnz << " nonzeros." << std::endl;
            initialized = true;
            if (const char* ps = std::getenv("HNH_PERMUTE_SEED")) permuteVertices(std::strtoull(ps, nullptr, 10));
            return;
        }
        hnh::read_matrix_market(filename, m, n, all);
        M = m;
        N = n;
        dist_nnz = all.size();
        for (size_t e = rank; e < all.size(); e += p) coords.push_back(all[e]);
        if (rank == 0) std::cout << "File reader read " << dist_nnz << " nonzeros." << std::endl;
    } else {
        const uint64_t m = 1ull << logM;
        uint64_t seed = 12345;
        if (const char* s = std::getenv("HNH_ER_SEED")) seed = std::strtoull(s, nullptr, 10);
        // The reference calls GenGraph500Data with the initiator {.25, .25, .25, .25} (SpmatLocal.hpp:502-505): Erdos-Renyi.  HNH_RMAT="a,b,c"
        // (e.g. .57,.19,.19: Graph500's) selects a SKEWED initiator instead — the same generator call with other constants, hub rows
        // included; HNH_RMAT_SCRAMBLE=0 leaves the hubs at the low vertex numbers.
        double ra = 0.25, rb = 0.25, rc = 0.25;
        bool skewed = false, scramble = true;
        if (const char* rm = std::getenv("HNH_RMAT")) {
            if (std::sscanf(rm, "%lf,%lf,%lf", &ra, &rb, &rc) != 3 || ra < 0 || rb < 0 || rc < 0 || ra + rb + rc > 1.0)
                hnh::fatal("Error, HNH_RMAT must be three probabilities a,b,c with a + b + c <= 1!");
            skewed = true;
        }
        if (const char* sc = std::getenv("HNH_RMAT_SCRAMBLE")) scramble = std::atoi(sc) != 0;
        if (device_setup()) {
            // the same generator evaluated on this rank's GPU: draws, radix sort, de-duplication, strided slice — the
            // tuples are born device-resident (hnh_generate_er_keys / hnh_generate_rmat_keys / hnh_tuples_from_keys)
            const uint64_t draws = m * (uint64_t)nnz_per_row;
            hnh::DeviceArray keys(world, std::max<uint64_t>(draws, 1) * sizeof(uint64_t));
            int64_t unique = 0;
            if (skewed)
                world->check(world->be->hnh_generate_rmat_keys(world->ctx, logM, draws, ra, rb, rc, seed, scramble ? 1 : 0, static_cast<uint64_t*>(keys.ptr()),
                                                               &unique, HNH_STREAM_COMPUTE), "hnh_generate_rmat_keys");
            else
            world->check(world->be->hnh_generate_er_keys(world->ctx, m, m, draws, seed, static_cast<uint64_t*>(keys.ptr()), &unique,
                                                         HNH_STREAM_COMPUTE), "hnh_generate_er_keys");
            M = N = m;
            dist_nnz = (uint64_t)unique;
            n_resident = unique > rank ? (size_t)((unique - rank + p - 1) / p) : 0;
            dcoords = hnh::DeviceArray(world, std::max<size_t>(n_resident, 1) * sizeof(spcoord_t));
            world->check(world->be->hnh_tuples_from_keys(world->ctx, static_cast<const uint64_t*>(keys.ptr()), m, rank, p, 1.0, dptr(),
                                                         (int64_t)n_resident, HNH_STREAM_COMPUTE), "hnh_tuples_from_keys");
            world->sync(HNH_STREAM_COMPUTE);
            resident = true;
            if (rank == 0) std::cout << "R-mat generator created " << dist_nnz << " nonzeros." << std::endl;
            initialized = true;
            if (const char* ps = std::getenv("HNH_PERMUTE_SEED")) permuteVertices(std::strtoull(ps, nullptr, 10));
            return;
        }
        std::vector<uint64_t> keys = skewed ? hnh::rmat_keys(logM, m * (uint64_t)nnz_per_row, ra, rb, rc, seed, scramble)
                                            : hnh::erdos_renyi_keys(m, m, m * (uint64_t)nnz_per_row, seed);
        M = N = m;
        dist_nnz = keys.size();
        coords.reserve(keys.size() / p + 1);
        for (size_t e = rank; e < keys.size(); e += p) coords.push_back({keys[e] / m, keys[e] % m, 1.0});
        if (rank == 0) std::cout << "R-mat generator created " << dist_nnz << " nonzeros." << std::endl;
    }
    initialized = true;
    if (const char* ps = std::getenv("HNH_PERMUTE_SEED")) permuteVertices(std::strtoull(ps, nullptr, 10));
}

// Relabels rows and columns with a seeded random permutation (one permutation for both when the matrix is square,
// like RenameVertices on a graph).  Every rank derives the same permutation, so it is applied to local tuples only.
void SpmatLocal::permuteVertices(uint64_t seed) {
    std::vector<uint64_t> rp = hnh::vertex_permutation(M, seed);
    std::vector<uint64_t> cp = (M == N) ? rp : hnh::vertex_permutation(N, seed + 1);
    if (resident) {
        hnh::DeviceArray drp(world, rp.size() * sizeof(uint64_t)), dcp(world, cp.size() * sizeof(uint64_t));
        world->copy(drp.ptr(), rp.data(), rp.size() * sizeof(uint64_t), HNH_COPY_H2D, HNH_STREAM_COMPUTE);
        world->copy(dcp.ptr(), cp.data(), cp.size() * sizeof(uint64_t), HNH_COPY_H2D, HNH_STREAM_COMPUTE);
        world->check(world->be->hnh_tuples_relabel(world->ctx, dptr(), (int64_t)n_resident, static_cast<const uint64_t*>(drp.ptr()),
                                                   static_cast<const uint64_t*>(dcp.ptr()), HNH_STREAM_COMPUTE), "hnh_tuples_relabel");
        world->sync(HNH_STREAM_COMPUTE);  // the label tables die here
        return;
    }
#pragma omp parallel for
    for (size_t e = 0; e < coords.size(); e++) {
        coords[e].r = rp[coords[e].r];
        coords[e].c = cp[coords[e].c];
    }
}
