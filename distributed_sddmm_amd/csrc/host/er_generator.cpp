#include "er_generator.hpp"

#include <parallel/algorithm>

#include <algorithm>
#include <fstream>
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

void read_matrix_market(const std::string& path, uint64_t& m, uint64_t& n, std::vector<spcoord_t>& tuples) {
    std::ifstream in(path);
    if (!in) fatal("Error, cannot open matrix file " + path);
    std::string line;
    if (!std::getline(in, line) || line.rfind("%%MatrixMarket", 0) != 0) fatal("Error, " + path + " is not a MatrixMarket file");
    std::string lower = line;
    std::transform(lower.begin(), lower.end(), lower.begin(), ::tolower);
    if (lower.find("coordinate") == std::string::npos) fatal("Error, only coordinate MatrixMarket files are supported");
    const bool pattern = lower.find("pattern") != std::string::npos;
    const bool symmetric = lower.find("symmetric") != std::string::npos;
    while (std::getline(in, line))
        if (!line.empty() && line[0] != '%') break;
    uint64_t entries = 0;
    {
        std::istringstream hdr(line);
        if (!(hdr >> m >> n >> entries)) fatal("Error, bad MatrixMarket size line in " + path);
    }
    tuples.clear();
    tuples.reserve(symmetric ? 2 * entries : entries);
    for (uint64_t e = 0; e < entries; e++) {
        uint64_t r, c;
        double v = 1.0;
        if (!(in >> r >> c)) fatal("Error, truncated MatrixMarket file " + path);
        if (!pattern && !(in >> v)) fatal("Error, truncated MatrixMarket file " + path);
        if (r < 1 || r > m || c < 1 || c > n) fatal("Error, MatrixMarket index out of range in " + path);
        tuples.push_back({r - 1, c - 1, v});
        if (symmetric && r != c) tuples.push_back({c - 1, r - 1, v});
    }
    // duplicates -> maximum
    std::sort(tuples.begin(), tuples.end(), [](const spcoord_t& a, const spcoord_t& b) { return row_major(a, b); });
    size_t out = 0;
    for (size_t e = 0; e < tuples.size(); e++) {
        if (out > 0 && tuples[out - 1].r == tuples[e].r && tuples[out - 1].c == tuples[e].c)
            tuples[out - 1].value = std::max(tuples[out - 1].value, tuples[e].value);
        else
            tuples[out++] = tuples[e];
    }
    tuples.resize(out);
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
        if (device_setup()) {
            // the same generator evaluated on this rank's GPU: draws, radix sort, de-duplication, strided slice — the
            // tuples are born device-resident (hnh_generate_er_keys / hnh_tuples_from_keys)
            const uint64_t draws = m * (uint64_t)nnz_per_row;
            hnh::DeviceArray keys(world, std::max<uint64_t>(draws, 1) * sizeof(uint64_t));
            int64_t unique = 0;
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
        std::vector<uint64_t> keys = hnh::erdos_renyi_keys(m, m, m * (uint64_t)nnz_per_row, seed);
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
