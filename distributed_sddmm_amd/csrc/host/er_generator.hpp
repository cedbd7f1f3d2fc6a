// Deterministic synthetic inputs (replaces CombBLAS GenGraph500Data with initiator {.25,.25,.25,.25},
// SpmatLocal.hpp:502-505, and its file reader).  Counter-based: draw k of an (m x n, draws, seed) matrix is
//     row = splitmix64(seed + 2k*G) % m,  col = splitmix64(seed + (2k+1)*G) % n,   G = 0x9E3779B97F4A7C15
// followed by de-duplication; every rank (and oracle/oracle.py:erdos_renyi_mn, bit for bit) evaluates the
// same function, so all transports and rank counts see the same global matrix.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "common.hpp"

namespace hnh {

inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Sorted (row-major), de-duplicated keys row * n + col of the whole matrix.
std::vector<uint64_t> erdos_renyi_keys(uint64_t m, uint64_t n, uint64_t draws, uint64_t seed);

// Skewed stand-in for real graphs (BASELINE config 4 names com-Orkut, which is not available offline):
// Graph500-style R-MAT on 2^logm vertices with initiator (a, b, c, 1-a-b-c).  Edge k, level l draws
//     u = (splitmix64(seed + (k*logm + l)*G) >> 11) * 2^-53   and picks the quadrant by thresholds;
// with `scramble` vertex v is renamed to (v * 0x9E3779B1 + 0x7F4A7C15) mod 2^logm (a bijection), the
// counterpart of the reference's PermEdges/RenameVertices load-balancing step (SpmatLocal.hpp:506-507).
// Bit-identical twin: oracle/oracle.py:rmat.
std::vector<uint64_t> rmat_keys(int logm, uint64_t edges, double a, double b, double c, uint64_t seed, bool scramble);

// Seeded random relabelling of n vertices for load balance on real graphs (the reference applies CombBLAS's
// PermEdges / RenameVertices to generated graphs, SpmatLocal.hpp:506-507, and ships random_permute.cpp for files):
// new_label[v] = position of v when vertices are ordered by splitmix64(seed + v * G) (ties by v).
// Twin: oracle/oracle.py:vertex_permutation.
std::vector<uint64_t> vertex_permutation(uint64_t n, uint64_t seed);

// MatrixMarket coordinate reader (general / symmetric; pattern, integer or real); duplicates keep the
// maximum, as the reference's `maximum<double>()` reduction does (SpmatLocal.hpp:487).  Returns all tuples.
void read_matrix_market(const std::string& path, uint64_t& m, uint64_t& n, std::vector<spcoord_t>& tuples);
// its two halves: the parallel parser (symmetric entries mirrored, duplicates still present) and the duplicate merge on the host
// (the default set-up does the merge on the GPU instead: hnh_tuples_sort + hnh_tuples_dedup_max)
void parse_matrix_market(const std::string& path, uint64_t& m, uint64_t& n, std::vector<spcoord_t>& tuples);
// the same, the parsing threads' pieces kept apart (file order)
void parse_matrix_market_parts(const std::string& path, uint64_t& m, uint64_t& n, std::vector<std::vector<spcoord_t>>& parts);
void merge_duplicates_max(std::vector<spcoord_t>& tuples);

}  // namespace hnh
