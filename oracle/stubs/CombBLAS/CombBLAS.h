// TEST INFRASTRUCTURE ONLY (oracle build). Declarations-only stand-in for the CombBLAS names that
// the reference's common.h:14 and SpmatLocal::loadTuples/unpack_tuples (SpmatLocal.hpp:358-370,
// 467-533) mention, so the unmodified reference headers PARSE.  CombBLAS contributes only input
// generation / file I/O to the reference (never arithmetic); the oracle driver feeds coordinates
// straight into SpmatLocal::coords, so every function here aborts if it is ever reached.
#pragma once
#include <mpi.h>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <memory>
#include <string>
#include <tuple>
#include <array>

namespace combblas {

#define MAXVERTNAME 64

[[noreturn]] inline void hnh_stub_unreachable(const char* what) {
    std::fprintf(stderr, "CombBLAS stand-in: %s is not available in the oracle build\n", what);
    std::abort();
}

template <typename T> struct maximum { T operator()(const T& a, const T& b) const { return a < b ? b : a; } };

class CommGrid {
public:
    CommGrid(MPI_Comm, int, int) {}
};

template <typename IT, typename NT> class SpDCCols {};

template <typename IT>
class DistEdgeList {
public:
    explicit DistEdgeList(std::shared_ptr<CommGrid>) {}
    void GenGraph500Data(double*, unsigned long, int) { hnh_stub_unreachable("GenGraph500Data"); }
};
template <typename IT> void PermEdges(DistEdgeList<IT>&) { hnh_stub_unreachable("PermEdges"); }
template <typename IT> void RenameVertices(DistEdgeList<IT>&) { hnh_stub_unreachable("RenameVertices"); }

template <typename IT, typename NT, typename DER>
class SpParMat {
public:
    explicit SpParMat(std::shared_ptr<CommGrid>) {}
    template <typename ET> SpParMat(DistEdgeList<ET>&, bool) {}
    template <typename OP> void ParallelReadMM(const std::string&, bool, OP) { hnh_stub_unreachable("ParallelReadMM"); }
    int64_t getnnz() const { return 0; }
    int64_t getnrow() const { return 0; }
    int64_t getncol() const { return 0; }
    DER seq() { return DER(); }
};

template <typename IT, typename NT>
class SpTuples {
public:
    std::tuple<IT, IT, NT>* tuples = nullptr;
    template <typename DER> explicit SpTuples(const DER&) {}
    int64_t getnnz() const { return 0; }
};

}  // namespace combblas
