// Host-side utilities mirroring the reference's common.h / common.cpp (L0 of SURVEY §1):
//   spcoord_t (common.h:27-33), MatMode (common.h:21), pMod / divideAndRoundUp / divideIntoSegments
//   (common.cpp:16-28,68-83), steady-clock timers (common.cpp:6-14), comparators (common.cpp:49-66).
// Error convention of the reference is "print to cout and exit(1)" (15D_dense_shift.hpp:60-65,
// sparse_kernels.cpp:76-83): hnh::fatal() does exactly that unless a caller that must survive
// (the C API used by the Python tests) switched it to throwing.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

typedef enum { Amat, Bmat } MatMode;
typedef enum { k_sddmmA, k_spmmA, k_spmmB, k_sddmmB } KernelMode;  // sparse_kernels.h:13

struct spcoord_t {
    uint64_t r;
    uint64_t c;
    double value;
    std::string string_rep() const { return std::to_string(r) + " " + std::to_string(c) + " " + std::to_string(value); }
};

inline bool column_major(const spcoord_t& a, const spcoord_t& b) { return a.c == b.c ? a.r < b.r : a.c < b.c; }
inline bool row_major(const spcoord_t& a, const spcoord_t& b) { return a.r == b.r ? a.c < b.c : a.r < b.r; }

typedef std::chrono::time_point<std::chrono::steady_clock> my_timer_t;
inline my_timer_t start_clock() { return std::chrono::steady_clock::now(); }
inline double stop_clock_get_elapsed(my_timer_t& start) {
    std::chrono::duration<double> diff = std::chrono::steady_clock::now() - start;
    return diff.count();
}

inline int pMod(int num, int denom) { return ((num % denom) + denom) % denom; }
inline int divideAndRoundUp(int num, int denom) { return num / denom + (num % denom > 0 ? 1 : 0); }

// roughly equal segments; segment_starts has num_segments + 1 entries (common.cpp:68-83)
inline void divideIntoSegments(int total, int num_segments, std::vector<int>& segment_starts, std::vector<int>& segment_sizes) {
    const int share = divideAndRoundUp(total, num_segments);
    segment_starts.clear();
    segment_sizes.clear();
    for (int i = 0; i < num_segments; i++) segment_starts.push_back(std::min(share * i, total));
    segment_starts.push_back(total);
    for (int i = 0; i < num_segments; i++) segment_sizes.push_back(segment_starts[i + 1] - segment_starts[i]);
}

namespace hnh {

// setup-phase stopwatch: prints "<label>: x.xx s" on destruction when HNH_VERBOSE_SETUP=1
struct PhaseTimer {
    const char* label;
    my_timer_t t0;
    bool on;
    explicit PhaseTimer(const char* l) : label(l), t0(start_clock()), on(std::getenv("HNH_VERBOSE_SETUP") != nullptr) {}
    ~PhaseTimer() {
        if (on) std::cout << "[setup] " << label << ": " << stop_clock_get_elapsed(t0) << " s" << std::endl;
    }
};

struct Error : public std::runtime_error {
    using std::runtime_error::runtime_error;
};

void set_throw_on_error(bool on);  // default false: print + exit(1) like the reference
[[noreturn]] void fatal(const std::string& msg);

}  // namespace hnh
