// Forwarding header for code written against the reference's `#include "json.hpp"` + `using json = nlohmann::json;`
// (distributed_sparse.h:14-16, benchmark_dist.cpp:20-22): the reporting calls of this engine return hnh::json, an independent
// value type with the operations the reference's harness uses (distributed_sddmm_amd/csrc/host/json.hpp).  nlohmann/json itself
// is a third-party header that the reference vendors; a program that ships it includes it BEFORE these headers and converts
// with nlohmann::json::parse(x.dump()).
#pragma once
#include "../../distributed_sddmm_amd/csrc/host/json.hpp"
#ifndef INCLUDE_NLOHMANN_JSON_HPP_  // (nlohmann's own include guard: never redefine the real thing)
namespace nlohmann {
using json = hnh::json;
}
#endif
