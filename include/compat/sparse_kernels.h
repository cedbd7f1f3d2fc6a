// Forwarding header: lets code written against the reference's include names build against this engine
// (add -I include/compat -I distributed_sddmm_amd/csrc/host -I include).  See INTEGRATION.md section A.
#pragma once
#include "../../distributed_sddmm_amd/csrc/host/sparse_kernels.hpp"
#include "common.h"  // sparse_kernels.h:7-8
#include "SpmatLocal.hpp"
using namespace std;  // the reference's headers say so at global scope, and code written against them relies on it
