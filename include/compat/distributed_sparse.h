// Forwarding header: lets code written against the reference's include names build against this engine
// (add -I include/compat -I distributed_sddmm_amd/csrc/host -I include).  See INTEGRATION.md section A.
#pragma once
#include "../../distributed_sddmm_amd/csrc/host/distributed_sparse.hpp"
#include <cassert>
#include "mpi.h"  // distributed_sparse.h:7-14
#include "common.h"
#include "SpmatLocal.hpp"
#include "FlexibleGrid.hpp"
#include "sparse_kernels.h"
#include "json.hpp"
using namespace std;  // the reference's headers say so at global scope, and code written against them relies on it
