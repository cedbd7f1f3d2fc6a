// Forwarding header: lets code written against the reference's include names build against this engine
// (add -I include/compat -I distributed_sddmm_amd/csrc/host -I include).  See INTEGRATION.md section A.
#pragma once
#include "../../distributed_sddmm_amd/csrc/host/cannon_dense_25d.hpp"
#include "distributed_sparse.h"  // (the reference's header includes it, and with it <mpi.h>, common.h, <cassert>)
using namespace std;  // the reference's headers say so at global scope, and code written against them relies on it
