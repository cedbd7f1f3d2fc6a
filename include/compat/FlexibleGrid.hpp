// Forwarding header: lets code written against the reference's include names build against this engine
// (add -I include/compat -I distributed_sddmm_amd/csrc/host -I include).  See INTEGRATION.md section A.
#pragma once
#include "../../distributed_sddmm_amd/csrc/host/flexible_grid.hpp"
#include "mpi.h"  // FlexibleGrid.hpp:4
using namespace std;  // the reference's headers say so at global scope, and code written against them relies on it
