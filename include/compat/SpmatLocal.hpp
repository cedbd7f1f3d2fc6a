// Forwarding header: lets code written against the reference's include names build against this engine
// (add -I include/compat -I distributed_sddmm_amd/csrc/host -I include).  See INTEGRATION.md section A.
#pragma once
#include "../../distributed_sddmm_amd/csrc/host/spmat_local.hpp"
#include <cassert>
#include <string.h>
#include "mpi.h"     // the reference's SpmatLocal.hpp:11,13 bring these in for every file that includes it
#include "common.h"
using namespace std;  // the reference's headers say so at global scope, and code written against them relies on it
