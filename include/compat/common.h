// Forwarding header: lets code written against the reference's include names build against this engine
// (add -I include/compat -I distributed_sddmm_amd/csrc/host -I include).  See INTEGRATION.md section A.
#pragma once
#include "../../distributed_sddmm_amd/csrc/host/common.hpp"
#include "../../distributed_sddmm_amd/csrc/host/dense.hpp"
using hnh::BufferPair;
using hnh::DenseMatrix;
using hnh::VectorXd;
using namespace std;  // the reference's headers say so at global scope, and code written against them relies on it
// common.cpp:37 builds the MPI datatype of an SPCOORD tuple; tuples travel as bytes here, so there is nothing to register.  Kept
// because every main of the reference calls it right after MPI_Init (bench_erdos_renyi.cpp:21, bench_file.cpp:21, scratch.cpp:80).
inline void initialize_mpi_datatypes() {}
