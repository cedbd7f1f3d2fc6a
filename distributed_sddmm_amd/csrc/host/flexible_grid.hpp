// 3-D process grid with selectable rank adjacency — same public surface as the reference's FlexibleGrid
// (FlexibleGrid.hpp:41-135): i, j, k, nr, nc, nh, adjacency, row/col/fiber worlds and the three slice
// communicators, rankIn{Row,Col,Fiber}, get_global_rank().  MPI communicators become hnh::Comm (ordered
// rank lists; RCCL sub-communicators are attached when the world is an RcclWorld).
//
// adjacency orders the grid dimensions from fastest- to slowest-varying world rank:
//   1 crf (i fastest)  2 cfr  3 rcf (j fastest)  4 rfc  5 fcr  6 frc      (FlexibleGrid.hpp:30-38)
// On one MI355X node every pair of GPUs is an xGMI peer (full mesh), so adjacency changes which physical
// link a ring uses but not its bandwidth; the numbering is kept for drop-in parity of rank layouts.
#pragma once
#include <iostream>
#include <vector>
#include "world.hpp"

class FlexibleGrid {
public:
    int i, j, k;
    int adjacency;
    int global_rank, num_procs;
    int dim_list[3];
    int nr, nc, nh;
    int permutation[3];

    hnh::Comm row_world, col_world, fiber_world;
    hnh::Comm rowcol_slice, rowfiber_slice, colfiber_slice;
    int rankInRow, rankInCol, rankInFiber;
    hnh::World* world;

    FlexibleGrid(int nr, int nc, int nh, int adjacency) {
        world = hnh::current_world();
        num_procs = world->size;
        global_rank = world->rank;
        if (nr * nc * nh != num_procs) hnh::fatal("Error, grid dimensions do not multiply to the number of processes!");
        dim_list[0] = this->nr = nr;
        dim_list[1] = this->nc = nc;
        dim_list[2] = this->nh = nh;
        this->adjacency = adjacency;
        static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
        if (adjacency < 1 || adjacency > 6) hnh::fatal("Error, adjacency must be between 1 and 6!");
        for (int t = 0; t < 3; t++) permutation[t] = perms[adjacency - 1][t];

        get_ijk_indices(&i, &j, &k);
        if (global_rank != get_global_rank(i, j, k)) hnh::fatal("Error, grid rank map is not a bijection!");

        // same colours and keys as the MPI_Comm_split calls of FlexibleGrid.hpp:80-88
        row_world = world->split(i + k * nr, j);
        col_world = world->split(j + k * nc, i);
        fiber_world = world->split(i + j * nr, k);
        rowcol_slice = world->split(k, i + j * nr);
        rowfiber_slice = world->split(j, i + k * nr);
        colfiber_slice = world->split(i, j + k * nc);
        rankInRow = row_world.me;
        rankInCol = col_world.me;
        rankInFiber = fiber_world.me;
    }

    ~FlexibleGrid() {
        for (hnh::Comm* c : {&row_world, &col_world, &fiber_world, &rowcol_slice, &rowfiber_slice, &colfiber_slice})
            world->free_comm(*c);
    }
    FlexibleGrid(const FlexibleGrid&) = delete;
    FlexibleGrid& operator=(const FlexibleGrid&) = delete;

    void get_ijk_indices(int rank, int* i_out, int* j_out, int* k_out) const {
        int t[3];
        t[permutation[0]] = rank % dim_list[permutation[0]];
        t[permutation[1]] = (rank / dim_list[permutation[0]]) % dim_list[permutation[1]];
        t[permutation[2]] = (rank / (dim_list[permutation[0]] * dim_list[permutation[1]])) % dim_list[permutation[2]];
        *i_out = t[0];
        *j_out = t[1];
        *k_out = t[2];
    }
    void get_ijk_indices(int* i_out, int* j_out, int* k_out) const { get_ijk_indices(global_rank, i_out, j_out, k_out); }

    int get_global_rank(int i_in, int j_in, int k_in) const {
        const int t[3] = {i_in, j_in, k_in};
        return t[permutation[0]] + t[permutation[1]] * dim_list[permutation[0]] +
               t[permutation[2]] * dim_list[permutation[0]] * dim_list[permutation[1]];
    }

    void print_rank_information() const {
        std::cout << "Global Rank: " << global_rank << "i, j, k: (" << i << ", " << j << ", " << k << ")" << std::endl;
    }

    // Collects one int per rank and prints it layer by layer on rank 0 (FlexibleGrid.hpp:141-167).
    void gather_and_pretty_print(const std::string& title, int msg) {
        std::vector<int> all(num_procs, 0);
        world->host_allgather(&msg, all.data(), sizeof(int));
        if (global_rank != 0) return;
        std::cout << title << std::endl;
        for (int kk = 0; kk < nh; kk++) {
            std::cout << "========= Layer " << kk << " ==========" << std::endl;
            for (int ii = 0; ii < nr; ii++) {
                for (int jj = 0; jj < nc; jj++) std::cout << all[get_global_rank(ii, jj, kk)] << "\t";
                std::cout << std::endl;
            }
            std::cout << "============================" << std::endl;
        }
    }

    // Broadcast-based self check of every sub-communicator (FlexibleGrid.hpp:169-201); returns true if
    // each rank received the value its communicator's root must have sent.
    bool self_test(bool print = false) {
        bool ok = true;
        auto bc = [&](hnh::Comm& c, int value, int expect, const char* title) {
            int buf = value;
            world->host_bcast(c, 0, &buf, sizeof(int));
            ok = ok && (buf == expect);
            if (print) gather_and_pretty_print(title, buf);
        };
        if (print) {
            gather_and_pretty_print("Global Ranks:", global_rank);
            gather_and_pretty_print("i Values:", i);
            gather_and_pretty_print("j Values:", j);
            gather_and_pretty_print("k Values:", k);
        }
        bc(row_world, i, i, "Row Broadcast:");              // all members of a row share i (and k)
        bc(col_world, j, j, "Col Broadcast:");              // all members of a column share j
        bc(fiber_world, i + nr * j, i + nr * j, "Fiber Broadcast:");
        bc(rowcol_slice, k, k, "Row Column Slice Broadcast:");
        bc(colfiber_slice, i, i, "Column Fiber Slice Broadcast:");
        bc(rowfiber_slice, j, j, "Row Fiber Slice Broadcast:");
        return ok;
    }
};
