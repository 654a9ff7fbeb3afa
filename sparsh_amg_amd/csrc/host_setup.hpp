// host_setup.hpp -- host side of the AMG setup phase (coarsening, Galerkin product, coarse
// factorisation).  The north star keeps setup on the host; MKL/PARDISO are replaced by the
// code in host_setup.cpp.  Reference behaviour followed: src/AMG_phases.cpp:35-90,
// src/AMG_coarsening.cpp:14-97 (HEM), :269-339 (Beck), src/AMG_cycle_utilities.cpp:126-146.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace sparsh {

// Non-owning or owning CSR view on host memory.
struct HostCsr {
    int nrow = 0, ncol = 0;
    const int *rowptr = nullptr;
    const int *col = nullptr;
    const double *val = nullptr;
    // storage when owning
    std::vector<int> rp_store, col_store;
    std::vector<double> val_store;
    // rank-local blocks of a partitioned operator only: the global column of every entry and the
    // global index of local row 0 (layout builders need the original entry order)
    std::vector<int> gcol_store;
    int grow0 = 0;
    // deep-halo local operators: rows beyond the own block are ghost rows with arbitrary global indices
    std::vector<int> grow_store;  // global row of every local row (empty: grow0 + r)
    int grow(int r) const { return grow_store.empty() ? grow0 + r : grow_store[r]; }

    int nnz() const { return rowptr ? rowptr[nrow] : 0; }
    void adopt()
    {
        rowptr = rp_store.data();
        col = col_store.data();
        val = val_store.data();
    }
    static HostCsr alias(int nrow, int ncol, const int *rp, const int *ci, const double *v)
    {
        HostCsr A;
        A.nrow = nrow;
        A.ncol = ncol;
        A.rowptr = rp;
        A.col = ci;
        A.val = v;
        return A;
    }
};

struct SetupParams {
    int max_levels = 6;
    int limit_upper = 4000;
    int limit_lower = 2000;
    int coarsening = 0;  // 0 HEM, 1 Beck
    int coarse_limit = 40000;  // largest coarsest level max_levels may leave (above: keep coarsening)
    int dense_limit = 8192;    // largest coarsest level solved with an explicit dense inverse
    int extend_until = 0;      // an extended hierarchy stops at the first level of at most this many rows (0: coarse_limit)
    size_t coarse_factor_bytes = (size_t)1 << 30;  // the reference's own coarsest level is kept, whatever its rows, while its estimated
                                                   // nested-dissection factors stay below this (0: coarse_limit alone decides)
    int host_threads = 0;
    bool print = true;
};

struct HostLevel {
    HostCsr A;             // level operator (level 0 aliases the caller's arrays)
    HostCsr P;             // prolongator to the next level (empty on the last level)
    HostCsr R;             // explicit P^T in CSR (gather form of the restriction)
    std::vector<double> diag;
    bool P_is_aggregation = false;  // one entry of value 1.0 per row (HEM)
};

struct HostHierarchy {
    std::vector<HostLevel> levels;
    // coarsest level: explicit inverse, row-major nL x nL, when nL <= dense_limit (coarse_dense);
    // otherwise the device factors it (coarse.cpp) and coarse_inverse stays empty
    int nL = 0;
    bool coarse_dense = true;
    std::vector<double> coarse_inverse;
    double seconds = 0.0;
    bool extended = false;  // hierarchy continued past max_levels because of coarse_limit
    std::string error;
};

int effective_cpus();  // CPUs usable by this process (affinity and cgroup quota)

// individual steps (exposed for tests through the C ABI)
HostCsr hem_prolongator(const HostCsr &A, int level);
HostCsr beck_prolongator(const HostCsr &A);
HostCsr transpose(const HostCsr &A);
HostCsr galerkin(const HostCsr &A, const HostCsr &P, const HostCsr &R, bool P_is_aggregation);
std::vector<double> extract_diagonal(const HostCsr &A);
// reverse Cuthill-McKee ordering of the symmetrised pattern: order[new] = old
std::vector<int> rcm_order(const HostCsr &A);
// dense inverse of a sparse matrix through RCM + banded LU (partial pivoting); false if singular
bool sparse_inverse(const HostCsr &A, std::vector<double> &inv);

// rough size of the nested-dissection factors of A (nd_plan.cpp; one breadth-first search, no dissection)
size_t nd_estimate_factor_bytes(const HostCsr &A);

// whole setup
bool build_hierarchy(const HostCsr &A0, const SetupParams &prm, HostHierarchy &H);

// Flat byte image of a hierarchy (multi-GPU: rank 0 builds it once, the other ranks receive it instead of repeating
// the setup).  Level 0's operator is not part of the image: every rank aliases the caller's arrays, as build_hierarchy does.
void serialize_hierarchy(const HostHierarchy &H, std::vector<char> &out);
bool deserialize_hierarchy(const char *buf, size_t bytes, const HostCsr &A0, HostHierarchy &H);

}  // namespace sparsh
