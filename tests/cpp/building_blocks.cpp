// A two-grid cycle written by hand from the reference's operator-level building blocks
// (AMG_smoothers.hpp / AMG_cycle_utilities.hpp / AMG_coarsening.hpp / AMG_coarse_level_solver.hpp),
// the way src/AMG_phases.cpp composes them inside AMG_solver.
#include "AMG.hpp"
#include "AMG_coarse_level_solver.hpp"
#include "AMG_coarsening.hpp"
#include "AMG_cycle_utilities.hpp"
#include "AMG_smoothers.hpp"

#include <cmath>
#include <cstdio>
#include <vector>

static double sum_abs(const double *v, int n)
{
    double s = 0;
    for (int i = 0; i < n; i++) s += std::fabs(v[i]);
    return s;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    sp_matrix_mg *A = new sp_matrix_mg();
    double *b;
    readcoo(argv[1], argv[2], A, b);
    A->sp_matrix_fill();
    A->sp_matrix_fill_diagonal();
    const int n = A->nrow;

    sp_matrix_mg *P = nullptr, *Ac = nullptr, *Pb = nullptr;
    sequential::HEM_Prolongator(*A, P, 0);
    parallel::coarsen_matrix(*A, Ac, *P);
    sequential::beck_prolongator(*A, Pb);
    const int nc = Ac->nrow;
    std::printf("SHAPES %d %d %d %d %d %d %d\n", P->nrow, P->ncol, P->rowptr[P->nrow], Ac->nrow, Ac->rowptr[nc], Pb->ncol,
                Pb->rowptr[Pb->nrow]);
    std::printf("ACSUM %.17g %.17g\n", sum_abs(Ac->val, Ac->rowptr[nc]), sum_abs(Ac->diagonal, nc));

    std::vector<double> xv((size_t)n, 0.0), rv((size_t)n), bcv((size_t)nc), xcv((size_t)nc);
    double *x = xv.data(), *r = rv.data(), *bc = bcv.data(), *xc = xcv.data();
    const double r0 = parallel::residual(*A, b, x);
    parallel::jacobi_smoother(*A, b, x, 6);
    parallel::store_residual(*A, b, x, r);
    parallel::transfer_residual(*P, r, bc);
    Direct_Solver_Pardiso S(*Ac);
    S.Direct_Solver_Pardiso_solve(bc, xc);
    parallel::transfer_solution(*P, xc, x);
    sequential::jacobi_smoother(*A, b, x, 6);
    const double r1 = sequential::residual(*A, b, x);
    std::printf("TWOGRID %.17g %.17g %.17g %.17g %.17g %.17g\n", r0, r1, x[0], x[n / 2], bc[0], xc[nc / 2]);

    sp_matrix_mg *Q = nullptr;
    sequential::mis_prolongator(*A, Q);
    if (Q != nullptr) return 4;
    delete Pb;
    delete Ac;
    delete P;
    delete A;
    return 0;
}
