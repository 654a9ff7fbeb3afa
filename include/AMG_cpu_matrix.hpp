// AMG_cpu_matrix.hpp -- sp_matrix_mg of the drop-in C++ API.
// Public members and methods keep the names of the reference's include/AMG_cpu_matrix.hpp:12-51.
// The three MKL-typed members (A1, sA, des) become opaque: user code never touches them
// (README.md:76-94, main.cpp:16-42 of the reference).
#ifndef AMG_CPU_MATRIX_HPP_
#define AMG_CPU_MATRIX_HPP_

#include "AMG_matrix.hpp"

struct sparsh_matrix_descr {
    int type = 0, mode = 0, diag = 0;
};

class sp_matrix_mg : public sp_matrix
{
  public:
    void *A1 = nullptr;       // opaque backend handle (was sparse_matrix_t)
    int sA = 0;               // last backend status (was sparse_status_t)
    sparsh_matrix_descr des;  // (was matrix_descr)

    double *diagonal = nullptr;
    double *helper = nullptr;
    double *entries = nullptr;

    int *color = nullptr;
    int *color_count = nullptr;
    int total_colors = 0;
    int max_color_row = 0;

  public:
    using sp_matrix::sp_matrix;

    // registers the CSR with the solver backend; sorts the columns of every row
    void sp_matrix_fill();

    // fills `diagonal` and allocates the `helper` scratch vector
    void sp_matrix_fill_diagonal();

    // multi-colour reordering for the SOR smoother: not part of the MI355X build (prints a notice)
    void color_matrix_and_reorder();

    // divides every row by its diagonal entry and b by sqrt(diagonal)
    void scale_system(double *&b);

    // divides every entry by the squared 2-norm of its column
    void normalize_matrix();

    ~sp_matrix_mg();
};

#endif /* AMG_CPU_MATRIX_HPP_ */
