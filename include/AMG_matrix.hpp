// AMG_matrix.hpp -- host CSR container of the drop-in C++ API.
// Same public layout and names as the reference's include/AMG_matrix.hpp:6-32 so user code
// (e.g. the reference's main.cpp) compiles unchanged against libsparsh_amg.so.
#ifndef AMG_MATRIX_HPP_
#define AMG_MATRIX_HPP_

// CSR storage: rowptr[nrow+1], colindex[nnz], val[nnz]; 0-based.
class sp_matrix
{
  public:
    int nrow;  // rows
    int ncol;  // columns
    int nnz;   // stored entries

    int *rowptr = nullptr;
    int *colindex = nullptr;
    double *val = nullptr;

  public:
    // allocates zero-filled CSR arrays for an r x c matrix with n entries
    sp_matrix(int r, int c, int n);
    sp_matrix();

    // prints the matrix row by row
    void check_sp_matrix();

    // As in the reference there is no destructor: the CSR arrays belong to the caller.
};

#endif /* AMG_MATRIX_HPP_ */
