// A STAND-IN for <g2o/solvers/dense/linear_solver_dense.h> (tests/test_reference_compile.py).  NOT g2o.
#pragma once
namespace g2o {
template <class M>
class LinearSolverDense {};
}  // namespace g2o
