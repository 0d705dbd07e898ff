// A STAND-IN for <g2o/solvers/eigen/linear_solver_eigen.h> (tests/test_reference_compile.py).  NOT g2o.
#pragma once
namespace g2o {
template <class M>
class LinearSolverEigen {};
}  // namespace g2o
