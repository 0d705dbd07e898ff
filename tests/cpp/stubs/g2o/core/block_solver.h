// A STAND-IN for <g2o/core/block_solver.h> (tests/test_reference_compile.py): the names the reference's Optimizer.h mentions in typedefs.  NOT g2o.
#pragma once
#include <map>
#include <set>
#include <tuple>
#include <unordered_map>
namespace g2o {
struct PoseMatrix6 {};
struct PoseMatrix7 {};
struct BlockSolver_6_3 {
  typedef PoseMatrix6 PoseMatrixType;
};
struct BlockSolver_7_3 {
  typedef PoseMatrix7 PoseMatrixType;
};
struct OptimizableGraph {
  struct Edge {};
};
}  // namespace g2o
