// A STAND-IN for <g2o/types/slam3d/se3quat.h> (tests/test_reference_compile.py).  NOT g2o.
#pragma once
namespace g2o {
struct SE3Quat {};
struct Vector3 {};
}  // namespace g2o
