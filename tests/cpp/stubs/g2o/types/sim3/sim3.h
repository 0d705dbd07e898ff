// A STAND-IN for <g2o/types/sim3/sim3.h> (tests/test_reference_compile.py).  NOT g2o.
#pragma once
namespace g2o {
struct Sim3 {};
}  // namespace g2o
