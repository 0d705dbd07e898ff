// A STAND-IN for <rclcpp/rclcpp.hpp> (tests/test_reference_compile.py): the logging macros and get_logger the reference's Error.h and
// frame code use.  NOT rclcpp.
#pragma once
#include <cstdio>
#include <memory>
#include <string>

namespace rclcpp {
class Logger {};
inline Logger get_logger(const std::string&) { return Logger(); }
class Node {
 public:
  typedef std::shared_ptr<Node> SharedPtr;
};
template <class T>
class Publisher {
 public:
  typedef std::shared_ptr<Publisher> SharedPtr;
};
template <class T>
class Subscription {
 public:
  typedef std::shared_ptr<Subscription> SharedPtr;
};
}  // namespace rclcpp
#define RCLCPP_INFO(logger, ...) ((void)(logger), (void)std::snprintf(nullptr, 0, __VA_ARGS__))
#define RCLCPP_WARN(logger, ...) ((void)(logger), (void)std::snprintf(nullptr, 0, __VA_ARGS__))
#define RCLCPP_ERROR(logger, ...) ((void)(logger), (void)std::snprintf(nullptr, 0, __VA_ARGS__))
#define RCLCPP_FATAL(logger, ...) ((void)(logger), (void)std::snprintf(nullptr, 0, __VA_ARGS__))
