// prost/exception.hpp -- same role as the reference's include/prost/exception.hpp:29-41.
#ifndef PROST_EXCEPTION_HPP_
#define PROST_EXCEPTION_HPP_
#include <exception>
#include <string>

namespace prost {

class Exception : public std::exception {
 public:
  explicit Exception(const std::string& msg) : msg_(msg) {}
  virtual ~Exception() throw() {}
  virtual const char* what() const throw() { return msg_.c_str(); }

 private:
  std::string msg_;
};

}  // namespace prost
#endif
