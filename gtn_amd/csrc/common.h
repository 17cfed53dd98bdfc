// common.h -- error plumbing shared by the engine's translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>

#include "gtn_amd.h"

namespace gtnx {

// One exception type inside the engine; capi.cpp maps it to gtnx_status_t and the
// C++ shim (include/gtn/) maps that back to the reference's std:: exception types.
struct Error : public std::runtime_error {
  gtnx_status_t status;
  Error(gtnx_status_t s, const std::string& m) : std::runtime_error(m), status(s) {}
};

[[noreturn]] inline void throw_invalid(const std::string& m) { throw Error(GTNX_INVALID_ARGUMENT, m); }
[[noreturn]] inline void throw_logic(const std::string& m) { throw Error(GTNX_LOGIC_ERROR, m); }
[[noreturn]] inline void throw_runtime(const std::string& m) { throw Error(GTNX_RUNTIME_ERROR, m); }
[[noreturn]] inline void throw_range(const std::string& m) { throw Error(GTNX_OUT_OF_RANGE, m); }
[[noreturn]] inline void throw_device(const std::string& m) { throw Error(GTNX_DEVICE_ERROR, m); }

#define HIP_CHECK(expr)                                                              \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      ::gtnx::throw_device(std::string("HIP error: ") + hipGetErrorString(_e) +      \
                           " at " __FILE__ ":" + std::to_string(__LINE__) + " (" #expr ")"); \
    }                                                                                \
  } while (0)

constexpr const char* kCycleMsg = "Graph has a cycle, self-loop or is disconnected!"; // shortest.cpp:150-151

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

} // namespace gtnx
