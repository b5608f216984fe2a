// slamx_runtime.hpp -- shared plumbing of the C++ facade: one process-wide GPU context and the
// status-code -> exception mapping (std::invalid_argument for empty descriptors,
// std::runtime_error for everything else, as the reference throws).
#pragma once

#include <cstdlib>
#include <stdexcept>
#include <string>

#include <slamx.h>

namespace slam {
namespace detail {

inline void check(int rc) {
    if (rc == SLAMX_OK) {
        return;
    }
    const std::string msg = slamx_last_error();
    if (rc == SLAMX_ERR_INVALID_ARGUMENT) {
        throw std::invalid_argument(msg);
    }
    throw std::runtime_error(msg);
}

// Device 0 unless SLAMX_DEVICE is set; created on first use, lives for the process.
inline slamx_ctx* default_context() {
    static slamx_ctx* ctx = [] {
        int dev = 0;
        if (const char* e = std::getenv("SLAMX_DEVICE")) {
            dev = std::atoi(e);
        }
        slamx_ctx* c = nullptr;
        check(slamx_ctx_create(dev, nullptr, &c));
        return c;
    }();
    return ctx;
}

}  // namespace detail
}  // namespace slam
