// slamx_runtime.hpp -- shared plumbing of the C++ facade: one process-wide GPU context and the
// status-code -> exception mapping (std::invalid_argument for empty descriptors,
// std::runtime_error for everything else, as the reference throws).
#pragma once

#include <cstdlib>
#include <map>
#include <mutex>
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

// The context the facade objects of the CALLING THREAD are created on.  Default: one process-wide context on
// device 0 (or SLAMX_DEVICE), created on first use, living for the process -- the reference's single-threaded
// model.  A multi-GPU host runs one thread per GPU and calls slam::useDevice(g) first in each (tools/cli -g N):
// everything the thread constructs or calls afterwards uses that GPU's context (one per device, shared by the
// threads that name the same device, created on first use).
inline slamx_ctx*& thread_context() {
    thread_local slamx_ctx* ctx = nullptr;
    return ctx;
}

inline slamx_ctx* context_of_device(int device) {
    static std::mutex m;
    static std::map<int, slamx_ctx*> contexts;
    std::lock_guard<std::mutex> lock(m);
    auto it = contexts.find(device);
    if (it == contexts.end()) {
        slamx_ctx* c = nullptr;
        check(slamx_ctx_create(device, nullptr, &c));
        it = contexts.emplace(device, c).first;
    }
    return it->second;
}

inline slamx_ctx* default_context() {
    if (thread_context() != nullptr) {
        return thread_context();
    }
    static slamx_ctx* ctx = [] {
        int dev = 0;
        if (const char* e = std::getenv("SLAMX_DEVICE")) {
            dev = std::atoi(e);
        }
        return context_of_device(dev);
    }();
    return ctx;
}

}  // namespace detail

// Binds the calling thread to GPU `device` (see detail::thread_context); returns its context for C-ABI calls.
inline slamx_ctx* useDevice(int device) {
    detail::thread_context() = detail::context_of_device(device);
    return detail::thread_context();
}
}  // namespace slam
