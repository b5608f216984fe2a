// slam::Backend -- facade over the reference's declaration-only class (include/slam/backend/backend.hpp:10-18:
// `explicit Backend(slam::Map& map)`, `void run()  // starts the backend thread, performs the optimisations`,
// a reference to the map and a std::thread).  run() here does what the comment promises: on the backend
// thread it takes the newest window of keyframes (<= 10) with every map point they observe, runs
// damped Gauss-Newton steps on the GPU (slamx_gn_*: J^T J / J^T r and the landmark elimination on the FP64
// matrix cores, see include/slamx.h) and writes the optimised poses and points back into the map.
// The optimisation itself has no counterpart in the reference: build-defined.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <exception>
#include <functional>
#include <limits>
#include <stdexcept>
#include <thread>
#include <string>
#include <vector>

#include <slam/backend/map.hpp>
#include <slam/frontend/slamx_runtime.hpp>

namespace slam {

class Backend {
public:
    struct Options {
        double fx = 0, fy = 0, cx = 0, cy = 0;  // pinhole intrinsics (Camera K0: K(0,0) K(1,1) K(0,2) K(1,2))
        int window = 10;                        // newest keyframes optimised together (<= SLAMX_GN_MAX_CAMS)
        int heldKeyframes = 2;                  // oldest keyframes of the window kept fixed (gauge + scale)
        int iterations = 5;
        double damping = 1e-4;                  // lambda added to every diagonal
    };

    explicit Backend(slam::Map& map) : m_map(map) {}
    Backend(slam::Map& map, const Options& o) : m_map(map), m_options(o) {}
    ~Backend() { wait(); }
    Backend(const Backend&) = delete;
    Backend& operator=(const Backend&) = delete;

    void setOptions(const Options& o) { m_options = o; }

    // Starts the backend thread; one optimisation pass per call.  join() waits for it and rethrows what the pass
    // threw (an exception must not leave the thread function: that would be std::terminate); the destructor waits
    // and drops it.  Intrinsics are checked here, before a thread exists: the reference-signature constructor
    // `Backend(Map&)` leaves them at zero, and setOptions() must be called first.
    void run() {
        join();
        if (!(m_options.fx > 0.0) || !(m_options.fy > 0.0)) {
            throw std::invalid_argument("slam::Backend::run: focal lengths must be positive (setOptions / Options ctor)");
        }
        m_failed = false;
        m_error = nullptr;
        m_backendThread = std::thread([this] {
            try {
                optimise();
            } catch (...) {
                m_error = std::current_exception();
                m_failed = true;
            }
        });
    }
    void join() {
        wait();
        if (m_error) {
            std::exception_ptr e = m_error;
            m_error = nullptr;
            std::rethrow_exception(e);
        }
    }
    // true iff the last pass ended in an error that join() has reported or will report (callable while the pass runs:
    // the flag is atomic; join() itself belongs to ONE thread)
    bool failed() const { return m_failed; }
    // sum of squared reprojection residuals before the first and after the last step of the last pass
    double initialCost() const { return m_cost0; }
    double finalCost() const { return m_cost1; }

private:
    void wait() {
        if (m_backendThread.joinable()) {
            m_backendThread.join();
        }
    }
    void optimise() {
        const Options o = m_options;
        std::vector<Map::Keyframe> kfs;
        std::vector<std::array<double, 3>> pts;
        std::vector<Map::Observation> obs;
        {
            std::lock_guard<std::mutex> lock(m_map.get().m_mapMutex);
            kfs = m_map.get().m_keyframes;
            pts = m_map.get().m_points;
            obs = m_map.get().m_observations;
        }
        const int W = std::min<int>({o.window, SLAMX_GN_MAX_CAMS, static_cast<int>(kfs.size())});
        if (W < 1 || pts.empty()) {
            return;
        }
        const int first = static_cast<int>(kfs.size()) - W;
        // landmarks of the window: every map point seen from one of its keyframes
        std::vector<int> slot(pts.size(), -1);
        std::vector<int> ids;
        for (const auto& ob : obs) {
            if (ob.keyframe >= first && slot[static_cast<std::size_t>(ob.point)] < 0) {
                slot[static_cast<std::size_t>(ob.point)] = static_cast<int>(ids.size());
                ids.push_back(ob.point);
            }
        }
        const int L = static_cast<int>(ids.size());
        if (L == 0) {
            return;
        }
        std::vector<double> poses(static_cast<std::size_t>(W) * 12), X(static_cast<std::size_t>(L) * 3);
        std::vector<float> track(static_cast<std::size_t>(L) * W * 2, std::numeric_limits<float>::quiet_NaN());
        for (int c = 0; c < W; ++c) {
            const auto& k = kfs[static_cast<std::size_t>(first + c)];
            std::copy(k.R.begin(), k.R.end(), poses.begin() + 12 * c);
            std::copy(k.t.begin(), k.t.end(), poses.begin() + 12 * c + 9);
        }
        for (int l = 0; l < L; ++l) {
            std::copy(pts[static_cast<std::size_t>(ids[static_cast<std::size_t>(l)])].begin(),
                      pts[static_cast<std::size_t>(ids[static_cast<std::size_t>(l)])].end(), X.begin() + 3 * l);
        }
        for (const auto& ob : obs) {
            if (ob.keyframe >= first) {
                const std::size_t at = (static_cast<std::size_t>(slot[static_cast<std::size_t>(ob.point)]) * W + (ob.keyframe - first)) * 2;
                track[at] = ob.u;
                track[at + 1] = ob.v;
            }
        }
        slamx_gn* gn = nullptr;
        detail::check(slamx_gn_create(detail::default_context(), 1, W, L, &gn));
        struct Guard {
            slamx_gn* g;
            ~Guard() { slamx_gn_destroy(g); }
        } guard{gn};
        detail::check(slamx_gn_upload(gn, poses.data(), X.data(), track.data(), nullptr));
        const double K[4] = {o.fx, o.fy, o.cx, o.cy};
        const int its = std::max(0, std::min(o.iterations, SLAMX_GN_MAX_ITERATIONS));
        detail::check(slamx_gn_iterate(gn, K, o.damping, std::min(o.heldKeyframes, W), its));
        std::vector<double> cost(static_cast<std::size_t>(its) + 1);
        int32_t status = 0;
        detail::check(slamx_gn_download(gn, poses.data(), X.data(), cost.data(), &status));
        if (status != 0) {
            // bit 0: a step was skipped (reduced system not positive definite); nothing is written back
            throw std::runtime_error("slam::Backend: Gauss-Newton window status " + std::to_string(status) +
                                     " (reduced camera system not positive definite); the map is unchanged");
        }
        m_cost0 = cost.front();
        m_cost1 = cost.back();
        std::lock_guard<std::mutex> lock(m_map.get().m_mapMutex);
        for (int c = 0; c < W; ++c) {
            auto& k = m_map.get().m_keyframes[static_cast<std::size_t>(first + c)];
            std::copy(poses.begin() + 12 * c, poses.begin() + 12 * c + 9, k.R.begin());
            std::copy(poses.begin() + 12 * c + 9, poses.begin() + 12 * c + 12, k.t.begin());
        }
        for (int l = 0; l < L; ++l) {
            std::copy(X.begin() + 3 * l, X.begin() + 3 * l + 3, m_map.get().m_points[static_cast<std::size_t>(ids[static_cast<std::size_t>(l)])].begin());
        }
    }

    // not const: the backend calls non-const functions of Map (as the reference notes)
    std::reference_wrapper<slam::Map> m_map;
    std::thread m_backendThread;
    Options m_options;
    double m_cost0 = 0.0, m_cost1 = 0.0;
    std::exception_ptr m_error;  // written by the backend thread, read after it has been joined
    std::atomic<bool> m_failed{false};  // written by the backend thread, read by failed() from any thread
};

}  // namespace slam
