// slam::Map -- facade over the reference's declaration-only class (include/slam/backend/map.hpp:9-21:
// `Map()`, `insertKeyframe(/* ... */)`, `insertMapPoint(/* ... */)`, a mutex, "getters?").  The reference
// never says what a keyframe or a map point holds; this build stores exactly what its backend
// (slam::Backend -> slamx_gn_*) optimises: keyframe poses (world -> camera), map points (world), and the
// pixel observations that tie them together.  Thread-safe the way the reference intends (one mutex).
#pragma once

#include <array>
#include <cstddef>
#include <mutex>
#include <stdexcept>
#include <vector>

namespace slam {

class Backend;

class Map {
public:
    struct Keyframe {
        std::array<double, 9> R;  // row-major, world -> camera
        std::array<double, 3> t;
    };
    struct Observation {
        int keyframe;
        int point;
        float u, v;  // pixel coordinates of the map point in the keyframe
    };

    Map() = default;

    // returns the index of the new keyframe / map point
    int insertKeyframe(const std::array<double, 9>& R, const std::array<double, 3>& t) {
        std::lock_guard<std::mutex> lock(m_mapMutex);
        m_keyframes.push_back(Keyframe{R, t});
        return static_cast<int>(m_keyframes.size()) - 1;
    }
    int insertMapPoint(const std::array<double, 3>& xyz) {
        std::lock_guard<std::mutex> lock(m_mapMutex);
        m_points.push_back(xyz);
        return static_cast<int>(m_points.size()) - 1;
    }
    // throws std::out_of_range for a keyframe / map point that does not exist (the backend indexes with both)
    void addObservation(int keyframe, int point, float u, float v) {
        std::lock_guard<std::mutex> lock(m_mapMutex);
        if (keyframe < 0 || static_cast<std::size_t>(keyframe) >= m_keyframes.size() || point < 0 ||
            static_cast<std::size_t>(point) >= m_points.size()) {
            throw std::out_of_range("slam::Map::addObservation: no such keyframe / map point");
        }
        m_observations.push_back(Observation{keyframe, point, u, v});
    }

    std::size_t keyframeCount() const {
        std::lock_guard<std::mutex> lock(m_mapMutex);
        return m_keyframes.size();
    }
    std::size_t mapPointCount() const {
        std::lock_guard<std::mutex> lock(m_mapMutex);
        return m_points.size();
    }
    Keyframe keyframe(int i) const {
        std::lock_guard<std::mutex> lock(m_mapMutex);
        return m_keyframes.at(static_cast<std::size_t>(i));
    }
    std::array<double, 3> mapPoint(int i) const {
        std::lock_guard<std::mutex> lock(m_mapMutex);
        return m_points.at(static_cast<std::size_t>(i));
    }

private:
    friend class Backend;
    std::vector<Keyframe> m_keyframes;
    std::vector<std::array<double, 3>> m_points;
    std::vector<Observation> m_observations;
    mutable std::mutex m_mapMutex;
};

}  // namespace slam
