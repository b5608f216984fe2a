// reference_call_sequence -- the reference's OWN call sequences against the pose / backend interfaces, written with
// the reference's types (cv::Mat, cv::KeyPoint, cv::DMatch, cv::Point3d, Eigen::Vector3d, Eigen::Matrix4d):
//   test/frontend/test_pose_estimator.cpp:113-177   detectAndCompute x2 -> match -> convert to cv types ->
//                                                   PoseEstimator::estimate(pairs1, pairs2, matches, R, t) ->
//                                                   isRotationMatrix(R) -> triangulatePoints(kps1, kps2, matches, R, t)
//   test/backend/test_loop_closure.cpp:46-79        addKeyframe(i, desc, kps, std::vector<Eigen::Vector3d>) x N -> detect
// Built with -DSLAMX_USE_OPENCV -DSLAMX_USE_EIGEN.  In this image the two libraries are absent, so the Makefile
// puts tests/compile_check/ (API-subset headers, NOT the libraries) on the include path; with the real libraries
// that directory is simply left out.  Every reference-typed result is compared with the plain-typed overload of
// the same call: exit code 0 iff they agree bit for bit and R is a rotation (the reference's only check,
// test_pose_estimator.cpp:34-43, 1e-6).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include <slam/backend/loop_closure.hpp>
#include <slam/frontend/feature_detector.hpp>
#include <slam/frontend/feature_matcher.hpp>
#include <slam/frontend/pose_estimator.hpp>
#include <slam/preprocessing/png_gray.hpp>

#if !defined(SLAMX_HAVE_OPENCV) || !defined(SLAMX_HAVE_EIGEN)
#error "build with -DSLAMX_USE_OPENCV -DSLAMX_USE_EIGEN and OpenCV / Eigen (or tests/compile_check) on the include path"
#endif

// test_pose_estimator.cpp:34-43
static bool isRotationMatrix(const cv::Mat& R) {
    if (R.empty()) {
        return false;
    }
    double worst = 0.0;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) {
                s += R.at<double>(k, i) * R.at<double>(k, j);
            }
            worst = std::fmax(worst, std::fabs(s - (i == j ? 1.0 : 0.0)));
        }
    }
    return worst < 1e-6;
}

static slam::EigenGrayMatrix load(const std::string& path) { return slamx_tools::read_png_gray(path); }

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <data dir with camera.yml, feature_*.yml, loop_closure.yml, images/>\n", argv[0]);
        return 2;
    }
    const std::string data = argv[1];
    slam::Camera camera(data + "/camera.yml");
    slam::PoseEstimator pose_estimator(camera);
    slam::FeatureDetector detector(data + "/feature_detector.yml");
    slam::FeatureMatcher matcher(data + "/feature_matcher.yml");
    const slam::EigenGrayMatrix img1 = load(data + "/images/0000000000.png");
    const slam::EigenGrayMatrix img2 = load(data + "/images/0000000001.png");
    std::vector<slam::Keypoint> slamKps1, slamKps2;
    slam::DescriptorMatrix slamDesc1, slamDesc2;
    detector.detectAndCompute(img1, slamKps1, slamDesc1);
    detector.detectAndCompute(img2, slamKps2, slamDesc2);
    std::vector<slam::Match> slamMatches;
    matcher.match(slamDesc1, slamDesc2, slamMatches);
    // convertToCvKeypoints / convertToCvMatches of the reference test (:46-67)
    std::vector<cv::KeyPoint> kps1, kps2;
    for (const auto& k : slamKps1) {
        kps1.emplace_back(k.x, k.y, k.size, k.angle, k.response);
    }
    for (const auto& k : slamKps2) {
        kps2.emplace_back(k.x, k.y, k.size, k.angle, k.response);
    }
    std::vector<cv::DMatch> all_matches;
    for (const auto& m : slamMatches) {
        all_matches.emplace_back(m.queryIdx, m.trainIdx, m.distance);
    }
    std::vector<slam::KeyDescriptorPair> pairs1, pairs2;
    for (const auto& k : kps1) {
        pairs1.push_back({k, cv::Mat()});  // estimate() never reads the descriptor half
    }
    for (const auto& k : kps2) {
        pairs2.push_back({k, cv::Mat()});
    }
    std::vector<std::pair<int, int>> matches_for_estimation;
    for (const auto& m : all_matches) {
        matches_for_estimation.push_back({m.queryIdx, m.trainIdx});
    }
    int bad = 0;
    // fewer than 8 matches: R, t untouched (pose_estimator.cpp:22-26)
    {
        cv::Mat R0, t0;
        const std::vector<std::pair<int, int>> few(matches_for_estimation.begin(), matches_for_estimation.begin() + 7);
        pose_estimator.estimate(pairs1, pairs2, few, R0, t0);
        if (!R0.empty() || !t0.empty()) {
            std::printf("FAIL: estimate with 7 matches touched R / t\n");
            ++bad;
        }
    }
    cv::Mat R, t;
    pose_estimator.estimate(pairs1, pairs2, matches_for_estimation, R, t);
    if (!isRotationMatrix(R) || t.rows != 3 || t.cols != 1 || t.type() != CV_64F) {
        std::printf("FAIL: estimate did not return a rotation / a 3x1 CV_64F translation\n");
        return 1;
    }
    // the same call through the plain-typed overload
    slam::Mat3d Rp{};
    slam::Vec3d tp{};
    pose_estimator.estimate(slamKps1, slamKps2, matches_for_estimation, Rp, tp);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            bad += R.at<double>(i, j) != Rp[static_cast<size_t>(i * 3 + j)];
        }
        bad += t.at<double>(i, 0) != tp[static_cast<size_t>(i)];
    }
    const std::vector<cv::Point3d> points_3d = pose_estimator.triangulatePoints(kps1, kps2, all_matches, R, t);
    const std::vector<slam::Point3d> points_p = pose_estimator.triangulatePoints(slamKps1, slamKps2, slamMatches, Rp, tp);
    bad += points_3d.size() != points_p.size() || points_3d.size() != all_matches.size();
    int in_front = 0;
    for (size_t i = 0; i < points_3d.size() && i < points_p.size(); ++i) {
        bad += points_3d[i].x != points_p[i].x || points_3d[i].y != points_p[i].y || points_3d[i].z != points_p[i].z;
        in_front += points_3d[i].z > 0;
    }
    // simpleRecoverPose with cv types, on the essential matrix of the motion just found: E = [t]x R
    {
        cv::Mat E(3, 3, CV_64F), K(3, 3, CV_64F), R2, t2;
        const double tx[9] = {0, -tp[2], tp[1], tp[2], 0, -tp[0], -tp[1], tp[0], 0};
        const slam::Mat3d Kp = camera.intrinsics();
        slam::Mat3d Ep{};
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) {
                    s += tx[i * 3 + k] * Rp[static_cast<size_t>(k * 3 + j)];
                }
                E.at<double>(i, j) = s;
                Ep[static_cast<size_t>(i * 3 + j)] = s;
                K.at<double>(i, j) = camera.getIntrinsicMatrix()(i, j);  // const Eigen::Matrix3d&, as in the reference
                bad += K.at<double>(i, j) != Kp[static_cast<size_t>(i * 3 + j)];
            }
        }
        std::vector<cv::Point2f> n1, n2;
        std::vector<slam::Point2f> m1, m2;
        for (const auto& m : matches_for_estimation) {
            const cv::Point2f a((kps1[static_cast<size_t>(m.first)].pt.x - static_cast<float>(Kp[2])) / static_cast<float>(Kp[0]),
                                (kps1[static_cast<size_t>(m.first)].pt.y - static_cast<float>(Kp[5])) / static_cast<float>(Kp[4]));
            const cv::Point2f b((kps2[static_cast<size_t>(m.second)].pt.x - static_cast<float>(Kp[2])) / static_cast<float>(Kp[0]),
                                (kps2[static_cast<size_t>(m.second)].pt.y - static_cast<float>(Kp[5])) / static_cast<float>(Kp[4]));
            n1.push_back(a);
            n2.push_back(b);
            m1.push_back({a.x, a.y});
            m2.push_back({b.x, b.y});
        }
        simpleRecoverPose(E, n1, n2, K, R2, t2);
        slam::Mat3d R2p{};
        slam::Vec3d t2p{};
        simpleRecoverPose(Ep, m1, m2, Kp, R2p, t2p);
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                bad += R2.at<double>(i, j) != R2p[static_cast<size_t>(i * 3 + j)];
            }
            bad += t2.at<double>(i, 0) != t2p[static_cast<size_t>(i)];
        }
        bad += !isRotationMatrix(R2);
    }
    std::printf("pose: %zu matches, %zu points triangulated, %d in front, trace(R) = %.6f, cv-typed == plain-typed: %s\n",
                matches_for_estimation.size(), points_3d.size(), in_front,
                R.at<double>(0, 0) + R.at<double>(1, 1) + R.at<double>(2, 2), bad == 0 ? "yes" : "NO");
    // ---- test_loop_closure.cpp:46-79 with std::vector<Eigen::Vector3d> map points
    slam::LoopClosure lcEigen("unused.fbow", data + "/loop_closure.yml", matcher, 7);
    slam::LoopClosure lcPlain("unused.fbow", data + "/loop_closure.yml", matcher, 7);
    std::vector<slam::Keypoint> lastKeypoints;
    slam::DescriptorMatrix lastDescriptors;
    for (int i = 0; i < 10; ++i) {
        char name[32];
        std::snprintf(name, sizeof(name), "/images/%010d.png", i);
        const slam::EigenGrayMatrix img = load(data + name);
        std::vector<slam::Keypoint> kps;
        slam::DescriptorMatrix desc;
        detector.detectAndCompute(img, kps, desc);
        std::vector<Eigen::Vector3d> mapPoints;
        std::vector<slam::Vec3d> mapPlain;
        for (const auto& kp : kps) {
            mapPoints.emplace_back(kp.x, kp.y, 1.0);
            mapPlain.push_back({kp.x, kp.y, 1.0});
        }
        lcEigen.addKeyframe(i, desc, kps, mapPoints);
        lcPlain.addKeyframe(i, desc, kps, mapPlain);
        if (i == 9) {
            lastKeypoints = kps;
            lastDescriptors = desc;
        }
    }
    const auto re = lcEigen.detect(lastDescriptors, lastKeypoints, camera);
    const auto rp = lcPlain.detect(lastDescriptors, lastKeypoints, camera);
    bad += re.has_value() != rp.has_value();
    if (re && rp) {
        bad += re->matchedKeyframeId != rp->matchedKeyframeId;
        const Eigen::Matrix4d& T = re->relativeTransform;  // the reference's type (loop_closure.hpp:19)
        bad += T(3, 3) != 1.0;
    }
    std::printf("loop closure: %s%s; Eigen-typed == plain-typed: %s\n", re ? "keyframe " : "no loop",
                re ? std::to_string(re->matchedKeyframeId).c_str() : "", bad == 0 ? "yes" : "NO");
    return bad == 0 ? 0 : 1;
}
