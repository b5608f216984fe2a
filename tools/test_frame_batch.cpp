// test_frame_batch -- slam::FrameBatch (include/slam/frontend/frame_batch.hpp, an extension: the reference has no batch
// interface) against the reference-signature calls on the same frames: every frame's keypoints and descriptors must equal
// FeatureDetector::detectAndCompute's, every frame's matches FeatureMatcher::match(desc k, desc k-1, matches, kps k, kps k-1)'s,
// byte for byte.  Prints what a caller gains by handing over the sequence at once.
//   test_frame_batch <feature_detector.yml> <feature_matcher.yml> <repeat> <img0.png> <img1.png> ...   (images of one size)
// `repeat` cycles through the images that often (a longer sequence from the reference's ten KITTI frames).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <string>
#include <vector>

#include <slam/frontend/frame_batch.hpp>

#include <slam/preprocessing/png_gray.hpp>

namespace {
double seconds_since(const std::chrono::steady_clock::time_point& t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s <feature_detector.yml> <feature_matcher.yml> <repeat> <img0.png> <img1.png> ...\n", argv[0]);
        return -1;
    }
    try {
        slam::FeatureDetector detector(argv[1]);
        slam::FeatureMatcher matcher(argv[2]);
        const int repeat = std::max(1, std::atoi(argv[3]));
        std::vector<slam::EigenGrayMatrix> images;
        // "synth:ROWSxCOLS:NRECT:RMIN:RMAX:COUNT" instead of image files: COUNT frames of bench.py's synthetic sequence
        // (slamx_synth_frame) -- e.g. synth:1080x1920:6800:3:24:64 = frames of the headline workload
        int srows = 0, scols = 0, snrect = 0, srmin = 0, srmax = 0, scount = 0;
        if (argc == 5 && std::sscanf(argv[4], "synth:%dx%d:%d:%d:%d:%d", &srows, &scols, &snrect, &srmin, &srmax, &scount) == 6) {
            for (int k = 0; k < scount; ++k) {
                slam::EigenGrayMatrix im(srows, scols);
                if (slamx_synth_frame(0x5EED0051ull, k, srows, scols, snrect, srmin, srmax, im.data()) != SLAMX_OK) {
                    throw std::runtime_error("slamx_synth_frame failed");
                }
                images.push_back(std::move(im));
            }
        }
        for (int i = 4; i < argc && scount == 0; ++i) {
            images.push_back(slamx_tools::read_png_gray(argv[i]));
            if (images.back().rows() != images[0].rows() || images.back().cols() != images[0].cols()) {
                throw std::runtime_error(std::string("image of another size: ") + argv[i]);
            }
        }
        const int n = static_cast<int>(images.size()) * repeat;
        auto image_of = [&](int k) -> const slam::EigenGrayMatrix& { return images[static_cast<size_t>(k) % images.size()]; };

        // ---- the reference's way: one call per frame, one per frame pair
        std::vector<std::vector<slam::Keypoint>> kps(static_cast<size_t>(n));
        std::vector<slam::DescriptorMatrix> desc(static_cast<size_t>(n));
        std::vector<std::vector<slam::Match>> matches(static_cast<size_t>(n));
        detector.detectAndCompute(image_of(0), kps[0], desc[0]);  // warm-up: the library's lazy allocations
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < n; ++k) {
            detector.detectAndCompute(image_of(k), kps[static_cast<size_t>(k)], desc[static_cast<size_t>(k)]);
            if (k > 0) {
                matcher.match(desc[static_cast<size_t>(k)], desc[static_cast<size_t>(k) - 1], matches[static_cast<size_t>(k)],
                              kps[static_cast<size_t>(k)], kps[static_cast<size_t>(k) - 1]);
            }
        }
        const double t_single = seconds_since(t0);

        // ---- the batch
        // (synthetic frames: bench.py's capacities for them; image files: the facade's defaults)
        slam::FrameBatch batch(detector, static_cast<int>(images[0].rows()), static_cast<int>(images[0].cols()), n, scount > 0 ? 32768 : 0,
                               scount > 0 ? 4096 : 0);
        double t_batch = 0.0, t_pipe = 0.0;
        for (int pass = 0; pass < 2; ++pass) {  // pass 0 warms up (first-use allocations of the batch)
            t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < n; ++k) {
                batch.setFrame(k, image_of(k));
            }
            batch.detectAndCompute(n);
            batch.matchConsecutive(matcher, n);
            batch.fetch(n);
            t_batch = seconds_since(t0);
        }
        // what the three calls left, frame by frame, to compare the pipelined run with
        std::vector<std::vector<slam::Keypoint>> sk(static_cast<size_t>(n));
        std::vector<slam::DescriptorMatrix> sd(static_cast<size_t>(n));
        std::vector<std::vector<slam::Match>> sm(static_cast<size_t>(n));
        for (int k = 0; k < n; ++k) {
            batch.keypoints(k, sk[static_cast<size_t>(k)]);
            batch.descriptors(k, sd[static_cast<size_t>(k)]);
            if (k > 0) {
                batch.matches(matcher, k, sm[static_cast<size_t>(k)]);
            }
        }
        // ---- the same sequence PIPELINED: host copy | upload | kernels overlap chunk by chunk (FrameBatch::processSequence)
        std::vector<const slam::EigenGrayMatrix*> seq(static_cast<size_t>(n));
        for (int k = 0; k < n; ++k) {
            seq[static_cast<size_t>(k)] = &image_of(k);
        }
        // (experiment switches: FB_CHUNK = frames per chunk, FB_THREADS = copy threads; 0 / unset = the defaults)
        const int fb_chunk = std::getenv("FB_CHUNK") != nullptr ? std::atoi(std::getenv("FB_CHUNK")) : 0;
        const int fb_threads = std::getenv("FB_THREADS") != nullptr ? std::atoi(std::getenv("FB_THREADS")) : 0;
        std::vector<int> fb_sizes;  // FB_SIZES=16,32,32,...: the chunks' sizes one by one (the last repeats)
        if (const char* e = std::getenv("FB_SIZES")) {
            for (const char* q = e; *q != 0;) {
                fb_sizes.push_back(std::atoi(q));
                while (*q != 0 && *q != ',') {
                    ++q;
                }
                q += *q == ',' ? 1 : 0;
            }
        }
        double t_queued = 0.0;
        for (int pass = 0; pass < 6; ++pass) {
            t0 = std::chrono::steady_clock::now();
            batch.processSequence(seq, &matcher, true, -1, fb_chunk, fb_threads, fb_sizes);
            const double tq = seconds_since(t0);
            batch.fetch(n);
            const double t = seconds_since(t0);
            if (pass == 0 || t < t_pipe) {
                t_pipe = t;
                t_queued = tq;
            }
        }
        if (n <= 64) {
            // a short sequence is one chunk by default: once more in chunks of three frames on three copy threads, so that the
            // comparison below sees a pipeline with chunk borders (frame 3 k matched against a frame of the chunk before)
            batch.processSequence(seq, &matcher, true, -1, 3, 3);
            batch.fetch(n);
        }
        size_t total_kps = 0, total_matches = 0;
        int bad = 0;
        for (int k = 0; k < n; ++k) {
            std::vector<slam::Keypoint> bk;
            slam::DescriptorMatrix bd;
            std::vector<slam::Match> bm;
            batch.keypoints(k, bk);
            batch.descriptors(k, bd);
            if (k > 0) {
                batch.matches(matcher, k, bm);
            }
            const auto& rk = kps[static_cast<size_t>(k)];
            const auto& rd = desc[static_cast<size_t>(k)];
            const auto& rm = matches[static_cast<size_t>(k)];
            const bool same_k = bk.size() == rk.size() && (bk.empty() || std::memcmp(bk.data(), rk.data(), bk.size() * sizeof(slam::Keypoint)) == 0);
            const bool same_d = bd.rows() == rd.rows() && bd.cols() == rd.cols() &&
                                (bd.size() == 0 || std::memcmp(bd.data(), rd.data(), static_cast<size_t>(bd.size())) == 0);
            bool same_m = bm.size() == rm.size() && (bm.empty() || std::memcmp(bm.data(), rm.data(), bm.size() * sizeof(slam::Match)) == 0);
            // ... and the unpipelined batch calls left the same
            const auto& qk = sk[static_cast<size_t>(k)];
            const auto& qd = sd[static_cast<size_t>(k)];
            const auto& qm = sm[static_cast<size_t>(k)];
            same_m = same_m && qk.size() == rk.size() && (qk.empty() || std::memcmp(qk.data(), rk.data(), qk.size() * sizeof(slam::Keypoint)) == 0) &&
                     qd.rows() == rd.rows() && (qd.size() == 0 || std::memcmp(qd.data(), rd.data(), static_cast<size_t>(qd.size())) == 0) &&
                     qm.size() == rm.size() && (qm.empty() || std::memcmp(qm.data(), rm.data(), qm.size() * sizeof(slam::Match)) == 0);
            if (!same_k || !same_d || !same_m) {
                std::fprintf(stderr, "frame %d differs: keypoints %s (%zu / %zu), descriptors %s, matches %s (%zu / %zu)\n", k,
                             same_k ? "equal" : "DIFFER", bk.size(), rk.size(), same_d ? "equal" : "DIFFER", same_m ? "equal" : "DIFFER",
                             bm.size(), rm.size());
                ++bad;
            }
            total_kps += bk.size();
            total_matches += bm.size();
        }
        std::printf("%d frames of %td x %td: %zu keypoints, %zu matches\n", n, images[0].cols(), images[0].rows(), total_kps, total_matches);
        std::printf("frame by frame (detectAndCompute + match): %.3f ms, %.0f frames/s\n", t_single * 1e3, n / t_single);
        std::printf("one FrameBatch, its calls one after another (setFrame x %d, detectAndCompute, matchConsecutive, fetch): %.3f ms, %.0f frames/s\n",
                    n, t_batch * 1e3, n / t_batch);
        std::printf("one FrameBatch, pipelined (processSequence: host copy | upload | kernels chunk by chunk, then fetch): %.3f ms, %.0f frames/s\n",
                    t_pipe * 1e3, n / t_pipe);
        std::printf("   (of that: %.3f ms until the last chunk was queued, %.3f ms waiting in fetch)\n", t_queued * 1e3, (t_pipe - t_queued) * 1e3);
        std::printf("batch == frame by frame: %s\n", bad == 0 ? "yes" : "NO");
        // ---- the pose stage of the same batch, when the configuration directory holds a camera.yml: every frame pair's
        //      essential-matrix RANSAC, cheirality vote, map points and PnP in twelve launches, from the arrays in HBM
        const std::filesystem::path cam = std::filesystem::path(argv[1]).parent_path() / "camera.yml";
        if (std::filesystem::exists(cam)) {
            const slam::Camera camera(cam);
            slam::PoseStageOptions opt;
            std::vector<slam::FramePose> poses;
            t0 = std::chrono::steady_clock::now();
            batch.estimatePoses(camera.intrinsics(), n, opt, poses);
            const double t_pose = seconds_since(t0);
            int valid = 0, sane = 0, forward = 0;
            size_t points = 0;
            for (int k = 1; k < n; ++k) {
                const slam::FramePose& p = poses[static_cast<size_t>(k)];
                if (!p.valid) {
                    continue;
                }
                ++valid;
                double ortho = 0.0;
                for (int i = 0; i < 3; ++i) {
                    for (int j = 0; j < 3; ++j) {
                        double d = 0.0;
                        for (int q = 0; q < 3; ++q) {
                            d += p.R[static_cast<size_t>(3 * q + i)] * p.R[static_cast<size_t>(3 * q + j)];
                        }
                        ortho = std::max(ortho, std::fabs(d - (i == j ? 1.0 : 0.0)));
                    }
                }
                const double tn = std::sqrt(p.t[0] * p.t[0] + p.t[1] * p.t[1] + p.t[2] * p.t[2]);
                std::vector<slam::Point3d> pts;
                std::vector<uint8_t> mask;
                batch.mapPoints(k, pts, &mask);
                int flagged = 0;
                for (uint8_t m : mask) {
                    flagged += m != 0 ? 1 : 0;
                }
                if (ortho < 1e-9 && std::fabs(tn - 1.0) < 1e-9 && p.inliers <= p.matches && static_cast<int>(pts.size()) == p.matches && flagged == p.inliers) {
                    ++sane;
                }
                if (std::fabs(p.t[2]) > 0.9) {
                    ++forward;  // (the KITTI car drives forward; a repeated frame pair has no motion to find)
                }
                points += pts.size();
            }
            std::printf("pose stage (%d + %d hypotheses per pair): %.3f ms for %d pairs, %d with a pose, %d consistent (R orthonormal, |t| = 1, "
                        "inlier flags = inlier count, a map point per match), %d with |t_z| > 0.9, %zu map points\n",
                        opt.iterations, opt.pnpIterations, t_pose * 1e3, n - 1, valid, sane, forward, points);
            if (sane != valid) {
                ++bad;
            }
        }
        return bad == 0 ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "An error occurred: %s\n", e.what());
        return -1;
    }
}
