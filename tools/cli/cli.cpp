// cli -- keeps the reference's flag surface (tools/cli/cli.cpp:20: getopt "hv:c:") but, unlike the
// reference's stub SLAMModel (src/model/model.cpp:5-7 is empty, run() is never defined), actually
// drives the hot path: frame directory -> detectAndCompute -> match(frame k, frame k-1) and, when the
// config directory also holds a camera.yml, PoseEstimator::estimate + triangulatePoints on the matches.
//   cli -c <config dir with feature_detector.yml + feature_matcher.yml [+ camera.yml]> -v <directory of PNG frames>
//       [-p <levels>] [-g <gpus>] [-b <frames>]
// -p (an extension: the reference has no pyramid) runs the detector on an image pyramid of that many levels.
// -g N (an extension: the reference has no multi-GPU code): the sequence is cut into N contiguous
//    chunks of frames, one host thread + one context per GPU (slam::useDevice), each thread recomputing the frame
//    before its chunk (a 1-frame halo) -- no traffic between GPUs while frames are processed -- and at the end the
//    map points triangulated on every GPU are exchanged with ONE all-gather over RCCL (slamx_mapgather_*).
// -b N (an extension: the reference's detector and matcher take one frame per call): frames go through slam::FrameBatch
//    N at a time -- one upload, one detectAndCompute over all of them, one match of every frame against its predecessor,
//    one copy of the results -- and the output is the frame-by-frame run's, line for line (the last frame of a batch is
//    recomputed as the first of the next one, so that every frame has its predecessor in the same batch).
#include <getopt.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <stdexcept>
#include <filesystem>
#include <string>
#include <thread>
#include <vector>

#include <slam/frontend/feature_detector.hpp>
#include <slam/frontend/feature_matcher.hpp>
#include <slam/frontend/frame_batch.hpp>
#include <slam/frontend/pose_estimator.hpp>

#include <slam/preprocessing/png_gray.hpp>

namespace fs = std::filesystem;

static void printHelp() {
    std::printf("cli -c <config_dir> -v <stream_path> [-p <levels>] [-g <gpus>] [-b <frames>]\n"
                "  -c  directory holding feature_detector.yml and feature_matcher.yml\n"
                "  -v  directory of 8-bit PNG frames (processed in lexical order)\n"
                "  -p  detect on an image pyramid of this many levels (scale 1.2); default: full resolution only\n"
                "  -g  shard the frames over this many GPUs (one host thread + context each, RCCL all-gather of the\n"
                "      map points at the end); default 1\n"
                "  -b  hand the frames to the GPU this many at a time (slam::FrameBatch; frames of one size, no pyramid);\n"
                "      default 1: one detectAndCompute and one match call per frame, as the reference's classes work\n"
                "  -h  this help\n");
}

struct ChunkResult {
    std::string log;                // the per-frame lines of the chunk, printed in frame order by main
    size_t keypoints = 0, matches = 0;
    std::vector<double> mapPoints;  // x y z of every point triangulated in the chunk
    std::vector<int32_t> gatheredCounts;
    int gatheredTotal = -1;         // points of ALL ranks after the gather (-1: no gather ran)
    std::exception_ptr error;
};

// Frames [lo, hi) of `files` on GPU `rank` (of `world`); frame lo - 1 is recomputed as the predecessor of frame lo.
static void runChunk(int rank, int world, const uint8_t* gatherId, const std::string& configPath, const std::vector<fs::path>& files,
                     size_t lo, size_t hi, int pyramidLevels, int batchFrames, ChunkResult* out) {
    slamx_mapgather* gather = nullptr;
    try {
        slamx_ctx* ctx = slam::useDevice(rank);  // everything this thread creates below lives on GPU `rank`
        if (world > 1 || gatherId != nullptr) {
            slam::detail::check(slamx_mapgather_create(ctx, gatherId, rank, world, &gather));  // every rank must arrive here
            // the capacity every rank's block has in the one-collective exchange at the end: an equal share of the 2^22
            // doubles the gathered points may fill (1.4 M points in all)
            slam::detail::check(slamx_mapgather_reserve(gather, (1 << 22) / 3 / world));
        }
        slam::FeatureDetector detector(fs::path(configPath) / "feature_detector.yml");
        slam::FeatureMatcher matcher(fs::path(configPath) / "feature_matcher.yml");
        const bool havePose = fs::exists(fs::path(configPath) / "camera.yml");
        slam::Camera camera = havePose ? slam::Camera(fs::path(configPath) / "camera.yml") : slam::Camera();
        slam::PoseEstimator poseEstimator(camera);
        std::vector<slam::Keypoint> prevKps;
        slam::DescriptorMatrix prevDesc;
        auto detect = [&](size_t i, std::vector<slam::Keypoint>& kps, slam::DescriptorMatrix& desc) {
            const slam::EigenGrayMatrix image = slamx_tools::read_png_gray(files[i].string());
            if (pyramidLevels > 1) {
                detector.detectAndComputePyramid(image, kps, desc, pyramidLevels, 1.2);
            } else {
                detector.detectAndCompute(image, kps, desc);
            }
        };
        if (lo > 0 && lo < hi && batchFrames <= 1) {
            detect(lo - 1, prevKps, prevDesc);  // the halo: owned (and reported) by the previous chunk
        }
        char line[512];
        // what a frame contributes to the output, given its results and its predecessor's
        auto report = [&](size_t i, const std::vector<slam::Keypoint>& kps, const std::vector<slam::Match>& matches) {
            std::snprintf(line, sizeof(line), "frame %zu %s: %zu keypoints, %zu matches to previous\n", i,
                          files[i].filename().c_str(), kps.size(), matches.size());
            out->log += line;
            if (havePose && matches.size() >= 8) {
                std::vector<std::pair<int, int>> pairs;
                for (const auto& m : matches) {
                    pairs.emplace_back(m.trainIdx, m.queryIdx);  // (previous frame, this frame)
                }
                slam::Mat3d R{1, 0, 0, 0, 1, 0, 0, 0, 1};
                slam::Vec3d t{0, 0, 0};
                if (poseEstimator.estimate(prevKps, kps, pairs, R, t)) {
                    std::vector<slam::Match> swapped;
                    for (const auto& m : matches) {
                        swapped.push_back({m.trainIdx, m.queryIdx, m.distance});
                    }
                    const auto pts = poseEstimator.triangulatePoints(prevKps, kps, swapped, R, t);
                    std::snprintf(line, sizeof(line), "  pose: t = (%.4f %.4f %.4f), trace(R) = %.6f, %zu map points\n", t[0], t[1],
                                  t[2], R[0] + R[4] + R[8], pts.size());
                    out->log += line;
                    for (const auto& p : pts) {
                        out->mapPoints.insert(out->mapPoints.end(), {p.x, p.y, p.z});
                    }
                }
            }
            out->keypoints += kps.size();
            out->matches += matches.size();
        };
        if (batchFrames > 1 && lo < hi) {
            // slot 0 of every batch is the predecessor of its first new frame (frame lo - 1 for the first batch, the last
            // frame of the batch before otherwise): recomputed, never reported twice
            const slam::EigenGrayMatrix first = slamx_tools::read_png_gray(files[lo > 0 ? lo - 1 : lo].string());
            slam::FrameBatch batch(detector, static_cast<int>(first.rows()), static_cast<int>(first.cols()), batchFrames + 1);
            slam::EigenGrayMatrix carry = first;  // image of the frame that goes into slot 0
            bool haveCarry = lo > 0;
            for (size_t i0 = lo; i0 < hi;) {
                const int slot0 = haveCarry ? 1 : 0;
                if (haveCarry) {
                    batch.setFrame(0, carry);
                }
                const size_t count = std::min(hi - i0, static_cast<size_t>(batchFrames));
                for (size_t k = 0; k < count; ++k) {
                    carry = slamx_tools::read_png_gray(files[i0 + k].string());
                    batch.setFrame(slot0 + static_cast<int>(k), carry);
                }
                const int n = slot0 + static_cast<int>(count);
                batch.detectAndCompute(n);
                batch.matchConsecutive(matcher, n);
                batch.fetch(n);
                if (haveCarry) {
                    batch.keypoints(0, prevKps);
                }
                for (size_t k = 0; k < count; ++k) {
                    const int slot = slot0 + static_cast<int>(k);
                    std::vector<slam::Keypoint> kps;
                    std::vector<slam::Match> matches;
                    batch.keypoints(slot, kps);
                    if (slot > 0 && !kps.empty() && !prevKps.empty()) {
                        batch.matches(matcher, slot, matches);
                    }
                    report(i0 + k, kps, matches);
                    prevKps = std::move(kps);
                }
                i0 += count;
                haveCarry = true;
            }
        } else {
            for (size_t i = lo; i < hi; ++i) {
                std::vector<slam::Keypoint> kps;
                slam::DescriptorMatrix desc;
                detect(i, kps, desc);
                std::vector<slam::Match> matches;
                if (i > 0 && !kps.empty() && !prevKps.empty()) {
                    matcher.match(desc, prevDesc, matches, kps, prevKps);
                }
                report(i, kps, matches);
                prevKps = std::move(kps);
                prevDesc = std::move(desc);
            }
        }
    } catch (...) {
        out->error = std::current_exception();
    }
    // The gather is entered even after a local failure (with zero points): no rank may be left waiting in a collective.
    if (gather != nullptr) {
        try {
            const int n = out->error ? 0 : static_cast<int>(out->mapPoints.size() / 3);
            out->gatheredCounts.assign(static_cast<size_t>(world), 0);
            std::vector<double> all(static_cast<size_t>(1) << 22);
            int total = 0;
            // ONE collective: the per-rank capacity was agreed (reserved) right after the communicator was made
            slam::detail::check(slamx_mapgather_allgather_fixed(gather, out->mapPoints.data(), n, out->gatheredCounts.data(), all.data(),
                                                                static_cast<int>(all.size() / 3), &total));
            out->gatheredTotal = total;
        } catch (...) {
            if (!out->error) {
                out->error = std::current_exception();
            }
        }
        slamx_mapgather_destroy(gather);
    }
}

int main(int argc, char** argv) {
    int c{};
    std::string streamPath{};
    std::string configPath{};
    int pyramidLevels = 0;
    int gpus = 1;
    int batchFrames = 1;
    bool gpusGiven = false;
    while ((c = getopt(argc, argv, "hv:c:p:g:b:")) != -1) {
        switch (c) {
            case 'h':
                printHelp();
                return 0;
            case 'v':
                streamPath = optarg;
                break;
            case 'c':
                configPath = optarg;
                break;
            case 'p':
                pyramidLevels = std::atoi(optarg);
                break;
            case 'g':
                gpus = std::atoi(optarg);
                gpusGiven = true;
                break;
            case 'b':
                batchFrames = std::atoi(optarg);
                break;
            default:
                printHelp();
                return 0;
        }
    }
    if (streamPath.empty() || configPath.empty() || gpus < 1 || batchFrames < 1) {
        printHelp();
        return 0;
    }
    if (batchFrames > 1 && pyramidLevels > 1) {
        std::fprintf(stderr, "An error occurred: -b and -p cannot be combined (the batch interface of the facade has no pyramid)\n");
        return -1;
    }
    try {
        // Preprocessor::prepareDirectory, src/preprocessing/preprocessor.cpp:24-41
        std::vector<fs::path> files;
        for (const auto& entry : fs::directory_iterator(streamPath)) {
            if (entry.is_regular_file() && entry.path().extension() == ".png") {
                files.push_back(entry.path());
            }
        }
        std::sort(files.begin(), files.end());
        const int world = std::max(1, std::min<int>(gpus, static_cast<int>(files.size())));
        if (world > slamx_device_count()) {
            // refused here, before any thread exists: a rank that cannot get its GPU would leave the others waiting
            // in the communicator's rendezvous
            throw std::runtime_error("-g " + std::to_string(gpus) + ": only " + std::to_string(slamx_device_count()) +
                                     " GPU(s) visible");
        }
        if (world > 1) {
            // ... and every rank's context is created HERE, so that a GPU that cannot be opened fails the run before a
            // thread waits for it in ncclCommInitRank (the threads find these contexts through slam::useDevice)
            for (int r = 0; r < world; ++r) {
                (void)slam::detail::context_of_device(r);
            }
        }
        std::vector<uint8_t> id(SLAMX_MAPGATHER_ID_BYTES, 0);
        if (gpusGiven) {
            slam::detail::check(slamx_mapgather_unique_id(id.data()));  // handed to every rank's thread
        }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<ChunkResult> results(static_cast<size_t>(world));
        std::vector<std::thread> threads;
        for (int r = 0; r < world; ++r) {
            const size_t lo = files.size() * static_cast<size_t>(r) / static_cast<size_t>(world);
            const size_t hi = files.size() * static_cast<size_t>(r + 1) / static_cast<size_t>(world);
            if (world == 1) {
                runChunk(0, 1, gpusGiven ? id.data() : nullptr, configPath, files, lo, hi, pyramidLevels, batchFrames, &results[0]);
            } else {
                threads.emplace_back(runChunk, r, world, id.data(), std::cref(configPath), std::cref(files), lo, hi, pyramidLevels,
                                     batchFrames, &results[static_cast<size_t>(r)]);
            }
        }
        for (auto& th : threads) {
            th.join();
        }
        size_t totalKps = 0, totalMatches = 0;
        for (const auto& r : results) {
            if (r.error) {
                std::rethrow_exception(r.error);
            }
            std::fputs(r.log.c_str(), stdout);
            totalKps += r.keypoints;
            totalMatches += r.matches;
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (results[0].gatheredTotal >= 0) {
            std::printf("%d GPU%s, map points gathered over RCCL: %d (per GPU:", world, world > 1 ? "s" : "", results[0].gatheredTotal);
            for (int32_t n : results[0].gatheredCounts) {
                std::printf(" %d", n);
            }
            std::printf(")\n");
        }
        std::printf("%zu frames, %zu keypoints, %zu matches in %.3f s\n", files.size(), totalKps, totalMatches, dt);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "An error occurred: %s\n", e.what());
        return -1;
    }
    return 0;
}
