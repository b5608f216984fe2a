// pyramid_oracle.cpp -- CPU restatement of the build-defined pyramid downscale (pyramid.hip).
//
// TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED / NO REFERENCE COUNTERPART: the reference has no image
// pyramid (src/frontend/feature_detector.cpp:8-18 detects on the full-resolution image; SURVEY.md
// section 0).  The arithmetic is defined by this build and is integer-only, so the
// GPU result is compared bit for bit.  Written independently of the GPU's table-driven form: the
// source position is evaluated per pixel in exact rational arithmetic.
#include <cmath>
#include <cstdint>

extern "C" {

// level size rule shared with the product: cvRound(size / scale_factor^level), computed in double
int oracle_pyramid_level_size(int size, double scale_factor, int level) {
    return static_cast<int>(std::lrint(size / std::pow(scale_factor, level)));
}

// dst (drows x dcols, dense) = bilinear downscale of src (rows x cols, dense), 11-bit weights:
//   s = ((2 d + 1) S - D) / (2 D) per axis; i0 = floor(s) clamped to [0, S - 1]; f = floor(frac(s) 2048), 0 at the clamps
//   v = ((a (2048 - fx) + b fx)(2048 - fy) + (c (2048 - fx) + d fx) fy + 2^21) >> 22
void oracle_downscale(const uint8_t* src, int rows, int cols, uint8_t* dst, int drows, int dcols) {
    auto pos = [](int d, int D, int S, int* i0, int* f) {
        const long long num = (2LL * d + 1) * S - D;
        const long long den = 2LL * D;
        if (num <= 0) {
            *i0 = 0;
            *f = 0;
        } else {
            *i0 = static_cast<int>(num / den);
            *f = static_cast<int>(((num - (num / den) * den) * 2048) / den);
        }
        if (*i0 >= S - 1) {
            *i0 = S - 1;
            *f = 0;
        }
    };
    for (int y = 0; y < drows; ++y) {
        int y0, fy;
        pos(y, drows, rows, &y0, &fy);
        const int y1 = y0 + 1 < rows ? y0 + 1 : y0;
        for (int x = 0; x < dcols; ++x) {
            int x0, fx;
            pos(x, dcols, cols, &x0, &fx);
            const int x1 = x0 + 1 < cols ? x0 + 1 : x0;
            const long long a = src[static_cast<size_t>(y0) * cols + x0], b = src[static_cast<size_t>(y0) * cols + x1];
            const long long c = src[static_cast<size_t>(y1) * cols + x0], d = src[static_cast<size_t>(y1) * cols + x1];
            const long long top = a * (2048 - fx) + b * fx;
            const long long bot = c * (2048 - fx) + d * fx;
            dst[static_cast<size_t>(y) * dcols + x] = static_cast<uint8_t>((top * (2048 - fy) + bot * fy + (1LL << 21)) >> 22);
        }
    }
}

}  // extern "C"
