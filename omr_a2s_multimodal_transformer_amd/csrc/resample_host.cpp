// Host side of the score-image front end: the coefficient tables of a separable antialiased bicubic resize in 8-bit fixed
// point, as Pillow (the reference's `Image.resize`, src/data/preprocessing.py:44-52; Pillow's documented default filter for
// mode "L" is BICUBIC) defines them: per output pixel the window [xmin, xmin + n) of contributing input pixels and their
// weights, Keys cubic (a = -0.5) stretched by the down-scale factor, normalised in double precision and rounded to
// 22 fractional bits.  Compiled with -ffp-contract=off: the table must not depend on whether the host has FMA units.
#include <cmath>
#include <vector>

#include "omr_hip.h"

#define OMR_ERR_ARG (-1)      /* same code as omr_common.h (not included: device header) */

namespace {
constexpr int kPrecisionBits = 32 - 8 - 2;
constexpr double kSupport = 2.0;

inline double keys_cubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
}  // namespace

extern "C" int omr_resample_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return OMR_ERR_ARG;
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)std::ceil(kSupport * filterscale) * 2 + 1;
}

extern "C" int omr_resample_coeffs(int in_size, int out_size, int* bounds, int* coefs) {
    const int ksize = omr_resample_ksize(in_size, out_size);
    if (ksize < 0 || !bounds || !coefs) return OMR_ERR_ARG;
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = kSupport * filterscale, ss = 1.0 / filterscale;
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double w = keys_cubic((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (int x = xmax; x < ksize; ++x) k[x] = 0.0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
        for (int x = 0; x < ksize; ++x)
            coefs[(long)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << kPrecisionBits)) : (int)(0.5 + k[x] * (1 << kPrecisionBits));
    }
    return ksize;
}
