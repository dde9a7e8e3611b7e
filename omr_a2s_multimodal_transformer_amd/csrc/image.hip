// Score-image front end on the GPU (src/data/preprocessing.py:44-52): convert("L") -> resize(BICUBIC) -> ToTensor.
// Byte / integer work, HBM-bound and tiny; bit-exact with Pillow's 8-bit resampler:
//   gray   L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16                      (ITU-R 601-2 luma in 16-bit fixed point)
//   pass   out = clip8((sum_k in[xmin + k] * coef[k] + 2^21) >> 22)             horizontal first, then vertical, a uint8
//                                                                               image between the two passes
//   float  out / 255 (fp32 division), stored as fp32 or bf16 straight into the (padded) batch tensor
// One thread per output pixel; consecutive threads walk consecutive output columns, so the horizontal pass reads
// overlapping windows of one source row (cache hits) and the vertical pass reads coalesced rows.
#include "omr_common.h"
#include "omr_hip.h"

namespace {

constexpr int kPrecisionBits = 22;

__device__ __forceinline__ int clip8(int v) {
    v >>= kPrecisionBits;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

template <int CH> __device__ __forceinline__ int luma(const unsigned char* p) {
    if constexpr (CH == 1) return p[0];
    else return (int)((p[0] * 19595u + p[1] * 38470u + p[2] * 7471u + 0x8000u) >> 16);
}

template <int CH>
__global__ __launch_bounds__(256) void gray_hpass_kernel(const unsigned char* __restrict__ src, int h, int w, long row_stride,
                                                         const int* __restrict__ bounds, const int* __restrict__ coefs, int ksize, int out_w,
                                                         unsigned char* __restrict__ dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)h * out_w) return;
    const int y = (int)(i / out_w), x = (int)(i - (long)y * out_w);
    const unsigned char* row = src + (long)y * row_stride;
    if (coefs == nullptr) {                       // same width: conversion only
        dst[i] = (unsigned char)luma<CH>(row + (long)x * CH);
        return;
    }
    const int xmin = bounds[2 * x], n = bounds[2 * x + 1];
    const int* k = coefs + (long)x * ksize;
    int ss = 1 << (kPrecisionBits - 1);
    for (int t = 0; t < n; ++t) ss += luma<CH>(row + (long)(xmin + t) * CH) * k[t];
    dst[i] = (unsigned char)clip8(ss);
}

template <typename T>
__global__ __launch_bounds__(256) void vpass_to_float_kernel(const unsigned char* __restrict__ src, int h, int w, const int* __restrict__ bounds,
                                                             const int* __restrict__ coefs, int ksize, int out_h, T* __restrict__ dst, long dst_row_stride) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)out_h * w) return;
    const int y = (int)(i / w), x = (int)(i - (long)y * w);
    int v;
    if (coefs == nullptr) {
        v = src[i];
    } else {
        const int ymin = bounds[2 * y], n = bounds[2 * y + 1];
        const int* k = coefs + (long)y * ksize;
        int ss = 1 << (kPrecisionBits - 1);
        for (int t = 0; t < n; ++t) ss += (int)src[(long)(ymin + t) * w + x] * k[t];
        v = clip8(ss);
    }
    dst[(long)y * dst_row_stride + x] = from_f32<T>(__fdiv_rn((float)v, 255.0f));
}

}  // namespace

extern "C" int omr_image_gray_hpass(const unsigned char* src, int h, int w, int channels, long row_stride, const int* bounds, const int* coefs,
                                    int ksize, int out_w, unsigned char* dst, void* stream) {
    if (!src || !dst || h <= 0 || w <= 0 || out_w <= 0 || (channels != 1 && channels != 3 && channels != 4)) return OMR_ERR_ARG;
    if ((coefs == nullptr) != (bounds == nullptr) || (coefs == nullptr && out_w != w) || (coefs && ksize <= 0)) return OMR_ERR_ARG;
    if (row_stride < (long)w * channels) return OMR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)(((long)h * out_w + 255) / 256);
    if (channels == 1) hipLaunchKernelGGL(gray_hpass_kernel<1>, dim3(blocks), dim3(256), 0, s, src, h, w, row_stride, bounds, coefs, ksize, out_w, dst);
    else if (channels == 3) hipLaunchKernelGGL(gray_hpass_kernel<3>, dim3(blocks), dim3(256), 0, s, src, h, w, row_stride, bounds, coefs, ksize, out_w, dst);
    else hipLaunchKernelGGL(gray_hpass_kernel<4>, dim3(blocks), dim3(256), 0, s, src, h, w, row_stride, bounds, coefs, ksize, out_w, dst);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_image_vpass_to_float(const unsigned char* src, int h, int w, const int* bounds, const int* coefs, int ksize, int out_h,
                                        int out_dtype, void* dst, long dst_row_stride, void* stream) {
    if (!src || !dst || h <= 0 || w <= 0 || out_h <= 0 || dst_row_stride < w) return OMR_ERR_ARG;
    if ((coefs == nullptr) != (bounds == nullptr) || (coefs == nullptr && out_h != h) || (coefs && ksize <= 0)) return OMR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)(((long)out_h * w + 255) / 256);
    if (out_dtype == OMR_F32) hipLaunchKernelGGL(vpass_to_float_kernel<float>, dim3(blocks), dim3(256), 0, s, src, h, w, bounds, coefs, ksize, out_h, (float*)dst, dst_row_stride);
    else if (out_dtype == OMR_BF16) hipLaunchKernelGGL(vpass_to_float_kernel<bf16>, dim3(blocks), dim3(256), 0, s, src, h, w, bounds, coefs, ksize, out_h, (bf16*)dst, dst_row_stride);
    else return OMR_ERR_UNSUPPORTED;
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}
