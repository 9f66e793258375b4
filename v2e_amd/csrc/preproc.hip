// preproc.hip -- stage 1 of v2e.py (:687-738) on device: cv2.resize(..., interpolation=cv2.INTER_AREA) of a uint8 frame
// (1 or 3 interleaved channels) followed by cv2.cvtColor(..., cv2.COLOR_BGR2GRAY).
//
// PARITY UNPINNED: OpenCV is a third-party dependency of the reference (setup.py: opencv-python, no version pinned) that is
// not in either tree and not in this image, and the reference holds no vectors for this stage; what is restated here is
// OpenCV 4.x's published algorithm (modules/imgproc/src/resize.cpp: computeResizeAreaTab, resizeArea_, resizeAreaFast_;
// color_yuv.simd.hpp: RGB2Gray<uchar>), with the reading of it stated line by line in oracle/preproc_oracle.py:
//   * shrinking in both directions only (scale_x, scale_y >= 1: the INTER_AREA branch proper);
//   * integer scale factors ("area fast"): the integer sum of the iscale_x x iscale_y box; 2 x 2: (sum + 2) >> 2, otherwise
//     saturate_cast<uchar>(sum * (1.f / area)) (float product, round half to even);
//   * otherwise: per source row the horizontal weighted sums buf[dx] = sum_k S[si_k] * alpha_k (float32, in table order, each
//     product and sum rounded), rows combined as sum[dx] = beta_0 buf_0 (+= beta_j buf_j ...), saturate_cast<uchar>(sum);
//     the (source index, weight) tables are computed on the host exactly as computeResizeAreaTab does (v2e_amd/preproc.py);
//   * BGR2GRAY, gray_shift 15 (the default: RGB2Gray<uchar> of OpenCV 4.x, color_rgb.simd.hpp: BY15 3735, GY15 19235, RY15 9798):
//     (B * 3735 + G * 19235 + R * 9798 + (1 << 14)) >> 15; gray_shift 14 (OpenCV 3.x's table form, and 4.x's YUV path):
//     (B * 1868 + G * 9617 + R * 4899 + (1 << 13)) >> 14.  The two differ by one grey level on ~1 % of random pixels; which one a
//     given cv2 computes is what scripts/check_stage1_against_cv2.py prints.
// The HIP kernels are tested bit for bit against that restatement (tests/test_preproc.py); neither has been compared with
// cv2 itself.
#include "common.h"

namespace {

__device__ __forceinline__ uint8_t sat_u8_rint(float v)
{
    const int i = __float2int_rn(v); // cvRound: round half to even
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

// general INTER_AREA: one thread per (image, dy, dx, channel)
__global__ __launch_bounds__(256) void k_area(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int nimg, int sh, int sw, int dh,
                                              int dw, int cn, const int *__restrict__ xofs, const int *__restrict__ xsi,
                                              const float *__restrict__ xal, const int *__restrict__ yofs, const int *__restrict__ ysi,
                                              const float *__restrict__ yal)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)nimg * dh * dw * cn;
    if (i >= total) return;
    const int c = (int)(i % cn);
    long long r = i / cn;
    const int dx = (int)(r % dw); r /= dw;
    const int dy = (int)(r % dh);
    const long long img = r / dh;
    const uint8_t *S0 = src + img * (long long)sh * sw * cn;
    const int k0 = xofs[dx], k1 = xofs[dx + 1], j0 = yofs[dy], j1 = yofs[dy + 1];
    float sum = 0.f;
    for (int j = j0; j < j1; ++j) {
        const uint8_t *S = S0 + (size_t)ysi[j] * sw * cn + c;
        float buf = 0.f;
        for (int k = k0; k < k1; ++k) buf = __fadd_rn(buf, __fmul_rn((float)S[(size_t)xsi[k] * cn], xal[k]));
        const float t = __fmul_rn(yal[j], buf);
        sum = j == j0 ? t : __fadd_rn(sum, t);
    }
    dst[i] = sat_u8_rint(sum);
}

// integer scale factors: the box sum
__global__ __launch_bounds__(256) void k_area_fast(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int nimg, int sh, int sw,
                                                   int dh, int dw, int cn, int isx, int isy)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)nimg * dh * dw * cn;
    if (i >= total) return;
    const int c = (int)(i % cn);
    long long r = i / cn;
    const int dx = (int)(r % dw); r /= dw;
    const int dy = (int)(r % dh);
    const long long img = r / dh;
    const uint8_t *S = src + ((img * sh + (long long)dy * isy) * sw + (long long)dx * isx) * cn + c;
    int sum = 0;
    for (int y = 0; y < isy; ++y)
        for (int x = 0; x < isx; ++x) sum += S[((size_t)y * sw + x) * cn];
    if (isx == 2 && isy == 2) dst[i] = (uint8_t)((sum + 2) >> 2);
    else dst[i] = sat_u8_rint(__fmul_rn((float)sum, 1.f / (float)(isx * isy)));
}

template <int SHIFT>
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, long long npx)
{
    constexpr int BY = SHIFT == 15 ? 3735 : 1868, GY = SHIFT == 15 ? 19235 : 9617, RY = SHIFT == 15 ? 9798 : 4899;
    static_assert(BY + GY + RY == (1 << SHIFT), "the coefficients sum to one");
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    const uint8_t *p = src + i * 3;
    dst[i] = (uint8_t)(((int)p[0] * BY + (int)p[1] * GY + (int)p[2] * RY + (1 << (SHIFT - 1))) >> SHIFT);
}

} // namespace

extern "C" {

int v2e_resize_area_u8(const uint8_t *src, uint8_t *dst, int n, int sh, int sw, int dh, int dw, int cn, const int32_t *xofs,
                       const int32_t *xsi, const float *xalpha, const int32_t *yofs, const int32_t *ysi, const float *yalpha, void *stream)
{
    V2E_REQUIRE(src && dst && n > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && (cn == 1 || cn == 3), "bad resize args");
    V2E_REQUIRE(dh <= sh && dw <= sw, "INTER_AREA is restated for shrinking only");
    const long long total = (long long)n * dh * dw * cn;
    hipStream_t s = (hipStream_t)stream;
    if (!xofs) { // integer scale factors
        V2E_REQUIRE(sw % dw == 0 && sh % dh == 0, "the box path needs integer scale factors");
        k_area_fast<<<v2e_cdiv(total, 256), 256, 0, s>>>(src, dst, n, sh, sw, dh, dw, cn, sw / dw, sh / dh);
    } else {
        V2E_REQUIRE(xsi && xalpha && yofs && ysi && yalpha, "null table");
        k_area<<<v2e_cdiv(total, 256), 256, 0, s>>>(src, dst, n, sh, sw, dh, dw, cn, xofs, xsi, xalpha, yofs, ysi, yalpha);
    }
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_bgr2gray_u8(const uint8_t *src_bgr, uint8_t *dst, int64_t npx, int gray_shift, void *stream)
{
    V2E_REQUIRE(src_bgr && dst && npx > 0, "bad args");
    V2E_REQUIRE(gray_shift == 14 || gray_shift == 15, "gray_shift is 15 (OpenCV 4.x RGB2Gray<uchar>) or 14 (OpenCV 3.x)");
    if (gray_shift == 15) k_bgr2gray<15><<<v2e_cdiv(npx, 256), 256, 0, (hipStream_t)stream>>>(src_bgr, dst, npx);
    else k_bgr2gray<14><<<v2e_cdiv(npx, 256), 256, 0, (hipStream_t)stream>>>(src_bgr, dst, npx);
    V2E_HIP(hipGetLastError());
    return 0;
}

} // extern "C"
