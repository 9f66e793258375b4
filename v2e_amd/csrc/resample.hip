// resample.hip -- the reference's host-side hand-off between its two hot kernels, on device.
//
// v2ecore/dataloader.py:136-147 (np.load -> PIL LANCZOS resize -> ToTensor [-> Normalize]) and
// v2ecore/slomo.py:437-444 ([revNormalize ->] ToPILImage (x*255 -> byte) -> PIL BILINEAR resize
// -> PNG -> cv2.imread) go through the CPU, PIL and the filesystem in the reference.  These
// kernels keep the frames in HBM and reproduce Pillow's 8-bit resampling bit for bit
// (libImaging/Resample.c: 22-bit fixed-point coefficients computed by v2e_amd/resample.py exactly
// as precompute_coeffs/normalize_coeffs_8bpc do; horizontal pass, uint8 clip8 intermediate,
// vertical pass).  Integer arithmetic: exact.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v)
{
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// out[n][y][xx] = clip8(sum_x in[n][y][xmin+x] * k[xx][x] + half)
__global__ __launch_bounds__(256) void k_resample_h(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, long long rows,
                                                    int iw, int ow, const int *__restrict__ bounds,
                                                    const int *__restrict__ coef, int ksize)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ow) return;
    const int xx = (int)(i % ow);
    const long long r = i / ow;
    const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
    const uint8_t *p = in + r * iw + x0;
    const int *k = coef + (size_t)xx * ksize;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < n; ++x) acc += (int)p[x] * k[x];
    out[i] = clip8(acc);
}

__global__ __launch_bounds__(256) void k_resample_v(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int nimg, int ih,
                                                    int oh, int w, const int *__restrict__ bounds,
                                                    const int *__restrict__ coef, int ksize)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)nimg * oh * w) return;
    const int x = (int)(i % w);
    const long long r = i / w;
    const int yy = (int)(r % oh);
    const long long img = r / oh;
    const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
    const uint8_t *p = in + (img * ih + y0) * w + x;
    const int *k = coef + (size_t)yy * ksize;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < n; ++y) acc += (int)p[(size_t)y * w] * k[y];
    out[i] = clip8(acc);
}

// ToTensor (uint8 -> float32 / 255) followed by Normalize(mean, std = 1)  (slomo.py:148-162)
__global__ __launch_bounds__(256) void k_u8_to_f32(const uint8_t *__restrict__ in, float *__restrict__ out, long long n, float mean)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)in[i] / 255.0f - mean;
}

// revNormalize (x + mean) followed by ToPILImage's pic.mul(255).byte(): truncation toward zero,
// then the low byte (what static_cast<uint8_t>(float) compiles to on the reference's CPU path)
__global__ __launch_bounds__(256) void k_f32_to_u8(const float *__restrict__ in, uint8_t *__restrict__ out, long long n, float mean)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (uint8_t)(int)((in[i] + mean) * 255.0f);
}

// [U][B] frame order of the interpolation batch -> time order [B][U] (slomo.py:441: idx = counter + U*b + k)
__global__ __launch_bounds__(256) void k_f32_to_u8_reorder(const float *__restrict__ in, uint8_t *__restrict__ out, int U, int B, int hw, float mean)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)U * B * hw) return;
    const int p = (int)(i % hw);
    const long long f = i / hw; // destination frame b*U + k
    const int b = (int)(f / U), k = (int)(f % U);
    out[i] = (uint8_t)(int)((in[((size_t)k * B + b) * hw + p] + mean) * 255.0f);
}

} // namespace

extern "C" {

int v2e_resample_u8(const uint8_t *in, uint8_t *tmp, uint8_t *out, int n, int ih, int iw, int oh, int ow,
                    const int32_t *hbounds, const int32_t *hcoef, int hksize, const int32_t *vbounds,
                    const int32_t *vcoef, int vksize, void *stream)
{
    V2E_REQUIRE(in && out && n > 0 && ih > 0 && iw > 0 && oh > 0 && ow > 0, "bad resample args");
    hipStream_t s = (hipStream_t)stream;
    const uint8_t *src = in;
    int cw = iw;
    if (ow != iw) {
        V2E_REQUIRE(hbounds && hcoef && (tmp || oh == ih), "horizontal tables / tmp missing");
        uint8_t *dst = (oh == ih) ? out : tmp;
        const long long rows = (long long)n * ih;
        k_resample_h<<<v2e_cdiv(rows * ow, 256), 256, 0, s>>>(src, dst, rows, iw, ow, hbounds, hcoef, hksize);
        src = dst;
        cw = ow;
    }
    if (oh != ih) {
        V2E_REQUIRE(vbounds && vcoef, "vertical tables missing");
        k_resample_v<<<v2e_cdiv((long long)n * oh * cw, 256), 256, 0, s>>>(src, out, n, ih, oh, cw, vbounds, vcoef, vksize);
    } else if (ow == iw) {
        V2E_HIP(hipMemcpyAsync(out, in, (size_t)n * ih * iw, hipMemcpyDeviceToDevice, s));
    }
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_u8_to_f32_norm(const uint8_t *in, float *out, int64_t n, float mean, void *stream)
{
    V2E_REQUIRE(in && out && n >= 0, "bad args");
    if (n == 0) return 0;
    k_u8_to_f32<<<v2e_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(in, out, n, mean);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_f32_to_u8_trunc(const float *in, uint8_t *out, int U, int B, int hw, float mean, int reorder, void *stream)
{
    V2E_REQUIRE(in && out && U > 0 && B > 0 && hw > 0, "bad args");
    const long long n = (long long)U * B * hw;
    if (reorder) k_f32_to_u8_reorder<<<v2e_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(in, out, U, B, hw, mean);
    else k_f32_to_u8<<<v2e_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(in, out, n, mean);
    V2E_HIP(hipGetLastError());
    return 0;
}

} // extern "C"
